"""Drop-ins for probs.fastlog (probs.py:78-85) and probs.prob (probs.py:49-62): same positional
signatures, in-place numpy semantics, seconds returned.  TPB/BPG are accepted and ignored."""
import sys

import numpy as np

from . import _abi


def _rows2d(a, name):
    if a.ndim == 2 and a.size == 0 and a.dtype in (np.float32, np.float64):
        return max(a.shape[1], 1)
    if a.ndim != 2 or a.dtype not in (np.float32, np.float64) or a.strides[1] != a.itemsize \
            or a.strides[0] % a.itemsize:
        raise ValueError("%s must be a 2-D float32/float64 array with contiguous rows" % name)
    return a.strides[0] // a.itemsize


def fastlog(plI, MIN=sys.float_info.min, TPB=None, BPG=None, device=0):
    """x <- log10(max(x, MIN)) in place (probs.py:64-85)."""
    ld = _rows2d(plI, "plI")
    sec = _abi.C.c_double(0.0)
    _abi.check(_abi.lib().trpl_log10_clamp(_abi.ptr(plI), plI.itemsize, plI.shape[0], plI.shape[1], ld, float(MIN),
                                           int(device), _abi.C.byref(sec)))
    return sec.value


def prob(P, plI, values, uncertainty=None, mag_grid=None, TPB=None, BPG=None, device=0):
    """P[j] -= sum_i (plI[j,i] + mag_grid[j] - values[i])**2 in place (probs.py:20-62).
    `uncertainty` is accepted for signature compatibility; the reference never reads it (:40)."""
    if not (isinstance(P, np.ndarray) and P.dtype == np.float64 and P.ndim == 1 and P.flags.c_contiguous):
        raise ValueError("P must be a contiguous 1-D float64 array (a view is fine)")
    if plI.ndim == 2 and plI.size and plI.strides[1] != plI.itemsize:
        plI = np.ascontiguousarray(plI)                  # read-only input: a copy is harmless
    ld = _rows2d(plI, "plI")
    values = np.ascontiguousarray(values, dtype=np.float64)
    mag = np.ascontiguousarray(mag_grid, dtype=np.float64)
    if plI.shape != (len(P), len(values)) or mag.shape != (len(P),):
        raise ValueError("shape mismatch: P %r, plI %r, values %r, mag %r"
                         % (P.shape, plI.shape, values.shape, mag.shape))
    sec = _abi.C.c_double(0.0)
    _abi.check(_abi.lib().trpl_sse_accumulate(_abi.ptr(P), _abi.ptr(plI), plI.itemsize, len(P), len(values), ld,
                                              _abi.ptr(values), _abi.ptr(mag), int(device), _abi.C.byref(sec)))
    return sec.value
