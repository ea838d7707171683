"""Drop-in for the reference's GPU `model` callable, pvSimPCR.pvSim (pvSimPCR.py:309-401):
same positional signature, same in-place numpy semantics, same return value (solver seconds).
The work is done by hand-written gfx950 kernels behind the C ABI (trpl_solve_pl)."""
import numpy as np

from . import _abi


def _as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def solve_pl(matPar, Length, Time, L, T, dN, plT=1, tol=7, MAX=10000, out=None, dtype=np.float64,
             strict=False, device=0, fp32=False):
    """PL(t) for S systems of one curve.  matPar (S,12) and dN (L,) in nm/ns units.
    Returns (plI, status, iters_total, seconds)."""
    matPar = _as_f64(matPar)
    if matPar.ndim != 2 or matPar.shape[1] != 12:
        raise ValueError("matPar must have shape (S, 12)")
    dN = _as_f64(dN)
    if dN.shape != (L,):
        raise ValueError("excitation must have L=%d points, got %r" % (L, dN.shape))
    S = matPar.shape[0]
    ncol = T // plT + 1
    if out is None:
        out = np.empty((S, ncol), dtype=dtype)
    if out.ndim != 2 or out.shape != (S, ncol) or out.dtype not in (np.float32, np.float64) \
            or out.strides[1] != out.itemsize or out.strides[0] % out.itemsize:
        raise ValueError("plI must be a (S, T//plT+1) float32/float64 array with contiguous rows")
    status = np.zeros(S, dtype=np.int32)
    iters = np.zeros(S, dtype=np.int64)
    sec = _abi.C.c_double(0.0)
    flags = (_abi.FLAG_STRICT if strict else 0) | (_abi.FLAG_FP32 if fp32 else 0)
    _abi.check(_abi.lib().trpl_solve_pl(_abi.ptr(matPar), S, float(Length), float(Time), int(L), int(T), int(plT),
                                        int(tol), int(MAX), _abi.ptr(dN), _abi.ptr(out), out.itemsize,
                                        out.strides[0] // out.itemsize, _abi.ptr(status), _abi.ptr(iters), flags,
                                        int(device), _abi.C.byref(sec)))
    return out, status, iters, sec.value


def pvSim(plI_main, plN_main, plP_main, plE_main, matPar, simPar, iniPar, TPB=None, BPG=None,
          max_sims_per_block=1, init_mode="exp", strict=False, device=0, info=None):
    """pvSimPCR.pvSim (pvSimPCR.py:309).  plN/plP/plE (unused debug buffers there, :368-370),
    TPB, BPG and max_sims_per_block (CUDA launch shape) are accepted and ignored: one
    wavefront owns one system.  `info`, if a dict, receives 'status' and 'iters_total'."""
    Length, Time, L, T, plT, pT, tol, MAX = simPar
    if max_sims_per_block != 1:
        # bundling couples the convergence of unrelated samples in the reference
        # (pvSimPCR.py:213-216); the per-system result is the MSPB = 1 one
        pass
    dx = Length / L
    if init_mode == "exp":                                   # pvSimPCR.py:347-353
        a, l = iniPar
        x = np.arange(L) + 0.5
        dN = a * np.exp(-x / (l / dx))
    elif init_mode == "points":                              # :355-356
        dN = np.asarray(iniPar, dtype=np.float64)
    else:
        raise ValueError("init_mode %r is not supported (the reference's 'continue' is broken, "
                         "pvSimPCR.py:357-362)" % (init_mode,))
    _, status, iters, sec = solve_pl(matPar, Length, Time, int(L), int(T), dN, plT=int(plT), tol=int(tol),
                                     MAX=int(MAX), out=plI_main, strict=strict, device=device)
    if info is not None:
        info["status"] = status
        info["iters_total"] = iters
    return sec
