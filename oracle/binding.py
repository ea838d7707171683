"""ctypes binding of the CPU oracle (oracle/liboracle.so) + a restatement of the
reference's host control loop (bayeslib.simulate) on top of it.

TEST INFRASTRUCTURE, NOT PRODUCT -- see oracle/__init__.py.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)


def _ptr(a, ty=C.c_void_p):
    return a.ctypes.data_as(ty) if a is not None else None


class OracleLib:
    def __init__(self, path):
        self.dll = C.CDLL(path)
        d = self.dll
        d.oracle_version.restype = C.c_int
        d.oracle_pcreduce.argtypes = [_dp] * 6 + [C.c_int]
        d.oracle_pcreduce.restype = None
        d.oracle_norm2.argtypes = [_dp] * 6 + [C.c_int]
        d.oracle_norm2.restype = C.c_double
        d.oracle_scales.argtypes = [C.c_double, C.c_double, C.c_int, C.c_long, _dp, _dp, _dp]
        d.oracle_scales.restype = None
        d.oracle_pvsim.argtypes = [_dp, C.c_long, C.c_double, C.c_double, C.c_int, C.c_long, C.c_int,
                                   C.c_int, C.c_int, _dp, C.c_void_p, C.c_int, C.c_long,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        d.oracle_pvsim.restype = C.c_int
        d.oracle_pvsim_snap.argtypes = d.oracle_pvsim.argtypes[:-1] + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                                      C.c_void_p, C.c_int]
        d.oracle_pvsim_snap.restype = C.c_int
        d.oracle_pvsim_bundle.argtypes = d.oracle_pvsim.argtypes[:-1] + [C.c_int, C.c_int]
        d.oracle_pvsim_bundle.restype = C.c_int
        d.oracle_fastlog.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_long, C.c_double]
        d.oracle_fastlog.restype = None
        d.oracle_prob.argtypes = [_dp, C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_long, _dp, _dp]
        d.oracle_prob.restype = None
        d.oracle_set_max_order.argtypes = [C.c_int]
        d.oracle_set_max_order.restype = C.c_int


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "trpl_oracle.c")
    if force or not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def load():
    global _LIB
    if _LIB is None:
        _LIB = OracleLib(build())
    return _LIB


class fma_variant:
    """Context manager: inside it every oracle call runs the FMA-contracting build of the same source
    (oracle/Makefile: liboracle_fma.so) -- the reference's arithmetic as numba-CUDA / nvcc would contract it.  Not the
    pinned oracle; used to measure the distance between two legitimate evaluations of the reference."""

    def __enter__(self):
        global _LIB
        so = os.path.join(_HERE, "liboracle_fma.so")
        src = os.path.join(_HERE, "trpl_oracle.c")
        if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_fma.so"])
        self._saved = load()
        _LIB = OracleLib(so)
        return self

    def __exit__(self, *exc):
        global _LIB
        _LIB = self._saved


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def pcreduce(ld, d, ud, B):
    """Solve one tridiagonal system (reference call form pcreduce(A2, A1, A0, bb, x))."""
    ld, d, ud, B = (_f64(v).copy() for v in (ld, d, ud, B))
    n = len(d)
    x = np.zeros(n)
    buf = np.zeros(4 * n)
    load().dll.oracle_pcreduce(_ptr(ld, _dp), _ptr(d, _dp), _ptr(ud, _dp), _ptr(B, _dp), _ptr(x, _dp),
                               _ptr(buf, _dp), n)
    return x


def norm2(A0, A1, A2, b, c):
    A0, A1, A2, b, c = (_f64(v) for v in (A0, A1, A2, b, c))
    n = len(b)
    buf = np.zeros(2 * n)
    return load().dll.oracle_norm2(_ptr(A0, _dp), _ptr(A1, _dp), _ptr(A2, _dp), _ptr(b, _dp), _ptr(c, _dp),
                                   _ptr(buf, _dp), n)


def scales(length, time_ns, L, T):
    s = np.zeros(12)
    dx3 = C.c_double()
    pln = C.c_double()
    load().dll.oracle_scales(length, time_ns, L, T, _ptr(s, _dp), C.byref(dx3), C.byref(pln))
    return s, dx3.value, pln.value


def pvsim(mat12, length, time_ns, L, T, ini, plT=1, tol=7, MAX=10000, dtype=np.float64, nthreads=1,
          want_step_iters=False, snap_steps=None, mspb=1, max_order=5):
    """pvSim(..., init_mode="points") semantics.  Returns dict(plI, status, iters_total, iters_max[, step_iters]
    [, plN, plP, plE: the state at the time steps snap_steps, pvSimPCR.py:283-288 / Legacy/pvSim.py:121-126]).
    mspb > 1: max_sims_per_block consecutive samples share one convergence test (pvSimPCR.py:213-216,:258-266).
    max_order < 5 caps the BDF order ramp of pvSimPCR.py:241-250 (2: Euler, then BDF2 -- Legacy/pvSim.py:94-97)."""
    old_order = load().dll.oracle_set_max_order(int(max_order))
    try:
        return _pvsim(mat12, length, time_ns, L, T, ini, plT, tol, MAX, dtype, nthreads, want_step_iters, snap_steps, mspb)
    finally:
        load().dll.oracle_set_max_order(old_order)


def _pvsim(mat12, length, time_ns, L, T, ini, plT, tol, MAX, dtype, nthreads, want_step_iters, snap_steps, mspb):
    mat12 = _f64(mat12)
    S = mat12.shape[0]
    assert mat12.shape[1] == 12
    ini = _f64(ini)
    assert ini.shape == (L,)
    ncol = T // plT + 1
    pl = np.empty((S, ncol), dtype=dtype)
    status = np.zeros(S, dtype=np.int32)
    itot = np.zeros(S, dtype=np.int64)
    imax = np.zeros(S, dtype=np.int32)
    steps = np.zeros((S, T + 1), dtype=np.int32) if want_step_iters else None
    if mspb != 1:
        assert snap_steps is None, "the bundled oracle records no snapshots"
        rc = load().dll.oracle_pvsim_bundle(_ptr(mat12, _dp), S, float(length), float(time_ns), int(L), int(T), int(plT),
                                            int(tol), int(MAX), _ptr(ini, _dp), _ptr(pl), pl.dtype.itemsize, ncol,
                                            _ptr(status), _ptr(itot), _ptr(imax), _ptr(steps), int(mspb), int(nthreads))
        if rc != 0:
            raise ValueError("oracle_pvsim_bundle: bad arguments")
        out = {"plI": pl, "status": status, "iters_total": itot, "iters_max": imax}
        if want_step_iters:
            out["step_iters"] = steps
        return out
    snaps = np.ascontiguousarray(snap_steps if snap_steps is not None else [], dtype=np.int64)   # C long
    ns = len(snaps)
    plN, plP, plE = np.zeros((S, ns, L)), np.zeros((S, ns, L)), np.zeros((S, ns, L + 1))
    rc = load().dll.oracle_pvsim_snap(_ptr(mat12, _dp), S, float(length), float(time_ns), int(L), int(T), int(plT),
                                      int(tol), int(MAX), _ptr(ini, _dp), _ptr(pl), pl.dtype.itemsize, ncol,
                                      _ptr(status), _ptr(itot), _ptr(imax), _ptr(steps), _ptr(snaps) if ns else None,
                                      ns, _ptr(plN) if ns else None, _ptr(plP) if ns else None,
                                      _ptr(plE) if ns else None, int(nthreads))
    if rc != 0:
        raise ValueError("oracle_pvsim: bad arguments")
    out = {"plI": pl, "status": status, "iters_total": itot, "iters_max": imax}
    if ns:
        out.update(plN=plN, plP=plP, plE=plE)
    if want_step_iters:
        out["step_iters"] = steps
    return out


def fastlog(x, MIN=sys.float_info.min):
    """In place, like probs.fastlog (probs.py:78-85)."""
    assert x.flags.c_contiguous and x.ndim == 2 and x.dtype in (np.float32, np.float64)
    load().dll.oracle_fastlog(_ptr(x), x.dtype.itemsize, x.shape[0], x.shape[1], x.shape[1], float(MIN))
    return x


def prob(P, plI, values, mag):
    """In place on P, like probs.prob (probs.py:49-62); `uncertainty` is unused there."""
    assert P.flags.c_contiguous and P.dtype == np.float64
    plI = np.ascontiguousarray(plI)
    values = _f64(values)
    mag = _f64(mag)
    assert plI.shape == (len(P), len(values)) and plI.dtype in (np.float32, np.float64)
    load().dll.oracle_prob(_ptr(P, _dp), _ptr(plI), plI.dtype.itemsize, plI.shape[0], plI.shape[1], plI.shape[1],
                           _ptr(values, _dp), _ptr(mag, _dp))
    return P


def _almost_equal(x, x0, threshold=1e-10):
    """bayeslib.almost_equal (bayeslib.py:78-81)."""
    if x.shape != x0.shape:
        return False
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.abs(np.nanmax((x - x0) / x0)) < threshold


def simulate_loglik(X, ini, lengths, time_ns, L, T, e_data, tol=7, MAX=10000, sims_per_gpu=1024,
                    pl_dtype=np.float32, normalize=False, log_pl=True, nthreads=1, mspb=1):
    """Restatement of bayeslib.simulate's GPU branch (bayeslib.py:83-205) over the oracle:
    curves -> sample blocks -> experiments; fp32 plI buffer (:137); X[:, :-1] to the model and
    X[:, -1] as the log offset (:144,:195); optional self-normalisation (:150-154); log clamp
    (:155-157); bypass or per-row scipy griddata time interpolation (:173-191); prob (:195).

    X (S,13) in nm/ns units; ini (C,L); lengths scalar or (C,); e_data = list of
    (times[c], log10 values[c]) per experiment.  Returns P (n_exp, S).  mspb: gpu_info["max_sims_per_block"]
    (:93,:146), bundles restart with every block of sims_per_gpu samples.
    """
    from scipy.interpolate import griddata
    X = _f64(X)
    S = len(X)
    ncurves = len(ini)
    lengths = [float(lengths)] * ncurves if np.isscalar(lengths) else [float(v) for v in lengths]
    P = np.zeros((len(e_data), S))
    sim_t = np.linspace(0, time_ns, T + 1)                                     # :115
    for c in range(ncurves):
        for blk in range(0, S, sims_per_gpu):                                  # :131
            size = min(sims_per_gpu, S - blk)
            r = pvsim(X[blk:blk + size, :-1], lengths[c], time_ns, L, T, ini[c], tol=tol, MAX=MAX,
                      dtype=pl_dtype, nthreads=nthreads, mspb=mspb)
            pl = r["plI"]
            if normalize:                                                      # :150-154
                pl = (pl.T / pl.T[0]).T.astype(pl_dtype)
            if log_pl:
                fastlog(pl)                                                    # :157
            for e, exp in enumerate(e_data):
                times = np.asarray(exp[0][c], dtype=float)
                values = np.asarray(exp[1][c], dtype=float)
                if _almost_equal(sim_t, times):                                # :173,:182-183
                    pl_int = pl
                else:                                                          # :186-189
                    pl_int = np.empty((size, len(times)))
                    for i, row in enumerate(pl):
                        pl_int[i] = griddata(sim_t, row, times)
                prob(P[e, blk:blk + size], pl_int, values, np.ascontiguousarray(X[blk:blk + size, -1]))
    return P
