"""Drop-in for the reference's GPU `model` callable, pvSimPCR.pvSim (pvSimPCR.py:309-401):
same positional signature, same in-place numpy semantics, same return value (solver seconds).
The work is done by hand-written gfx950 kernels behind the C ABI (trpl_solve_pl)."""
import numpy as np

from . import _abi


def _as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def solve_pl(matPar, Length, Time, L, T, dN, plT=1, tol=7, MAX=10000, out=None, dtype=np.float64,
             strict=False, device=0, fp32=False, kernel=None, mixed=False, snap_steps=None, plN=None, plP=None, plE=None,
             snapshots=None, resume=None, snap_raw=False, bundle=1, hist32=False, bdf_order=None, extra_flags=0):
    """PL(t) for S systems of one curve.  matPar (S,12) and dN (L,) in nm/ns units.
    Returns (plI, status, iters_total, seconds).

    kernel: None (the library picks by launch size), "pair" or "single" (TRPL_FLAG_KERNEL_*).
    snap_steps: time-step indices at which the state is recorded (the reference's pT after
    bayeslib.py:123); the snapshots are written into plN, plP (S, len(snap_steps), L) and plE
    (S, len(snap_steps), L+1) when given (float64, C-contiguous, filled in place like the reference's
    plN_main / plP_main / plE_main), or returned in the dict `snapshots` (keys 'plN', 'plP', 'plE').
    snap_raw: snapshots in solver units (TRPL_FLAG_SNAP_RAW), the form a resume reads back bit for bit.
    resume: (t0, N5, P5, E5) -- continue at step t0 >= 4 from the five newest time levels t0-4 .. t0 of every
    system, arrays (S, 5, L), (S, 5, L), (S, 5, L+1) as recorded by snap_steps=checkpoint_steps(t0) with
    snap_raw=True (trpl_solve_pl_resume; the reference's init_mode="continue", pvSimPCR.py:357-358); dN is
    ignored, PL columns before t0 keep what `out` holds.  Pin `kernel` to repeat an uninterrupted run bit for bit
    (the automatic choice looks at the number of steps left).
    bundle: the reference's max_sims_per_block (TRPL_FLAG_BUNDLE, up to 4): `bundle` consecutive samples share one
    convergence test per inner iteration (pvSimPCR.py:211-216) -- strict=True bit-identical to the reference run that
    way, otherwise to rounding (L <= 128, one-system kernel).
    bdf_order: cap the BDF order ramp (pvSimPCR.py:241-250) at 1 .. 5 (TRPL_FLAG_BDF_ORDER; 2 = Legacy/pvSim.py's scheme).
    extra_flags: further TRPL_FLAG_* bits, ORed in (tests / measurements: _abi.FLAG_PAIR_ALWAYS_SEAM, ...)."""
    matPar = _as_f64(matPar)
    if matPar.ndim != 2 or matPar.shape[1] != 12:
        raise ValueError("matPar must have shape (S, 12)")
    S = matPar.shape[0]
    if resume is None:
        dN = _as_f64(dN)
        if dN.shape != (L,):
            raise ValueError("excitation must have L=%d points, got %r" % (L, dN.shape))
    else:
        t0, rN, rP, rE = resume
        rN, rP, rE = _as_f64(rN), _as_f64(rP), _as_f64(rE)
        if rN.shape != (S, 5, L) or rP.shape != (S, 5, L) or rE.shape != (S, 5, L + 1):
            raise ValueError("resume levels must have shapes (S, 5, L), (S, 5, L), (S, 5, L+1)")
    ncol = T // plT + 1
    if out is None:
        out = np.empty((S, ncol), dtype=dtype)
    if out.ndim != 2 or out.shape != (S, ncol) or out.dtype not in (np.float32, np.float64) \
            or out.strides[1] != out.itemsize or out.strides[0] % out.itemsize:
        raise ValueError("plI must be a (S, T//plT+1) float32/float64 array with contiguous rows")
    status = np.zeros(S, dtype=np.int32)
    iters = np.zeros(S, dtype=np.int64)
    sec = _abi.C.c_double(0.0)
    flags = (_abi.FLAG_STRICT if strict else 0) | _abi.fp32_flags(fp32) | _abi.kernel_flag(kernel) \
        | (_abi.FLAG_MIXED if mixed else 0) | (_abi.FLAG_SNAP_RAW if snap_raw else 0) | _abi.flag_bundle(bundle, L) \
        | (_abi.FLAG_HIST32 if hist32 else 0) | _abi.flag_bdf_order(bdf_order) | int(extra_flags)
    steps = None
    n_snap = 0
    if snap_steps is not None and len(snap_steps):
        steps = np.ascontiguousarray(snap_steps, dtype=np.int64)
        n_snap = len(steps)
        if steps.ndim != 1 or n_snap > _abi.MAX_SNAPS:
            raise ValueError("snap_steps must be a 1-D list of at most %d time-step indices" % _abi.MAX_SNAPS)
        if snapshots is not None:
            plN = np.zeros((S, n_snap, L)) if plN is None else plN
            plP = np.zeros((S, n_snap, L)) if plP is None else plP
            plE = np.zeros((S, n_snap, L + 1)) if plE is None else plE
        for name, arr, width in (("plN", plN, L), ("plP", plP, L), ("plE", plE, L + 1)):
            if arr is not None and not (isinstance(arr, np.ndarray) and arr.dtype == np.float64 and arr.flags.c_contiguous
                                        and arr.shape == (S, n_snap, width)):
                raise ValueError("%s must be a C-contiguous float64 array of shape (%d, %d, %d)" % (name, S, n_snap, width))
    head = (_abi.ptr(matPar), S, float(Length), float(Time), int(L), int(T), int(plT), int(tol), int(MAX))
    tail = (_abi.ptr(out), out.itemsize, out.strides[0] // out.itemsize, _abi.ptr(status), _abi.ptr(iters),
            _abi.ptr(steps), n_snap, _abi.ptr(plN) if n_snap else None, _abi.ptr(plP) if n_snap else None,
            _abi.ptr(plE) if n_snap else None, flags, int(device), _abi.C.byref(sec))
    if resume is None:
        _abi.check(_abi.lib().trpl_solve_pl_snap(*head, _abi.ptr(dN), *tail))
    else:
        _abi.check(_abi.lib().trpl_solve_pl_resume(*head, int(t0), _abi.ptr(rN), _abi.ptr(rP), _abi.ptr(rE), *tail))
    if snapshots is not None:
        snapshots.update(plN=plN, plP=plP, plE=plE)
    return out, status, iters, sec.value


def checkpoint_steps(t0):
    """The five time steps whose raw snapshots (snap_raw=True) let a run be continued at step t0."""
    if t0 < 4:
        raise ValueError("a checkpoint needs five time levels: t0 >= 4")
    return [int(t0) - 4 + m for m in range(5)]


def _snapshot_target(arr, S, n, width):
    """The caller's plN/plP/plE buffer if the kernel can fill it in place (pvSimPCR.py:366-368 ships whatever
    it is given to the device): a float64 C-contiguous (S, len(pT), width) array."""
    return arr if (isinstance(arr, np.ndarray) and arr.dtype == np.float64 and arr.flags.c_contiguous
                   and arr.shape == (S, n, width)) else None


def pvSim(plI_main, plN_main, plP_main, plE_main, matPar, simPar, iniPar, TPB=None, BPG=None,
          max_sims_per_block=1, init_mode="exp", strict=False, device=0, info=None, kernel=None):
    """pvSimPCR.pvSim (pvSimPCR.py:309).  TPB and BPG (CUDA launch shape) are accepted and ignored: one wavefront owns
    one system (or two).  max_sims_per_block = 2 .. 4 (2 .. 16 on grids of up to 64 nodes) -- neighbouring samples
    sharing one convergence test -- is honoured (bit for bit with strict=True).  init_mode "continue" (a stub in the reference,
    pvSimPCR.py:357-358) works here: iniPar = (t0, N5, P5, E5), see solve_pl(resume=...).  plN_main / plP_main / plE_main, the reference's
    debug outputs (recording hook pvSimPCR.py:283-288, disabled there; working form Legacy/pvSim.py:121-126,
    :169-171), are FILLED when they are float64 arrays of shape (S, len(pT), L) / (S, len(pT), L+1): the
    densities (nm^-3) and the field (nm^-1) of the state at the time steps pT = simPar[5]; anything else
    (None, the dummies bayeslib passes) is ignored as before.  `info`, if a dict, receives 'status' and
    'iters_total'.  kernel: None (the library picks the FAST stepper by launch size) | "pair" | "single" (TRPL_FLAG_KERNEL_*;
    driver.simulate pins it for the launches it overlaps)."""
    Length, Time, L, T, plT, pT, tol, MAX = simPar
    # max_sims_per_block > 1 couples the convergence of neighbouring samples in the reference (pvSimPCR.py:213-216):
    # honoured up to 4 per bundle from L = 128 on and 16 up to L = 64 (strict: any L, bit for bit; otherwise L <= 128, to
    # rounding) -- the reference's 48 KB of shared memory hold 3 / 6 / 13 at L = 128 / 64 / 32; beyond that every sample
    # converges on its own
    bundle = int(max_sims_per_block)
    if not (1 <= bundle <= _abi.bundle_cap(L) and (strict or int(L) <= 128)):
        bundle = 1
    dx = Length / L
    if init_mode == "exp":                                   # pvSimPCR.py:347-353
        a, l = iniPar
        x = np.arange(L) + 0.5
        dN = a * np.exp(-x / (l / dx))
    elif init_mode == "points":                              # :355-356
        dN = np.asarray(iniPar, dtype=np.float64)
    elif init_mode == "continue":                            # :357-358 (`pass` there: the call then fails on dN)
        dN = None
    else:
        raise ValueError("init_mode %r is not supported (pvSimPCR.py:359-362)" % (init_mode,))
    S = len(matPar)
    steps = None
    try:
        steps = [int(v) for v in pT] if pT is not None else None
    except TypeError:
        steps = None
    snaps = {}
    if steps and 0 < len(steps) <= _abi.MAX_SNAPS:
        snaps = {"plN": _snapshot_target(plN_main, S, len(steps), int(L)),
                 "plP": _snapshot_target(plP_main, S, len(steps), int(L)),
                 "plE": _snapshot_target(plE_main, S, len(steps), int(L) + 1)}
    want = any(v is not None for v in snaps.values())
    _, status, iters, sec = solve_pl(matPar, Length, Time, int(L), int(T), dN, plT=int(plT), tol=int(tol),
                                     MAX=int(MAX), out=plI_main, strict=strict, device=device,
                                     resume=tuple(iniPar) if init_mode == "continue" else None, bundle=bundle,
                                     kernel=kernel if (bundle == 1 and not strict and int(L) == 128) else None,
                                     snap_steps=steps if want else None, **(snaps if want else {}))
    if info is not None:
        info["status"] = status
        info["iters_total"] = iters
    return sec
