"""Device-resident entry points: inputs and outputs are torch tensors already in HBM; nothing is
allocated, copied or synchronised by the library (trpl_*_dev in include/trpl.h).  torch is used
for device memory, streams and torch.distributed only -- plumbing, not compute."""
import numpy as np

from . import _abi


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype, name):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError("%s must be a contiguous CUDA tensor of dtype %s" % (name, dtype))
    return t.data_ptr()


def loglik_device(X, init_params, lengths, Time, L, T, obs, n_obs, P, sse, status=None, iters_total=None,
                  tol=7, MAX=10000, plT=1, flags=0, floor_col=None):
    """trpl_loglik_dev on the current device and stream.  X (S,13) f64, init_params (C,L) f64,
    obs (C,obs_ld) f64, P (S,) f64 accumulated in place, sse (C,S) f64 out, optional status
    (C,S) int32, iters_total (C,S) int64 and floor_col (C,S) int32 (first compared PL column below the
    cancellation floor, -1 if none: include/trpl.h).  lengths / n_obs are host sequences."""
    import torch
    S, Cn = X.shape[0], init_params.shape[0]
    if X.shape[1] != 13 or init_params.shape[1] != L or obs.shape[0] != Cn or tuple(sse.shape) != (Cn, S) \
            or tuple(P.shape) != (S,):
        raise ValueError("shape mismatch")
    lengths = np.ascontiguousarray(np.broadcast_to(np.asarray(lengths, dtype=np.float64), (Cn,)))
    n_obs = np.ascontiguousarray(np.broadcast_to(np.asarray(n_obs, dtype=np.int64), (Cn,)))
    _abi.check(_abi.lib().trpl_loglik_dev(
        _chk(X, torch.float64, "X"), S, Cn, _abi.ptr(lengths), float(Time), int(L), int(T), int(plT), int(tol),
        int(MAX), _chk(init_params, torch.float64, "init_params"), _chk(obs, torch.float64, "obs"),
        obs.shape[1], _abi.ptr(n_obs), _chk(P, torch.float64, "P"), _chk(sse, torch.float64, "sse"),
        None if status is None else _chk(status, torch.int32, "status"),
        None if iters_total is None else _chk(iters_total, torch.int64, "iters_total"),
        None if floor_col is None else _chk(floor_col, torch.int32, "floor_col"), int(flags), _stream()))


def loglik_obs_device(X, init_params, lengths, Time, L, T, obs, obs_hi, obs_dx, obs_h, n_obs, P, sse, status=None,
                      iters_total=None, tol=7, MAX=10000, flags=0, floor_col=None):
    """trpl_loglik_obs_dev: observation times off the simulation grid.  obs / obs_dx / obs_h (C,obs_ld)
    f64 and obs_hi (C,obs_ld) int32 are the bracketing arrays of driver.bracket_times, on the device."""
    import torch
    S, Cn = X.shape[0], init_params.shape[0]
    if X.shape[1] != 13 or init_params.shape[1] != L or tuple(sse.shape) != (Cn, S) or tuple(P.shape) != (S,) \
            or not (obs.shape == obs_hi.shape == obs_dx.shape == obs_h.shape) or obs.shape[0] != Cn:
        raise ValueError("shape mismatch")
    lengths = np.ascontiguousarray(np.broadcast_to(np.asarray(lengths, dtype=np.float64), (Cn,)))
    n_obs = np.ascontiguousarray(np.broadcast_to(np.asarray(n_obs, dtype=np.int64), (Cn,)))
    _abi.check(_abi.lib().trpl_loglik_obs_dev(
        _chk(X, torch.float64, "X"), S, Cn, _abi.ptr(lengths), float(Time), int(L), int(T), int(tol), int(MAX),
        _chk(init_params, torch.float64, "init_params"), _chk(obs, torch.float64, "obs"),
        _chk(obs_hi, torch.int32, "obs_hi"), _chk(obs_dx, torch.float64, "obs_dx"), _chk(obs_h, torch.float64, "obs_h"),
        obs.shape[1], _abi.ptr(n_obs), _chk(P, torch.float64, "P"), _chk(sse, torch.float64, "sse"),
        None if status is None else _chk(status, torch.int32, "status"),
        None if iters_total is None else _chk(iters_total, torch.int64, "iters_total"),
        None if floor_col is None else _chk(floor_col, torch.int32, "floor_col"), int(flags), _stream()))


def solve_pl_device(matPar, Length, Time, L, T, dN, plI, status=None, iters_total=None, tol=7, MAX=10000, plT=1,
                    flags=0):
    """trpl_solve_pl_dev: matPar (S,12) f64, dN (L,) f64, plI (S, T//plT+1) f32/f64 out."""
    import torch
    S = matPar.shape[0]
    if matPar.shape[1] != 12 or tuple(dN.shape) != (L,) or tuple(plI.shape) != (S, T // plT + 1):
        raise ValueError("shape mismatch")
    if plI.dtype not in (torch.float32, torch.float64):
        raise ValueError("plI must be float32 or float64")
    _abi.check(_abi.lib().trpl_solve_pl_dev(
        _chk(matPar, torch.float64, "matPar"), S, float(Length), float(Time), int(L), int(T), int(plT), int(tol),
        int(MAX), _chk(dN, torch.float64, "dN"), _chk(plI, plI.dtype, "plI"), plI.element_size(), plI.shape[1],
        None if status is None else _chk(status, torch.int32, "status"),
        None if iters_total is None else _chk(iters_total, torch.int64, "iters_total"), int(flags), _stream()))


def pcr_solve_device(ld, d, ud, b, x, flags=0):
    """trpl_pcr_solve_batched_dev: all (S,L) tensors of one dtype (f64 or f32)."""
    import torch
    S, L = d.shape
    dt = d.dtype
    if dt not in (torch.float32, torch.float64):
        raise ValueError("dtype must be float32 or float64")
    for t in (ld, ud, b, x):
        if tuple(t.shape) != (S, L):
            raise ValueError("shape mismatch")
    _abi.check(_abi.lib().trpl_pcr_solve_batched_dev(_chk(ld, dt, "ld"), _chk(d, dt, "d"), _chk(ud, dt, "ud"),
                                                     _chk(b, dt, "b"), _chk(x, dt, "x"), S, L, d.element_size(),
                                                     int(flags), _stream()))


# ---- posterior core, device-resident (trpl_posterior_*_dev) ----
def posterior_workspace(D=16):
    """A workspace tensor large enough for any posterior_*_device call with up to D columns."""
    import torch
    n = int(_abi.lib().trpl_posterior_workspace_bytes(int(D)))
    return torch.empty(n // 8, dtype=torch.float64, device="cuda")


def posterior_weights_device(LL, tf, W, workspace, stats=None):
    """W <- normalize(LL / tf) (Visualization/utils.py:157-166); LL, W (S,) f64; stats (2,) f64 optional
    {max, raw sum}."""
    import torch
    if LL.shape != W.shape or LL.dim() != 1:
        raise ValueError("LL and W must be (S,)")
    _abi.check(_abi.lib().trpl_posterior_weights_dev(
        _chk(LL, torch.float64, "LL"), LL.shape[0], float(tf), _chk(W, torch.float64, "W"),
        None if stats is None else _chk(stats, torch.float64, "stats"), _chk(workspace, torch.float64, "workspace"),
        workspace.numel() * 8, _stream()))


def posterior_moments_device(V, W, sums, central, workspace, mean_in=None):
    """V (D,S), W (S,) -> sums (2+D,), central (D, D+2) as trpl_posterior_moments defines them."""
    import torch
    D, S = V.shape
    if tuple(W.shape) != (S,) or tuple(sums.shape) != (2 + D,) or tuple(central.shape) != (D, D + 2):
        raise ValueError("shape mismatch")
    _abi.check(_abi.lib().trpl_posterior_moments_dev(
        _chk(V, torch.float64, "V"), S, D, _chk(W, torch.float64, "W"),
        None if mean_in is None else _chk(mean_in, torch.float64, "mean_in"), _chk(sums, torch.float64, "sums"),
        _chk(central, torch.float64, "central"), _chk(workspace, torch.float64, "workspace"), workspace.numel() * 8,
        _stream()))


def posterior_hist_device(x, W, lo, hi, out, y=None, ylo=0.0, yhi=1.0):
    """out (bins,) or (bins, ybins) += weighted counts (W None: counts); the caller zeroes out."""
    import torch
    xb = out.shape[0]
    yb = out.shape[1] if y is not None else 1
    _abi.check(_abi.lib().trpl_posterior_hist_dev(
        _chk(x, torch.float64, "x"), None if y is None else _chk(y, torch.float64, "y"),
        None if W is None else _chk(W, torch.float64, "W"), x.shape[0], float(lo), float(hi), int(xb), float(ylo),
        float(yhi), int(yb), _chk(out, torch.float64, "out"), _stream()))


def credible_interval_device(x, W, lo=0.025, hi=0.975):
    """utils.py:185-196 on the device: sort by x, cumulate the weights, last point below `lo` and first
    above `hi` (torch.sort / cumsum: library plumbing, no custom kernel)."""
    import torch
    xs, order = torch.sort(x)
    cs = torch.cumsum(W[order], 0)
    below = torch.nonzero(cs < lo)
    above = torch.nonzero(cs > hi)
    return float(xs[below[-1, 0]]), float(xs[above[0, 0]])


def sample_box_device(X, minX, maxX, do_log, seed=42, flags=0):
    """Fill X (S, ncol) f64 in HBM with bayeslib.random_grid's draws (trpl_sample_box_dev)."""
    import torch
    lo = np.ascontiguousarray(minX, dtype=np.float64)
    hi = np.ascontiguousarray(maxX, dtype=np.float64)
    lg = np.ascontiguousarray(do_log, dtype=np.int32)
    if X.dim() != 2 or X.shape[1] != lo.size or not (lo.shape == hi.shape == lg.shape):
        raise ValueError("X must be (S, ncol) with ncol = len(minX) = len(maxX) = len(do_log)")
    _abi.check(_abi.lib().trpl_sample_box_dev(int(seed) & 0xFFFFFFFF, X.shape[0], lo.size, _abi.ptr(lo), _abi.ptr(hi),
                                              _abi.ptr(lg), int(flags), _chk(X, torch.float64, "X"), _stream()))


def loglik_from_pl_device(pl, obs, mag, P=None, sse=None, obs_hi=None, obs_dx=None, obs_h=None, ncol=None, flags=0,
                          status=None):
    """trpl_loglik_from_pl_dev: likelihood of PL rows resident in HBM (pl (rows, ld) f32/f64, as written by
    solve_pl_device) against one observation set obs (n_obs,) f64 -- on the grid, or off-grid with the
    bracketing arrays obs_hi (int32), obs_dx, obs_h of driver.bracket_times.  mag (rows,) f64 log offsets;
    P (rows,) is decremented in place and/or sse (rows,) receives the squared-error sums; status (rows,)
    int32 from solve_pl_device makes flagged systems score +inf."""
    import torch
    if pl.dim() != 2 or pl.dtype not in (torch.float32, torch.float64):
        raise ValueError("pl must be a 2-D float32/float64 tensor")
    rows, ld = pl.shape
    interp = obs_hi is not None
    _abi.check(_abi.lib().trpl_loglik_from_pl_dev(
        _chk(pl, pl.dtype, "pl"), pl.element_size(), rows, int(ld if ncol is None else ncol), ld,
        _chk(obs, torch.float64, "obs"), _chk(obs_hi, torch.int32, "obs_hi") if interp else None,
        _chk(obs_dx, torch.float64, "obs_dx") if interp else None, _chk(obs_h, torch.float64, "obs_h") if interp else None,
        obs.shape[0], _chk(mag, torch.float64, "mag"), None if status is None else _chk(status, torch.int32, "status"),
        None if P is None else _chk(P, torch.float64, "P"),
        None if sse is None else _chk(sse, torch.float64, "sse"), int(flags), _stream()))


def solve_pl_snap_device(matPar, Length, Time, L, T, dN, plI, snap_steps, plN=None, plP=None, plE=None, status=None,
                         iters_total=None, tol=7, MAX=10000, plT=1, flags=0):
    """trpl_solve_pl_snap_dev: solve_pl_device that also records the state at the time steps snap_steps
    (host sequence) into plN, plP (S, len(snap_steps), L) and plE (S, len(snap_steps), L+1), f64 tensors."""
    import torch
    S = matPar.shape[0]
    steps = np.ascontiguousarray(snap_steps, dtype=np.int64)
    n = len(steps)
    if matPar.shape[1] != 12 or tuple(dN.shape) != (L,) or tuple(plI.shape) != (S, T // plT + 1):
        raise ValueError("shape mismatch")
    for t, w in ((plN, L), (plP, L), (plE, L + 1)):
        if t is not None and tuple(t.shape) != (S, n, w):
            raise ValueError("snapshot tensors must be (S, len(snap_steps), L) / (.., L+1)")
    _abi.check(_abi.lib().trpl_solve_pl_snap_dev(
        _chk(matPar, torch.float64, "matPar"), S, float(Length), float(Time), int(L), int(T), int(plT), int(tol),
        int(MAX), _chk(dN, torch.float64, "dN"), _chk(plI, plI.dtype, "plI"), plI.element_size(), plI.shape[1],
        None if status is None else _chk(status, torch.int32, "status"),
        None if iters_total is None else _chk(iters_total, torch.int64, "iters_total"), _abi.ptr(steps), n,
        None if plN is None else _chk(plN, torch.float64, "plN"), None if plP is None else _chk(plP, torch.float64, "plP"),
        None if plE is None else _chk(plE, torch.float64, "plE"), int(flags), _stream()))


def solve_pl_resume_device(matPar, Length, Time, L, T, t0, resN, resP, resE, plI, snap_steps=(), plN=None, plP=None,
                           plE=None, status=None, iters_total=None, tol=7, MAX=10000, plT=1, flags=0):
    """trpl_solve_pl_resume_dev: continue at step t0 from the five raw time levels resN, resP (S, 5, L) and resE
    (S, 5, L+1), f64 tensors as recorded by solve_pl_snap_device(..., snap_steps=[t0-4 .. t0], flags=FLAG_SNAP_RAW)."""
    import torch
    S = matPar.shape[0]
    steps = np.ascontiguousarray(snap_steps, dtype=np.int64)
    n = len(steps)
    if matPar.shape[1] != 12 or tuple(plI.shape) != (S, T // plT + 1):
        raise ValueError("shape mismatch")
    for t, w in ((resN, L), (resP, L), (resE, L + 1)):
        if tuple(t.shape) != (S, 5, w):
            raise ValueError("resume tensors must be (S, 5, L) / (S, 5, L+1)")
    for t, w in ((plN, L), (plP, L), (plE, L + 1)):
        if t is not None and tuple(t.shape) != (S, n, w):
            raise ValueError("snapshot tensors must be (S, len(snap_steps), L) / (.., L+1)")
    _abi.check(_abi.lib().trpl_solve_pl_resume_dev(
        _chk(matPar, torch.float64, "matPar"), S, float(Length), float(Time), int(L), int(T), int(plT), int(tol),
        int(MAX), int(t0), _chk(resN, torch.float64, "resN"), _chk(resP, torch.float64, "resP"),
        _chk(resE, torch.float64, "resE"), _chk(plI, plI.dtype, "plI"), plI.element_size(), plI.shape[1],
        None if status is None else _chk(status, torch.int32, "status"),
        None if iters_total is None else _chk(iters_total, torch.int64, "iters_total"), _abi.ptr(steps) if n else None, n,
        None if plN is None else _chk(plN, torch.float64, "plN"), None if plP is None else _chk(plP, torch.float64, "plP"),
        None if plE is None else _chk(plE, torch.float64, "plE"), int(flags), _stream()))


class MultiDevice:
    """One process, several GPUs, results resident on every GPU (trpl_multi_* / trpl_loglik_multi_dev, SURVEY
    8e): contiguous sample shards, one RCCL all-gather of the per-sample likelihoods over xGMI.  The handle
    owns one stream and one RCCL rank per device; create it once and reuse it."""

    def __init__(self, devices=None, allow_duplicate_devices=False):
        """allow_duplicate_devices (tests): several ranks on one GPU, against a stand-in collective library named by
        TRPL_RCCL_LIBRARY (trpl_multi_create_ex, TRPL_MULTI_ALLOW_DUPLICATE_DEVICES); real RCCL refuses them."""
        self._h = _abi.C.c_void_p()
        dev = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        _abi.check(_abi.lib().trpl_multi_create_ex(_abi.ptr(dev), 0 if dev is None else len(dev),
                                                   _abi.MULTI_ALLOW_DUPLICATE_DEVICES if allow_duplicate_devices else 0,
                                                   _abi.C.byref(self._h)))
        self.n = int(_abi.lib().trpl_multi_device_count(self._h))
        self.devices = list(range(self.n)) if dev is None else [int(d) for d in dev]

    def close(self):
        if self._h:
            _abi.lib().trpl_multi_destroy(self._h)
            self._h = _abi.C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def synchronize(self):
        _abi.check(_abi.lib().trpl_multi_synchronize(self._h))

    def shard_bounds(self, S):
        from .dist import shard_bounds
        return [shard_bounds(S, self.n, r) for r in range(self.n)]

    def _table(self, tensors, dtype, name, optional=False):
        import torch
        if tensors is None:
            if optional:
                return None
            raise ValueError("%s: one tensor per device is required" % name)
        if len(tensors) != self.n:
            raise ValueError("%s: need %d per-device tensors" % (name, self.n))
        tab = (_abi.C.c_void_p * self.n)()
        for r, t in enumerate(tensors):
            if t is None or t.numel() == 0:
                tab[r] = None
                continue
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()
                    and t.device.index == self.devices[r]):
                raise ValueError("%s[%d] must be a contiguous %s tensor on cuda:%d" % (name, r, dtype, self.devices[r]))
            tab[r] = t.data_ptr()
        return tab

    def loglik(self, X, init_params, lengths, Time, L, T, obs, n_obs, P_full, sse=None, status=None, iters_total=None,
               obs_hi=None, obs_dx=None, obs_h=None, tol=7, MAX=10000, plT=1, flags=0, floor_col=None, order=True):
        """Enqueue the sharded fused likelihood + the all-gather; returns at once (synchronize() waits).
        X: list of per-device shards (n_r, 13); init_params / obs (/ obs_hi, obs_dx, obs_h): lists of per-device
        replicas; P_full: list of per-device (S,) f64 outputs; sse / status / iters_total / floor_col: optional
        lists of per-device (C, n_r) outputs.  S is taken from P_full.
        The handle works on its own streams.  With order=True (default) they are ordered on the device against
        torch's current stream of every device, both ways (trpl_multi_wait_stream before, trpl_multi_release_stream
        after): inputs just produced on the torch stream are complete when the solve reads them, and work given to
        the torch stream afterwards sees the gathered P_full -- no host wait.  order=False leaves both to the
        caller (synchronize())."""
        import torch
        S = int(P_full[0].shape[0])
        Cn = int(init_params[0].shape[0])
        bounds = self.shard_bounds(S)
        for r, (lo, hi) in enumerate(bounds):
            if tuple(X[r].shape) != (hi - lo, 13) or tuple(P_full[r].shape) != (S,) \
                    or tuple(init_params[r].shape) != (Cn, L) or obs[r].shape[0] != Cn:
                raise ValueError("rank %d: shapes do not match the shard [%d, %d) of S=%d" % (r, lo, hi, S))
            for name, lst, dt in (("sse", sse, torch.float64), ("status", status, torch.int32),
                                  ("iters_total", iters_total, torch.int64)):
                if lst is not None and tuple(lst[r].shape) != (Cn, hi - lo):
                    raise ValueError("%s[%d] must be (C, %d)" % (name, r, hi - lo))
            if floor_col is not None and tuple(floor_col[r].shape) != (Cn, hi - lo):
                raise ValueError("floor_col[%d] must be (C, %d)" % (r, hi - lo))
        lengths = np.ascontiguousarray(np.broadcast_to(np.asarray(lengths, dtype=np.float64), (Cn,)))
        n_obs = np.ascontiguousarray(np.broadcast_to(np.asarray(n_obs, dtype=np.int64), (Cn,)))
        streams = [torch.cuda.current_stream(torch.device("cuda", d)).cuda_stream for d in self.devices] if order else []
        for r, s in enumerate(streams):
            _abi.check(_abi.lib().trpl_multi_wait_stream(self._h, r, s))
        _abi.check(_abi.lib().trpl_loglik_multi_dev(
            self._h, self._table(X, torch.float64, "X"), S, Cn, _abi.ptr(lengths), float(Time), int(L), int(T), int(plT),
            int(tol), int(MAX), self._table(init_params, torch.float64, "init_params"),
            self._table(obs, torch.float64, "obs"), self._table(obs_hi, torch.int32, "obs_hi", True),
            self._table(obs_dx, torch.float64, "obs_dx", True), self._table(obs_h, torch.float64, "obs_h", True),
            int(obs[0].shape[1]), _abi.ptr(n_obs), self._table(P_full, torch.float64, "P_full"),
            self._table(sse, torch.float64, "sse", True), self._table(status, torch.int32, "status", True),
            self._table(iters_total, torch.int64, "iters_total", True),
            self._table(floor_col, torch.int32, "floor_col", True), int(flags)))
        for r, s in enumerate(streams):
            _abi.check(_abi.lib().trpl_multi_release_stream(self._h, r, s))
