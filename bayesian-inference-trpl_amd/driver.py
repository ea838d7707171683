"""Host control loop: the semantics of bayeslib.simulate / bayes (bayeslib.py:83-252) over the
gfx950 kernels, plus the fused single-call likelihood the reference does not have.

Two paths compute the same P:
  * unfused (drop-in order of operations): model -> fastlog -> [time interpolation] -> prob per
    curve x sample block x experiment, exactly the reference's loop nest and dtypes;
  * fused (`loglik`): one launch per experiment, PL never leaves the registers; usable when
    every observation time grid is a prefix of the simulation grid (the shipped example data).
"""
import os
import sys
import time

import numpy as np

from . import _abi
from .likelihood import fastlog, prob
from .model import pvSim
from .sampler import make_grid


def almost_equal(x, x0, threshold=1e-10):
    """bayeslib.almost_equal (bayeslib.py:78-81)."""
    x = np.asarray(x)
    x0 = np.asarray(x0)
    if x.shape != x0.shape:
        return False
    with np.errstate(divide="ignore", invalid="ignore"):
        return bool(np.abs(np.nanmax((x - x0) / x0)) < threshold)


def is_grid_prefix(times, sim_t, threshold=1e-10):
    """True when `times` coincides with the first len(times) points of the simulation grid."""
    times = np.asarray(times, dtype=float)
    n = len(times)
    if n < 1 or n > len(sim_t):
        return False
    ref = sim_t[:n]
    scale = np.maximum(np.abs(ref), np.abs(sim_t[1]) if len(sim_t) > 1 else 1.0)
    return bool(np.all(np.abs(times - ref) <= threshold * scale))


def observations_on_grid(times, sim_t, literal=False):
    """How the fused level routes one curve's observation times.  True: compared point by point with the first
    len(times) simulation columns (entry points trpl_loglik / the on-grid form of trpl_loglik_from_pl_dev); False:
    interpolated in the kernel (trpl_loglik_obs / the bracket form).  Default rule: a PREFIX of the simulation grid is on
    the grid.  literal (gpu_info["interpolate_prefix"]): the reference's own test -- only the FULL grid bypasses the
    interpolation (bayeslib.py:173,:182-183), a prefix goes through it like any other set of times (:184-191)."""
    if literal:
        return almost_equal(sim_t, np.asarray(times, dtype=float))
    return is_grid_prefix(times, sim_t)


def fused_entry_point(exp_times, sim_t, num_curves, literal=False):
    """Name of the C entry point driver.simulate's fused level calls for one experiment (single-experiment branch): every
    curve on the grid -> "trpl_loglik", else "trpl_loglik_obs".  bench.py labels its e2e record with it."""
    on = all(observations_on_grid(exp_times[c], sim_t, literal) for c in range(num_curves))
    return "trpl_loglik" if on else "trpl_loglik_obs"


def _interp_rows_numpy(sim_t, pl, times):
    """interp_rows in NumPy: the arithmetic trpl_interp_rows restates (and the route for matrices it does not take)."""
    hi = np.clip(np.searchsorted(sim_t, times), 1, len(sim_t) - 1)
    lo = hi - 1
    slope = (pl[:, hi] - pl[:, lo]) / (sim_t[hi] - sim_t[lo])[None, :]
    return np.ascontiguousarray(slope * (times - sim_t[lo])[None, :] + pl[:, lo], dtype=np.float64)


def interp_rows(sim_t, pl, times):
    """1-D linear interpolation of every row of `pl` from sim_t onto `times`; the arithmetic of scipy's interp1d / griddata,
    which the reference applies row by row (bayeslib.py:186-189): slope = (y_hi - y_lo) / (x_hi - x_lo) with the difference
    taken in pl's own dtype (float32 for the reference's buffer), then slope * (x - x_lo) + y_lo in float64; NaN outside the
    grid.  Returns float64.  Row-contiguous float32 / float64 matrices go through trpl_interp_rows -- the same operations in
    plain host C++, bit for bit, without the interpreter lock, so the worker threads of simulate() interpolate side by side."""
    sim_t = np.asarray(sim_t, dtype=float)
    times = np.asarray(times, dtype=float)
    if (isinstance(pl, np.ndarray) and pl.ndim == 2 and pl.dtype in (np.float32, np.float64) and pl.shape[0] > 0
            and pl.strides[1] == pl.itemsize and pl.strides[0] % pl.itemsize == 0 and pl.strides[0] >= pl.shape[1] * pl.itemsize
            and pl.shape[1] == len(sim_t) and len(sim_t) >= 2 and len(times) > 0):
        hi, dx, h = bracket_times(sim_t, times)
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        dx = np.ascontiguousarray(dx, dtype=np.float64)
        h = np.ascontiguousarray(h, dtype=np.float64)
        out = np.empty((pl.shape[0], len(times)), dtype=np.float64)
        _abi.check(_abi.lib().trpl_interp_rows(_abi.ptr(pl), pl.itemsize, pl.shape[0], pl.shape[1], pl.strides[0] // pl.itemsize,
                                               _abi.ptr(hi), _abi.ptr(dx), _abi.ptr(h), len(times), _abi.ptr(out), out.shape[1]))
    else:
        out = _interp_rows_numpy(sim_t, pl, times)
    outside = (times < sim_t[0]) | (times > sim_t[-1])
    out[:, outside] = np.nan
    return out


def bracket_times(sim_t, times):
    """The bracketing scipy's interp1d uses for linear interpolation (and griddata with it,
    bayeslib.py:189): hi = clip(searchsorted(x, x_new), 1, n-1), lo = hi - 1.  Returns
    (hi int32, x_new - x_lo, x_hi - x_lo)."""
    sim_t = np.asarray(sim_t, dtype=float)
    times = np.asarray(times, dtype=float)
    hi = np.clip(np.searchsorted(sim_t, times), 1, len(sim_t) - 1)
    lo = hi - 1
    return hi.astype(np.int32), times - sim_t[lo], sim_t[hi] - sim_t[lo]


DEFAULT_MAX_HOST_BYTES = 32 << 30      # gpu_info["max_host_bytes"]: what the results of the unfused overlapped path may hold at once


def unfused_curve_bytes(size, ncol, pl_dtype, n_interp, num_curves):
    """Host bytes of ONE (block, curve) result of the unfused path, as (full, done).  full: while it is being made -- the
    block's PL matrix (size x ncol in the caller's PL dtype, bayeslib.py:137) + a float64 interpolated copy (size x n_obs) per
    experiment whose times are off the grid (:184-191).  done: once it waits for prob() -- the same, minus the PL matrix when
    no experiment compares this curve on the grid (the matrix is then dropped as soon as it has been interpolated).
    n_interp: the n_obs of every (experiment, curve) that is interpolated, 0 for those compared on the grid,
    experiment-major; the largest curve counts."""
    n_interp = np.asarray(n_interp, dtype=np.int64).reshape(-1, max(int(num_curves), 1))
    pl = int(ncol) * np.dtype(pl_dtype).itemsize
    if not n_interp.size:
        return int(size) * pl, int(size) * pl
    interp = 8 * n_interp.sum(axis=0)                                     # per curve
    keeps_pl = (n_interp == 0).any(axis=0)                                # some experiment reads the PL matrix itself
    full = int(size) * int((pl + interp).max())
    done = int(size) * int(np.where(keeps_pl, pl + interp, interp).max())
    return full, done


def overlap_window(full, done, budget, num_curves):
    """How many (block, curve) results the overlapped unfused path may hold at once -- being made by a worker thread, waiting
    for prob(), or being consumed -- and how many worker threads make them: the largest window <= 2 * num_curves (this block's
    curves + the next block's) whose bytes, workers * full + (window - workers) * done, fit the budget.  (0, 0): not even two
    results fit -- the path then runs inline, one result at a time, like the reference (bayeslib.py:137)."""
    for w in range(2 * int(num_curves), 1, -1):
        workers = min(int(num_curves), 8, w - 1)
        if workers * full + (w - workers) * done <= budget:
            return w, workers
    return 0, 0


def loglik(X, init_params, lengths, Time, L, T, obs, tol=7, MAX=10000, plT=1, P=None, pl_f32=False,
           normalize=False, strict=False, device=0, info=None, times=None, fp32=False, devices=None, kernel=None,
           mixed=False, bundle=1, hist32=False, bdf_order=None, extra_flags=0):
    """Fused likelihood of one experiment (trpl_loglik / trpl_loglik_obs / trpl_loglik_multi).

    X (S,13) solver units; init_params (C,L) nm^-3; lengths scalar or (C,); obs = list of C
    arrays of log10 observations.  With times=None they sit on the first len(obs[c])
    simulation-grid points; otherwise times = list of C arrays of observation times in
    [0, Time] (any spacing), interpolated like the reference does (bayeslib.py:184-191).
    Accumulates into P (S,) if given (like probs.prob), else starts from zeros.  Returns P.
    devices: None = the single `device`; "all" = every visible device; or a list of device ordinals
    (contiguous sample shards, one per entry, run concurrently from this host thread).
    kernel: None (the library picks the stepper by launch size), "pair" or "single" (TRPL_FLAG_KERNEL_*).
    bundle: the reference's max_sims_per_block (TRPL_FLAG_BUNDLE; single-device calls only).
    bdf_order: cap the BDF order ramp at 1 .. 5 (TRPL_FLAG_BDF_ORDER); extra_flags: further TRPL_FLAG_* bits, ORed in
    (tests / measurements: _abi.FLAG_PAIR_ALWAYS_SEAM, _abi.FLAG_PAIR_ADJACENT, ...).
    """
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim != 2 or X.shape[1] != 13:
        raise ValueError("X must have shape (S, 13)")
    S = X.shape[0]
    ini = np.ascontiguousarray(init_params, dtype=np.float64)
    if ini.ndim != 2 or ini.shape[1] != L:
        raise ValueError("init_params must have shape (C, L=%d), got %r" % (L, ini.shape))
    Cn = ini.shape[0]
    lengths = np.full(Cn, float(lengths)) if np.isscalar(lengths) else np.ascontiguousarray(lengths, dtype=float)
    if lengths.shape != (Cn,) or len(obs) != Cn or (times is not None and len(times) != Cn):
        raise ValueError("need one length and one observation set per curve")
    n_obs = np.array([len(o) for o in obs], dtype=np.int64)
    obs_ld = int(n_obs.max())
    obs_mat = np.zeros((Cn, obs_ld))
    hi_mat = np.ones((Cn, obs_ld), dtype=np.int32)
    dx_mat = np.zeros((Cn, obs_ld))
    h_mat = np.ones((Cn, obs_ld))
    if times is not None:
        if plT != 1:
            raise ValueError("off-grid observation times need plT = 1")
        sim_t = np.linspace(0, Time, T + 1)                                   # bayeslib.py:115
    for c, o in enumerate(obs):
        o = np.asarray(o, dtype=float)
        if times is not None:
            tc = np.asarray(times[c], dtype=float)
            if tc.shape != o.shape:
                raise ValueError("curve %d: %d times for %d observations" % (c, len(tc), len(o)))
            if len(tc) and (tc.min() < sim_t[0] or tc.max() > sim_t[-1]):
                raise ValueError("curve %d: observation times outside [0, Time] (the reference would "
                                 "interpolate them to NaN)" % c)
            order = np.argsort(tc, kind="stable")
            tc, o = tc[order], o[order]
            hi_mat[c, :len(o)], dx_mat[c, :len(o)], h_mat[c, :len(o)] = bracket_times(sim_t, tc)
        obs_mat[c, :len(o)] = o
    if P is None:
        P = np.zeros(S)
    if not (P.dtype == np.float64 and P.flags.c_contiguous and P.shape == (S,)):
        raise ValueError("P must be a contiguous float64 array of shape (S,)")
    sse = np.zeros((Cn, S))
    status = np.zeros((Cn, S), dtype=np.int32)
    iters = np.zeros((Cn, S), dtype=np.int64)
    floor_col = np.full((Cn, S), -1, dtype=np.int32)
    flags = (_abi.FLAG_STRICT if strict else 0) | (_abi.FLAG_PL_F32 if pl_f32 else 0) \
        | (_abi.FLAG_NORMALIZE if normalize else 0) | _abi.fp32_flags(fp32) | _abi.kernel_flag(kernel) \
        | (_abi.FLAG_MIXED if mixed else 0) | _abi.flag_bundle(bundle, L) | (_abi.FLAG_HIST32 if hist32 else 0) \
        | _abi.flag_bdf_order(bdf_order) | int(extra_flags)
    sec = _abi.C.c_double(0.0)
    lib = _abi.lib()
    if devices is not None:
        dev = None if isinstance(devices, str) else np.ascontiguousarray(devices, dtype=np.int32)
        if isinstance(devices, str) and devices != "all":
            raise ValueError("devices must be None, 'all' or a list of device ordinals")
        if dev is not None and (dev.ndim != 1 or len(dev) < 1):
            raise ValueError("devices must name at least one device")
        off = times is not None
        rc = lib.trpl_loglik_multi(_abi.ptr(X), S, Cn, _abi.ptr(lengths), float(Time), int(L), int(T), int(plT),
                                   int(tol), int(MAX), _abi.ptr(ini), _abi.ptr(obs_mat),
                                   _abi.ptr(hi_mat) if off else None, _abi.ptr(dx_mat) if off else None,
                                   _abi.ptr(h_mat) if off else None, obs_ld, _abi.ptr(n_obs), _abi.ptr(P),
                                   _abi.ptr(sse), _abi.ptr(status), _abi.ptr(iters), _abi.ptr(floor_col), flags,
                                   None if dev is None else _abi.ptr(dev), 0 if dev is None else len(dev),
                                   _abi.C.byref(sec))
    elif times is None:
        rc = lib.trpl_loglik(_abi.ptr(X), S, Cn, _abi.ptr(lengths), float(Time), int(L), int(T), int(plT), int(tol),
                             int(MAX), _abi.ptr(ini), _abi.ptr(obs_mat), obs_ld, _abi.ptr(n_obs), _abi.ptr(P),
                             _abi.ptr(sse), _abi.ptr(status), _abi.ptr(iters), _abi.ptr(floor_col), flags, int(device),
                             _abi.C.byref(sec))
    else:
        rc = lib.trpl_loglik_obs(_abi.ptr(X), S, Cn, _abi.ptr(lengths), float(Time), int(L), int(T), int(tol), int(MAX),
                                 _abi.ptr(ini), _abi.ptr(obs_mat), _abi.ptr(hi_mat), _abi.ptr(dx_mat), _abi.ptr(h_mat),
                                 obs_ld, _abi.ptr(n_obs), _abi.ptr(P), _abi.ptr(sse), _abi.ptr(status),
                                 _abi.ptr(iters), _abi.ptr(floor_col), flags, int(device), _abi.C.byref(sec))
    _abi.check(rc)
    if info is not None:
        info.update(sse=sse, status=status, iters_total=iters, floor_col=floor_col, seconds=sec.value)
    return P


def _bundle_of(gpu_info, L):
    """gpu_info['max_sims_per_block'] (bayeslib.py:93) as a TRPL_FLAG_BUNDLE size: 2 .. 4 at L = 128 and 2 .. 16 on
    grids of up to 64 nodes are honoured by the default arithmetic, anything else means every sample on its own
    (model.pvSim applies the same rule)."""
    m = int(gpu_info.get("max_sims_per_block", 1))
    return m if (1 <= m <= _abi.bundle_cap(L) and int(L) <= 128 and gpu_info.get("devices") is None) else 1


def _simulate_resident(e_data, P, X, num_curves, thicknesses, sim_params, init_params, normalize, pl_dtype, group,
                       num_gpus, gpu_id, device, solver_time, err_sq_time, sim_t, bundle=1, literal=False):
    """Several experiments, fused option on: the reference's own loop order -- curves -> sample blocks ->
    experiments (bayeslib.py:117-171) -- with the block's PL matrix kept in HBM: one solve per (curve, block)
    (trpl_solve_pl_dev), then one pass over it per experiment (trpl_loglik_from_pl_dev: normalise, clamp,
    log10, time interpolation, squared error).  Only X goes in and P comes out.  literal: observations_on_grid's switch
    (gpu_info["interpolate_prefix"])."""
    import time

    import torch

    from . import device as tdev
    dev = torch.device("cuda", device)
    Time, L, T = sim_params[1], sim_params[2], sim_params[3]
    tdt = torch.float32 if pl_dtype == np.float32 else torch.float64
    flags = _abi.FLAG_NORMALIZE if normalize else 0
    with torch.cuda.device(dev):
        ini_d = torch.from_numpy(np.ascontiguousarray(init_params, dtype=np.float64)).to(dev)
        # per (experiment, curve): observations and, off the grid, their brackets -- staged once
        staged = []
        for exp in e_data:
            per_curve = []
            for c in range(num_curves):
                t = np.asarray(exp[0][c], dtype=float)
                o = np.asarray(exp[1][c], dtype=float)
                if observations_on_grid(t, sim_t, literal):                   # bayeslib.py:182-183, and prefixes of the grid (below)
                    per_curve.append((torch.from_numpy(np.ascontiguousarray(o)).to(dev), None))
                else:
                    order = np.argsort(t, kind="stable")
                    hi, dx, h = bracket_times(sim_t, t[order])
                    per_curve.append((torch.from_numpy(np.ascontiguousarray(o[order])).to(dev),
                                      tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (hi, dx, h))))
            staged.append(per_curve)
        for blk in range(gpu_id * group, len(X), num_gpus * group):           # :131
            size = min(group, len(X) - blk)
            X_d = torch.from_numpy(np.ascontiguousarray(X[blk:blk + size], dtype=np.float64)).to(dev)
            mat_d, mag_d = X_d[:, :12].contiguous(), X_d[:, 12].contiguous()   # :144,:195
            P_d = torch.from_numpy(np.ascontiguousarray(P[:, blk:blk + size])).to(dev)
            pl_d = torch.empty((size, T // sim_params[4] + 1), dtype=tdt, device=dev)     # :137
            st_d = torch.empty(size, dtype=torch.int32, device=dev)
            for c in range(num_curves):                                       # :117
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                tdev.solve_pl_device(mat_d, thicknesses[c], Time, L, T, ini_d[c].contiguous(), pl_d, status=st_d,
                                     tol=sim_params[6], MAX=sim_params[7], plT=sim_params[4],
                                     flags=_abi.flag_bundle(bundle, L))
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for e, per_curve in enumerate(staged):                        # :171
                    o_d, br = per_curve[c]
                    tdev.loglik_from_pl_device(pl_d, o_d, mag_d, P=P_d[e], flags=flags, status=st_d,
                                               obs_hi=None if br is None else br[0],
                                               obs_dx=None if br is None else br[1],
                                               obs_h=None if br is None else br[2])
                torch.cuda.synchronize(dev)
                solver_time[gpu_id] += t1 - t0
                err_sq_time[gpu_id] += time.perf_counter() - t1
            P[:, blk:blk + size] = P_d.cpu().numpy()


def simulate(model, e_data, P, X, plI, plI_int, num_curves, sim_params, init_params, sim_flags, gpu_info, gpu_id,
             solver_time, err_sq_time, misc_time, logger=None):
    """bayeslib.simulate (bayeslib.py:83-205), GPU branch, same arguments and in-place effects.

    Loop order curves -> sample blocks of gpu_info['sims_per_gpu'] (blocks gpu_id, gpu_id +
    num_gpus, ...) -> experiments; float32 PL buffer (:137); X[:, :-1] to the model and
    X[:, -1] as the log offset (:144,:195).  With gpu_info['fused'] true and every observation
    grid a prefix of the simulation grid, each (curve-set, block, experiment) is one fused launch
    (spread over gpu_info['devices'] when that is given, see loglik).
    """
    group = int(gpu_info["sims_per_gpu"])
    num_gpus = int(gpu_info["num_gpus"])
    device = int(gpu_info.get("device", 0))
    LOG_PL = sim_flags["log_pl"]
    NORMALIZE = sim_flags["self_normalize"]
    if sim_flags.get("load_PL_from_file"):
        raise NotImplementedError("load PL not implemented")              # bayeslib.py:163-164
    if isinstance(sim_params[0], (int, float)):                           # :109-112
        thicknesses = [sim_params[0]] * num_curves
    else:
        thicknesses = list(sim_params[0])
    Time, L, T = sim_params[1], sim_params[2], sim_params[3]
    sim_t = np.linspace(0, Time, T + 1)                                   # :115
    pl_dtype = np.dtype(gpu_info.get("pl_dtype", np.float32))

    def in_range(t):
        t = np.asarray(t, dtype=float)
        return len(t) > 0 and t.min() >= sim_t[0] and t.max() <= sim_t[-1]

    fused = bool(gpu_info.get("fused", False)) and LOG_PL and sim_params[4] == 1 and all(
        in_range(exp[0][c]) for exp in e_data for c in range(num_curves))
    if fused and len(e_data) > 1 and gpu_info.get("devices") is None:
        _simulate_resident(e_data, P, X, num_curves, thicknesses, sim_params, init_params, NORMALIZE, pl_dtype,
                           group, num_gpus, gpu_id, device, solver_time, err_sq_time, sim_t,
                           bundle=_bundle_of(gpu_info, L), literal=bool(gpu_info.get("interpolate_prefix", False)))
        return
    if fused:
        # An experiment sampled exactly on the full simulation grid is compared point by point (the reference's bypass,
        # bayeslib.py:182-183); anything else is interpolated there (:184-191).  Observation times that are a PREFIX of the
        # simulation grid -- the shipped example data: 0.025 ns spacing, 140 .. 320 ns of a 2000 ns window; the reference's
        # shape test sends them to griddata -- are interpolated AT grid nodes, where the interp1d form returns the node's own
        # value to one rounding: they take the on-grid entry point too (batched emission, curve-pair table: 21 % faster on
        # the production shape, likelihoods equal to 7e-15, tools/bench_prefix_vs_interp.py); gpu_info["interpolate_prefix"]
        # = True keeps the literal interpolation (both fused branches: this one and _simulate_resident).
        literal = bool(gpu_info.get("interpolate_prefix", False))
        for blk in range(gpu_id * group, len(X), num_gpus * group):
            size = min(group, len(X) - blk)
            for e, exp in enumerate(e_data):
                on_grid = fused_entry_point(exp[0], sim_t, num_curves, literal) == "trpl_loglik"
                info = {}
                loglik(X[blk:blk + size], init_params, thicknesses, Time, L, T,
                       [exp[1][c] for c in range(num_curves)], tol=sim_params[6], MAX=sim_params[7],
                       P=P[e, blk:blk + size], pl_f32=(pl_dtype == np.float32), normalize=NORMALIZE,
                       device=device, info=info, devices=gpu_info.get("devices"), bundle=_bundle_of(gpu_info, L),
                       times=None if on_grid else [exp[0][c] for c in range(num_curves)])
                solver_time[gpu_id] += info["seconds"]
        return

    # The reference walks curves -> blocks (:117,:131).  A sample only ever meets its own block, and every
    # P[e, j] receives its curves in the same order either way, so the loops can be swapped: blocks outside,
    # and -- when the model is this package's own (re-entrant: every call runs on a private stream) -- the
    # block's curves solved concurrently from a few host threads, which is what fills the chip when the
    # blocks are as small as the reference's default 1024 samples.  Since round 5 a worker also does its curve's
    # normalisation, fastlog and time interpolation (the calls release the GIL), and the next block's curves are
    # submitted while this thread accumulates the current block's squared errors with prob() in curve order: the
    # same calls on the same data in the same order per P[e, j], overlapped (production shape, sims_per_gpu 1024:
    # 147 s -> see DESIGN.md section 5).
    overlap = (model is pvSim or getattr(model, "reentrant", False)) and bool(gpu_info.get("overlap_curves", True)) \
        and num_curves > 1
    ncol = T // sim_params[4] + 1
    obs_times = [[np.asarray(exp[0][c], dtype=float) for c in range(num_curves)] for exp in e_data]
    on_grid = [[almost_equal(sim_t, t) for t in per_curve] for per_curve in obs_times]        # :173,:182-183

    # gpu_info["kernel"] = "pair" / "single" pins the FAST stepper of every pvSim launch (measurements; default None: the
    # library picks per launch -- the one-system kernel for the reference's 1024-sample blocks.  Pinning the paired kernel for
    # the overlapped launches was measured and rejected: three worker threads keep 3 x 1024 systems in flight, which under-fills
    # it -- production shape 126 s against 119 s, profiles/r6_e2e_production_ab.json)
    model_kw = {}
    if model is pvSim and gpu_info.get("kernel") is not None and int(gpu_info.get("max_sims_per_block", 1)) == 1:
        model_kw["kernel"] = gpu_info["kernel"]
    reads_pl = [any(on_grid[e][c] for e in range(len(e_data))) for c in range(num_curves)]

    def process_curve(ic_num, blk, size, last=True):
        par = list(sim_params)
        par[0] = thicknesses[ic_num]                                      # :119
        # a FRESH matrix per task, like the reference (:137).  Reusing the matrices of consumed results was measured and
        # rejected: at sims_per_gpu = 1024 the reused address ranges -- page-locked and released again by every pvSim and
        # fastlog call -- serialise the worker threads' launches (production shape / 4: 33 s against 27.5 s, shared or
        # per-thread reuse alike; profiles/r6_levelb_task_phases.txt)
        buf = np.empty((size, ncol), dtype=pl_dtype)
        sec = model(buf, None, None, None, X[blk:blk + size, :-1], par, init_params[ic_num], None, None,
                    int(gpu_info.get("max_sims_per_block", 1)), init_mode="points", **model_kw)      # :93,:146
        misc = 0.0
        if NORMALIZE:                                                     # :150-154
            buf /= buf[:, :1].copy()
        if LOG_PL:                                                        # :155-157
            misc += fastlog(buf, sys.float_info.min, device=device)
        ints = []
        for e in range(len(e_data)):                                      # :168
            if on_grid[e][ic_num]:
                ints.append(buf)
            else:                                                         # :184-191
                clock0 = time.perf_counter()
                ints.append(interp_rows(sim_t, buf, obs_times[e][ic_num]))
                misc += time.perf_counter() - clock0
        # the PL matrix itself is only read again when an experiment compares this curve on the grid; otherwise it is let go
        # here, while the result waits for prob() (the run's LAST result keeps it: plI[gpu_id] ends up holding the last PL
        # matrix, as in the reference)
        return (buf if (reads_pl[ic_num] or last) else None), ints, sec, misc

    blocks = list(range(gpu_id * group, len(X), num_gpus * group))        # :131
    tasks = [(blk, min(group, len(X) - blk), c) for blk in blocks for c in range(num_curves)]   # the order P receives them in
    # Host memory of the overlapped path is bounded by BYTES (gpu_info["max_host_bytes"], default 32 GiB), not by a block
    # count: overlap_window() picks how many results may exist at once -- at most two blocks' curves, the round-5 schedule --
    # and how many worker threads make them.
    full, done = unfused_curve_bytes(min(group, len(X)) if len(X) else 0, ncol, pl_dtype,
                                     [0 if on_grid[e][c] else len(obs_times[e][c]) for e in range(len(e_data))
                                      for c in range(num_curves)], num_curves)
    window, workers = overlap_window(full, done, int(gpu_info.get("max_host_bytes", DEFAULT_MAX_HOST_BYTES)), num_curves)
    if window < 2:
        overlap = False
    pool = None
    if overlap:
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=workers)
    try:
        nxt, queue = 0, None

        def submit_next():
            b2, s2, c2 = tasks[nxt]
            queue.append(pool.submit(process_curve, c2, b2, s2, nxt == len(tasks) - 1))

        if overlap:
            queue = deque()
            while nxt < len(tasks) and len(queue) < window - 1:          # + the result being consumed = window
                submit_next()
                nxt += 1
        for blk, size, ic_num in tasks:
            if ic_num == 0:
                if logger is not None:
                    logger.info("Calculating {} of {}".format(blk, len(X)))
                mag = np.ascontiguousarray(X[blk:blk + size, -1])
            sim_params[0] = thicknesses[ic_num]                           # :119 (the caller's list is mutated, as there)
            if overlap:
                res = queue.popleft().result()
            else:
                plI[gpu_id] = plI_int[gpu_id] = None                      # one result at a time
                res = process_curve(ic_num, blk, size)
            plI[gpu_id], ints, sec, misc = res
            del res
            solver_time[gpu_id] += sec
            misc_time[gpu_id] += misc
            for e, exp in enumerate(e_data):
                plI_int[gpu_id] = ints[e]
                err_sq_time[gpu_id] += prob(P[e, blk:blk + size], ints[e], exp[1][ic_num], None, mag, device=device)
            del ints                                                      # plI_int[gpu_id] keeps the last matrix, as there
            if overlap and nxt < len(tasks):                              # the previous result is gone: the window has room
                submit_next()
                nxt += 1
    finally:
        if pool is not None:
            pool.shutdown(wait=True)


def bayes(model, N, P, minX, maxX, do_log, init_params, sim_params, e_data, sim_flags, gpu_info, logger=None,
          rng=None):
    """bayeslib.bayes (bayeslib.py:207-252): sample the box, run simulate() for this process's
    share of the sample blocks, return (N, P, X).  The process's block index comes from
    SLURM_ARRAY_TASK_ID as in the reference (:231), falling back to RANK, then 0."""
    num_gpus = int(gpu_info["num_gpus"])
    solver_time, err_sq_time, misc_time = np.zeros(num_gpus), np.zeros(num_gpus), np.zeros(num_gpus)
    N, P, X = make_grid(len(e_data), minX, maxX, do_log, sim_flags, rng=rng)
    gpu_id = int(os.getenv("SLURM_ARRAY_TASK_ID", os.getenv("RANK", "0")))
    if not 0 <= gpu_id < num_gpus:
        raise ValueError("process index %d outside num_gpus=%d" % (gpu_id, num_gpus))
    plI, plI_int = [None] * num_gpus, [None] * num_gpus
    simulate(model if model is not None else pvSim, e_data, P, X, plI, plI_int, len(init_params),
             list(sim_params), init_params, sim_flags, gpu_info, gpu_id, solver_time, err_sq_time, misc_time,
             logger=logger)
    if logger is not None:
        logger.info("Total tEvol time: {}".format(solver_time))
        logger.info("Total err_sq time: {}".format(err_sq_time))
        logger.info("Total misc time: {}".format(misc_time))
    return N, P, X
