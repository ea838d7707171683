"""Host-side logic of the product package that needs no GPU: sampler, time-grid mapping,
interpolation, sharding, and the control flow of simulate() with stand-in compute callables."""
import numpy as np
import pytest
from scipy.interpolate import griddata


def test_sampler_matches_reference_draws(trpl, golden):
    g = golden("sampler")
    assert np.array_equal(trpl.UNIT_CONVERSIONS, g["unit"])
    for S in (4, 64):
        assert np.array_equal(trpl.default_box(42, S), g[f"X{S}"])
    np.random.seed(42)                                   # legacy global state, like the reference
    X = trpl.random_grid(g["minX"] * g["unit"], g["maxX"] * g["unit"], g["do_log"], 4)
    assert np.array_equal(X, g["X4"])
    # SURVEY Appendix B result 2 (row 0 of the S=4 draw, common units)
    row0 = g["X4"][0] / g["unit"]
    assert np.allclose(row0[:5], [1e8, 5.61151642e14, 7.80093202, 30.0557506, 4.622589e-10], rtol=1e-8)


def test_make_grid_overrides(trpl):
    flags = {"random_sample": True, "num_points": 5, "override_equal_mu": True, "override_equal_s": True,
             "override_equal_auger": True}
    rng = np.random.RandomState(1)
    N, P, X = trpl.make_grid(2, trpl.DEFAULT_MINX, trpl.DEFAULT_MAXX, trpl.DEFAULT_DO_LOG, flags, rng=rng)
    assert P.shape == (2, 5) and not P.any() and len(N) == 5
    assert np.array_equal(X[:, 2], X[:, 3]) and np.array_equal(X[:, 6], X[:, 5]) and np.array_equal(X[:, 8], X[:, 7])
    with pytest.raises(NotImplementedError):
        trpl.make_grid(1, trpl.DEFAULT_MINX, trpl.DEFAULT_MAXX, trpl.DEFAULT_DO_LOG, {"random_sample": False})


def test_interp_rows_matches_scipy_griddata(trpl):
    rng = np.random.default_rng(0)
    sim_t = np.linspace(0, 10, 401)
    pl = rng.normal(size=(5, 401)).astype(np.float32)
    times = np.concatenate([sim_t[:50], rng.uniform(0, 10, 30), [10.0, 0.0]])
    got = trpl.interp_rows(sim_t, pl, times)
    for i in range(len(pl)):
        want = griddata(sim_t, pl[i], times)
        assert np.array_equal(got[i], want)
    out = trpl.interp_rows(sim_t, pl, np.array([-1.0, 11.0]))
    assert np.isnan(out).all()


def test_grid_prefix_and_almost_equal(trpl):
    sim_t = np.linspace(0, 2000, 80001)
    assert trpl.is_grid_prefix(sim_t[:5601], sim_t) and trpl.is_grid_prefix(sim_t, sim_t)
    assert not trpl.is_grid_prefix(sim_t[1:10], sim_t)
    assert not trpl.is_grid_prefix(sim_t[:10] * 1.001, sim_t)
    assert not trpl.is_grid_prefix(np.linspace(0, 1, 5), np.linspace(0, 1, 3))
    assert trpl.almost_equal(sim_t, sim_t.copy()) and not trpl.almost_equal(sim_t, sim_t[:-1])


def test_shard_bounds_partition(trpl):
    for S in (0, 1, 7, 64, 65536, 524288 + 3):
        for world in (1, 2, 3, 8):
            spans = [trpl.dist.shard_bounds(S, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == S
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        trpl.dist.shard_bounds(4, 2, 2)


def test_simulate_control_flow_with_standin_compute(trpl, monkeypatch):
    """simulate(): curve -> block -> experiment order, block striding over processes, float32 PL
    buffer, log + interpolation + offset, all checked with numpy stand-ins for the three device
    calls (the real ones are exercised by the -m gpu tests)."""
    drv = trpl.driver
    S, L, T, Time = 10, 8, 20, 2.0
    rng = np.random.default_rng(3)
    X = rng.uniform(1, 2, (S, 13))
    ini = rng.uniform(1, 2, (2, L))
    sim_t = np.linspace(0, Time, T + 1)
    calls = []

    def fake_model(plI, plN, plP, plE, matPar, simPar, iniPar, TPB, BPG, mspb, init_mode="exp"):
        assert plI.dtype == np.float32 and matPar.shape[1] == 12 and init_mode == "points"
        calls.append((simPar[0], len(matPar)))
        plI[:] = (matPar[:, :1] * iniPar.sum() * np.exp(-sim_t))[:, :T + 1]
        return 0.5

    def fake_fastlog(plI, MIN, device=0):
        plI[:] = np.log10(np.maximum(plI, MIN))
        return 0.25

    def fake_prob(P, plI, values, unc, mag, device=0):
        P -= np.sum((plI.astype(np.float64) + mag[:, None] - values) ** 2, axis=1)
        return 0.125

    monkeypatch.setattr(drv, "fastlog", fake_fastlog)
    monkeypatch.setattr(drv, "prob", fake_prob)
    e_data = [([sim_t, sim_t], [np.zeros(T + 1), np.ones(T + 1)]),
              ([sim_t[:7], sim_t[:5] + 0.01], [np.zeros(7), np.ones(5)])]
    flags = {"load_PL_from_file": False, "log_pl": True, "self_normalize": False}
    P_parts = []
    for gpu_id in range(2):
        P = np.zeros((2, S))
        st, et, mt = np.zeros(2), np.zeros(2), np.zeros(2)
        sim_params = [[100.0, 200.0], Time, L, T, 1, (0,), 7, 50]
        drv.simulate(fake_model, e_data, P, X, [None, None], [None, None], 2, sim_params, ini, flags,
                     {"sims_per_gpu": 3, "num_gpus": 2}, gpu_id, st, et, mt)
        P_parts.append(P)
        assert st[gpu_id] > 0 and et[gpu_id] > 0 and mt[gpu_id] > 0 and st[1 - gpu_id] == 0
    # process 0 takes blocks starting at 0 and 6, process 1 at 3 and 9 (bayeslib.py:131)
    assert np.all(P_parts[0][:, [3, 4, 5, 9]] == 0) and np.all(P_parts[0][:, [0, 1, 2, 6, 7, 8]] != 0)
    assert np.all(P_parts[1][:, [0, 1, 2, 6, 7, 8]] == 0) and np.all(P_parts[1][:, [3, 4, 5, 9]] != 0)
    assert calls[0] == (100.0, 3) and (200.0, 1) in calls
    # direct evaluation of the same quantity
    P = P_parts[0] + P_parts[1]
    want = np.zeros((2, S))
    for c in range(2):
        pl = (X[:, :1] * ini[c].sum() * np.exp(-sim_t)).astype(np.float32)
        lg = np.log10(pl)
        want[0] -= np.sum((lg.astype(np.float64) + X[:, -1:] - e_data[0][1][c]) ** 2, axis=1)
        t1 = e_data[1][0][c]
        want[1] -= np.sum((drv.interp_rows(sim_t, lg, t1) + X[:, -1:] - e_data[1][1][c]) ** 2, axis=1)
    assert np.allclose(P, want, rtol=1e-12)
    with pytest.raises(NotImplementedError):
        drv.simulate(fake_model, e_data, P, X, [None], [None], 2, sim_params, ini,
                     dict(flags, load_PL_from_file=True), {"sims_per_gpu": 3, "num_gpus": 1}, 0, st, et, mt)


def test_host_interpolation_entry_point_is_the_numpy_arithmetic_bit_for_bit(trpl):
    """trpl_interp_rows (plain host C++, no device) against the NumPy form it replaces in driver.interp_rows -- the reference's
    row-by-row griddata (bayeslib.py:184-191; interp1d's slope * (x - x_lo) + y_lo with the difference in the matrix's own
    dtype): float32 and float64 matrices, observation times that are grid nodes, off-grid, repeated, at both window edges and
    outside the window (NaN), a matrix whose rows are strided (ld > ncol), and the inputs the entry point does not take
    (integer matrix, empty time list) through the NumPy route.  Bit patterns, NaNs included."""
    import ctypes as C
    drv, A = trpl.driver, trpl._abi
    rng = np.random.default_rng(4)
    sim_t = np.linspace(0.0, 50.0, 2001)
    times_sets = [sim_t[:321], np.sort(rng.uniform(0.0, 50.0, 777)),
                  np.array([0.0, 0.0, 50.0, 49.999999, 0.0125, 0.0125, -1.0, 51.0, 25.0])]
    for dt in (np.float32, np.float64):
        wide = np.log10(rng.lognormal(-3, 2, (37, 2001 + 5))).astype(dt)
        for pl in (np.ascontiguousarray(wide[:, :2001]), wide[:, 3:2004]):          # contiguous, and row-strided (ld = 2006)
            for times in times_sets:
                got = drv.interp_rows(sim_t, pl, times)
                want = drv._interp_rows_numpy(sim_t, pl, times)
                want[:, (times < sim_t[0]) | (times > sim_t[-1])] = np.nan
                assert got.dtype == np.float64 and got.shape == want.shape
                assert np.array_equal(got.view(np.int64), want.view(np.int64)), (dt, len(times))
                assert np.isnan(got[:, times < 0]).all() and np.isnan(got[:, times > 50]).all()
    ints = rng.integers(0, 9, (4, 2001))
    assert np.array_equal(drv.interp_rows(sim_t, ints, sim_t[:5] + 0.001), drv._interp_rows_numpy(sim_t, ints, sim_t[:5] + 0.001))
    assert drv.interp_rows(sim_t, wide[:, :2001], np.zeros(0)).shape == (37, 0)
    # the entry point validates its brackets and shapes; nothing is written on a refusal
    lib = A.lib()
    pl = np.zeros((2, 10), dtype=np.float32); out = np.full((2, 3), 7.0)
    hi = np.array([1, 10, 2], dtype=np.int32); one = np.ones(3)
    assert lib.trpl_interp_rows(A.ptr(pl), 4, 2, 10, 10, A.ptr(hi), A.ptr(one), A.ptr(one), 3, A.ptr(out), 3) == A.ERR_ARG
    assert b"hi[1]" in lib.trpl_last_error() and (out == 7.0).all()
    assert lib.trpl_interp_rows(A.ptr(pl), 2, 2, 10, 10, A.ptr(hi), A.ptr(one), A.ptr(one), 3, A.ptr(out), 3) == A.ERR_ARG
    assert lib.trpl_interp_rows(A.ptr(pl), 4, 2, 10, 9, A.ptr(hi), A.ptr(one), A.ptr(one), 3, A.ptr(out), 3) == A.ERR_ARG
    assert lib.trpl_interp_rows(None, 4, 0, 10, 10, None, None, None, 3, None, 3) == A.OK            # empty batch


def test_fused_routing_rule_and_its_literal_switch(trpl):
    """driver.observations_on_grid / fused_entry_point: ONE rule for both fused branches of simulate() (one experiment per
    call, several experiments over a resident PL block).  Default: observation times that are a prefix of the simulation grid
    are compared on the grid (trpl_loglik); literal = gpu_info["interpolate_prefix"]: the reference's own test -- only the FULL
    grid bypasses the interpolation (bayeslib.py:173,:182-183), a prefix is interpolated like any other set (trpl_loglik_obs)."""
    drv = trpl.driver
    sim_t = np.linspace(0, 2.0, 81)
    full, prefix, off = sim_t, sim_t[:33], sim_t[:33] + 1e-3
    for times, default, literal in ((full, True, True), (prefix, True, False), (off, False, False), (sim_t[::2], False, False)):
        assert drv.observations_on_grid(times, sim_t) is default
        assert drv.observations_on_grid(times, sim_t, literal=True) is literal
    assert drv.fused_entry_point([prefix, full, prefix], sim_t, 3) == "trpl_loglik"
    assert drv.fused_entry_point([prefix, full, prefix], sim_t, 3, literal=True) == "trpl_loglik_obs"
    assert drv.fused_entry_point([full, full], sim_t, 2, literal=True) == "trpl_loglik"
    assert drv.fused_entry_point([prefix, off], sim_t, 2) == "trpl_loglik_obs"


def test_unfused_overlap_is_bounded_by_host_bytes(trpl, monkeypatch):
    """gpu_info["max_host_bytes"] bounds the PL results the unfused overlapped path holds at once (bayeslib.py:131-137 holds
    one; round 5's overlap held two blocks' curves whatever their size).  A re-entrant stand-in model counts the PL matrices
    in existence (in use or waiting in the path's reuse pool; not yet garbage-collected) at each of its calls; P is the same for
    every budget."""
    import gc
    import threading
    import weakref
    drv = trpl.driver
    S, L, T, Time, C = 24, 8, 40, 2.0, 3
    rng = np.random.default_rng(5)
    X = rng.uniform(1, 2, (S, 13))
    ini = rng.uniform(1, 2, (C, L))
    sim_t = np.linspace(0, Time, T + 1)
    lock = threading.Lock()
    state = {"peak": 0, "calls": 0}
    alive = {}                                                   # id -> True of every PL matrix that exists (in use or pooled)

    def model(plI, plN, plP, plE, matPar, simPar, iniPar, TPB, BPG, mspb, init_mode="exp"):
        gc.collect()
        key = id(plI)
        with lock:
            if key not in alive:                                 # a reused matrix is the same object: counted once
                alive[key] = True
                weakref.finalize(plI, alive.pop, key, None)
            state["calls"] += 1
            state["peak"] = max(state["peak"], len(alive))
        plI[:] = (matPar[:, :1] * iniPar.sum() * np.exp(-sim_t))[:, :T + 1]
        return 0.5
    model.reentrant = True

    def fake_fastlog(plI, MIN, device=0):
        plI[:] = np.log10(np.maximum(plI, MIN))
        return 0.25

    def fake_prob(P, plI, values, unc, mag, device=0):
        P -= np.sum((plI.astype(np.float64) + mag[:, None] - values) ** 2, axis=1)
        return 0.125

    monkeypatch.setattr(drv, "fastlog", fake_fastlog)
    monkeypatch.setattr(drv, "prob", fake_prob)
    e_data = [([sim_t] * C, [np.zeros(T + 1)] * C), ([sim_t[:9] + 0.01] * C, [np.ones(9)] * C)]
    flags = {"load_PL_from_file": False, "log_pl": True, "self_normalize": False}
    group = 4                                                    # 6 blocks x 3 curves = 18 results
    full, done = drv.unfused_curve_bytes(group, T + 1, np.float32, [0] * C + [9] * C, C)
    assert full == done == group * ((T + 1) * 4 + 9 * 8)         # experiment 0 reads the PL matrix itself: nothing to drop
    assert drv.overlap_window(full, done, 3 * full + full // 2, C) == (3, 2) and drv.overlap_window(full, done, full, C) == (0, 0)
    assert drv.overlap_window(full, done, drv.DEFAULT_MAX_HOST_BYTES, C) == (2 * C, C)
    results = {}

    def run(label, info, bound, data):
        gc.collect()
        state.update(peak=0, calls=0)
        P = np.zeros((2, S))
        st, et, mt = np.zeros(1), np.zeros(1), np.zeros(1)
        plI, plI_int = [None], [None]
        drv.simulate(model, data, P, X, plI, plI_int, C, [100.0, Time, L, T, 1, (0,), 7, 50], ini, flags,
                     dict({"sims_per_gpu": group, "num_gpus": 1}, **info), 0, st, et, mt)
        assert state["calls"] == 18 and state["peak"] <= bound, (label, state)
        assert plI[0].shape == (group, T + 1) and plI_int[0].shape == (group, 9)      # the last matrices stay, as in the reference
        results[label] = P
        peak = state["peak"]
        del plI, plI_int
        gc.collect()
        return peak

    run("serial", {"overlap_curves": False}, 1, e_data)
    run("one", {"max_host_bytes": full}, 1, e_data)
    run("three", {"max_host_bytes": 3 * full + full // 2}, 3, e_data)
    assert run("default", {}, 2 * C, e_data) > 3                 # the default budget does overlap two blocks
    for label in ("one", "three", "default"):
        assert np.array_equal(results[label], results["serial"]), label
    # every experiment off the grid (the production shape: observation times that are not the full grid): a result waiting for
    # prob() has let its PL matrix go -- only the worker threads' matrices (and the run's last one) are alive, whatever the window
    off = [([sim_t[:9] + 0.01] * C, [np.ones(9)] * C), ([sim_t[:7] + 0.02] * C, [np.zeros(7)] * C)]
    full2, done2 = drv.unfused_curve_bytes(group, T + 1, np.float32, [9] * C + [7] * C, C)
    assert full2 == group * ((T + 1) * 4 + 16 * 8) and done2 == group * 16 * 8
    assert drv.overlap_window(full2, done2, C * full2 + C * done2, C) == (2 * C, C)
    gc.collect()
    state.update(peak=0, calls=0)
    P = np.zeros((2, S))
    plI, plI_int = [None], [None]
    drv.simulate(model, off, P, X, plI, plI_int, C, [100.0, Time, L, T, 1, (0,), 7, 50], ini, flags,
                 {"sims_per_gpu": group, "num_gpus": 1}, 0, np.zeros(1), np.zeros(1), np.zeros(1))
    assert state["calls"] == 18 and state["peak"] <= C + 1, state
    assert plI[0] is not None and plI[0].shape == (group, T + 1) and plI_int[0].shape == (group, 7)
    Ps = np.zeros((2, S))
    drv.simulate(model, off, Ps, X, [None], [None], C, [100.0, Time, L, T, 1, (0,), 7, 50], ini, flags,
                 {"sims_per_gpu": group, "num_gpus": 1, "overlap_curves": False}, 0, np.zeros(1), np.zeros(1), np.zeros(1))
    assert np.array_equal(P, Ps)


def test_csv_ingestion_matches_reference(trpl, golden, tmp_path):
    """dataio.get_initpoints / get_data on data files cut from the reference's shipped examples
    against the arrays the reference's own bayes_io produced from them (bayes_io.py:15-119)."""
    import os
    from conftest import GOLDEN
    g = golden("bayes_realdata")
    ini = trpl.get_initpoints(os.path.join(GOLDEN, "exc_power_scan.csv"), {"select_obs_sets": None})
    assert np.array_equal(ini, g["ini"])
    assert np.allclose(ini, trpl.workloads.power_scan(128)[0], rtol=1e-8)      # analytic regeneration
    ic = {"time_cutoff": 5, "select_obs_sets": None, "noise_level": None}
    sf = {"log_pl": True, "self_normalize": False}
    e = trpl.get_data([os.path.join(GOLDEN, "obs_balanced_6ns.csv")], ic, sf, scale_f=1e-23)
    assert len(e) == 1 and len(e[0][0]) == 3
    for c in range(3):
        assert np.array_equal(e[0][0][c], g[f"t_0_{c}"]) and np.array_equal(e[0][1][c], g[f"v_0_{c}"])
        assert np.array_equal(e[0][2][c], g[f"u_0_{c}"])
    # options: curve selection, no cutoff (6 ns = 241 points), self-normalisation, linear PL
    e2 = trpl.get_data([os.path.join(GOLDEN, "obs_balanced_6ns.csv")],
                       {"time_cutoff": None, "select_obs_sets": [2, 0], "noise_level": None},
                       {"log_pl": False, "self_normalize": True})
    assert [len(t) for t in e2[0][0]] == [241, 241] and np.isclose(max(e2[0][1][0]), 1.0)
    assert np.array_equal(e2[0][0][1][:201], g["t_0_0"])
    P = np.arange(6.0).reshape(2, 3); X = np.ones((3, 13))
    trpl.export(str(tmp_path / "runA"), P[0], X)
    assert np.array_equal(np.load(tmp_path / "runA" / "runA_BAYRAN_P.npy"), P[0])
    assert np.array_equal(np.load(tmp_path / "runA" / "runA_BAYRAN_X.npy"), X)


def test_bracket_times_is_interp1d_bracketing(trpl):
    sim_t = np.linspace(0, 5, 201)
    rng = np.random.default_rng(2)
    times = np.sort(np.concatenate([[0.0, 5.0, sim_t[17], sim_t[118]], rng.uniform(0, 5, 40)]))
    hi, dx, h = trpl.bracket_times(sim_t, times)
    # float32 rows, as in the reference's buffer: scipy takes its generic linear path (for float64
    # 1-D rows it delegates to np.interp, which brackets on-grid points differently -- same value
    # up to rounding)
    y = rng.normal(size=201).astype(np.float32)
    want = griddata(sim_t, y, times)
    got = ((y[hi] - y[hi - 1]) / h) * dx + y[hi - 1]
    assert np.array_equal(got, want) and hi.min() >= 1 and hi.max() <= 200 and hi.dtype == np.int32
    y64 = y.astype(np.float64)
    assert np.allclose(((y64[hi] - y64[hi - 1]) / h) * dx + y64[hi - 1], griddata(sim_t, y64, times), rtol=0, atol=1e-15)


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start two ranks itself (fresh children under
    torch.distributed.run, 127.0.0.1, a free port) BEFORE importing torch or touching a GPU, and exit with their status --
    round 2's form spent 30 s on CPU baselines and then exited without a line.  On a host without a GPU the ranks
    themselves stop with "needs a GPU": both must have been started, the parent must relay their failure, and no CPU
    baseline may have run first (the whole thing takes seconds)."""
    import os
    import subprocess
    import sys
    import time
    import torch
    from conftest import ROOT
    if torch.cuda.is_available():
        pytest.skip("the GPU form of this launch is tests/test_gpu_multi.py::test_bench_two_rank_rehearsal_...")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["TRPL_AUTOBUILD"] = "0"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--warmup", "0", "--T", "50", "--samples-per-gpu", "64"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-3000:]          # both ranks got as far as the device check
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]            # and no line was invented
    assert time.time() - t0 < 120
