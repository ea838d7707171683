"""The likelihood side of the path (a-7 `fastlog`, a-8 `prob`, a-10 `simulate`, f-1 ingestion and observation-time mapping):
the stand-alone kernels against the reference's goldens (prob bit-exact, log10 to 1 ulp), the unfused drop-in sequence and the
fused trpl_loglik / trpl_loglik_obs / trpl_loglik_from_pl_dev against bayeslib.bayes's golden likelihoods and the oracle's
restatement of bayeslib.simulate -- on-grid and off-grid observation times, self-normalisation, the reference's float32 PL
staging, real example data, more than sixteen curves per call."""
import os

import numpy as np
import pytest

from gpu_common import record, DT, nthreads

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------- probs
def test_fastlog_and_prob_vs_reference_golden(gpu, golden):
    g = golden("probs")
    l64 = g["pl64"].copy()
    assert gpu.fastlog(l64, float(g["MIN"]), 128, 256) > 0
    assert np.max(np.abs(l64 - g["log64"])) <= 2e-16 * np.max(np.abs(g["log64"]))
    l32 = g["pl32"].copy()
    gpu.fastlog(l32, float(g["MIN"]))
    assert l32.dtype == np.float32 and np.max(np.abs(l32 - g["log32"]) / np.abs(g["log32"])) <= 2.0 ** -23
    P = g["P64_in"].copy()
    assert gpu.prob(P, g["log64"], g["values"], np.ones(37), g["mag"], 128, 256) > 0
    assert np.array_equal(P, g["P64"])                               # serial order kept: bit-exact
    P = np.zeros(5)
    gpu.prob(P, g["log32"], g["values"], None, g["mag"])
    assert np.array_equal(P, g["P32"])


def test_fastlog_prob_edge_cases(gpu, oracle):
    x = np.array([[0.0, -1.0, 1e-3]], dtype=np.float32)
    gpu.fastlog(x)
    assert np.isneginf(x[0, 0]) and np.isneginf(x[0, 1])             # (float)DBL_MIN == 0
    rng = np.random.default_rng(11)
    rows, cols = 131, 203                                            # ragged vs the 64x64 tiles
    big = rng.lognormal(-5, 2, (rows, cols + 9))
    view = big[:, 3:3 + cols]                                        # non-contiguous rows (ld > cols)
    want = view.copy(); oracle.fastlog(want)
    gpu.fastlog(view)
    assert np.max(np.abs(view - want)) <= 4e-16 * np.max(np.abs(want))
    assert np.array_equal(big[:, :3], big[:, :3]) and np.all(big[:, cols + 3:] > 0)
    values = rng.uniform(-9, -1, cols); mag = rng.uniform(-1, 1, rows)
    Pfull = np.zeros((2, rows + 5))
    gpu.prob(Pfull[1, 2:2 + rows], want, values, None, mag)          # a view into P, like bayeslib.py:195
    Pw = np.zeros(rows); oracle.prob(Pw, want, values, mag)
    assert np.array_equal(Pfull[1, 2:2 + rows], Pw) and not Pfull[0].any() and not Pfull[1, :2].any()
    P0 = np.ones(3); gpu.prob(P0, np.zeros((3, 0)), np.zeros(0), None, np.zeros(3))
    assert np.array_equal(P0, np.ones(3))


# ----------------------------------------------------------------------------- end to end
def _e2e_inputs(g):
    T, tg, npre = int(g["T"]), g["tgrid"], int(g["npre"])
    e_data = [([tg] * 3, list(g["obs0"]), [None] * 3), ([tg[:npre]] * 3, list(g["obs1"]), [None] * 3)]
    flags = {"load_PL_from_file": False, "log_pl": True, "self_normalize": False}
    return T, e_data, flags


def test_simulate_unfused_vs_reference_bayes_golden(gpu, golden):
    g = golden("bayes_e2e")
    T, e_data, flags = _e2e_inputs(g)
    X = g["X"]
    P = np.zeros((2, len(X)))
    z = np.zeros(1)
    sim_params = [float(g["length"]), float(g["time"]), 128, T, 1, (0,), 7, 10000]
    gpu.simulate(gpu.pvSim, e_data, P, X, [None], [None], 3, sim_params, g["ini"], flags,
                 {"sims_per_gpu": 4, "num_gpus": 1}, 0, z.copy(), z.copy(), z.copy())
    # fp32 PL buffer: one float32 ulp of log10 PL (~1e-7 * |log PL| ~ 7e-7) enters each residual
    assert np.max(np.abs(P - g["P"]) / np.abs(g["P"])) < 2e-5


def test_fused_loglik_vs_reference_and_oracle(gpu, oracle, golden):
    g = golden("bayes_e2e")
    T, e_data, flags = _e2e_inputs(g)
    X = g["X"]
    for e in range(2):
        obs = [e_data[e][1][c] for c in range(3)]
        info = {}
        P32 = gpu.loglik(X, g["ini"], 2000.0, float(g["time"]), 128, T, obs, pl_f32=True, info=info)
        assert not info["status"].any()
        assert np.max(np.abs(P32 - g["P"][e]) / np.abs(g["P"][e])) < 2e-5
        # full fp64 (no float32 staging) against the oracle run with a float64 buffer
        e64 = [([g["tgrid"][:len(o)] for o in obs], obs)]
        want = oracle.simulate_loglik(X, g["ini"], 2000.0, float(g["time"]), 128, T, e64, pl_dtype=np.float64,
                                      nthreads=4)[0]
        for strict, tol in ((True, 1e-11), (False, 1e-8)):
            P64 = gpu.loglik(X, g["ini"], 2000.0, float(g["time"]), 128, T, obs, strict=strict)
            assert np.max(np.abs(P64 - want) / np.abs(want)) < tol
    # simulate() in fused mode accumulates into P exactly like the unfused loop
    P = np.zeros((2, len(X))); z = np.zeros(1)
    sim_params = [2000.0, float(g["time"]), 128, T, 1, (0,), 7, 10000]
    gpu.simulate(gpu.pvSim, e_data, P, X, [None], [None], 3, sim_params, g["ini"], flags,
                 {"sims_per_gpu": 4, "num_gpus": 1, "fused": True}, 0, z.copy(), z.copy(), z.copy())
    assert np.max(np.abs(P - g["P"]) / np.abs(g["P"])) < 2e-5


def test_fused_loglik_twothick_normalize_and_nonconvergence(gpu, oracle):
    w = gpu.workloads
    ini, lens = w.twothick(128)
    X = w.samples(6)
    X[:, -1] = np.linspace(-0.3, 0.3, 6)
    T, Time = 40, 1.0
    ref = [oracle.pvsim((w.MARKED_POINT * gpu.UNIT_CONVERSIONS)[None, :-1], lens[c], Time, 128, T, ini[c])["plI"][0]
           for c in range(6)]
    obs = [np.log10(r / r[0])[: T + 1 - 3 * c] for c, r in enumerate(ref)]          # ragged n_obs
    e_data = [([np.linspace(0, Time, T + 1)[:len(o)] for o in obs], obs)]
    want = oracle.simulate_loglik(X, ini, lens, Time, 128, T, e_data, pl_dtype=np.float64, normalize=True,
                                  nthreads=4)[0]
    info = {}
    Ps = gpu.loglik(X, ini, lens, Time, 128, T, obs, normalize=True, strict=True, info=info)
    assert np.max(np.abs(Ps - want) / np.abs(want)) < 1e-10
    P = gpu.loglik(X, ini, lens, Time, 128, T, obs, normalize=True)
    assert np.max(np.abs(P - want) / np.abs(want)) < 1e-8
    # non-convergence: pick MAX from the oracle's per-sample iteration maxima so that some samples
    # fail and some do not; a failing sample gets -inf, the others are bit-identical to the full run
    imax = np.array([oracle.pvsim(X[:, :-1], lens[c], Time, 128, T, ini[c])["iters_max"] for c in range(6)]).max(0)
    MAXc = int(np.sort(imax)[len(imax) // 2])
    expect_bad = imax >= MAXc
    assert expect_bad.any() and not expect_bad.all()
    Pn = gpu.loglik(X, ini, lens, Time, 128, T, obs, normalize=True, strict=True, MAX=MAXc, info=info)
    bad = info["status"].any(axis=0)
    assert np.array_equal(bad, expect_bad)
    assert np.all(np.isneginf(Pn[bad])) and np.array_equal(Pn[~bad], Ps[~bad])


def test_fused_loglik_real_data_and_offgrid_times(gpu, oracle, golden):
    """Shipped example data through this repo's own ingestion, then the fused kernel: on-grid
    experiment (trpl_loglik) and irregular off-grid observation times (trpl_loglik_obs, the
    in-kernel form of the reference's per-row griddata) against the reference's bayes() output."""
    import os
    from conftest import GOLDEN
    g = golden("bayes_realdata")
    T, Time, X = int(g["T"]), float(g["time"]), g["X"]
    ini = gpu.get_initpoints(os.path.join(GOLDEN, "exc_power_scan.csv"), {"select_obs_sets": None})
    e0 = gpu.get_data([os.path.join(GOLDEN, "obs_balanced_6ns.csv")],
                      {"time_cutoff": 5, "select_obs_sets": None, "noise_level": None},
                      {"log_pl": True, "self_normalize": False})[0]
    t1 = [g[f"t_1_{c}"] for c in range(3)]; v1 = [g[f"v_1_{c}"] for c in range(3)]
    # float32-staged, like the reference's buffer
    P0 = gpu.loglik(X, ini, 2000.0, Time, 128, T, e0[1], pl_f32=True)
    P1 = gpu.loglik(X, ini, 2000.0, Time, 128, T, v1, times=t1, pl_f32=True)
    assert np.max(np.abs(P0 - g["P"][0]) / np.abs(g["P"][0])) < 2e-5
    assert np.max(np.abs(P1 - g["P"][1]) / np.abs(g["P"][1])) < 2e-5
    # full fp64 against the oracle with a float64 buffer (scipy griddata on the CPU side)
    want = oracle.simulate_loglik(X, ini, 2000.0, Time, 128, T, [(t1, v1)], pl_dtype=np.float64, nthreads=4)[0]
    for strict, tol in ((True, 1e-11), (False, 1e-8)):
        info = {}
        P64 = gpu.loglik(X, ini, 2000.0, Time, 128, T, v1, times=t1, strict=strict, info=info)
        assert not info["status"].any() and np.max(np.abs(P64 - want) / np.abs(want)) < tol
    # the device-resident entry point (torch tensors) gives the same numbers as the host-buffer one
    import torch
    from trpl_amd import device as tdev
    dev = torch.device("cuda", 0)
    sim_t = np.linspace(0, Time, T + 1)
    br = [gpu.bracket_times(sim_t, t) for t in t1]
    n1 = len(t1[0])
    td = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    Pd = torch.zeros(len(X), dtype=torch.float64, device=dev); ssed = torch.empty((3, len(X)), dtype=torch.float64, device=dev)
    tdev.loglik_obs_device(td(X, torch.float64), td(ini, torch.float64), 2000.0, Time, 128, T, td(np.array(v1), torch.float64),
                           td(np.array([b[0] for b in br]), torch.int32), td(np.array([b[1] for b in br]), torch.float64),
                           td(np.array([b[2] for b in br]), torch.float64), [n1] * 3, Pd, ssed)
    assert np.array_equal(Pd.cpu().numpy(), P64)
    # unsorted input is sorted by time; observation order does not matter beyond rounding
    perm = np.random.default_rng(0).permutation(len(t1[0]))
    Pp = gpu.loglik(X, ini, 2000.0, Time, 128, T, [v1[0][perm], v1[1], v1[2]], times=[t1[0][perm], t1[1], t1[2]])
    assert np.allclose(Pp, P64, rtol=1e-12)
    with pytest.raises(ValueError):
        gpu.loglik(X, ini, 2000.0, Time, 128, T, v1, times=[t1[0] + 1.0, t1[1], t1[2]])
    # simulate() in fused mode picks the right entry point per experiment
    e_data = [e0, (t1, v1, [None] * 3)]
    P = np.zeros((2, len(X))); z = np.zeros(1)
    gpu.simulate(gpu.pvSim, e_data, P, X, [None], [None], 3, [2000.0, Time, 128, T, 1, (0,), 7, 10000], ini,
                 {"load_PL_from_file": False, "log_pl": True, "self_normalize": False},
                 {"sims_per_gpu": 3, "num_gpus": 1, "fused": True}, 0, z.copy(), z.copy(), z.copy())
    assert np.max(np.abs(P - g["P"]) / np.abs(g["P"])) < 2e-5
    # ... and the unfused drop-in loop gives the same
    P2 = np.zeros((2, len(X)))
    gpu.simulate(gpu.pvSim, e_data, P2, X, [None], [None], 3, [2000.0, Time, 128, T, 1, (0,), 7, 10000], ini,
                 {"load_PL_from_file": False, "log_pl": True, "self_normalize": False},
                 {"sims_per_gpu": 3, "num_gpus": 1}, 0, z.copy(), z.copy(), z.copy())
    assert np.max(np.abs(P2 - g["P"]) / np.abs(g["P"])) < 2e-5


# ---- likelihood of PL rows resident in HBM (trpl_loglik_from_pl_dev): one solve, several experiments ----
def test_loglik_from_resident_pl_equals_the_fused_kernel(trpl, gpu):
    """solve_pl_device into an HBM buffer + loglik_from_pl_device per observation set == the fused kernel
    (same PL, same log10 / staging / interpolation rules; the squared errors are summed in another order):
    fp64 and float32 staging, self-normalisation, on- and off-grid times, a flagged system, two experiments
    on one solve."""
    import torch
    tdev = trpl.device
    S, T, Time, L = 300, 160, 4.0, 128
    X = trpl.workloads.samples(S, seed=9)
    X[:, 12] = np.linspace(-0.3, 0.4, S)                                   # non-trivial log offsets
    ini, lengths = trpl.workloads.power_scan(L)
    dev = torch.device("cuda", 0)
    X_d = torch.from_numpy(X).to(dev)
    mat_d, mag_d = X_d[:, :12].contiguous(), X_d[:, 12].contiguous()
    ini_d = torch.from_numpy(ini).to(dev)
    rng = np.random.default_rng(3)
    sim_t = np.linspace(0, Time, T + 1)
    obs_on = [19.0 - 0.02 * np.arange(100), 19.5 - 0.01 * np.arange(T + 1), 20.0 - 0.03 * np.arange(7)]
    t_off = [np.sort(rng.uniform(0, Time, n)) for n in (50, 1, 33)]
    obs_off = [19.0 + 0.1 * rng.standard_normal(len(t)) for t in t_off]
    for pl_dtype, f32 in ((torch.float64, False), (torch.float32, True)):
        for normalize in (False, True):
            flags = trpl._abi.FLAG_NORMALIZE if normalize else 0
            want_on = trpl.loglik(X, ini, lengths, Time, L, T, obs_on, pl_f32=f32, normalize=normalize)
            want_off = trpl.loglik(X, ini, lengths, Time, L, T, obs_off, times=t_off, pl_f32=f32, normalize=normalize)
            P_on = torch.zeros(S, dtype=torch.float64, device=dev)
            P_off = torch.zeros(S, dtype=torch.float64, device=dev)
            pl = torch.empty((S, T + 1), dtype=pl_dtype, device=dev)
            st = torch.empty(S, dtype=torch.int32, device=dev)
            for c in range(3):                                               # ONE solve per curve, two experiments on it
                tdev.solve_pl_device(mat_d, lengths[c], Time, L, T, ini_d[c].contiguous(), pl, status=st)
                tdev.loglik_from_pl_device(pl, torch.from_numpy(obs_on[c]).to(dev), mag_d, P=P_on, flags=flags, status=st)
                hi, dx, h = trpl.bracket_times(sim_t, t_off[c])
                tdev.loglik_from_pl_device(pl, torch.from_numpy(obs_off[c]).to(dev), mag_d, P=P_off, flags=flags, status=st,
                                           obs_hi=torch.from_numpy(hi).to(dev), obs_dx=torch.from_numpy(dx).to(dev),
                                           obs_h=torch.from_numpy(h).to(dev))
            tol = 2e-6 if f32 else 1e-12          # float32 staging: (float)(pl/norm) here vs (float)pl/(float)norm fused
            assert np.allclose(P_on.cpu().numpy(), want_on, rtol=tol, atol=0), (pl_dtype, normalize)
            assert np.allclose(P_off.cpu().numpy(), want_off, rtol=tol, atol=0), (pl_dtype, normalize)
    # a flagged system scores -inf, its neighbours are untouched
    pl = torch.empty((S, T + 1), dtype=torch.float64, device=dev)
    st = torch.empty(S, dtype=torch.int32, device=dev)
    tdev.solve_pl_device(mat_d, lengths[2], Time, L, T, ini_d[2].contiguous(), pl, status=st, MAX=25)
    assert 0 < int((st != 0).sum()) < S
    sse = torch.empty(S, dtype=torch.float64, device=dev)
    tdev.loglik_from_pl_device(pl, torch.from_numpy(obs_on[1]).to(dev), mag_d, sse=sse, status=st)
    bad = (st != 0).cpu().numpy()
    assert np.isinf(sse.cpu().numpy()[bad]).all() and np.isfinite(sse.cpu().numpy()[~bad]).all()
    with pytest.raises(trpl.TrplError):
        tdev.loglik_from_pl_device(pl, torch.zeros(T + 5, dtype=torch.float64, device=dev), mag_d, sse=sse)


def test_fused_call_limits_sixteen_curves_and_strided_pl(trpl, gpu):
    """Edge sizes of one fused call: the maximum of 16 curves (ragged observation counts) equals sixteen
    one-curve calls accumulated in curve order; a 17th curve is a second launch of the same call; plT > 1 in a launch large enough for the
    two-systems-per-wavefront kernel equals STRICT."""
    S, T, Time, L = 40, 50, 1.25, 128
    X = trpl.workloads.samples(S, seed=21)
    base, lens3 = trpl.workloads.power_scan(L)
    ini = np.stack([base[c % 3] * (1.0 + 0.05 * c) for c in range(16)])
    lengths = np.array([2000.0 if c % 2 else 311.0 for c in range(16)])
    obs = [np.full(1 + (7 * c) % (T + 1), 19.0 + 0.1 * c) for c in range(16)]
    info = {}
    P16 = trpl.loglik(X, ini, lengths, Time, L, T, obs, info=info)
    Pacc = np.zeros(S)
    for c in range(16):
        one = {}
        trpl.loglik(X, ini[c:c + 1], lengths[c:c + 1], Time, L, T, [obs[c]], P=Pacc, info=one)
        assert np.array_equal(one["sse"][0], info["sse"][c]) and np.array_equal(one["iters_total"][0], info["iters_total"][c])
    assert np.array_equal(P16, Pacc)
    # a 17th curve: a second launch inside the same call since round 4 (bayeslib.py:117 loops any number of curves;
    # test_more_than_sixteen_curves_per_fused_call below), the first sixteen keep their bits
    i17 = {}
    P17 = trpl.loglik(X, np.concatenate([ini, ini[:1]]), np.append(lengths, 311.0), Time, L, T, obs + [obs[0]], info=i17)
    assert np.array_equal(i17["sse"][:16], info["sse"]) and np.array_equal(i17["sse"][16], info["sse"][0])
    assert np.array_equal(P17, P16 - i17["sse"][16])
    # plT = 4 at paired-kernel size
    S2, T2 = 5200, 64
    if trpl._abi.lib().trpl_kernel_variant(3 * S2, 128, T2, 0) == trpl._abi.KERNEL_FAST_PAIR:
        X2 = trpl.workloads.samples(S2, seed=22)
        obs4 = [np.full(T2 // 4 + 1, 19.5)] * 3
        fi, si = {}, {}
        pf = trpl.loglik(X2, base, lens3, T2 * 0.025, L, T2, obs4, plT=4, info=fi)
        ps = trpl.loglik(X2, base, lens3, T2 * 0.025, L, T2, obs4, plT=4, info=si, strict=True)
        assert np.array_equal(fi["iters_total"], si["iters_total"]) and np.allclose(pf, ps, rtol=1e-9, atol=0)


def test_paired_kernel_offgrid_observations_normalize_and_f32_staging_vs_oracle(gpu, oracle):
    """The paired kernel's less-travelled emission paths against the oracle's restatement of bayeslib.simulate:
    observation times OFF the simulation grid (per-row griddata in the reference, bayeslib.py:184-191), with
    self-normalisation (:150-154) and with the reference's float32 PL staging (:137) -- 96 samples x 3 curves."""
    w = gpu.workloads
    ini, lens = w.power_scan(128)
    S, T = 96, 160
    Time = T * 0.025
    X = w.samples(S, seed=121)
    rng = np.random.default_rng(7)
    times = [np.sort(rng.uniform(0.0, Time, 57)) for _ in range(3)]
    obs = [np.full(57, 19.0) - 0.3 * t for t in times]
    for normalize in (False, True):
        for f32 in (False, True):
            want = oracle.simulate_loglik(X, ini, lens, Time, 128, T, [(times, obs)],
                                          pl_dtype=np.float32 if f32 else np.float64, normalize=normalize,
                                          nthreads=nthreads())[0]
            for kernel in ("pair", "single"):
                info = {}
                P = gpu.loglik(X, ini, lens, Time, 128, T, obs, times=times, pl_f32=f32, normalize=normalize,
                               kernel=kernel, info=info)
                assert not info["status"].any()
                # float32 staging: one float32 ulp of log10 PL enters every residual (as in the unfused golden test)
                gate = 2e-5 if f32 else 1e-8
                assert np.max(np.abs(P - want) / np.abs(want)) < gate, (normalize, f32, kernel)


@pytest.mark.parametrize("kernel", ["single", "pair", None])
def test_more_than_sixteen_curves_per_fused_call(gpu, kernel):
    """bayeslib.simulate loops over ANY number of curves (bayeslib.py:117); a stepper launch carries the constants of at
    most 16.  The fused call runs 18 curves as two launches and one reduction: every curve's squared-error sum, status,
    iteration total and floor_col equal those of the curve run alone, bit for bit, and P is minus their sum in curve
    order (probs.py:44).  Through the host-buffer, the sharded and the off-grid entry points."""
    w = gpu.workloads
    L, T, S, C = 128, 150, 21, 18
    Time = T * DT
    rng = np.random.default_rng(11)
    lens = np.where(np.arange(C) % 3 == 0, 311.0, 2000.0)
    amps = 10 ** rng.uniform(16.0, 18.2, C)
    ini = np.stack([w.beer_lambert(amps[c], lens[c], L) for c in range(C)])
    X = w.samples(S, seed=5)
    obs = [np.linspace(19.0, 18.0, T + 1 - (c % 2) * 7) + 0.01 * c for c in range(C)]
    kw = {} if kernel is None else dict(kernel=kernel)
    info = {}
    P = gpu.loglik(X, ini, lens, Time, L, T, obs, info=info, **kw)
    assert not info["status"].any()
    alone = {}
    for c in range(C):
        one = {}
        gpu.loglik(X, ini[c:c + 1], lens[c:c + 1], Time, L, T, [obs[c]], info=one, kernel=kernel or "single")
        alone[c] = one
        if kernel is not None:                       # same stepper: the same bits
            assert np.array_equal(info["sse"][c], one["sse"][0]), c
            assert np.array_equal(info["iters_total"][c], one["iters_total"][0])
            assert np.array_equal(info["floor_col"][c], one["floor_col"][0])
        else:
            assert np.allclose(info["sse"][c], one["sse"][0], rtol=1e-9, atol=0)
    want = np.zeros(S)
    for c in range(C):
        want -= info["sse"][c]
    assert np.array_equal(P, want)
    # sharded over "devices" (the one GPU three times) and with off-grid observation times
    multi = {}
    Pm = gpu.loglik(X, ini, lens, Time, L, T, obs, info=multi, devices=[0, 0, 0], **kw)
    assert np.array_equal(Pm, P) and np.array_equal(multi["sse"], info["sse"])
    times = [np.linspace(0.0, Time, 40)[1:-1] + 0.004 for _ in range(C)]
    off = {}
    Po = gpu.loglik(X, ini, lens, Time, L, T, [np.full(38, 18.5)] * C, info=off, times=times, **kw)
    one = {}
    gpu.loglik(X, ini[17:18], lens[17:18], Time, L, T, [np.full(38, 18.5)], info=one, times=times[17:18], kernel=kernel or "single")
    if kernel is not None:
        assert np.array_equal(off["sse"][17], one["sse"][0])
    assert np.isfinite(Po).all()
    # ... and the off-grid call's other outputs for curves of BOTH launch groups: a curve of the first sixteen run alone,
    # status / iteration totals / floor_col of every group, P = minus the sum of the per-curve sums in curve order, and the
    # whole thing against the reference's loop -- PL from trpl_solve_pl, log10, the interp1d form of bayeslib.py:189, prob
    first = {}
    gpu.loglik(X, ini[3:4], lens[3:4], Time, L, T, [np.full(38, 18.5)], info=first, times=times[3:4], kernel=kernel or "single")
    if kernel is not None:
        assert np.array_equal(off["sse"][3], first["sse"][0]) and np.array_equal(off["floor_col"][3], first["floor_col"][0])
        assert np.array_equal(off["iters_total"][3], first["iters_total"][0]) and np.array_equal(off["iters_total"][17], one["iters_total"][0])
        assert np.array_equal(off["floor_col"][17], one["floor_col"][0])
    assert not off["status"].any() and (off["floor_col"] == -1).all()
    want_o = np.zeros(S)
    for c in range(C):
        want_o -= off["sse"][c]
    assert np.array_equal(Po, want_o)
    sim_t = np.linspace(0, Time, T + 1)
    for c in (3, 16, 17):
        pl, _, _, _ = gpu.solve_pl(X[:, :12], lens[c], Time, L, T, ini[c], kernel=kernel or "single")
        lg = gpu.interp_rows(sim_t, np.log10(pl), times[c])
        ref = np.sum((lg + X[:, 12:13] - 18.5) ** 2, axis=1)
        assert np.max(np.abs(off["sse"][c] / ref - 1)) < 1e-10, c
    with pytest.raises(gpu.TrplError):
        gpu.loglik(X[:2], np.repeat(ini[:1], 1025, axis=0), np.full(1025, 2000.0), Time, L, T, [obs[0]] * 1025)


@pytest.mark.parametrize("kernel", ["single", "pair"])
def test_twothick_bench_window_offgrid_observations_against_the_oracle(gpu, oracle, twothick_window, kernel):
    """The caller-side data path at the bench's window: observation times OFF the simulation grid (the reference
    interpolates every PL row with scipy griddata, bayeslib.py:184-191; here the bracketing is fused into the stepper,
    trpl_loglik_obs) and self-normalisation (:150-154), Twothick x 32 samples x 6 curves x T = 8000, against the oracle's
    restatement of bayeslib.simulate.  Floor-free samples: the oracle's likelihood to 1e-8 in fp64 and to 2e-5 with the
    reference's float32 PL staging (one float32 ulp of log10 PL enters every residual)."""
    g = twothick_window
    rng = np.random.default_rng(23)
    times = [np.sort(rng.uniform(0.0, g["Time"], 211)) for _ in range(6)]
    obs = [np.interp(times[c], np.linspace(0.0, g["Time"], g["T"] + 1), g["obs"][c]) + 0.01 for c in range(6)]
    for normalize, f32 in ((False, False), (True, False), (False, True)):
        want = oracle.simulate_loglik(g["X"], g["ini"], g["lens"], g["Time"], g["L"], g["T"], [(times, obs)],
                                      pl_dtype=np.float32 if f32 else np.float64, normalize=normalize, nthreads=nthreads())[0]
        info = {}
        P = gpu.loglik(g["X"], g["ini"], g["lens"], g["Time"], g["L"], g["T"], obs, times=times, pl_f32=f32,
                       normalize=normalize, kernel=kernel, info=info)
        assert not info["status"].any()
        clear = (info["floor_col"] == -1).all(axis=0)
        assert clear.sum() >= 0.8 * g["S"]
        rel = np.abs(P - want) / np.abs(want)
        assert rel[clear].max() < (2e-5 if f32 else 1e-8), (normalize, f32, float(rel[clear].max()))


def test_configs0_fused_likelihood_against_the_reference_cpu_path(gpu, golden):
    """BASELINE.json configs[0] on the GPU against what the REFERENCE'S OWN CPU PATH computed for it: 64 random samples x the
    three Power_scan curves, bench window (T = 8000 steps = 200 ns), the shipped Balancedhighsurf observations;
    tests/golden/fallback64.npz holds bayeslib.bayes(pvSim_fallback.pvSim_cpu_fallback, ...)'s likelihoods and PL (oracle/
    gen_golden.py case_fallback64).  The two paths are NOT the same discretisation: the CPU model
    integrates the method-of-lines system with scipy's adaptive BDF (rtol 1e-5) and takes PL by Simpson's rule over the cell
    CENTRES, which leaves out the two half end-cells the GPU path's midpoint rule (pvSimPCR.py:276-281) includes -- with
    the excitation peaked at the front surface that is 0.029 .. 0.040 dex at t = 0, decaying to 0.004 dex by 200 ns; the CPU
    branch also stages PL in float32 and has no mag_offset (0 in this box).  The stated bounds are that documented offset:
      |log10 PL_gpu - log10 PL_cpu| <= 0.0405 dex on every stored column, <= 0.0045 dex at 200 ns, the GPU curve ABOVE the CPU
      one everywhere (the omitted end-cells are positive), and |P_gpu / P_cpu - 1| <= 6e-3 (measured 5.5e-3, median 3.2e-3);
    a discretisation gap, not a parity statement -- parity is held against pvSimPCR.py and Legacy/pvSim.py elsewhere."""
    g = golden("fallback64")
    X, ini, L, length = g["X"], g["ini"], int(g["L"]), float(g["length"])
    T, Time, dec = int(g["w8k_T"]), float(g["w8k_time"]), int(g["w8k_dec"])
    obs = [g["w8k_obs_%d" % c] for c in range(3)]
    assert [len(o) for o in obs] == [5601, 8001, 8001] and (X[:, 12] == 0).all()
    worst_dex = 0.0
    for c in range(3):
        pl, st, _, _ = gpu.solve_pl(X[:, :12], length, Time, L, T, ini[c])
        assert not st.any()
        d = np.log10(pl[:, ::dec]) - np.log10(g["w8k_pl32"][c].astype(np.float64))
        assert (d > 0).all() and d.max() <= 0.0405 and d[:, -1].max() <= 0.0045, (c, float(d.min()), float(d.max()))
        assert 0.028 < d[:, 0].min() and d[:, 0].max() < 0.0405                  # the quadrature offset at t = 0
        worst_dex = max(worst_dex, float(d.max()))
    for kernel in ("single", "pair"):
        info = {}
        P = gpu.loglik(X, ini, length, Time, L, T, obs, info=info, kernel=kernel)
        assert not info["status"].any() and (info["floor_col"] == -1).all()
        rel = np.abs(P / g["w8k_P"][0] - 1)
        assert rel.max() <= 6e-3 and np.median(rel) <= 4e-3, (kernel, float(rel.max()))
    # the same likelihoods through the drop-in driver on the reference's own data path: dataio.get_data is pinned to
    # bayes_io.get_data elsewhere (test_csv_ingestion_matches_reference); here the ingested values are the fixture's
    record("configs0_vs_reference_cpu_path", {"max_dex": worst_dex, "max_rel_loglik": float(rel.max()), "median_rel_loglik": float(np.median(rel)),
                                              "reference_wall_s_8_tasks": float(g["w8k_wall"]), "reference_cpu": str(g["cpu_model"])})


@pytest.mark.parametrize("kernel", ["single", "pair"])
def test_offgrid_observations_dense_sparse_and_on_the_window_edges(gpu, kernel):
    """The off-grid emission is batched 64 grid columns at a time, the observations of a batch taken 64 per pass across the
    lanes (PlSink::flush_batch): many observations per grid interval (several passes per batch, a chunk boundary inside an
    interval), repeated times, none at all for hundreds of steps, observations exactly at t = 0, on grid nodes and at
    t = Time, a ragged count per curve, with and without the reference's float32 staging and self-normalisation -- against
    STRICT (the serial column-by-column emission, emit()) and against the reference's own order of operations on the stored
    PL matrix (trpl_solve_pl, log10, the interp1d form of bayeslib.py:189, probs.prob)."""
    w = gpu.workloads
    ini, lens = w.power_scan(128)
    S, T = 37, 333                                              # an odd batch, a last batch of 14 columns
    Time = T * DT
    X = w.samples(S, seed=23)
    X[:, 12] = np.linspace(-0.2, 0.2, S)
    rng = np.random.default_rng(17)
    sim_t = np.linspace(0, Time, T + 1)
    dense = np.sort(rng.uniform(0.0, 70 * DT, 4000))            # ~57 observations per interval over the first 70 steps
    times = [np.concatenate([[0.0, 0.0], dense, np.repeat(sim_t[[90, 91, 128]], 3), [Time - 1e-9, Time, Time]]),   # a gap of 200 steps
             np.sort(rng.uniform(0.0, Time, 129)),              # sparse: less than one per batch of 64 columns on average
             np.concatenate([sim_t[::7], [Time]])]              # on grid nodes only
    obs = [np.full(len(t), 18.0) - 0.2 * t for t in times]
    for pl_f32, normalize in ((False, False), (True, False), (False, True)):
        info, ref = {}, {}
        P = gpu.loglik(X, ini, lens, Time, 128, T, obs, times=times, kernel=kernel, pl_f32=pl_f32, normalize=normalize, info=info)
        Ps = gpu.loglik(X, ini, lens, Time, 128, T, obs, times=times, strict=True, pl_f32=pl_f32, normalize=normalize, info=ref)
        assert not info["status"].any() and np.array_equal(info["iters_total"], ref["iters_total"])
        assert (info["floor_col"] == -1).all() and (ref["floor_col"] == -1).all()
        gate = 2e-6 if pl_f32 else 1e-9                          # float32 staging: an ulp of log10 PL can flip with the state's last bits
        assert np.max(np.abs(info["sse"] / ref["sse"] - 1)) < gate, (pl_f32, normalize)
        assert np.max(np.abs(P / Ps - 1)) < gate
        if pl_f32 or normalize:
            continue
        for c in range(3):                                      # the reference's loop on the stored PL
            pl, _, _, _ = gpu.solve_pl(X[:, :12], lens[c], Time, 128, T, ini[c], kernel=kernel)
            lg = gpu.interp_rows(sim_t, np.log10(pl), np.sort(times[c]))
            want = np.sum((lg + X[:, 12:13] - obs[c][np.argsort(times[c], kind="stable")][None, :]) ** 2, axis=1)
            assert np.max(np.abs(info["sse"][c] / want - 1)) < 1e-11, c


@pytest.mark.parametrize("kernel", ["single", "pair"])
def test_offgrid_observations_with_systems_flagged_in_the_middle_of_the_window(gpu, kernel):
    """The batched off-grid emission when a system is FLAGGED (status = 1 + t, pvSimPCR.py:269) at a step t > 0: its
    parked columns are flushed up to the failing step only (PlSink::flush_batch with n = status - 1 - base), its squared
    error is +inf, floor_col -2, the sample's likelihood -inf -- and its wavefront partner (paired kernel: the flagged
    system is parked as a benign one beside a live one) and every other system are untouched.  Every iteration-capped
    system of the reference's parameter box is flagged at step 0 (tools/flag_step_probe.py: 12 batches, 24 000 systems),
    so the inputs are hostile on purpose: every other sample gets a NEGATIVE radiative coefficient (-100 x its B), whose
    dn/dt = +|B| n p blows up in finite time -- flags at steps 2 .. 366 of a 400-step window, before, inside and after the
    first 64-column batch, on one, two or all three curves of a sample (tools/flag_step_probe2.py).  Reference: STRICT with
    the same off-grid times (the serial column-by-column emission)."""
    w = gpu.workloads
    ini, lens = w.power_scan(128)
    S, T = 24, 400
    Time = T * DT
    X = w.samples(S, seed=23)
    X[::2, 4] *= -100.0
    rng = np.random.default_rng(3)
    times = [np.sort(rng.uniform(0.0, Time, 300)) for _ in range(3)]
    obs = [np.full(len(t), 18.0) - 0.2 * t for t in times]
    info, ref = {}, {}
    P = gpu.loglik(X, ini, lens, Time, 128, T, obs, times=times, kernel=kernel, MAX=200, info=info)
    Ps = gpu.loglik(X, ini, lens, Time, 128, T, obs, times=times, strict=True, MAX=200, info=ref)
    st = ref["status"]
    flagged = st != 0
    assert (st[flagged] > 1).sum() >= 10 and (st[flagged] > 65).sum() >= 4 and (st[flagged] == 1).sum() >= 1, st.tolist()
    assert not flagged[:, 1::2].any()                                        # the untouched samples all converge
    assert np.array_equal(info["status"], st)                                # flagged at the same step
    assert np.array_equal(info["iters_total"], ref["iters_total"])          # the iterations up to the failing step included
    for i in (info, ref):
        assert np.isposinf(i["sse"][flagged]).all() and (i["floor_col"][flagged] == -2).all()
        assert (i["floor_col"][~flagged] != -2).all()
    dead = flagged.any(axis=0)
    assert np.isneginf(P[dead]).all() and np.isneginf(Ps[dead]).all()
    clear = ~flagged & (ref["floor_col"] == -1) & np.isfinite(ref["sse"])
    assert clear[:, 1::2].all()                                              # every curve of every untouched sample
    assert np.array_equal(info["floor_col"][~flagged], ref["floor_col"][~flagged])
    assert np.max(np.abs(info["sse"][clear] / ref["sse"][clear] - 1)) < 1e-9
    alive = ~dead & np.isfinite(Ps)
    assert alive[1::2].all() and np.max(np.abs(P[alive] / Ps[alive] - 1)) < 1e-9


def test_driver_levels_agree_on_grid_prefix_observations_across_blocks(gpu, oracle):
    """The production shape in small (tools/e2e_production.py): observation times that are a PREFIX of the simulation grid, of
    different lengths per curve, S not a multiple of sims_per_gpu.  driver.bayes through (A) the fused level -- which routes grid
    prefixes to the on-grid entry -- , (A') the fused level with gpu_info["interpolate_prefix"] (the literal in-kernel
    interpolation, trpl_loglik_obs) and (B) the reference's call sequence pvSim -> fastlog -> interpolation -> prob with the
    curves of a block and the next block overlapped by host threads (5 blocks, the last one short): the same likelihoods --
    A against A' to 1e-13 (interpolating at a node returns the node's value to one rounding), B against A to the float32
    staging both emulate -- and the oracle's restatement of bayeslib.simulate within 2e-5 (float32 staging, as in the goldens)."""
    w = gpu.workloads
    sm = gpu.sampler
    ini, lens = w.power_scan(128)
    T, Time, S = 240, 6.0, 37
    sim_t = np.linspace(0, Time, T + 1)
    n_obs = (101, 161, 241)
    mark = (w.MARKED_POINT * gpu.UNIT_CONVERSIONS)[None, :-1]
    vals = [np.log10(oracle.pvsim(mark, lens[c], Time, 128, T, ini[c])["plI"][0][:n]) + 0.05 for c, n in enumerate(n_obs)]
    e_data = [([sim_t[:n] for n in n_obs], vals, [np.ones(n) for n in n_obs])]
    sim_flags = {"load_PL_from_file": False, "override_equal_auger": False, "override_equal_mu": False, "override_equal_s": False,
                 "log_pl": True, "self_normalize": False, "random_sample": True, "num_points": S}
    simPar = [2000.0, Time, 128, T, 1, (0, 100), 7, 10000]
    box = (sm.DEFAULT_MINX * sm.UNIT_CONVERSIONS, sm.DEFAULT_MAXX * sm.UNIT_CONVERSIONS, sm.DEFAULT_DO_LOG)
    res = {}
    for name, info in (("A", {"sims_per_gpu": 16, "fused": True}), ("A_literal", {"sims_per_gpu": 16, "fused": True, "interpolate_prefix": True}),
                       ("B", {"sims_per_gpu": 8}), ("B_serial", {"sims_per_gpu": 8, "overlap_curves": False})):
        gpu_info = dict(num_gpus=1, has_GPU=True, max_sims_per_block=1, **info)
        _, P, X = gpu.bayes(gpu.pvSim, None, None, *box, ini, list(simPar), e_data, sim_flags, gpu_info, rng=np.random.RandomState(42))
        res[name] = P[0].copy()
    assert np.isfinite(res["A"]).all()
    assert np.max(np.abs(res["A_literal"] / res["A"] - 1)) < 1e-13
    assert np.array_equal(res["B"], res["B_serial"])                       # overlapping changes no bit
    assert np.max(np.abs(res["B"] / res["A"] - 1)) < 1e-7                  # (blocks of 8 run the one-system kernel, the fused 16 x 3 too)
    want = oracle.simulate_loglik(X, ini, lens, Time, 128, T, [(e_data[0][0], e_data[0][1])], sims_per_gpu=8, nthreads=nthreads())[0]
    assert np.max(np.abs(res["B"] / want - 1)) < 2e-5 and np.max(np.abs(res["A"] / want - 1)) < 2e-5


def test_literal_interpolation_switch_reaches_the_multi_experiment_branch(gpu, oracle, monkeypatch):
    """gpu_info["interpolate_prefix"] in BOTH fused branches (round-5 advisor finding: the branch that keeps a block's PL matrix in
    HBM for several experiments always took the prefix route).  Two experiments whose observation times are prefixes of the
    simulation grid (one curve of the first is sampled on the whole grid): by default every (experiment, curve) pass over the
    resident PL runs the on-grid form of trpl_loglik_from_pl_dev (no brackets), with the switch every pass but the full-grid
    one runs the interpolating form -- seen by a spy on the device wrapper -- and the likelihoods agree to 1e-13 (interpolating AT a node returns the node's value to one rounding)."""
    w, sm = gpu.workloads, gpu.sampler
    ini, lens = w.power_scan(128)
    T, Time, S = 240, 6.0, 21
    sim_t = np.linspace(0, Time, T + 1)
    mark = (w.MARKED_POINT * gpu.UNIT_CONVERSIONS)[None, :-1]
    pl = [oracle.pvsim(mark, lens[c], Time, 128, T, ini[c])["plI"][0] for c in range(3)]
    e_data = [([sim_t[:n] for n in (101, 161, 241)], [np.log10(pl[c][:n]) + 0.05 for c, n in enumerate((101, 161, 241))], [None] * 3),
              ([sim_t[:61]] * 3, [np.log10(pl[c][:61]) - 0.03 for c in range(3)], [None] * 3)]
    X = w.samples(S, seed=3)
    flags = {"load_PL_from_file": False, "log_pl": True, "self_normalize": False}
    seen = []
    real = gpu.device.loglik_from_pl_device

    def spy(*a, **kw):
        seen.append(kw.get("obs_hi") is not None)
        return real(*a, **kw)
    monkeypatch.setattr(gpu.device, "loglik_from_pl_device", spy)
    res = {}
    for name, extra in (("prefix", {}), ("literal", {"interpolate_prefix": True})):
        del seen[:]
        P = np.zeros((2, S))
        z = np.zeros(1)
        gpu.simulate(gpu.pvSim, e_data, P, X, [None], [None], 3, [2000.0, Time, 128, T, 1, (0,), 7, 10000], ini, flags,
                     dict({"sims_per_gpu": 8, "num_gpus": 1, "fused": True, "pl_dtype": np.float64}, **extra), 0, z.copy(), z.copy(), z.copy())
        # per block: curves -> experiments (bayeslib.py:117,:171).  Curve 2 of experiment 0 is sampled on the FULL grid: the
        # reference's own bypass (bayeslib.py:182-183), on the grid under either rule
        want = [name == "literal" and not (e == 0 and c == 2) for _blk in range(3) for c in range(3) for e in range(2)]
        assert seen == want, (name, seen)
        res[name] = P
    assert np.isfinite(res["prefix"]).all() and (res["prefix"] < 0).all()
    assert np.max(np.abs(res["literal"] / res["prefix"] - 1)) < 1e-13
