"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against the golden
vectors produced by the reference and against the CPU oracle on the same seeded inputs.

Bars
  STRICT mode (TRPL_FLAG_STRICT): N/P/E state, every convergence decision and the PL quadrature
      (summed node by node like the reference) are bit-identical: iteration counts EQUAL and
      PL(t) EQUAL in fp64.
  FAST mode (default): FMA contraction + reciprocal arithmetic: PL rtol 1e-9 (north_star's fp64
      tolerance), iteration totals within 1 %.
  byte/serial-order kernels (sse accumulation): bit-exact.  log10: 1 ulp of the buffer dtype.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL_STRICT = 0.0          # bit-identical: relerr(...) <= RTOL_STRICT
RTOL_FAST = 1e-9


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.abs(b)))


# ----------------------------------------------------------------------------- batched PCR
@pytest.mark.parametrize("N", [4, 8, 32, 128, 512])
def test_pcr_batched_vs_reference_golden(gpu, golden, N):
    g = golden("pcr_norm")
    ld, d, ud, B, want = (np.ascontiguousarray(g[f"{k}{N}"]) for k in ("ld", "d", "ud", "B", "x"))
    lib = gpu._abi.lib()
    for flags, exact in ((gpu.FLAG_STRICT, True), (0, False)):
        x = np.zeros_like(d)
        keep = [a.copy() for a in (ld, d, ud, B)]
        gpu._abi.check(lib.trpl_pcr_solve_batched(ld.ctypes.data, d.ctypes.data, ud.ctypes.data, B.ctypes.data,
                                                  x.ctypes.data, d.shape[0], N, 8, flags, 0, None))
        assert all(np.array_equal(a, b) for a, b in zip(keep, (ld, d, ud, B)))       # inputs untouched
        if exact:
            assert np.array_equal(x, want)
        else:
            assert np.max(np.abs(x - want)) <= 1e-13 * np.max(np.abs(want))


def test_pcr_batched_fp32_and_many_systems(gpu, oracle):
    rng = np.random.default_rng(5)
    S, L = 1000, 128
    ld = rng.uniform(-1, 1, (S, L)); ud = rng.uniform(-1, 1, (S, L)); d = rng.uniform(2.5, 4, (S, L))
    ld[:, 0] = 0; ud[:, -1] = 0
    b = rng.normal(size=(S, L))
    lib = gpu._abi.lib()
    x = np.zeros((S, L))
    gpu._abi.check(lib.trpl_pcr_solve_batched(ld.ctypes.data, d.ctypes.data, ud.ctypes.data, b.ctypes.data,
                                              x.ctypes.data, S, L, 8, gpu.FLAG_STRICT, 0, None))
    for s in (0, 1, 499, 999):
        assert np.array_equal(x[s], oracle.pcreduce(ld[s], d[s], ud[s], b[s]))
    r = d * x; r[:, 1:] += ld[:, 1:] * x[:, :-1]; r[:, :-1] += ud[:, :-1] * x[:, 1:]
    assert np.max(np.abs(r - b)) < 1e-12
    f = [a.astype(np.float32) for a in (ld, d, ud, b)]
    for flags in (0, gpu.FLAG_STRICT):                               # interleaved/LDS-staged and blocked fp32 paths
        x32 = np.zeros((S, L), dtype=np.float32)
        gpu._abi.check(lib.trpl_pcr_solve_batched(*(a.ctypes.data for a in f), x32.ctypes.data, S, L, 4, flags, 0, None))
        assert np.max(np.abs(x32 - x)) < 2e-5
    # configs[4] shape: L = 512, fp32
    L5 = 512
    g5 = [rng.uniform(-1, 1, (64, L5)), rng.uniform(2.5, 4, (64, L5)), rng.uniform(-1, 1, (64, L5)), rng.normal(size=(64, L5))]
    g5[0][:, 0] = 0; g5[2][:, -1] = 0
    f5 = [a.astype(np.float32) for a in g5]
    x5 = np.zeros((64, L5), dtype=np.float32)
    gpu._abi.check(lib.trpl_pcr_solve_batched(*(a.ctypes.data for a in f5), x5.ctypes.data, 64, L5, 4, 0, 0, None))
    want5 = np.array([oracle.pcreduce(g5[0][s], g5[1][s], g5[2][s], g5[3][s]) for s in range(64)])
    assert np.max(np.abs(x5 - want5)) < 2e-5


# ----------------------------------------------------------------------------- pvSim
def _run(gpu, X12, length, time_ns, L, T, ini, **kw):
    pl, status, iters, sec = gpu.solve_pl(X12, length, time_ns, L, T, ini, **kw)
    assert sec > 0
    return pl, status, iters


def test_pvsim_power_scan_vs_reference_golden(gpu, golden):
    g = golden("pvsim_power")
    X, T = g["X"], int(g["T"])
    for c in range(3):
        want, want_it = g["plI"][c], g["iters"][c].sum(axis=1)
        pl, st, it = _run(gpu, X[:, :-1], 2000.0, float(g["time"]), 128, T, g["ini"][c], strict=True)
        assert not st.any() and np.array_equal(it, want_it)
        assert relerr(pl, want) <= RTOL_STRICT
        pl, st, it = _run(gpu, X[:, :-1], 2000.0, float(g["time"]), 128, T, g["ini"][c])
        assert not st.any() and np.all(np.abs(it - want_it) <= 0.01 * want_it + 1)
        assert relerr(pl, want) < RTOL_FAST


def test_pvsim_twothick_vs_reference_golden(gpu, golden):
    g = golden("pvsim_twothick")
    X, T = g["X"], int(g["T"])
    for c, length in enumerate(g["lengths"]):
        want, want_it = g["plI"][c], g["iters"][c].sum(axis=1)
        pl, st, it = _run(gpu, X[:, :-1], float(length), float(g["time"]), 128, T, g["ini"][c], strict=True)
        assert not st.any() and np.array_equal(it, want_it)
        assert relerr(pl, want) <= RTOL_STRICT
        pl, st, it = _run(gpu, X[:, :-1], float(length), float(g["time"]), 128, T, g["ini"][c])
        assert not st.any() and np.all(np.abs(it - want_it) <= 0.01 * want_it + 1)
        assert relerr(pl, want) < RTOL_FAST


def test_pvsim_64_random_samples_vs_oracle(gpu, oracle):
    """BASELINE configs[0] shape: Power_scan (3 excitations, 128 nodes) x 64 random parameter samples,
    here against the pinned CPU oracle: STRICT iteration counts identical and PL to 1e-13, FAST PL to
    1e-9 and iteration totals within 1 %."""
    w = gpu.workloads
    ini, lens = w.power_scan(128)
    X = w.samples(64)
    T, Time = 200, 200 * 0.025
    for c in range(3):
        r = oracle.pvsim(X[:, :-1], lens[c], Time, 128, T, ini[c], nthreads=8)
        pl, st, it, _ = gpu.solve_pl(X[:, :-1], lens[c], Time, 128, T, ini[c], strict=True)
        assert not st.any() and not r["status"].any()
        assert np.array_equal(it, r["iters_total"]) and relerr(pl, r["plI"]) <= RTOL_STRICT
        pl, st, it, _ = gpu.solve_pl(X[:, :-1], lens[c], Time, 128, T, ini[c])
        assert not st.any() and relerr(pl, r["plI"]) < RTOL_FAST
        assert np.all(np.abs(it - r["iters_total"]) <= 0.01 * r["iters_total"] + 1)


def test_against_legacy_pvsim_and_odeint(gpu, golden):
    """North-star parity references run by the reference itself (oracle/gen_golden.py):
    Legacy/pvSim.pvSim -- bit-level agreement on the steps where the schemes coincide (PL[0..2]), BDF
    order difference afterwards; PV_tester2.dydt + odeint -- the time-converged solution."""
    g = golden("legacy_odeint")
    X = g["X"].copy(); X[:, 7] = 0.0; X[:, 8] = 0.0
    L, T, length, Time = int(g["L"]), int(g["T"]), float(g["length"]), float(g["time"])
    sim_params = [length, Time, L, T, 1, (0,), 7, 10000]
    for strict in (True, False):
        plI = np.empty((len(X), T + 1))
        gpu.pvSim(plI, None, None, None, X[:, :-1], sim_params, (float(g["a_nm3"]), float(g["l_nm"])),
                  init_mode="exp", strict=strict)
        assert np.max(np.abs(plI[:, :3] / g["plI_legacy"][:, :3] - 1)) < 1e-12
        assert np.max(np.abs(plI / g["plI_legacy"] - 1)) < 1e-3
        assert np.max(np.abs(plI / g["plI_odeint"] - 1)) < 2e-2
        assert np.max(np.abs(plI[:, -1] / g["plI_odeint"][:, -1] - 1)) < 5e-4


def test_pvsim_float32_buffer_matches_reference(gpu, golden):
    g = golden("pvsim_power")
    T = int(g["T32"])
    pl, st, _ = _run(gpu, g["X"][:2, :-1], 2000.0, T * 0.025, 128, T, g["ini"][2], dtype=np.float32, strict=True)
    assert pl.dtype == np.float32 and not st.any()
    # same two float32 roundings as the reference (store, then divide); the fp64 value in front
    # of them differs by ~1e-16, so allow one float32 ulp
    assert np.max(np.abs(pl - g["plI32"]) / g["plI32"]) <= 2.0 ** -23


def test_pvsim_small_grids_plT_and_nonconvergence(gpu, golden):
    g = golden("pvsim_small")
    X = g["X"]
    for L in (8, 32, 64):
        for strict, tol in ((True, RTOL_STRICT), (False, RTOL_FAST)):
            pl, st, it = _run(gpu, X[:, :-1], 500.0, 30 * 0.05, L, 30, g[f"ini_L{L}"], tol=6, strict=strict)
            assert not st.any() and relerr(pl, g[f"plI_L{L}"]) <= tol
            if strict:
                assert np.array_equal(it, g[f"it_L{L}"].sum(axis=1))
    pl, st, it = _run(gpu, X[:, :-1], 500.0, 40 * 0.05, 32, 40, g["ini_L32"], tol=6, plT=4, strict=True)
    assert pl.shape == (3, 11) and relerr(pl, g["plI_plT4"]) <= RTOL_STRICT
    assert np.array_equal(it, g["it_plT4"].sum(axis=1))
    # forced non-convergence: status = 1 + step, remaining PL = NaN, other systems unaffected
    p, t, n = g["nc_log"][-1]
    Xnc = np.vstack([X[2, :-1], X[0, :-1]])
    pl, st, it = _run(gpu, Xnc, 311.0, 10 * 0.025, 32, 10, g["nc_ini"], MAX=3, strict=True)
    assert st[0] == 1 + t and np.isnan(pl[0, t:]).all()


def test_fast_mode_plT_and_midrun_nonconvergence(gpu, oracle):
    """FAST mode batches its PL output over 64 time points: check plT > 1 (columns != steps), a column
    count that is not a multiple of 64, and a non-convergence in the middle of a run -- PL before the
    failing step must be valid and everything from it on NaN -- against the oracle."""
    w = gpu.workloads
    ini, lens = w.power_scan(128)
    X = w.samples(4)
    T, Time = 150, 150 * 0.025
    r = oracle.pvsim(X[:, :-1], lens[2], Time, 128, T, ini[2], plT=3, want_step_iters=True)
    pl, st, it, _ = gpu.solve_pl(X[:, :-1], lens[2], Time, 128, T, ini[2], plT=3)
    assert pl.shape == (4, 51) and not st.any() and relerr(pl, r["plI"]) < RTOL_FAST
    # MAX just above the iteration count of the late steps: the early (stiffer) steps pass only for some samples
    steps = oracle.pvsim(X[:, :-1], lens[2], Time, 128, T, ini[2], want_step_iters=True)["step_iters"]
    MAXc = int(np.sort(steps.max(axis=1))[1]) + 1          # at least one sample exceeds it, at least one does not
    ro = oracle.pvsim(X[:, :-1], lens[2], Time, 128, T, ini[2], MAX=MAXc)
    assert ro["status"].any() and not ro["status"].all()
    for dtype in (np.float64, np.float32):
        pl, st, it, _ = gpu.solve_pl(X[:, :-1], lens[2], Time, 128, T, ini[2], MAX=MAXc, dtype=dtype)
        assert np.array_equal(st, ro["status"])
        for s_ in range(4):
            t_fail = st[s_] - 1 if st[s_] else T + 1
            good = slice(0, t_fail)
            tol = RTOL_FAST if dtype == np.float64 else 2.0 ** -22
            assert np.all(np.abs(pl[s_, good] / ro["plI"][s_, good] - 1) < tol)
            assert np.isnan(pl[s_, t_fail:]).all()
    # likelihood mode: the failing samples get -inf, the others match the oracle
    obs = [np.log10(ro["plI"][~ro["status"].astype(bool)][0]) + 0.05]
    info = {}
    P = gpu.loglik(X, ini[2:3], lens[2:3], Time, 128, T, obs, MAX=MAXc, info=info)
    assert np.array_equal(info["status"][0], ro["status"]) and np.all(np.isneginf(P[ro["status"] != 0]))
    want = oracle.simulate_loglik(X, ini[2:3], lens[2:3], Time, 128, T, [([np.linspace(0, Time, T + 1)], obs)],
                                  pl_dtype=np.float64, MAX=MAXc)[0]
    ok = ro["status"] == 0
    assert np.max(np.abs(P[ok] - want[ok]) / np.abs(want[ok])) < 1e-8


def test_pvsim_dropin_signature(gpu, golden):
    """Called exactly the way bayeslib.simulate calls the model (bayeslib.py:144-146)."""
    g = golden("pvsim_power")
    T = 24
    sim_params = [2000, T * 0.025, 128, T, 1, (0, 1, 3, 10, 30, 100), 7, 10000]
    plI = np.empty((5, T + 1), dtype=np.float32)
    plN = np.empty((5, 2, 128)); plE = np.empty((5, 2, 129))
    sec = gpu.pvSim(plI, plN, plN.copy(), plE, g["X"][:, :-1], sim_params, g["ini"][1], (128,), 8 * 256, 1,
                    init_mode="points")
    assert isinstance(sec, float) and sec > 0
    want = g["plI"][1][:, :T + 1]
    assert np.max(np.abs(plI / want - 1)) < 2e-7
    with pytest.raises(ValueError):
        gpu.pvSim(plI, None, None, None, g["X"][:, :-1], sim_params, g["ini"][1], init_mode="continue")
    with pytest.raises(ValueError):
        gpu.pvSim(plI, None, None, None, g["X"][:, :-1], sim_params, g["ini"][1][:64], init_mode="points")


@pytest.mark.parametrize("L", [256, 512])
def test_pvsim_fine_grids_vs_oracle(gpu, oracle, L):
    """L = 256 / 512 (4 and 8 rows per lane; the reference cannot run 512: its shared arrays exceed
    the 48 KB static limit).  No reference golden exists, so the pinned oracle is the
    check: STRICT iteration counts equal, PL to 1e-13; FAST to 1e-9."""
    w = gpu.workloads
    X = w.samples(3)
    T, Time, length = 12, 12 * 0.025, 2000.0
    ini = w.beer_lambert(w.POWER_SCAN_A_CM3[2], length, L)
    r = oracle.pvsim(X[:, :-1], length, Time, L, T, ini)
    pl, st, it, _ = gpu.solve_pl(X[:, :-1], length, Time, L, T, ini, strict=True)
    assert not st.any() and np.array_equal(it, r["iters_total"]) and relerr(pl, r["plI"]) <= RTOL_STRICT
    pl, st, it, _ = gpu.solve_pl(X[:, :-1], length, Time, L, T, ini)
    assert not st.any() and relerr(pl, r["plI"]) < RTOL_FAST
    assert np.all(np.abs(it - r["iters_total"]) <= 0.01 * r["iters_total"] + 1)


@pytest.mark.parametrize("L,tol,pl_gate,ll_gate", [(128, 4, 2e-4, 1e-3), (512, 3, 2e-3, 1e-2)])
def test_fp32_stepper_vs_fp64_oracle(gpu, oracle, L, tol, pl_gate, ll_gate):
    """TRPL_FLAG_FP32 (BASELINE configs[4]: L = 512, fp32).  No reference exists for fp32 (the
    reference is fp64 only and cannot run L = 512); the bar is the fp64 oracle at the accuracy an fp32
    state allows.  Measured (tools/fp32_probe.py): L = 128, tol 4-5: 2-3e-5 relative PL error
    (SURVEY App. B result 5 found <= 3.3e-5 dex on the emulated reference); L = 512: the diffusion
    stencil amplifies fp32 rounding by D dt/dx^2 ~ 200, tol 3 converges everywhere with <= 1e-3
    relative (4e-4 dex) PL error, tol >= 4 no longer converges for the high-mobility samples."""
    w = gpu.workloads
    X = w.samples(6)
    T, Time, length = 60, 60 * 0.025, 2000.0
    ini = np.stack([w.beer_lambert(A, length, L) for A in w.POWER_SCAN_A_CM3])
    ref = [oracle.pvsim(X[:, :-1], length, Time, L, T, ini[c], nthreads=4) for c in range(3)]
    for c in range(3):
        pl, st, it, _ = gpu.solve_pl(X[:, :-1], length, Time, L, T, ini[c], tol=tol, fp32=True)
        assert not st.any()
        assert relerr(pl, ref[c]["plI"]) < pl_gate
        assert np.all(it <= ref[c]["iters_total"])            # looser tolerance: never more iterations than tol 7
    obs = [np.log10(r["plI"][-1]) + 0.03 for r in ref]
    want = oracle.simulate_loglik(X, ini, length, Time, L, T, [([np.linspace(0, Time, T + 1)] * 3, obs)],
                                  pl_dtype=np.float64, nthreads=4)[0]
    info = {}
    P = gpu.loglik(X, ini, length, Time, L, T, obs, tol=tol, fp32=True, info=info)
    assert not info["status"].any() and np.max(np.abs(P - want) / np.abs(want)) < ll_gate
    with pytest.raises(gpu.TrplError):
        gpu.solve_pl(X[:, :-1], length, Time, 64, T, w.beer_lambert(1e17, length, 64), fp32=True)   # L < 128
    with pytest.raises(gpu.TrplError):
        gpu.solve_pl(X[:, :-1], length, Time, L, T, ini[0], fp32=True, strict=True)


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


# ----------------------------------------------------------------------------- full-size properties
def test_full_size_properties(gpu):
    """At sizes the CPU oracle cannot reach: (i) FAST vs STRICT agree on thousands of random
    samples; (ii) shard invariance: a sample's likelihood does not depend on its batch;
    (iii) determinism: two runs are bit-identical; (iv) the offset identity
    P(m) = P(0) - sum_c [ n_c m^2 + 2 m r_c ] holds through the fused kernel."""
    w = gpu.workloads
    ini, lens = w.power_scan(128)
    S, T, Time = 65536, 24, 24 * 0.025            # BASELINE configs[1] sample count
    X = w.samples(S)
    mark = (w.MARKED_POINT * gpu.UNIT_CONVERSIONS)[None, :]
    obs = []
    for c in range(3):
        pl, st, _, _ = gpu.solve_pl(mark[:, :-1], lens[c], Time, 128, T, ini[c], strict=True)
        obs.append(np.log10(pl[0]))
    info_f, info_s = {}, {}
    Pf = gpu.loglik(X, ini, lens, Time, 128, T, obs, info=info_f)
    Ps = gpu.loglik(X, ini, lens, Time, 128, T, obs, strict=True, info=info_s)
    ok = ~(info_f["status"].any(axis=0) | info_s["status"].any(axis=0))
    assert ok.mean() > 0.99
    assert np.max(np.abs(Pf[ok] - Ps[ok]) / np.abs(Ps[ok])) < 1e-8
    assert abs(info_f["iters_total"].sum() / info_s["iters_total"].sum() - 1) < 1e-3
    Pf2 = gpu.loglik(X, ini, lens, Time, 128, T, obs)
    assert np.array_equal(Pf, Pf2)                                            # (iii)
    sub = gpu.loglik(X[60001:65300], ini, lens, Time, 128, T, obs)            # same kernel, other partners
    assert np.array_equal(sub, Pf[60001:65300])                               # (ii)
    # a launch that cannot keep the chip full runs the one-system-per-wavefront kernel, whose node sums
    # associate differently: same likelihoods to rounding
    small = gpu.loglik(X[61000:61300], ini, lens, Time, 128, T, obs)
    assert np.allclose(small, Pf[61000:61300], rtol=1e-10, atol=1e-10)
    m = 0.25
    Xm = X.copy(); Xm[:, -1] = m
    Pm = gpu.loglik(Xm, ini, lens, Time, 128, T, obs)
    # residual sums r_c from sse(m=0): sum (a+m)^2 = sum a^2 + 2 m sum a + n m^2; check via a third offset
    Xm2 = X.copy(); Xm2[:, -1] = -m
    Pm2 = gpu.loglik(Xm2, ini, lens, Time, 128, T, obs)
    n_tot = 3 * (T + 1)
    assert np.max(np.abs((Pm[ok] + Pm2[ok]) / 2 - (Pf[ok] - n_tot * m * m)) / np.abs(Pf[ok])) < 1e-9


def test_multi_device_entry_point_equals_single_device(trpl, gpu):
    """trpl_loglik_multi: the shards of one host thread's call (three streams on this box's one device,
    uneven shard sizes) give bit-for-bit the single-launch result, on- and off-grid, all outputs."""
    X = trpl.workloads.samples(50, seed=5)
    ini, lengths = trpl.workloads.power_scan(128)
    T, Time = 120, 3.0
    ref_info = {}
    obs0 = [np.full(T + 1, 20.0) - 0.01 * np.arange(T + 1)] * 3
    want = trpl.loglik(X, ini, lengths, Time, 128, T, obs0, info=ref_info)
    for devices in ([0], [0, 0, 0], "all"):
        info = {}
        got = trpl.loglik(X, ini, lengths, Time, 128, T, obs0, info=info, devices=devices)
        assert np.array_equal(got, want)
        for k in ("sse", "status", "iters_total"):
            assert np.array_equal(info[k], ref_info[k]), (devices, k)
    times = [np.sort(np.random.default_rng(c).uniform(0, Time, 40)) for c in range(3)]
    obs1 = [np.full(40, 19.5)] * 3
    want = trpl.loglik(X, ini, lengths, Time, 128, T, obs1, times=times)
    got = trpl.loglik(X, ini, lengths, Time, 128, T, obs1, times=times, devices=[0, 0])
    assert np.array_equal(got, want)
    # more shards than samples: empty shards are skipped
    got = trpl.loglik(X[:2], ini, lengths, Time, 128, T, obs0, devices=[0, 0, 0, 0])
    assert np.array_equal(got, trpl.loglik(X[:2], ini, lengths, Time, 128, T, obs0))
    with pytest.raises(trpl.TrplError):
        trpl.loglik(X, ini, lengths, Time, 128, T, obs0, devices=[0, 99])
    # shards large enough for the two-systems-per-wavefront kernel: other partners, same bits
    Xb = trpl.workloads.samples(10243, seed=6)
    obs2 = [np.full(41, 20.0)] * 3
    one = trpl.loglik(Xb, ini, lengths, 1.0, 128, 40, obs2)
    assert np.array_equal(trpl.loglik(Xb, ini, lengths, 1.0, 128, 40, obs2, devices=[0, 0]), one)


# ---- two systems per wavefront (stepper_pair_impl.hpp): the kernel of every launch that fills the chip ----
def _pair_batch(trpl, S, T):
    lib = trpl._abi.lib()
    # asserted, not skipped: these sizes make the library choose the paired kernel on its own on an MI355X
    # (tests/test_gpu_round2.py forces it per call with TRPL_FLAG_KERNEL_PAIR and compares with the oracle)
    assert lib.trpl_kernel_variant(3 * S, 128, T, 0) == trpl._abi.KERNEL_FAST_PAIR
    assert lib.trpl_kernel_variant(3 * S, 128, T, trpl._abi.FLAG_STRICT) == trpl._abi.KERNEL_STRICT
    X = trpl.workloads.samples(S, seed=11)
    ini, lengths = trpl.workloads.power_scan(128)
    return X, ini, lengths


def test_paired_kernel_matches_strict_with_identical_iteration_counts(trpl, gpu):
    """Parity of the paired kernel at a size where it is the one that runs (5123 samples x 3 curves, odd
    tail included): against STRICT (bit-identical to the reference) every system takes exactly the
    same number of inner iterations and the likelihoods agree to 1e-9."""
    S, T, Time = 5123, 200, 5.0
    X, ini, lengths = _pair_batch(trpl, S, T)
    obs = [np.full(T + 1, 20.0) - 0.02 * np.arange(T + 1)] * 3
    fi, si = {}, {}
    pf = trpl.loglik(X, ini, lengths, Time, 128, T, obs, info=fi)
    ps = trpl.loglik(X, ini, lengths, Time, 128, T, obs, info=si, strict=True)
    assert not fi["status"].any() and not si["status"].any()
    assert np.array_equal(fi["iters_total"], si["iters_total"])
    assert np.max(np.abs(pf - ps) / np.abs(ps)) < 1e-9


def test_paired_kernel_result_does_not_depend_on_the_partner(trpl, gpu):
    """A system's result is bit-for-bit the same whichever sample shares its wavefront and whichever
    half it sits in: drop the first sample (every pairing changes, every system changes half)."""
    S, T, Time = 5122, 100, 2.5
    X, ini, lengths = _pair_batch(trpl, S, T)
    obs = [np.full(T + 1, 20.0)] * 3
    a, b = {}, {}
    pa = trpl.loglik(X, ini, lengths, Time, 128, T, obs, info=a)
    pb = trpl.loglik(X[1:], ini, lengths, Time, 128, T, obs, info=b)
    assert np.array_equal(pa[1:], pb)
    assert np.array_equal(a["sse"][:, 1:], b["sse"]) and np.array_equal(a["iters_total"][:, 1:], b["iters_total"])
    # and the PL-storing mode (pvSim): same kernel, same independence
    Xp = np.concatenate([X, X, X])[:, :12]                                   # one curve per call: 15 366 systems
    assert trpl._abi.lib().trpl_kernel_variant(len(Xp) - 1, 128, T, 0) == trpl._abi.KERNEL_FAST_PAIR
    pl_a = trpl.solve_pl(Xp, lengths[0], Time, 128, T, ini[0])[0]
    pl_b = trpl.solve_pl(Xp[1:], lengths[0], Time, 128, T, ini[0])[0]
    assert pl_a.shape == (3 * S, T + 1) and np.array_equal(pl_a[1:], pl_b)
    assert np.array_equal(pl_a[:S], pl_a[S:2 * S])                           # same sample, other partner and half


def test_paired_kernel_isolates_a_broken_system_from_its_partner(trpl, gpu):
    """NaN / zero-lifetime / non-converging samples are flagged (status, sse = inf) and their wavefront
    partners come out bit-identical to a run without them."""
    S, T, Time = 5120, 60, 1.5
    X, ini, lengths = _pair_batch(trpl, S, T)
    obs = [np.full(T + 1, 20.0)] * 3
    clean = {}
    pc = trpl.loglik(X, ini, lengths, Time, 128, T, obs, info=clean)
    bad = X.copy()
    bad[10, 9] = np.nan            # tau_n
    bad[21, 4] = np.inf            # radiative rate
    bad[300, 9] = 0.0              # degenerate lifetime (still solvable)
    bad[301, 2] = -1e9             # negative diffusivity: whatever the solve does, it stays in its half
    info = {}
    pb = trpl.loglik(bad, ini, lengths, Time, 128, T, obs, info=info)
    broken = np.array([10, 21, 300, 301])
    ok = np.setdiff1d(np.arange(S), broken)
    assert np.array_equal(pb[ok], pc[ok])
    assert np.array_equal(info["sse"][:, ok], clean["sse"][:, ok])
    assert np.array_equal(info["iters_total"][:, ok], clean["iters_total"][:, ok])
    assert (info["status"][:, [10, 21]] > 0).all() and np.isinf(info["sse"][:, [10, 21]]).all()
    assert not np.isfinite(pb[[10, 21]]).any()
    assert np.isfinite(pb[ok]).all()


def test_paired_kernel_mixed_convergence_matches_strict(trpl, gpu):
    """With a small iteration cap some systems are flagged at different steps while their partners go
    on: status (the step), iteration totals and the surviving likelihoods equal STRICT's."""
    S, T, Time = 5120, 30, 0.75
    X, ini, lengths = _pair_batch(trpl, S, T)
    obs = [np.full(T + 1, 20.0)] * 3
    fi, si = {}, {}
    pf = trpl.loglik(X, ini, lengths, Time, 128, T, obs, info=fi, MAX=60)
    ps = trpl.loglik(X, ini, lengths, Time, 128, T, obs, info=si, MAX=60, strict=True)
    frac = (si["status"] > 0).mean()
    assert 0.02 < frac < 0.98, frac                  # the cap must bite on some systems only
    assert np.array_equal(fi["status"], si["status"])
    assert np.array_equal(fi["iters_total"], si["iters_total"])
    live = ~(si["status"] > 0).any(axis=0)
    assert np.array_equal(np.isinf(pf), np.isinf(ps))
    assert np.max(np.abs(pf[live] - ps[live]) / np.abs(ps[live])) < 1e-9


# ---- posterior core (csrc/posterior.hip) against the reference's own outputs and the oracle ----
def test_posterior_core_matches_the_reference(trpl, gpu, golden):
    """trpl_amd.posterior (same names / arguments as Visualization/utils.py) on the GPU vs the golden made
    by the reference's functions: weights to 1e-13 relative, moments to 1e-11, histograms to 1e-10."""
    po = trpl.posterior
    g = golden("posterior")
    X, LL = po.filter_nan(g["X"], g["LL"])
    P = po.temper(LL, float(g["tf"]) / 2.0, 2.0)
    assert P.shape == g["P"].shape and np.allclose(P, g["P"], rtol=1e-13, atol=0) and abs(P.sum() - 1) < 1e-13
    assert (P[np.isinf(LL)] == 0).all()
    names = [str(n) for n in g["names"]]
    cols = {n: (np.log10(X[:, i]) if lg else X[:, i]) for n, i, lg in zip(names, g["col_index"], g["col_log"])}
    ws = float(np.sum(P ** 2))
    for k, n in enumerate(names):
        assert np.isclose(po.w_mean(cols[n], P), g["mean"][k], rtol=1e-12)
        assert np.isclose(po.w_variance(cols[n], P), g["var"][k], rtol=1e-10)
        assert np.isclose(po.w_sample_var(cols[n], P, ws), g["sstd"][k], rtol=1e-10)
        assert np.isclose(po.w_skew(cols[n], P), g["skew"][k], rtol=1e-9)
        assert np.isclose(po.w_kurtosis(cols[n], P), g["kurt"][k], rtol=1e-9)
        assert po.credible_interval(cols[n], g["P"]) == tuple(g["ci"][k])
    assert np.isclose(po.covariance(cols["p0"], cols["B"], P), g["cov"][0, 3], rtol=1e-9)
    summ = po.summarize(cols, P)
    assert np.allclose(summ["mean"], g["mean"], rtol=1e-12) and np.allclose(summ["covariance"], g["cov"], rtol=1e-9, atol=1e-18)
    assert np.allclose(summ["sample_std"], g["sstd"], rtol=1e-10) and np.isclose(summ["w2"], g["ws"], rtol=1e-12)
    limits = {n: tuple(g["limits"][k]) for k, n in enumerate(names)}
    secondary = {n: False for n in names}
    for k, n in enumerate(names):
        dens, e = po.marginalize_1D(P, limits, int(g["bins"]), secondary, n, cols[n])
        assert np.array_equal(e, g["edges"][k]) and np.allclose(dens, g["h1"][k], rtol=1e-10, atol=1e-14), n
    for (a, b), h in zip(g["pairs"], g["h2"]):
        dens, xc, yc = po.marginalize_2D(P, limits, int(g["bins"]), secondary, (names[a], names[b]), cols[names[a]], cols[names[b]])
        assert dens.shape == h.shape and xc.shape == yc.shape == (h.shape[1] + 1, h.shape[0] + 1)
        assert np.allclose(dens, h, rtol=1e-10, atol=1e-15)


def test_posterior_core_at_scale_and_edges(trpl, gpu, oracle):
    """1e6 samples (more than one pass of the fixed grid) against the CPU oracle; bin-edge rules on values
    that sit exactly on edges, outside the range and NaN; a 2-D histogram too large for the LDS bins."""
    from oracle import posterior as op
    po = trpl.posterior
    rng = np.random.default_rng(5)
    S = 1_000_003
    LL = -1e5 * rng.random(S) ** 2
    LL[::1000] = -np.inf
    V = np.stack([rng.normal(3.0, 2.0, S), rng.uniform(-1, 1, S), rng.lognormal(0, 1, S)])
    W = po.weights(LL, 4.0e3)
    Wo = op.weights(LL, 4.0e3)
    assert np.allclose(W, Wo, rtol=1e-12, atol=0)
    s, c = po.moments(V, W)
    assert abs(s[0] - 1) < 1e-12
    for d in range(3):
        assert np.isclose(s[2 + d] / s[0], op.w_mean(V[d], Wo), rtol=1e-11)
        assert np.isclose(c[d, d] / s[0], op.w_variance(V[d], Wo), rtol=1e-10)
        for e_ in range(3):
            assert np.isclose(c[d, e_] / s[0], op.covariance(V[d], V[e_], Wo), rtol=1e-8, atol=1e-12)
    # centring about given means (the sharded second pass)
    m = np.array([3.0, 0.0, 1.5])
    _, c2 = po.moments(V, W, mean_in=m)
    assert np.isclose(c2[0, 1], np.sum((V[0] - 3.0) * (V[1] - 0.0) * Wo), rtol=1e-8, atol=1e-12)
    # edges: numpy's own histogram on the reference's edge array is the checker here
    bins, lo, hi = 10, 0.1, 0.9
    e = po.bin_edges(lo, hi, bins)
    x = np.concatenate([e, e[:-1] + 1e-17, np.nextafter(e, -1), np.nextafter(e, 2), [np.nan, -1.0, 5.0], rng.uniform(0, 1, 5000)])
    w = rng.random(x.size)
    ok = ~np.isnan(x)
    want = np.histogram(x[ok], bins=e, weights=w[ok])[0]
    assert np.allclose(po.hist(x, w, lo, hi, bins), want, rtol=1e-12, atol=1e-15)
    assert np.array_equal(po.hist(x, None, lo, hi, bins), np.histogram(x[ok], bins=e)[0])
    xb = yb = 100                                                             # 10 000 bins: global-atomic path
    h2 = po.hist(V[1], W, -1, 1, xb, y=V[0], ylo=-3, yhi=9, ybins=yb)
    want2 = np.histogram2d(V[1], V[0], bins=[po.bin_edges(-1, 1, xb), po.bin_edges(-3, 9, yb)], weights=Wo)[0]
    assert np.allclose(h2, want2, rtol=1e-9, atol=1e-15)
    with pytest.raises(trpl.TrplError):
        po.hist(x, w, 1.0, 1.0, bins)
    assert po.weights(np.zeros(0)).shape == (0,)


# ---- device sampler (csrc/sampler.hip) against the reference's own draws ----
def test_device_sampler_draws_the_reference_stream(trpl, gpu, golden):
    """trpl_sample_box vs the reference's random_grid after numpy.random.seed(42) (sampler.npz holds its
    output): fixed and linear columns bit-identical, log-uniform columns to 2 ulp (device pow vs host pow);
    sizes that end inside / exactly on a 624-word block; the make_grid overrides."""
    sm = trpl.sampler
    lo, hi, lg = sm.DEFAULT_MINX * sm.UNIT_CONVERSIONS, sm.DEFAULT_MAXX * sm.UNIT_CONVERSIONS, sm.DEFAULT_DO_LOG
    g = golden("sampler")
    lin = np.array([not l for l in lg])
    for key in ("X4", "X64"):
        want = g[key]
        got = sm.random_grid_device(g["minX"] * g["unit"], g["maxX"] * g["unit"], g["do_log"], len(want), seed=42)
        assert np.array_equal(got[:, lin], want[:, lin]), key
        assert np.allclose(got[:, ~lin], want[:, ~lin], rtol=5e-16, atol=0), key
    for S in (1, 311, 312, 313, 624, 5000):                     # 312 doubles per regenerated block
        want = sm.default_box(42, S)
        got = sm.random_grid_device(lo, hi, lg, S, seed=42)
        assert np.array_equal(got[:, lin], want[:, lin]) and np.allclose(got, want, rtol=5e-16, atol=0), S
    want = sm.default_box(7, 1000)
    flags = {"override_equal_mu": True, "override_equal_s": True, "override_equal_auger": True}
    got = sm.random_grid_device(lo, hi, lg, 1000, seed=7, sim_flags=flags)
    assert np.array_equal(got[:, 2], want[:, 3]) and np.array_equal(got[:, 3], want[:, 3])
    assert np.allclose(got[:, 6], want[:, 5], rtol=5e-16) and np.allclose(got[:, 8], want[:, 7], rtol=5e-16)
    # the samples it produces drive the solver like the host-drawn ones
    import torch
    X = torch.empty((4096, 13), dtype=torch.float64, device="cuda")
    trpl.device.sample_box_device(X, lo, hi, lg, seed=42)
    assert np.allclose(X.cpu().numpy(), sm.default_box(42, 4096), rtol=5e-16, atol=0)
    with pytest.raises(trpl.TrplError):
        sm.random_grid_device(hi, lo, lg, 4)


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
    # tests/test_gpu_round4.py::test_more_than_sixteen_curves_per_fused_call), the first sixteen keep their bits
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


def test_host_calls_from_several_threads_overlap_and_agree(trpl, gpu):
    """Host-buffer entry points are re-entrant (private stream, stream-ordered allocations, thread-local
    error string): eight threads solving different curves / sample sets at once return exactly what
    the same calls return one after the other, and an error in one thread stays in that thread."""
    from concurrent.futures import ThreadPoolExecutor
    ini, lengths = trpl.workloads.power_scan(128)
    jobs = [(trpl.workloads.samples(200 + 17 * k, seed=30 + k)[:, :12], k % 3) for k in range(8)]

    def run(job):
        X, c = job
        pl, st, it, _ = trpl.solve_pl(X, lengths[c], 2.0, 128, 80, ini[c])
        lp = np.log10(np.maximum(pl, 1e-300))
        trpl.fastlog(pl, 1e-300)
        return pl, lp, st, it

    serial = [run(j) for j in jobs]
    with ThreadPoolExecutor(max_workers=8) as pool:
        threaded = list(pool.map(run, jobs))
    for a, b in zip(serial, threaded):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        assert np.allclose(a[0], a[1], rtol=1e-15, atol=0)

    def bad(_):
        try:
            trpl.solve_pl(jobs[0][0], lengths[0], 2.0, 100, 80, ini[0][:100])       # L not a power of two
        except trpl.TrplError as e:
            return str(e)
        return None
    with ThreadPoolExecutor(max_workers=2) as pool:
        msgs = list(pool.map(bad, range(4))) + [r[0].shape for r in pool.map(run, jobs[:2])]
    assert all(isinstance(m, str) and "power of two" in m for m in msgs[:4])
