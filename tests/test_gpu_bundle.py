"""max_sims_per_block > 1 (shared_array_max pvSimPCR.py:83-90): the reference lets the samples of a
bundle iterate until the slowest has converged.  STRICT mode reproduces that bit for bit -- against what the reference
itself produced (tests/golden/pvsim_bundle.npz) and against the oracle on larger batches; the FAST one-system kernel
follows the same path to rounding (L <= 128)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_bundled_strict_is_the_reference_bit_for_bit(gpu, golden):
    g = golden("pvsim_bundle")
    X, T, Time, L = g["X"][:, :12], int(g["T"]), float(g["time"]), int(g["L"])
    for tag, ini, length in (("P", g["iniP"], float(g["lengthP"])), ("T", g["iniT"], float(g["lengthT"]))):
        for m in (3, 2):
            pl, st, it, _ = gpu.solve_pl(X, length, Time, L, T, ini, strict=True, bundle=m)
            assert np.array_equal(pl, g["pl%s%d" % (tag, m)]), (tag, m)
            assert np.array_equal(it, g["it%s%d" % (tag, m)].sum(axis=1)), (tag, m)
            assert not st.any()
        # through the drop-in signature: max_sims_per_block is the 10th positional argument (bayeslib.py:144-146)
        plI = np.empty((len(X), T + 1))
        gpu.pvSim(plI, None, None, None, X, [length, Time, L, T, 1, None, 7, 10000], ini, (128,), 64, 3,
                  init_mode="points", strict=True)
        assert np.array_equal(plI, g["pl%s3" % tag])
        # FAST arithmetic, same bundling: the reference's PL to 1e-9, its iteration counts exactly
        info = {}
        gpu.pvSim(plI, None, None, None, X, [length, Time, L, T, 1, None, 7, 10000], ini, (128,), 64, 3, init_mode="points",
                  info=info)
        assert np.max(np.abs(plI / g["pl%s3" % tag] - 1)) < 1e-9
        assert np.array_equal(info["iters_total"], g["it%s3" % tag].sum(axis=1))
        # every sample on its own differs from that by the solver tolerance, not by rounding
        gpu.pvSim(plI, None, None, None, X, [length, Time, L, T, 1, None, 7, 10000], ini, (128,), 64, 1, init_mode="points")
        d = np.max(np.abs(plI / g["pl%s3" % tag] - 1))
        assert 1e-12 < d < 1e-5


@pytest.mark.parametrize("m,L", [(2, 128), (3, 128), (4, 128), (3, 32), (2, 256), (6, 64), (13, 32), (16, 64), (16, 8)])
def test_bundled_strict_vs_oracle_with_short_last_bundle_and_snapshots(gpu, oracle, m, L):
    w = gpu.workloads
    ini, lens = w.twothick(L)
    S = (4 if m <= 4 else 2) * m + 1                         # a last bundle of one system
    X = w.samples(S, seed=100 + m)[:, :12]
    T, Time = 48, 1.2
    want = oracle.pvsim(X, lens[0], Time, L, T, ini[0], mspb=m, nthreads=4)
    snaps = {}
    pl, st, it, _ = gpu.solve_pl(X, lens[0], Time, L, T, ini[0], strict=True, bundle=m, snap_steps=[0, 7, T],
                                 snapshots=snaps)
    assert np.array_equal(pl, want["plI"]) and np.array_equal(it, want["iters_total"]) and not st.any()
    assert np.isfinite(snaps["plN"]).all() and (snaps["plN"][:, 1] > 0).all()
    for b in range(0, S, m):                                 # a bundle's systems share their iteration total
        assert (it[b:b + m] == it[b]).all()
    # the coupling is real: unbundled totals are smaller for the faster members
    alone = gpu.solve_pl(X, lens[0], Time, L, T, ini[0], strict=True)[2]
    assert (alone <= it).all() and (alone < it).any()


@pytest.mark.parametrize("m,L", [(2, 128), (3, 128), (4, 128), (3, 32), (4, 64), (6, 64), (13, 32), (16, 16)])
def test_bundled_fast_kernel_vs_oracle(gpu, oracle, m, L):
    w = gpu.workloads
    ini, lens = w.twothick(L)
    S = (5 if m <= 4 else 2) * m + 2
    X = w.samples(S, seed=200 + m)[:, :12]
    T, Time = 64, 1.6
    want = oracle.pvsim(X, lens[0], Time, L, T, ini[0], mspb=m, nthreads=4)
    for plT in (1, 4):
        pl, st, it, _ = gpu.solve_pl(X, lens[0], Time, L, T, ini[0], bundle=m, plT=plT)
        assert not st.any() and np.array_equal(it, want["iters_total"])
        assert np.max(np.abs(pl / want["plI"][:, ::plT] - 1)) < 1e-9
    # fused likelihood through the same bundles
    obs = [np.log10(want["plI"][0]) + 0.01]
    info = {}
    P = gpu.loglik(np.hstack([X, np.zeros((S, 1))]), ini[:1], lens[:1], Time, L, T, obs, info=info, bundle=m)
    ref = -np.sum((np.log10(want["plI"]) - obs[0]) ** 2, axis=1)
    assert np.allclose(P, ref, rtol=1e-8) and np.array_equal(info["iters_total"][0], want["iters_total"])


def test_a_bundle_that_reaches_max_iter_is_flagged_as_a_whole(gpu, oracle):
    w = gpu.workloads
    ini, lens = w.twothick(128)
    X = w.samples(7, seed=3)[:, :12]
    T, Time = 40, 1.0
    probe = oracle.pvsim(X, lens[0], Time, 128, T, ini[0], mspb=3, want_step_iters=True)
    cap = int(probe["step_iters"][3].max())                  # the second bundle's slowest step fails at this cap
    want = oracle.pvsim(X, lens[0], Time, 128, T, ini[0], mspb=3, MAX=cap)
    pl, st, it, _ = gpu.solve_pl(X, lens[0], Time, 128, T, ini[0], strict=True, bundle=3, MAX=cap)
    assert np.array_equal(st, want["status"]) and st[3] > 0 and (st[3:6] == st[3]).all()
    assert np.array_equal(it, want["iters_total"])
    assert np.array_equal(np.isnan(pl), np.isnan(want["plI"])) and np.array_equal(pl[~np.isnan(pl)], want["plI"][~np.isnan(pl)])


def test_bundle_flag_is_validated(gpu):
    w = gpu.workloads
    ini, lens = w.twothick(128)
    X = w.samples(4, seed=1)[:, :12]
    with pytest.raises(gpu.TrplError, match="BUNDLE"):
        gpu.solve_pl(X, lens[0], 0.5, 128, 20, ini[0], bundle=2, kernel="pair")
    with pytest.raises(gpu.TrplError, match="BUNDLE" if gpu._abi.has_experimental() else "EXPERIMENTAL=1"):
        gpu.solve_pl(X, lens[0], 0.5, 128, 20, ini[0], bundle=2, mixed=True)
    ini256, lens256 = w.twothick(256)
    with pytest.raises(gpu.TrplError, match="L <= 128"):
        gpu.solve_pl(X, lens256[0], 0.5, 256, 20, ini256[0], bundle=2)
    with pytest.raises(ValueError):
        gpu.solve_pl(X, lens[0], 0.5, 128, 20, ini[0], strict=True, bundle=5)
    # the C ABI itself refuses what the binding refuses: 5 systems per bundle at L = 128, 16 are fine at L = 64
    lib, A = gpu._abi.lib(), gpu._abi
    pl, st = np.zeros((4, 21)), np.zeros(4, np.int32)
    rc = lib.trpl_solve_pl(X.ctypes.data, 4, float(lens[0]), 0.5, 128, 20, 1, 7, 100, ini[0].ctypes.data, pl.ctypes.data, 8, 21,
                           st.ctypes.data, None, A.FLAG_STRICT | (4 << 8), 0, None)
    assert rc == A.ERR_ARG and b"at most 4" in lib.trpl_last_error()
    ini64, lens64 = w.twothick(64)
    rc = lib.trpl_solve_pl(X.ctypes.data, 4, float(lens64[0]), 0.5, 64, 20, 1, 7, 100, ini64[0].ctypes.data, pl.ctypes.data, 8, 21,
                           st.ctypes.data, None, A.FLAG_STRICT | (15 << 8), 0, None)
    assert rc == A.OK and not st.any() and np.isfinite(pl).all()
    # bundle = 1 is the plain call
    a = gpu.solve_pl(X, lens[0], 0.5, 128, 20, ini[0], strict=True, bundle=1)[0]
    assert np.array_equal(a, gpu.solve_pl(X, lens[0], 0.5, 128, 20, ini[0], strict=True)[0])


def test_simulate_passes_max_sims_per_block_through_all_three_routes(gpu, oracle, golden):
    """bayeslib.simulate hands gpu_info['max_sims_per_block'] to the model (bayeslib.py:93,:146); the bundles restart
    with every block of sims_per_gpu samples.  The unfused drop-in loop, the fused single-launch route and the
    PL-resident multi-experiment route all follow the oracle's restatement run the same way (float64 PL buffer)."""
    g = golden("bayes_e2e")
    T, tg, npre = int(g["T"]), g["tgrid"], int(g["npre"])
    X = g["X"]
    e_data = [([tg] * 3, list(g["obs0"]), [None] * 3), ([tg[:npre]] * 3, list(g["obs1"]), [None] * 3)]
    flags = {"load_PL_from_file": False, "log_pl": True, "self_normalize": False}
    want = oracle.simulate_loglik(X, g["ini"], 2000.0, float(g["time"]), 128, T, [(e[0], e[1]) for e in e_data],
                                  pl_dtype=np.float64, sims_per_gpu=5, mspb=3, nthreads=4)
    alone = oracle.simulate_loglik(X, g["ini"], 2000.0, float(g["time"]), 128, T, [(e[0], e[1]) for e in e_data],
                                   pl_dtype=np.float64, sims_per_gpu=5, nthreads=4)
    assert np.max(np.abs(want - alone) / np.abs(alone)) > 1e-9               # the bundling is visible in P
    z = np.zeros(1)
    for extra in ({}, {"fused": True}):
        for n_exp in (2, 1):                                                 # 2 fused experiments: the PL-resident route
            P = np.zeros((n_exp, len(X)))
            sim_params = [2000.0, float(g["time"]), 128, T, 1, (0,), 7, 10000]
            info = {"sims_per_gpu": 5, "num_gpus": 1, "max_sims_per_block": 3, "pl_dtype": np.float64, **extra}
            gpu.simulate(gpu.pvSim, e_data[:n_exp], P, X, [None], [None], 3, sim_params, g["ini"], flags, info, 0,
                         z.copy(), z.copy(), z.copy())
            assert np.max(np.abs(P - want[:n_exp]) / np.abs(want[:n_exp])) < 1e-8, (extra, n_exp)


@pytest.mark.parametrize("strict", [True, False])
def test_bundles_whose_steps_take_one_iteration_hand_the_verdict_buffer_over_cleanly(gpu, oracle, strict):
    """The verdicts of a bundle are double-buffered in LDS by the parity of a counter that runs across time steps.
    Systems that START AT EQUILIBRIUM (no excitation) converge in the first inner iteration of every step, so every
    step ends on the same parity it began with -- the hand-over a per-step parity got wrong (a fast wavefront could
    overwrite the verdict a slow one was still reading; round-2 review).  Mixed with excited systems in the same
    launch so that bundles of both kinds alternate.  Iteration totals equal the oracle's, STRICT PL bit for bit."""
    w = gpu.workloads
    L, m = 128, 4
    ini, lens = w.twothick(L)
    S = 8 * m + 3
    X = w.samples(S, seed=77)[:, :12]
    T, Time = 400, 10.0
    for dN in (np.zeros(L), ini[0]):
        want = oracle.pvsim(X, lens[0], Time, L, T, dN, mspb=m, nthreads=8)
        pl, st, it, _ = gpu.solve_pl(X, lens[0], Time, L, T, dN, strict=strict, bundle=m)
        assert not st.any() and np.array_equal(it, want["iters_total"])
        if not dN.any():
            assert (it == T + 1).all()                       # one iteration per step, t = 0 .. T
            continue                                         # PL of an equilibrium state is pure cancellation
        if strict:
            assert np.array_equal(pl, want["plI"])
        else:
            assert np.max(np.abs(pl / want["plI"] - 1)) < 1e-9
