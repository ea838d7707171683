"""Round 3: the regime the benchmark times, against the ORACLE inside the GPU suite, and the cancellation floor as a
tested contract (round-2 review, "What's weak" 1).

  * Power_scan x 64 samples (BASELINE configs[0]) over the bench's window, T = 8000 steps of 0.025 ns: STRICT is the
    oracle bit for bit (PL as bit patterns, iteration totals); both FAST kernels hold the oracle's iteration totals,
    its PL to the stated bound and its likelihood to 1e-8 on every sample that stays above the floor;
  * short-lifetime samples that DO reach the floor: floor_col is the same in every arithmetic and equals the column
    the oracle's own PL gives; before it PL and the squared-error sum follow the oracle, and a window truncated there
    is floor-free;
  * floor_col through the multi-device and off-grid entry points.

The floor (include/trpl.h): r(t) = PL(t) / (B L n0p0) < TRPL_PL_FLOOR_EXCESS = 1e-4; the measured bound on the
deviation between FAST and the reference evaluation is 1e-9 + K / r with the K of include/trpl.h (profiles/r4_floor_study_*.json)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT = 0.025
KERNELS = [dict(strict=True), dict(kernel="single"), dict(kernel="pair")]
IDS = ["strict", "single", "pair"]


def excess_scale(X, length, L=128):
    """B L n0p0 in the units of PL (nm^-2 ns^-1): the non-dimensional rate * L * N0 * P0 of pvSimPCR.py:327-331 divided
    by dx^2 dt (:393) -- the time step cancels."""
    dx = length / L
    return X[:, 4] * L * X[:, 0] * X[:, 1] * dx


ENVELOPE_K = 5e-13          # TRPL_PL_ENVELOPE_K_THICK (include/trpl.h): every film of this module is 2000 nm at L = 128


def deviation_bound(pl_ref, scale):
    """The header's envelope, 1e-9 + TRPL_PL_ENVELOPE_K_THICK / r per point (measured prefactor 2e-13, tools/floor_study.py);
    inf where the reference PL is not positive."""
    r = pl_ref / scale[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        b = 1e-9 + ENVELOPE_K / r
    b[~(pl_ref > 0)] = np.inf
    return b


def first_below(pl, thr):
    """first column with pl < thr (or non-positive / NaN), -1 if none"""
    bad = ~(pl >= thr[:, None])
    return np.where(bad.any(axis=1), bad.argmax(axis=1), -1).astype(np.int32)


@pytest.fixture(scope="module")
def long_window(gpu, oracle):
    """configs[0] at the bench's window, solved once by the oracle (16 threads: seconds)."""
    w = gpu.workloads
    T, L, S = 8000, 128, 64
    Time = T * DT
    ini, lens = w.power_scan(L)
    X = w.samples(S)
    mark = (w.MARKED_POINT * gpu.UNIT_CONVERSIONS)[None, :-1]
    ref = [oracle.pvsim(X[:, :12], lens[c], Time, L, T, ini[c], nthreads=16) for c in range(3)]
    obs = [np.log10(oracle.pvsim(mark, lens[c], Time, L, T, ini[c])["plI"][0]) for c in range(3)]
    P = np.zeros(S)
    sse = np.zeros((3, S))
    for c in range(3):
        lg = ref[c]["plI"].copy()
        oracle.fastlog(lg)                                   # probs.fastlog
        Pc = np.zeros(S)
        oracle.prob(Pc, lg, obs[c], np.ascontiguousarray(X[:, -1]))      # probs.prob: P -= sum (lg + mag - obs)^2
        sse[c] = -Pc
        P += Pc
    return dict(T=T, L=L, S=S, Time=Time, ini=ini, lens=lens, X=X, ref=ref, obs=obs, P=P, sse=sse)


@pytest.mark.parametrize("mode", KERNELS, ids=IDS)
def test_bench_window_pl_and_iteration_totals_against_the_oracle(gpu, long_window, mode):
    g = long_window
    for c in range(3):
        want = g["ref"][c]
        assert not want["status"].any()
        pl, st, it, _ = gpu.solve_pl(g["X"][:, :12], g["lens"][c], g["Time"], g["L"], g["T"], g["ini"][c], **mode)
        assert not st.any()
        assert np.array_equal(it, want["iters_total"]), (c, int((it != want["iters_total"]).sum()))
        if mode.get("strict"):
            assert np.array_equal(pl.view(np.int64), want["plI"].view(np.int64))       # bit patterns
            continue
        dev = np.abs(pl / want["plI"] - 1)
        bound = deviation_bound(want["plI"], excess_scale(g["X"], g["lens"][c]))
        assert (dev <= bound).all(), (c, float(np.nanmax(dev / bound)))
        # the review's wording: every point >= 1e-12 of the curve's start that is also above the floor, to 2e-8
        above = (want["plI"] >= 1e-12 * want["plI"][:, :1]) & (want["plI"] >= 1e-4 * excess_scale(g["X"], g["lens"][c])[:, None])
        assert dev[above].max() < 2e-8 and above.mean() > 0.9


@pytest.mark.parametrize("mode", KERNELS, ids=IDS)
def test_bench_window_fused_likelihood_and_floor_indicator_against_the_oracle(gpu, long_window, mode):
    g = long_window
    info = {}
    P = gpu.loglik(g["X"], g["ini"], g["lens"], g["Time"], g["L"], g["T"], g["obs"], info=info, **mode)
    assert not info["status"].any()
    for c in range(3):
        assert np.array_equal(info["iters_total"][c], g["ref"][c]["iters_total"])
        # the indicator is what the oracle's own PL says
        want_col = first_below(g["ref"][c]["plI"], 1e-4 * excess_scale(g["X"], g["lens"][c]))
        assert np.array_equal(info["floor_col"][c], want_col), c
    clear = (info["floor_col"] < 0).all(axis=0)              # samples that never reach the floor, on any curve
    assert clear.sum() >= 0.8 * g["S"]
    rel = np.abs(P - g["P"]) / np.abs(g["P"])
    assert rel[clear].max() < 1e-8, float(rel[clear].max())
    rel_sse = np.abs(info["sse"] - g["sse"]) / g["sse"]
    assert rel_sse[info["floor_col"] < 0].max() < 1e-8
    if mode.get("strict"):                                   # the reference evaluation: every sample, floor or not
        assert rel.max() < 1e-12


@pytest.fixture(scope="module")
def decayed(gpu, oracle):
    """Samples with tau_n, tau_p of 0.3 .. 3 ns: gone by e^-17 .. e^-170 inside a 50 ns window."""
    w = gpu.workloads
    T, L, S = 2000, 128, 48
    Time = T * DT
    ini, lens = w.power_scan(L)
    X = w.samples(S, seed=7)
    rng = np.random.default_rng(3)
    X[:, 9] = 10 ** rng.uniform(np.log10(0.3), np.log10(3.0), S)
    X[:, 10] = X[:, 9] * 10 ** rng.uniform(-0.3, 0.3, S)
    ref = [oracle.pvsim(X[:, :12], lens[c], Time, L, T, ini[c], nthreads=16) for c in range(3)]
    return dict(T=T, L=L, S=S, Time=Time, ini=ini, lens=lens, X=X, ref=ref)


def test_floor_indicator_and_contract_on_samples_that_reach_the_floor(gpu, oracle, decayed):
    g = decayed
    T, S = g["T"], g["S"]
    obs = [np.linspace(18.0, 2.0, T + 1)] * 3                # any observation set: the contract is about PL and sse
    cols, sses, pls = {}, {}, {}
    for name, mode in zip(IDS, KERNELS):
        info = {}
        gpu.loglik(g["X"], g["ini"], g["lens"], g["Time"], g["L"], T, obs, info=info, **mode)
        assert not info["status"].any()
        cols[name], sses[name] = info["floor_col"], info["sse"]
        pls[name] = [gpu.solve_pl(g["X"][:, :12], g["lens"][c], g["Time"], g["L"], T, g["ini"][c], **mode)[0] for c in range(3)]
    # the same column in every arithmetic, and the one the oracle's PL gives
    want = np.stack([first_below(g["ref"][c]["plI"], 1e-4 * excess_scale(g["X"], g["lens"][c])) for c in range(3)])
    assert (want >= 0).mean() > 0.5 and (want[want >= 0] > 50).all()
    for name in IDS:
        assert np.array_equal(cols[name], want), name
    mag = np.ascontiguousarray(g["X"][:, -1])
    for c in range(3):
        ref = g["ref"][c]["plI"]
        assert np.array_equal(pls["strict"][c].view(np.int64), ref.view(np.int64))
        before = np.arange(T + 1)[None, :] < np.where(want[c] >= 0, want[c], T + 1)[:, None]
        for name in ("single", "pair"):
            dev = np.abs(pls[name][c] / ref - 1)
            assert dev[before].max() < 2e-8, (name, c, float(dev[before].max()))       # the contract, columns before floor_col
            scale = excess_scale(g["X"], g["lens"][c])
            physical = ref >= 1e-10 * scale[:, None]                 # towards r ~ 1e-13 both values become rounding noise
            worst = float(np.max((dev / deviation_bound(ref, scale))[physical]))
            assert worst <= 1.0, (name, c, worst)                    # the header's envelope itself, no extra factor
            # squared-error sum over the window before floor_col: the oracle's, to 1e-8
            def sse_before(pl):
                lg = np.log10(np.maximum(pl, np.finfo(float).tiny))
                e = np.where(before, lg + mag[:, None] - obs[c][None, :], 0.0)
                return (e * e).sum(axis=1)
            a, b = sse_before(pls[name][c]), sse_before(ref)
            assert np.max(np.abs(a - b) / b) < 1e-8
    # ... and past it the values are arbitrary, as documented: the reference order returns rounding noise of either
    # sign (some of it clamped), the default arithmetic follows the state onto the clamp -- their sse differ by far
    # more than any tolerance on at least some of these systems, which is why the indicator exists
    on_floor = want >= 0
    gap = np.abs(sses["pair"] - sses["strict"]) / sses["strict"]
    assert gap[on_floor].max() > 1e-3
    # a window that ends before the earliest floor column is floor-free, and there the fused likelihood is the oracle's
    n = int(want[want >= 0].min())
    info = {}
    P = gpu.loglik(g["X"], g["ini"], g["lens"], g["Time"], g["L"], T, [o[:n] for o in obs], info=info, kernel="pair")
    assert (info["floor_col"] == -1).all()
    Pw = np.zeros(S)
    for c in range(3):
        lg = g["ref"][c]["plI"][:, :n].copy()
        oracle.fastlog(lg)
        oracle.prob(Pw, lg, obs[c][:n], mag)
    assert np.max(np.abs(P - Pw) / np.abs(Pw)) < 1e-8


def test_floor_indicator_through_the_sharded_and_off_grid_entry_points(gpu, decayed):
    g = decayed
    T, S = 600, 21
    Time = T * DT
    X = g["X"][:S]
    obs = [np.linspace(18.0, 12.0, T + 1)] * 3
    one, multi, off = {}, {}, {}
    gpu.loglik(X, g["ini"], g["lens"], Time, g["L"], T, obs, info=one, kernel="single")
    gpu.loglik(X, g["ini"], g["lens"], Time, g["L"], T, obs, info=multi, kernel="single", devices=[0, 0, 0])
    assert np.array_equal(one["floor_col"], multi["floor_col"]) and (one["floor_col"] >= 0).any()
    # observation times off the grid: the indicator counts grid steps
    times = [np.linspace(0.0, Time, 301)[1:-1] + 0.003] * 3
    gpu.loglik(X, g["ini"], g["lens"], Time, g["L"], T, [np.full(299, 15.0)] * 3, info=off, times=times, kernel="single")
    reached = one["floor_col"] >= 0
    assert np.array_equal(off["floor_col"][reached & (one["floor_col"] < T - 2)], one["floor_col"][reached & (one["floor_col"] < T - 2)])
    # the device-resident call, with and without the output
    import torch
    dev = torch.device("cuda:0")
    tX = torch.from_numpy(X).to(dev); tini = torch.from_numpy(g["ini"]).to(dev)
    tobs = torch.from_numpy(np.stack(obs)).to(dev)
    P = torch.zeros(S, dtype=torch.float64, device=dev); sse = torch.empty((3, S), dtype=torch.float64, device=dev)
    fc = torch.full((3, S), -7, dtype=torch.int32, device=dev)
    gpu.device.loglik_device(tX, tini, g["lens"], Time, g["L"], T, tobs, [T + 1] * 3, P, sse, floor_col=fc,
                             flags=gpu.FLAG_KERNEL_SINGLE)
    torch.cuda.synchronize()
    assert np.array_equal(fc.cpu().numpy(), one["floor_col"]) and np.array_equal(sse.cpu().numpy(), one["sse"])


def test_fp32_state_drifts_over_long_windows_and_says_so(gpu):
    """BASELINE configs[4] as worded (L = 512, fp32).  An fp32 STATE cannot hold the BDF history differences (6e-8 per
    level against a change per step of dt / tau ~ 5e-5): the 60-step window of the round-1 test (2e-3) hid that the PL
    error grows to percents and, on the decayed tail, tens of percents.  The library therefore refuses
    TRPL_FLAG_FP32 beyond TRPL_FP32_MAX_STEPS = 256 steps unless TRPL_FLAG_FP32_LONG is given, and this test keeps the
    measured drift on record against the fp64 stepper (which the oracle pins at L = 512)."""
    w = gpu.workloads
    L, length = 512, 2000.0
    X = w.samples(24, seed=5)
    ini = np.stack([w.beer_lambert(A, length, L) for A in w.POWER_SCAN_A_CM3])
    with pytest.raises(gpu.TrplError, match="FP32_LONG"):
        gpu.solve_pl(X[:, :-1], length, 257 * DT, L, 257, ini[0], fp32=True, tol=3)
    worst = {}
    for T in (60, 256, 2000):
        ref = gpu.solve_pl(X[:, :-1], length, T * DT, L, T, ini[1], tol=7)[0]
        pl, st, _, _ = gpu.solve_pl(X[:, :-1], length, T * DT, L, T, ini[1], fp32="long" if T > 256 else True, tol=3)
        assert not st.any()
        ok = ref >= 1e-6 * ref[:, :1]                            # measurable PL only
        worst[T] = float(np.max(np.abs(pl / ref - 1)[ok]))
    assert worst[60] < 5e-3 and worst[256] < 5e-2               # the window the flag alone allows: screening quality
    assert worst[2000] > 3 * worst[256] and worst[2000] > 1e-2  # and it keeps growing: percents by 2000 steps
    # the likelihood entry points apply the same rule (steps up to the last observation)
    obs = [np.full(300, 15.0)] * 3
    with pytest.raises(gpu.TrplError, match="FP32_LONG"):
        gpu.loglik(X, ini, length, 400 * DT, L, 400, obs, fp32=True, tol=3)
    P = gpu.loglik(X, ini, length, 400 * DT, L, 400, [o[:200] for o in obs], fp32=True, tol=3)      # 199 steps: allowed
    assert np.isfinite(P).all()


def test_multi_device_call_is_ordered_against_the_callers_torch_stream(gpu):
    """trpl_loglik_multi_dev works on the handle's own streams.  MultiDevice.loglik(order=True) makes them wait for what
    the caller's torch stream holds (trpl_multi_wait_stream) and makes that stream wait for the result
    (trpl_multi_release_stream), on the device: parameters written by a copy that is still QUEUED behind ~0.3 s of
    other work when loglik() is called are the ones the solve reads, and a read of P_full queued right after the call
    sees the gathered vector -- no host synchronisation anywhere in between (round-2 advisor finding)."""
    import torch
    w = gpu.workloads
    dev = torch.device("cuda:0")
    L, T, S = 128, 40, 1500
    Time = T * DT
    ini, lens = w.power_scan(L)
    Xa, Xb = w.samples(S, seed=21), w.samples(S, seed=22)
    ini_d = torch.from_numpy(ini).to(dev)
    obs = torch.full((3, T + 1), 15.0, dtype=torch.float64, device=dev)
    want = {}
    for name, Xh in (("a", Xa), ("b", Xb)):
        Xd = torch.from_numpy(Xh).to(dev)
        P = torch.zeros(S, dtype=torch.float64, device=dev); sse = torch.empty((3, S), dtype=torch.float64, device=dev)
        gpu.device.loglik_device(Xd, ini_d, lens, Time, L, T, obs, [T + 1] * 3, P, sse, flags=gpu.FLAG_KERNEL_SINGLE)
        torch.cuda.synchronize()
        want[name] = P.clone()
    assert not torch.equal(want["a"], want["b"])
    X = torch.from_numpy(Xa).to(dev)
    Xb_pinned = torch.from_numpy(Xb).pin_memory()
    Pf = torch.zeros(S, dtype=torch.float64, device=dev)
    out = torch.empty(S, dtype=torch.float64, device=dev)
    with gpu.device.MultiDevice([0]) as md:
        md.loglik([X], [ini_d], lens, Time, L, T, [obs], [T + 1] * 3, [Pf], flags=gpu.FLAG_KERNEL_SINGLE)      # warm: RCCL channels up
        md.synchronize()
        torch.cuda.synchronize()
        torch.cuda._sleep(int(6e8))                          # ~0.3 s of work ahead of the copy on the torch stream
        X.copy_(Xb_pinned, non_blocking=True)                # still queued when loglik() is called
        md.loglik([X], [ini_d], lens, Time, L, T, [obs], [T + 1] * 3, [Pf], flags=gpu.FLAG_KERNEL_SINGLE)
        out.copy_(Pf)                                        # queued on the torch stream right behind the call
        torch.cuda.synchronize()
        md.synchronize()
    assert torch.equal(out, want["b"])


def test_likelihoods_do_not_depend_on_the_pairing_rule(gpu, tmp_path):
    """The paired stepper pairs two curves of one sample (trpl_pair_table) or, with TRPL_PAIR_CURVES=0, adjacent samples
    of one curve.  Scheduling only: likelihoods, per-curve sums, iteration totals, status and floor columns are the
    same bits -- Twothick (two groups of three curves: same-sample pairs and cross-sample leftovers), an odd batch (the
    last period has one sample), two child processes because the switch is read once per process."""
    import os, subprocess, sys
    from conftest import ROOT
    code = ("import sys, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "import trpl_amd\n"
            "w = trpl_amd.workloads\n"
            "ini, lens = w.twothick(128)\n"
            "X = w.samples(4099, seed=17)\n"
            "T = 300\n"
            "obs = [np.linspace(17.0, 14.0, T + 1)] * 6\n"
            "info = {}\n"
            "P = trpl_amd.loglik(X, ini, lens, T * 0.025, 128, T, obs, info=info, kernel='pair')\n"
            "np.savez(sys.argv[1], P=P, sse=info['sse'], it=info['iters_total'], st=info['status'], fc=info['floor_col'])\n") % ROOT
    out = {}
    for v in ("0", "1"):
        path = str(tmp_path / ("pair%s.npz" % v))
        env = dict(os.environ, TRPL_PAIR_CURVES=v, TRPL_AUTOBUILD="0")
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[v] = np.load(path)
    for k in ("P", "sse", "it", "st", "fc"):
        assert np.array_equal(out["0"][k], out["1"][k]), k
    assert np.isfinite(out["1"]["P"]).all() and not out["1"]["st"].any()


def test_wide_box_fuzz_of_the_two_fast_kernels(gpu):
    """Differential fuzz (tools/fuzz_pair.py in small): a parameter box 2-4 decades wider than the reference's on every
    axis, Twothick's six curves, a small iteration cap so that hundreds of systems are flagged -- the one-system and the
    paired kernel (curves of one sample in a wavefront, flagged partners parked beside live ones) flag the same systems
    at the same step, agree on the iteration totals of all but a handful of the others and on their sums to rounding;
    nothing non-finite leaks from a flagged system into its partner."""
    sm, w = gpu.sampler, gpu.workloads
    S, T = 3000, 120
    lo = np.array([1e8, 1e12, 0.01, 0.01, 1e-13, 1e-3, 1e-3, 1e-32, 1e-32, 0.1, 0.1, 0.1, 0])
    hi = np.array([1e8, 1e18, 500, 500, 1e-8, 1e5, 1e5, 1e-26, 1e-26, 1e4, 1e4, 0.1, 0])
    lg = np.array([1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0])
    X = sm.random_grid(lo * sm.UNIT_CONVERSIONS, hi * sm.UNIT_CONVERSIONS, lg, S, rng=np.random.RandomState(123))
    ini, lens = w.twothick(128)
    obs = [np.full(T + 1, 18.0) - 0.01 * np.arange(T + 1)] * len(lens)
    res = {}
    for k in ("single", "pair"):
        info = {}
        gpu.loglik(X, ini, lens, T * DT, 128, T, obs, info=info, MAX=400, kernel=k)
        res[k] = info
    a, b = res["single"], res["pair"]
    flagged = a["status"] != 0
    assert 50 < flagged.sum() < 0.5 * flagged.size
    assert np.array_equal(a["status"], b["status"])
    ok = ~flagged
    dit = np.abs(a["iters_total"][ok] - b["iters_total"][ok])
    assert (dit != 0).mean() < 2e-3 and dit.max() <= 3
    assert np.isfinite(a["sse"][ok]).all() and np.isfinite(b["sse"][ok]).all()
    assert np.isinf(a["sse"][flagged]).all() and np.isinf(b["sse"][flagged]).all()
    clear = ok & (a["floor_col"] < 0) & (b["floor_col"] < 0)
    rel = np.abs(a["sse"][clear] - b["sse"][clear]) / np.abs(a["sse"][clear])
    assert np.median(rel) < 1e-13 and np.quantile(rel, 0.999) < 1e-7
