"""Round 4: the two OTHER single-GPU configurations of BASELINE.json under the oracle at the window the benchmark times
(round-3 review, "Next" 1a) -- until now only Power_scan at L = 128 had a T = 8000 oracle test.

  * configs[2], Twothick (parallel_bayes_gpu.py:71: the three powers on 311 nm and 2000 nm films) x 32 samples x 6 curves
    x T = 8000: STRICT is the oracle bit for bit; both FAST kernels hold the oracle's iteration totals, its PL inside the
    floor envelope of include/trpl.h with the prefactor of the film's grid, its squared-error sums on the floor-free
    systems and its floor_col;
  * configs[4]'s grid, L = 512 (Power_scan profiles, 2000 nm) x 16 samples x 3 curves x T = 8000 through
    stepper_kernel<512>: at tol 7 the oracle's iteration totals and PL to FAST parity, at tol 6 (the setting DESIGN
    recommends for this grid) the oracle's tol-6 iteration totals and the documented 2e-5 against the tol-7 solution;
    the fp32-difference history (TRPL_FLAG_HIST32) is held to its own documented gate.

The oracle needs 7 s + 12 s for these on 8 threads."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from test_gpu_round3 import excess_scale, first_below

pytestmark = pytest.mark.gpu

DT = 0.025
T_BENCH = 8000
FLOOR = 1e-4                       # TRPL_PL_FLOOR_EXCESS
# prefactor k of the envelope |dPL / PL| <= 1e-9 + k / r between FAST and the reference evaluation, per film
# (include/trpl.h: the state gap that 1 / r amplifies grows with the stencil's stiffness D dt / dx^2)
ENVELOPE_K = {2000.0: 5e-13, 311.0: 1e-11}       # TRPL_PL_ENVELOPE_K_THICK / _THIN of include/trpl.h
SSE_GATE = {2000.0: 1e-9, 311.0: 1e-9}           # floor-free squared-error sums over 8000 steps (measured 4e-12)


def nthreads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 32))


def record(name, payload):
    """measured figures of a run, kept beside the logs (gpurun_out/ is merged back from the GPU box)"""
    d = os.path.join(ROOT, "gpurun_out", "r4")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "test_%s.json" % name), "w") as f:
            json.dump(payload, f, indent=1)
    except OSError:
        pass


@pytest.fixture(scope="module")
def twothick_window(gpu, oracle):
    w = gpu.workloads
    L, S, T = 128, 32, T_BENCH
    Time = T * DT
    ini, lens = w.twothick(L)
    X = w.samples(S)
    mark = (w.MARKED_POINT * gpu.UNIT_CONVERSIONS)[None, :-1]
    ref = [oracle.pvsim(X[:, :12], lens[c], Time, L, T, ini[c], nthreads=nthreads()) for c in range(6)]
    obs = [np.log10(oracle.pvsim(mark, lens[c], Time, L, T, ini[c])["plI"][0]) for c in range(6)]
    sse = np.zeros((6, S))
    mag = np.ascontiguousarray(X[:, -1])
    for c in range(6):
        lg = ref[c]["plI"].copy()
        oracle.fastlog(lg)
        Pc = np.zeros(S)
        oracle.prob(Pc, lg, obs[c], mag)
        sse[c] = -Pc
    return dict(L=L, S=S, T=T, Time=Time, ini=ini, lens=lens, X=X, ref=ref, obs=obs, sse=sse)


@pytest.mark.parametrize("mode", [dict(strict=True), dict(kernel="single"), dict(kernel="pair")], ids=["strict", "single", "pair"])
def test_twothick_bench_window_against_the_oracle(gpu, twothick_window, mode):
    g = twothick_window
    S = g["S"]
    rec = {}
    info = {}
    P = gpu.loglik(g["X"], g["ini"], g["lens"], g["Time"], g["L"], g["T"], g["obs"], info=info, **mode)
    assert not info["status"].any()
    for c in range(6):
        want = g["ref"][c]
        length = float(g["lens"][c])
        assert not want["status"].any()
        pl, st, it, _ = gpu.solve_pl(g["X"][:, :12], length, g["Time"], g["L"], g["T"], g["ini"][c], **mode)
        assert not st.any()
        scale = excess_scale(g["X"], length)
        want_col = first_below(want["plI"], FLOOR * scale)
        if mode.get("strict"):
            assert np.array_equal(it, want["iters_total"])
            assert np.array_equal(pl.view(np.int64), want["plI"].view(np.int64))       # bit patterns
            assert np.array_equal(info["iters_total"][c], want["iters_total"])
            assert np.array_equal(info["floor_col"][c], want_col)
            assert np.max(np.abs(info["sse"][c] - g["sse"][c]) / g["sse"][c]) < 1e-12
            continue
        # iteration totals: the oracle's (a knife-edge convergence decision may flip on one system of the 32, by one)
        differ = it != want["iters_total"]
        assert differ.sum() <= 1 and np.abs(it - want["iters_total"]).max() <= 1, (c, int(differ.sum()))
        assert np.array_equal(info["iters_total"][c], it)                  # fused and PL-storing launches agree
        r = want["plI"] / scale[:, None]
        dev = np.abs(pl / want["plI"] - 1)
        with np.errstate(divide="ignore", invalid="ignore"):
            bound = 1e-9 + ENVELOPE_K[length] / r
        physical = r >= 1e-10                                              # below: rounding noise in both evaluations
        worst = float(np.max((dev / bound)[physical])) if physical.any() else 0.0
        assert worst <= 1.0, (c, length, worst)
        above = r >= FLOOR
        k_meas = float(np.max((dev * r)[physical & (r < 0.1)])) if (physical & (r < 0.1)).any() else 0.0
        # floor_col: the column the oracle's own PL gives
        assert np.array_equal(info["floor_col"][c], want_col), (c, int((info["floor_col"][c] != want_col).sum()))
        clear = want_col < 0
        gap = np.abs(info["sse"][c] - g["sse"][c]) / g["sse"][c]
        assert gap[clear].max() < SSE_GATE[length], (c, length, float(gap[clear].max()))
        rec["curve%d" % c] = dict(length=length, iteration_totals_differ=int(differ.sum()), worst_over_bound=worst,
                                  max_dev_above_floor=float(dev[above].max()), envelope_k_measured=k_meas,
                                  floor_free=int(clear.sum()), max_sse_gap_floor_free=float(gap[clear].max()))
    if not mode.get("strict"):
        record("twothick_T8000_%s" % mode["kernel"], rec)
        # the likelihood of the floor-free samples: the oracle's, to the thin film's gate
        Pw = -g["sse"].sum(axis=0)
        clear_s = (info["floor_col"] < 0).all(axis=0)
        assert clear_s.sum() >= 0.8 * S
        assert np.max(np.abs(P[clear_s] - Pw[clear_s]) / np.abs(Pw[clear_s])) < SSE_GATE[311.0]


@pytest.fixture(scope="module")
def l512_window(gpu, oracle):
    w = gpu.workloads
    L, S, T, length = 512, 16, T_BENCH, 2000.0
    Time = T * DT
    ini = np.stack([w.beer_lambert(A, length, L) for A in w.POWER_SCAN_A_CM3])
    X = w.samples(S, seed=61)
    ref7 = [oracle.pvsim(X[:, :12], length, Time, L, T, ini[c], tol=7, nthreads=nthreads()) for c in range(3)]
    ref6 = [oracle.pvsim(X[:, :12], length, Time, L, T, ini[c], tol=6, nthreads=nthreads()) for c in range(3)]
    return dict(L=L, S=S, T=T, Time=Time, length=length, ini=ini, X=X, ref7=ref7, ref6=ref6)


def _loglik_from(oracle, pls, obs, mag):
    P = np.zeros(len(mag))
    for c, pl in enumerate(pls):
        lg = pl.copy()
        oracle.fastlog(lg)
        oracle.prob(P, lg, obs[c], mag)
    return P


@pytest.mark.parametrize("arith", ["fp64", "mixed", "hist32"])
def test_l512_bench_window_against_the_oracle(gpu, oracle, l512_window, arith):
    """stepper_kernel<512> (one system per wavefront, 8 rows per lane) over T = 8000.
    fp64:   tol 7 -- the oracle's iteration totals (+-1 on at most one system), PL within 1e-9 + 2e-12 / r (this grid's
            stencil is 16 times stiffer than the L = 128 one the header's K = 5e-13 is stated for; measured 5.5e-12 above the floor);
            tol 6 -- the tol-6 oracle's iteration totals, PL within 2e-5 and likelihood within 1e-5 of the tol-7 solution
    mixed:  PL within 1e-7 at tol 7, iteration totals within 4 per system of the oracle's ~18 000 (fp32 correction
            solves)
    hist32: the BDF history in difference form, the three older differences stored in fp32, each rounded once
            (TRPL_FLAG_HIST32; the round-3 review's gate): iteration totals within +-1 per system of the oracle's
            (measured: identical on all 48), PL within 1e-8 above the floor at tol 7 (measured 1.4e-9), likelihood within
            2e-8 (measured 7e-9).  A first form that re-referenced every difference to the newest level each step (four
            roundings per level, the newest difference rounded too) measured 5e-8: the first steps after the excitation,
            when a level differs from the next by O(1), round at 6e-8 of the state."""
    g = l512_window
    kw = dict(kernel="single") if arith == "fp64" else ({"mixed": True} if arith == "mixed" else {"kernel": "single", "hist32": True})
    if arith == "hist32" and not hasattr(gpu._abi, "FLAG_HIST32"):
        pytest.skip("library without TRPL_FLAG_HIST32")
    X, L, T, Time, length = g["X"], g["L"], g["T"], g["Time"], g["length"]
    scale = excess_scale(X, length, L)
    mag = np.ascontiguousarray(X[:, -1])
    obs = [np.log10(r["plI"][3]) + 0.02 for r in g["ref7"]]
    want_P = _loglik_from(oracle, [r["plI"] for r in g["ref7"]], obs, mag)
    rec = {}
    for tol, refs in ((7, g["ref7"]), (6, g["ref6"])):
        pls = []
        for c in range(3):
            want = refs[c]
            pl, st, it, _ = gpu.solve_pl(X[:, :12], length, Time, L, T, g["ini"][c], tol=tol, **kw)
            assert not st.any() and not want["status"].any()
            pls.append(pl)
            d_it = np.abs(it - want["iters_total"])
            if arith == "hist32":
                assert d_it.max() <= 1, (tol, c, int(d_it.max()))
            elif arith == "mixed":
                # an fp32 correction solve leaves ~1e-7 of the correction in the residual the next norm sees: a knife-edge
                # decision flips on most systems once or twice in ~18 000 iterations (measured: <= 3 per system)
                assert d_it.max() <= 4 and d_it.sum() <= 2 * len(d_it), (tol, c, int(d_it.max()), int(d_it.sum()))
            else:
                assert (d_it > 0).sum() <= 1 and d_it.max() <= 1, (tol, c, int((d_it > 0).sum()))
            ref7 = g["ref7"][c]["plI"]
            r = ref7 / scale[:, None]
            dev = np.abs(pl / ref7 - 1)
            above = r >= FLOOR
            if tol == 7:
                if arith == "fp64":
                    with np.errstate(divide="ignore", invalid="ignore"):
                        bound = 1e-9 + 2e-12 / r
                    assert np.max((dev / bound)[r >= 1e-10]) <= 1.0, (c, float(np.max((dev / bound)[r >= 1e-10])))
                else:
                    assert dev[above].max() < (1e-7 if arith == "mixed" else 1e-8), (arith, c, float(dev[above].max()))
            else:
                assert dev[above].max() < 2e-5, (arith, c, float(dev[above].max()))
            rec["tol%d_curve%d" % (tol, c)] = dict(max_dev_above_floor=float(dev[above].max()), iteration_totals_differ=int((d_it > 0).sum()))
        P = _loglik_from(oracle, pls, obs, mag)
        clear = np.all([first_below(g["ref7"][c]["plI"], FLOOR * scale) < 0 for c in range(3)], axis=0)
        gate = {("fp64", 7): 1e-8, ("mixed", 7): 1e-7, ("hist32", 7): 2e-8}.get((arith, tol), 1e-5)
        rel = np.abs(P - want_P) / np.abs(want_P)
        assert rel[clear].max() < gate, (arith, tol, float(rel[clear].max()))
        rec["tol%d_loglik_gap" % tol] = float(rel[clear].max())
    record("l512_T8000_%s" % arith, rec)


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
    with pytest.raises(gpu.TrplError):
        gpu.loglik(X[:2], np.repeat(ini[:1], 1025, axis=0), np.full(1025, 2000.0), Time, L, T, [obs[0]] * 1025)


@pytest.mark.parametrize("mode", [dict(strict=True), dict(kernel="single"), dict(kernel="pair")], ids=["strict", "single", "pair"])
def test_floor_col_of_a_flagged_system_is_the_sentinel(gpu, mode):
    """A system whose iteration hits MAX (pvSimPCR.py:269) has sse = +inf and no PL to compare: floor_col = -2 there
    (include/trpl.h), whatever was recorded before the failing step; the others keep their column or -1."""
    w = gpu.workloads
    L, T, S = 128, 60, 24
    ini, lens = w.power_scan(L)
    X = w.samples(S, seed=3)
    obs = [np.full(T + 1, 19.0)] * 3
    info = {}
    gpu.loglik(X, ini, lens, T * DT, L, T, obs, MAX=20, info=info, **mode)     # the oracle flags 11 + 24 + 24 of the 72 systems at this cap
    flagged = info["status"] != 0
    assert flagged.any() and not flagged.all()
    assert (info["floor_col"][flagged] == -2).all() and np.isinf(info["sse"][flagged]).all()
    assert (info["floor_col"][~flagged] >= -1).all() and np.isfinite(info["sse"][~flagged]).all()


def test_hist32_at_256_nodes_and_what_the_flag_refuses(gpu):
    """TRPL_FLAG_HIST32 has an L = 256 and an L = 512 instantiation (the grids whose history pins the occupancy): at
    L = 256 it follows the fp64-history stepper to 1e-8 with the same iteration totals (+-1) over the transient, where
    successive levels differ most; other grids, STRICT / FP32 / MIXED, the paired kernel, snapshots and bundles are
    refused with a message, not ignored."""
    w = gpu.workloads
    L, T, S, length = 256, 400, 12, 2000.0
    Time = T * DT
    X = w.samples(S, seed=9)
    for A in w.POWER_SCAN_A_CM3:
        ini = w.beer_lambert(A, length, L)
        pl64, st64, it64, _ = gpu.solve_pl(X[:, :12], length, Time, L, T, ini, kernel="single")
        pl32, st32, it32, _ = gpu.solve_pl(X[:, :12], length, Time, L, T, ini, kernel="single", hist32=True)
        assert not st64.any() and not st32.any()
        assert np.abs(it32 - it64).max() <= 1
        assert np.max(np.abs(pl32 / pl64 - 1)) < 1e-8
        assert not np.array_equal(pl32, pl64)                      # it IS another arithmetic
    ini = w.beer_lambert(w.POWER_SCAN_A_CM3[0], length, L)
    assert gpu._abi.lib().trpl_kernel_variant(S, L, T, gpu._abi.FLAG_HIST32) == gpu._abi.KERNEL_HIST32
    for bad in (dict(L=128), dict(strict=True), dict(mixed=True), dict(fp32=True), dict(bundle=2), dict(snap_steps=[3], snapshots={})):
        kw = dict(hist32=True)
        kw.update({k: v for k, v in bad.items() if k != "L"})
        Lb = bad.get("L", L)
        with pytest.raises(gpu.TrplError) as e:
            gpu.solve_pl(X[:, :12], length, 10 * DT, Lb, 10, w.beer_lambert(w.POWER_SCAN_A_CM3[0], length, Lb), **kw)
        assert "HIST32" in str(e.value) or "hist32" in str(e.value) or "history" in str(e.value), (bad, str(e.value))


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
        clear = (info["floor_col"] < 0).all(axis=0)
        assert clear.sum() >= 0.8 * g["S"]
        rel = np.abs(P - want) / np.abs(want)
        assert rel[clear].max() < (2e-5 if f32 else 1e-8), (normalize, f32, float(rel[clear].max()))


def test_paired_kernel_repeated_steps_leave_the_partners_bits_alone(gpu):
    """The paired kernel iterates without the seam selects and repeats a time step with them when a system is flagged in
    it (stepper_pair_impl.hpp, "optimistic seam").  Under a small iteration cap many systems are flagged at different
    steps, so many steps are repeated; with samples whose solve turns non-finite in between.  Dropping the first sample
    gives every system another wavefront partner and the other half of the wavefront: status, iteration totals, squared
    errors and likelihoods of every sample must not change by a bit, flagged or not."""
    w = gpu.workloads
    S, T, Time = 5121, 30, 0.75
    lib = gpu._abi.lib()
    assert lib.trpl_kernel_variant(3 * S, 128, T, 0) == gpu._abi.KERNEL_FAST_PAIR
    X = w.samples(S, seed=11)
    X[40, 9] = np.nan              # tau_n: non-finite from the first iteration on
    X[77, 4] = np.inf              # radiative rate
    X[301, 2] = -1e9               # negative diffusivity
    ini, lengths = w.power_scan(128)
    obs = [np.full(T + 1, 20.0)] * 3
    a, b = {}, {}
    pa = gpu.loglik(X, ini, lengths, Time, 128, T, obs, info=a, MAX=60)
    pb = gpu.loglik(X[1:], ini, lengths, Time, 128, T, obs, info=b, MAX=60)
    frac = (a["status"] > 0).mean()
    assert 0.02 < frac < 0.98, frac                  # the cap bites on some systems only: steps are repeated
    assert (a["status"][:, [40, 77]] > 0).all()
    assert np.array_equal(a["status"][:, 1:], b["status"])
    assert np.array_equal(a["iters_total"][:, 1:], b["iters_total"])
    assert np.array_equal(a["sse"][:, 1:], b["sse"])
    assert np.array_equal(pa[1:], pb)
    # and without the cap: the partners of the broken samples against a run that never had them.  A flagged system is
    # parked, but its lanes go on computing with its parameters (a NaN lifetime: NaN coefficients at every step), so beside
    # it the steps must run with the seam selects from the start -- not be repeated one by one after MAX iterations each:
    # the launch with the broken samples may not take much longer than the clean one (600 steps; 10 000 iterations per
    # repeated step would make it 100 x).
    T2 = 600
    obs2 = [np.full(T2 + 1, 20.0)] * 3
    X[500, 0] = np.nan             # n0: even the parked state is not finite
    X[900, 1] = np.inf             # p0
    clean = w.samples(S, seed=11)
    c, d = {}, {}
    pc = gpu.loglik(clean, ini, lengths, T2 * 0.025, 128, T2, obs2, info=c)
    pd = gpu.loglik(X, ini, lengths, T2 * 0.025, 128, T2, obs2, info=d)
    broken = [40, 77, 301, 500, 900]
    ok = np.setdiff1d(np.arange(S), broken)
    assert np.array_equal(pd[ok], pc[ok]) and np.array_equal(d["sse"][:, ok], c["sse"][:, ok])
    assert np.array_equal(d["iters_total"][:, ok], c["iters_total"][:, ok]) and not c["status"].any()
    assert (d["status"][:, [40, 77, 500, 900]] > 0).all() and np.isfinite(pd[ok]).all()
    assert d["seconds"] < 3.0 * c["seconds"] + 0.05, (d["seconds"], c["seconds"])


def test_paired_kernel_partner_that_turns_nonfinite_in_its_last_iteration(gpu):
    """The hole a first form of the optimistic seam had (found by tools/compare_builds.py --extreme): an iteration's
    convergence test precedes its solve, so a system can pass the test and turn non-finite in that same solve -- it is
    flagged only in the NEXT step, and without the seam selects its partner, polluted in the same solve, was marked
    converged with a NaN state.  The kernel now also repeats a step whose new state is not finite.  This sample (hostile:
    back-surface velocity 1e300, hole diffusivity 3e14) does exactly that on the strongest 2000 nm curve of Twothick at
    step 12 -> 13; its weaker curves live on.  They must come out as they do beside any other partner, bit for bit."""
    w = gpu.workloads
    x = np.array([[8.34540522577893e-05, 1.2115714113097759e-22, 8.558192569710279, 340469303561722.0, 350.2753963083046,
                   1.4435043418920274e-21, 1e+300, 1.425211096921706e-15, 28227.298775403244, 0.0015061769014185513,
                   35.96906672432525, 1.9644484713841463e-14, 0.0]])
    ini, lens = w.twothick(128)
    T = 40
    obs = [np.full(T + 1, 18.0)]
    runs = {}
    for name, curves in (("3+5", [3, 5]), ("3+1", [3, 1]), ("1+5", [1, 5])):
        info = {}
        gpu.loglik(x, ini[curves], lens[curves], T * 0.025, 128, T, obs * 2, info=info, MAX=1000, kernel="pair")
        runs[name] = info
    a, b, c = runs["3+5"], runs["3+1"], runs["1+5"]
    assert a["status"][1, 0] == 14 and c["status"][1, 0] == 14            # curve 5 is flagged in step 13, whoever is beside it
    assert a["iters_total"][1, 0] == c["iters_total"][1, 0] == 1016
    assert a["status"][0, 0] == 0 and b["status"][0, 0] == 0 and b["status"][1, 0] == 0 and c["status"][0, 0] == 0
    # curve 3 beside curve 5 (which turns non-finite) = curve 3 beside curve 1 (which does not); curve 1 likewise
    for k in ("sse", "iters_total", "floor_col"):
        assert a[k][0, 0].tobytes() == b[k][0, 0].tobytes(), k
        assert b[k][1, 0].tobytes() == c[k][0, 0].tobytes(), k
    assert a["iters_total"][0, 0] == T + 4                                 # one iteration per step after the first


@pytest.mark.parametrize("workload,seed", [("twothick", 12), ("power_scan", 31), ("twothick", 32)])
def test_optimistic_seam_equals_the_always_isolating_kernel_on_hostile_inputs(gpu, tmp_path, workload, seed):
    """Differential test of the paired kernel's optimistic seam against its always-isolating form -- both are in the
    library, TRPL_PAIR_ALWAYS_SEAM=1 selects the second per process (two child processes).  Inputs that are meant to break
    things (tools/compare_builds.py --extreme): every parameter of the box spread over 40 decades, one sample in eight
    with a zero, a negative value, an infinity, a NaN, 1e300 or a denormal in one column; a small iteration cap.  Tens of
    thousands of systems are flagged at every step of the window, beside partners that are not.  Every output array must
    be the same bits.  (Seed 12 of Twothick holds the sample by which the first form of the optimistic seam differed.)"""
    import subprocess
    import sys
    code = ("import sys, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "import trpl_amd\n"
            "w = trpl_amd.workloads\n"
            "workload, seed, S, T = sys.argv[2], int(sys.argv[3]), 12001, 120\n"
            "ini, lens = w.twothick(128) if workload == 'twothick' else w.power_scan(128)\n"
            "rng = np.random.RandomState(seed)\n"
            "X = w.samples(20001, seed=7)\n"                      # the generator of tools/compare_builds.py --extreme, its first S rows
            "X[:, :12] *= 10.0 ** rng.uniform(-20, 20, size=(20001, 12))\n"
            "special = np.array([0.0, -1.0, np.inf, -np.inf, np.nan, 1e-310, 1e300, -1e-300])\n"
            "rows = rng.choice(20001, size=20001 // 8, replace=False)\n"
            "X[rows, rng.randint(0, 12, size=rows.size)] = special[rng.randint(0, special.size, size=rows.size)]\n"
            "X = np.ascontiguousarray(X[:S]) if seed != 12 else np.ascontiguousarray(X[6000:6000 + S])\n"     # seed 12: rows around sample 6598
            "obs = [np.full(T + 1, 18.0)] * len(lens)\n"
            "info = {}\n"
            "P = trpl_amd.loglik(X, ini, lens, T * 0.025, 128, T, obs, info=info, MAX=1000, kernel='pair')\n"
            "np.savez(sys.argv[1], P=P, sse=info['sse'], it=info['iters_total'], st=info['status'], fc=info['floor_col'])\n") % ROOT
    out = {}
    for v in ("0", "1"):
        path = str(tmp_path / ("seam%s.npz" % v))
        env = dict(os.environ, TRPL_PAIR_ALWAYS_SEAM=v, TRPL_AUTOBUILD="0")
        r = subprocess.run([sys.executable, "-c", code, path, workload, str(seed)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[v] = np.load(path)
    flagged = int((out["1"]["st"] != 0).sum())
    assert flagged > 1000 and flagged < out["1"]["st"].size, flagged
    for k in ("P", "sse", "it", "st", "fc"):
        assert out["0"][k].tobytes() == out["1"][k].tobytes(), k
    record("optimistic_vs_always_seam_%s_%d" % (workload, seed), {"systems": int(out["1"]["st"].size), "flagged": flagged})
