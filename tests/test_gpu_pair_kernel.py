"""The two-systems-per-wavefront stepper (pair::stepper_pair_kernel, the kernel bench.py times): against the reference's goldens
and the oracle directly (forced per call with TRPL_FLAG_KERNEL_PAIR and at a size where the library selects it by itself);
against STRICT with identical iteration counts; a system's bits do not depend on its partner, the pairing rule
(TRPL_FLAG_PAIR_ADJACENT) or the seam form (TRPL_FLAG_PAIR_ALWAYS_SEAM, the differential test on hostile inputs -- both forms
in one process since the switch became a per-call flag in round 5); flagged partners, repeated steps, wide-box fuzz."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from gpu_common import DT, _check_pl_against, follows_iteration_path, nthreads, record

pytestmark = pytest.mark.gpu


# ---- two systems per wavefront (stepper_pair_impl.hpp): the kernel of every launch that fills the chip ----
def _pair_batch(trpl, S, T):
    lib = trpl._abi.lib()
    # asserted, not skipped: these sizes make the library choose the paired kernel on its own on an MI355X
    # (test_paired_kernel_reproduces_the_reference_goldens below forces it per call with TRPL_FLAG_KERNEL_PAIR and compares with the oracle)
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


def test_paired_kernel_reproduces_the_reference_goldens(gpu, golden):
    """pvsim_power.npz / pvsim_twothick.npz (the reference's own pvSim outputs, incl. the 311 nm curves
    with 567 iterations on step 0) pushed through the two-systems-per-wavefront kernel."""
    lib = gpu._abi.lib()
    assert lib.trpl_kernel_variant(5, 128, 160, gpu._abi.FLAG_KERNEL_PAIR) == gpu._abi.KERNEL_FAST_PAIR
    assert lib.trpl_kernel_variant(5, 128, 160, 0) == gpu._abi.KERNEL_FAST      # too small to be picked unforced
    for name in ("pvsim_power", "pvsim_twothick"):
        g = golden(name)
        X12, Time, L, T = g["X"][:, :12], float(g["time"]), int(g["L"]), int(g["T"])
        lengths = g["lengths"] if "lengths" in g.files else np.full(len(g["ini"]), float(g["length"]))
        for c in range(len(g["ini"])):
            want, iters = g["plI"][c], g["iters"][c].sum(axis=1)             # iterate()'s return per step, summed
            pl, st, it, _ = gpu.solve_pl(X12, float(lengths[c]), Time, L, T, g["ini"][c], kernel="pair")
            assert not st.any()
            follows_iteration_path(it, iters, "%s curve %d" % (name, c))
            assert np.max(np.abs(pl - want) / np.abs(want)) < 1e-9, (name, c)
    # bayes_e2e.npz: the likelihoods bayeslib.bayes(pvSim, ...) itself produced (two experiments: on-grid and a
    # prefix grid, float32 PL staging), through the paired kernel's fused path, and through its real-data sibling
    g = golden("bayes_e2e")
    T, tg, npre = int(g["T"]), g["tgrid"], int(g["npre"])
    for obs in (list(g["obs0"]), list(g["obs1"])):
        e = 0 if len(obs[0]) == len(tg) else 1
        info = {}
        P32 = gpu.loglik(g["X"], g["ini"], 2000.0, float(g["time"]), 128, T, obs, pl_f32=True, info=info, kernel="pair")
        assert not info["status"].any()
        assert np.max(np.abs(P32 - g["P"][e]) / np.abs(g["P"][e])) < 2e-5
        single = gpu.loglik(g["X"], g["ini"], 2000.0, float(g["time"]), 128, T, obs, pl_f32=True, kernel="single")
        assert np.allclose(P32, single, rtol=1e-6, atol=0)
    # an odd sample count (the last wavefront holds one system) and a single sample
    g = golden("pvsim_power")
    for n in (1, 3):
        pl, st, it, _ = gpu.solve_pl(g["X"][:n, :12], 2000.0, float(g["time"]), 128, int(g["T"]), g["ini"][1], kernel="pair")
        assert np.max(np.abs(pl - g["plI"][1][:n]) / np.abs(g["plI"][1][:n])) < 1e-9


def test_paired_kernel_vs_oracle_power_scan_at_natural_size(gpu, oracle):
    """Power_scan, 5 123 samples x 3 curves (odd tail included): the library selects the paired kernel by
    itself (asserted, not skipped); fused likelihoods against oracle.simulate_loglik on all host threads."""
    w = gpu.workloads
    S, T = 5123, 100
    Time = T * 0.025
    lib = gpu._abi.lib()
    assert lib.trpl_kernel_variant(3 * S, 128, T, 0) == gpu._abi.KERNEL_FAST_PAIR
    ini, lens = w.power_scan(128)
    X = w.samples(S, seed=21)
    mark = (w.MARKED_POINT * gpu.UNIT_CONVERSIONS)[None, :-1]
    obs = [np.log10(oracle.pvsim(mark, lens[c], Time, 128, T, ini[c])["plI"][0]) for c in range(3)]
    e_data = [([np.linspace(0, Time, T + 1)] * 3, obs)]
    want = oracle.simulate_loglik(X, ini, lens, Time, 128, T, e_data, pl_dtype=np.float64, nthreads=nthreads())[0]
    info = {}
    P = gpu.loglik(X, ini, lens, Time, 128, T, obs, info=info)
    assert not info["status"].any()
    rel = np.abs(P - want) / np.abs(want)
    assert rel.max() < 1e-8, rel.max()
    # and the same systems' PL and iteration counts, one curve, straight from the paired kernel
    r = oracle.pvsim(X[:, :12], lens[2], Time, 128, T, ini[2], nthreads=nthreads())
    assert lib.trpl_kernel_variant(S, 128, T, gpu._abi.FLAG_KERNEL_PAIR) == gpu._abi.KERNEL_FAST_PAIR
    err, same = _check_pl_against(gpu, X[:, :12], lens[2], Time, 128, T, ini[2], r["plI"], r["iters_total"], "pair")
    assert same > 0.99, same


def test_paired_kernel_vs_oracle_twothick(gpu, oracle):
    """Twothick (311 / 2000 nm alternating, the 311 nm stencil ~40x stiffer), 2 600 samples x 6 curves: the
    library selects the paired kernel by itself (asserted); against the oracle -- PL to 1e-9, iteration
    totals, likelihoods to 1e-8."""
    w = gpu.workloads
    S, T = 2600, 100
    Time = T * 0.025
    assert gpu._abi.lib().trpl_kernel_variant(6 * S, 128, T, 0) == gpu._abi.KERNEL_FAST_PAIR
    ini, lens = w.twothick(128)
    X = w.samples(S, seed=22)
    mark = (w.MARKED_POINT * gpu.UNIT_CONVERSIONS)[None, :-1]
    obs = [np.log10(oracle.pvsim(mark, lens[c], Time, 128, T, ini[c])["plI"][0]) for c in range(6)]
    e_data = [([np.linspace(0, Time, T + 1)] * 6, obs)]
    want = oracle.simulate_loglik(X, ini, lens, Time, 128, T, e_data, pl_dtype=np.float64, nthreads=nthreads())[0]
    info = {}
    P = gpu.loglik(X, ini, lens, Time, 128, T, obs, info=info)
    assert not info["status"].any()
    assert np.max(np.abs(P - want) / np.abs(want)) < 1e-8
    for c in (0, 4):                               # 311 nm at the lowest and at the highest power
        r = oracle.pvsim(X[:, :12], lens[c], Time, 128, T, ini[c], nthreads=nthreads())
        if c == 4:
            assert r["iters_max"].max() > 100      # the stiff start (hundreds of iterations on step 0) is in the comparison
        err, same = _check_pl_against(gpu, X[:, :12], lens[c], Time, 128, T, ini[c], r["plI"], r["iters_total"], "pair")
        assert same > 0.97, (c, same)
    # the one-system kernel on the same inputs: both FAST kernels sit within rounding of the oracle
    P1 = gpu.loglik(X, ini, lens, Time, 128, T, obs, kernel="single")
    assert np.max(np.abs(P1 - want) / np.abs(want)) < 1e-8


def test_variant_flags_are_validated(gpu):
    w = gpu.workloads
    ini, lens = w.power_scan(64)
    X = w.samples(4)
    with pytest.raises(gpu.TrplError):             # the paired kernel exists for L = 128 only
        gpu.solve_pl(X[:, :12], lens[0], 0.25, 64, 10, ini[0], kernel="pair")
    ini, lens = w.power_scan(128)
    with pytest.raises(gpu.TrplError):
        gpu.solve_pl(X[:, :12], lens[0], 0.25, 128, 10, ini[0], kernel="pair", strict=True)
    lib = gpu._abi.lib()
    pl = np.zeros((4, 11))
    both = gpu._abi.FLAG_KERNEL_PAIR | gpu._abi.FLAG_KERNEL_SINGLE
    rc = lib.trpl_solve_pl(X[:, :12].copy().ctypes.data, 4, 2000.0, 0.25, 128, 10, 1, 7, 100, ini[0].ctypes.data,
                           pl.ctypes.data, 8, 11, None, None, both, 0, None)
    assert rc == gpu._abi.ERR_ARG


def test_likelihoods_do_not_depend_on_the_pairing_rule(gpu):
    """The paired stepper pairs two curves of one sample (trpl_pair_table) or, with TRPL_FLAG_PAIR_ADJACENT, adjacent samples
    of one curve.  Scheduling only: likelihoods, per-curve sums, iteration totals, status and floor columns are the
    same bits -- Twothick (two groups of three curves: same-sample pairs and cross-sample leftovers), an odd batch (the
    last period has one sample).  One process: the switch is a per-call flag since round 5 (it was an environment variable
    read once per process)."""
    w = gpu.workloads
    ini, lens = w.twothick(128)
    X = w.samples(4099, seed=17)
    T = 300
    obs = [np.linspace(17.0, 14.0, T + 1)] * 6
    out = {}
    for name, extra in (("table", 0), ("adjacent", gpu._abi.FLAG_PAIR_ADJACENT)):
        info = {}
        P = gpu.loglik(X, ini, lens, T * 0.025, 128, T, obs, info=info, kernel="pair", extra_flags=extra)
        out[name] = dict(P=P, sse=info["sse"], it=info["iters_total"], st=info["status"], fc=info["floor_col"])
    for k in ("P", "sse", "it", "st", "fc"):
        assert out["table"][k].tobytes() == out["adjacent"][k].tobytes(), k
    assert np.isfinite(out["table"]["P"]).all() and not out["table"]["st"].any()


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
    clear = ok & (a["floor_col"] == -1) & (b["floor_col"] == -1)
    rel = np.abs(a["sse"][clear] - b["sse"][clear]) / np.abs(a["sse"][clear])
    assert np.median(rel) < 1e-13 and np.quantile(rel, 0.999) < 1e-7


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
def test_optimistic_seam_equals_the_always_isolating_kernel_on_hostile_inputs(gpu, workload, seed):
    """Differential test of the paired kernel's optimistic seam against its always-isolating form -- both are in the
    library, TRPL_FLAG_PAIR_ALWAYS_SEAM selects the second per call (one process; until round 5 an environment variable chose
    per process).  Inputs that are meant to break things (tools/compare_builds.py --extreme): every parameter of the box
    spread over 40 decades, one sample in eight with a zero, a negative value, an infinity, a NaN, 1e300 or a denormal in one
    column; a small iteration cap.  Tens of thousands of systems are flagged at every step of the window, beside partners
    that are not.  Every output array must be the same bits.  (Seed 12 of Twothick holds the sample by which the first form
    of the optimistic seam differed.)"""
    w = gpu.workloads
    S, T = 12001, 120
    ini, lens = w.twothick(128) if workload == "twothick" else w.power_scan(128)
    rng = np.random.RandomState(seed)
    X = w.samples(20001, seed=7)                       # the generator of tools/compare_builds.py --extreme, its first S rows
    X[:, :12] *= 10.0 ** rng.uniform(-20, 20, size=(20001, 12))
    special = np.array([0.0, -1.0, np.inf, -np.inf, np.nan, 1e-310, 1e300, -1e-300])
    rows = rng.choice(20001, size=20001 // 8, replace=False)
    X[rows, rng.randint(0, 12, size=rows.size)] = special[rng.randint(0, special.size, size=rows.size)]
    X = np.ascontiguousarray(X[:S]) if seed != 12 else np.ascontiguousarray(X[6000:6000 + S])     # seed 12: rows around sample 6598
    obs = [np.full(T + 1, 18.0)] * len(lens)
    out = {}
    for name, extra in (("optimistic", 0), ("always", gpu._abi.FLAG_PAIR_ALWAYS_SEAM)):
        info = {}
        P = gpu.loglik(X, ini, lens, T * 0.025, 128, T, obs, info=info, MAX=1000, kernel="pair", extra_flags=extra)
        out[name] = dict(P=P, sse=info["sse"], it=info["iters_total"], st=info["status"], fc=info["floor_col"])
    flagged = int((out["always"]["st"] != 0).sum())
    assert flagged > 1000 and flagged < out["always"]["st"].size, flagged
    for k in ("P", "sse", "it", "st", "fc"):
        assert out["optimistic"][k].tobytes() == out["always"][k].tobytes(), k
    record("optimistic_vs_always_seam_%s_%d" % (workload, seed), {"systems": int(out["always"]["st"].size), "flagged": flagged})
