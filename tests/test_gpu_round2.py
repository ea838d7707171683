"""GPU parity tests added in round 2 (all through the C ABI):

* the two-systems-per-wavefront stepper -- the kernel bench.py times -- DIRECTLY against the pinned CPU
  oracle and the reference's own golden vectors (Power_scan and Twothick 311 / 2000 nm), forced per call
  with TRPL_FLAG_KERNEL_PAIR and at a size where the library selects it by itself;
* sharding invariance: a logical batch above the pair threshold cut into shards below it gives the same
  bits as one launch (trpl_loglik_multi and the rank driver's pinned flags);
* state snapshots plN / plP / plE against the oracle and Legacy/pvSim.py's own output;
* the device-resident multi-GPU entry point (RCCL all-gather) on a one-rank communicator;
* bench.py's N = 2 control flow as child processes (gloo) against the single-rank run.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def nthreads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 64))


def above_floor(pl, rel=1e-12):
    """PL points above the cancellation floor: >= rel * PL(0)."""
    return np.abs(pl) >= rel * np.abs(pl[:, :1])


# ------------------------------------------------------------------ paired kernel vs oracle / goldens
def _check_pl_against(gpu, X12, length, Time, L, T, ini, want_pl, want_iters, kernel, rtol=1e-9):
    pl, st, it, _ = gpu.solve_pl(X12, length, Time, L, T, ini, kernel=kernel)
    assert not st.any()
    ok = above_floor(want_pl)
    assert ok.mean() > 0.95
    err = np.max(np.abs(pl[ok] - want_pl[ok]) / np.abs(want_pl[ok]))
    assert err < rtol, err
    # iteration totals: +-1 % in sum (the FAST kernels may flip a knife-edge convergence decision)
    assert abs(it.sum() / want_iters.sum() - 1) < 0.01
    return err, float((it == want_iters).mean())


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
            assert not st.any() and np.all(np.abs(it - iters) <= 0.01 * iters + 1), (name, c)
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


# ------------------------------------------------------------------ sharding invariance
def test_sharded_batch_equals_single_launch_across_the_pair_threshold(gpu):
    """The whole batch is above the paired kernel's threshold, every shard is below it: the variant is a
    property of the logical batch, so trpl_loglik_multi (8 shards) and a rank driver that pins its flags
    from the total (dist / bench.py) return bit for bit the single launch's likelihoods."""
    w = gpu.workloads
    S, T, Time = 5124, 40, 1.0
    lib = gpu._abi.lib()
    A = gpu._abi
    assert lib.trpl_kernel_variant(3 * S, 128, T, 0) == A.KERNEL_FAST_PAIR
    assert lib.trpl_kernel_variant(3 * (S // 8 + 1), 128, T, 0) == A.KERNEL_FAST
    ini, lens = w.power_scan(128)
    X = w.samples(S, seed=31)
    obs = [np.full(T + 1, 20.0) - 0.01 * np.arange(T + 1)] * 3
    one, multi = {}, {}
    want = gpu.loglik(X, ini, lens, Time, 128, T, obs, info=one)
    got = gpu.loglik(X, ini, lens, Time, 128, T, obs, info=multi, devices=[0] * 8)
    assert np.array_equal(got, want)
    for k in ("sse", "status", "iters_total"):
        assert np.array_equal(multi[k], one[k]), k
    # the rank driver: each rank launches its own shard with the flags pinned from the TOTAL
    flags = A.pin_variant(0, 3 * S, 128, T)
    assert flags == A.FLAG_KERNEL_PAIR
    for world in (2, 8):
        parts = []
        for r in range(world):
            lo, hi = gpu.dist.shard_bounds(S, world, r)
            parts.append(gpu.loglik(X[lo:hi], ini, lens, Time, 128, T, obs, kernel="pair"))
        assert np.array_equal(np.concatenate(parts), want)
    # without the pin the shards would run the other kernel: close (rounding), not identical -- the
    # documented reason for pinning
    lo, hi = gpu.dist.shard_bounds(S, 8, 1)
    unpinned = gpu.loglik(X[lo:hi], ini, lens, Time, 128, T, obs)
    assert np.allclose(unpinned, want[lo:hi], rtol=1e-10, atol=0)
    # a small batch stays on the one-system kernel in every shard
    small = gpu.loglik(X[:300], ini, lens, Time, 128, T, obs)
    assert np.array_equal(gpu.loglik(X[:300], ini, lens, Time, 128, T, obs, devices=[0, 0, 0]), small)


def test_multi_validates_observation_brackets_like_the_single_device_call(gpu):
    w = gpu.workloads
    ini, lens = w.power_scan(128)
    X = w.samples(6)
    T, Time = 40, 1.0
    lib = gpu._abi.lib()
    A = gpu._abi
    n = 5
    obs = np.full((3, n), 19.0)
    n_obs = np.full(3, n, dtype=np.int64)
    hi = np.tile(np.array([3, 2, 5, 7, 9], dtype=np.int32), (3, 1))        # not sorted
    dx = np.full((3, n), 0.01)
    h = np.full((3, n), 0.025)
    P = np.zeros(6)
    dev = np.zeros(2, dtype=np.int32)

    def call(hi_, dx_, h_):
        return lib.trpl_loglik_multi(X.ctypes.data, 6, 3, lens.ctypes.data, Time, 128, T, 1, 7, 1000, ini.ctypes.data,
                                     obs.ctypes.data, hi_.ctypes.data, dx_.ctypes.data, h_.ctypes.data, n,
                                     n_obs.ctypes.data, P.ctypes.data, None, None, None, None, 0, dev.ctypes.data, 2, None)
    assert call(hi, dx, h) == A.ERR_ARG and b"sorted" in lib.trpl_last_error()
    good = np.tile(np.array([2, 3, 5, 7, 9], dtype=np.int32), (3, 1))
    bad_hi = good.copy(); bad_hi[1, 4] = T + 1
    assert call(bad_hi, dx, h) == A.ERR_ARG
    bad_h = h.copy(); bad_h[2, 0] = 0.0
    assert call(good, dx, bad_h) == A.ERR_ARG
    bad_dx = dx.copy(); bad_dx[0, 1] = 0.05                                # beyond the bracket
    assert call(good, bad_dx, h) == A.ERR_ARG
    assert call(good, dx, h) == A.OK
    with pytest.raises(ValueError):                                         # the Python driver refuses earlier still
        gpu.loglik(X, ini, lens, Time, 128, T, [np.full(3, 19.0)] * 3, times=[np.array([0.1, 0.2, 2.0])] * 3,
                   devices=[0, 0])


# ------------------------------------------------------------------ state snapshots (f-4)
SNAPS = [0, 1, 2, 7, 24, 72, 100, 100, 400]          # a repeated step and one beyond T


def _snap_case(gpu, oracle, **kw):
    w = gpu.workloads
    ini, lens = w.twothick(128)
    X = w.samples(7, seed=41)[:, :12]
    T, Time = 100, 2.5
    want = oracle.pvsim(X, lens[0], Time, 128, T, ini[0], snap_steps=SNAPS)
    got = {}
    pl, st, it, _ = gpu.solve_pl(X, lens[0], Time, 128, T, ini[0], snap_steps=SNAPS, snapshots=got, **kw)
    return want, got, pl, st, it


def test_snapshots_strict_are_bit_identical_to_the_oracle(gpu, oracle):
    """plN / plP / plE of the STRICT kernel: the reference's state bit for bit at every recorded step, the
    repeated step fills its first slot only and the step beyond T is never reached (Legacy/pvSim.py:121-126)."""
    want, got, pl, st, it = _snap_case(gpu, oracle, strict=True)
    assert np.array_equal(pl, want["plI"]) and np.array_equal(it, want["iters_total"])
    for k in ("plN", "plP", "plE"):
        assert got[k].shape == want[k].shape
        assert np.array_equal(got[k], want[k]), k
    assert (got["plN"][:, 7] == 0).all() and (got["plN"][:, 8] == 0).all()          # untouched slots
    assert (got["plE"][:, :, 0] == 0).all() and (got["plE"][:, :, 128] == 0).all()   # E_0 = E_L = 0
    assert (got["plN"][:, :7] > 0).all()


@pytest.mark.parametrize("kernel", ["single", "pair"])
def test_snapshots_fast_kernels_vs_oracle(gpu, oracle, kernel):
    want, got, pl, st, it = _snap_case(gpu, oracle, kernel=kernel)
    assert not st.any()
    for k in ("plN", "plP"):
        assert np.max(np.abs(got[k][:, :7] - want[k][:, :7]) / want[k][:, :7]) < 1e-9, k
        assert (got[k][:, 7:] == 0).all()
    # the field is the integral of the (tiny) charge imbalance P - N, i.e. pure cancellation once the carriers
    # have relaxed (1e-14 of N is 1e-7 .. 1e-5 of E): compare against the largest field of the snapshot
    scale = np.abs(want["plE"]).max(axis=2, keepdims=True)
    assert np.max(np.abs(got["plE"][:, 1:7] - want["plE"][:, 1:7]) / scale[:, 1:7]) < 2e-5
    assert (got["plE"][:, 0] == 0).all()                                            # t = 0: no field yet


def test_snapshots_vs_legacy_pvsim_golden_and_dropin_signature(gpu, golden):
    """Against what Legacy/pvSim.pvSim itself returned (legacy_odeint.npz: BDF2 / Thomas, no Auger,
    exponential excitation): the steps on which the schemes coincide (t = 0, 1, 2) to 1e-12 for N and P,
    afterwards within the BDF-order gap at the five Testing/compare.py:22 sample points; through the
    drop-in pvSim(), which fills the caller's plN / plP / plE like the reference's signature promises."""
    g = golden("legacy_odeint")
    X = g["X"].copy()
    X[:, 7:9] = 0.0                                                         # Legacy has no Auger terms
    L, T, Length, Time = int(g["L"]), int(g["T"]), float(g["length"]), float(g["time"])
    pT = tuple(int(v) for v in g["pT"])
    S = len(X)
    for strict in (True, False):
        plI = np.empty((S, T + 1))
        plN = np.zeros((S, len(pT), L)); plP = np.zeros((S, len(pT), L)); plE = np.zeros((S, len(pT), L + 1))
        gpu.pvSim(plI, plN, plP, plE, X[:, :12], [Length, Time, L, T, 1, pT, 7, 10000],
                  (float(g["a_nm3"]), float(g["l_nm"])), (128,), 2048, 1, init_mode="exp", strict=strict)
        for mine, ref in ((plN, g["plN_legacy"]), (plP, g["plP_legacy"])):
            assert np.max(np.abs(mine[:, :3] - ref[:, :3]) / ref[:, :3]) < 1e-12
            locs = (np.array([0.1, 0.3, 0.5, 0.7, 0.9]) * L).astype(int)     # Testing/compare.py:22
            for thr in range(S):
                a, b = mine[thr][3:, locs].ravel(), ref[thr][3:, locs].ravel()
                assert np.linalg.norm(a - b) / np.linalg.norm(b) < 5e-3      # compare.py:43's norm
        scale = np.abs(g["plE_legacy"][:, :3]).max(axis=2, keepdims=True)
        scale[scale == 0] = 1.0
        assert np.max(np.abs(plE[:, :3] - g["plE_legacy"][:, :3]) / scale) < 1e-5
        assert np.max(np.abs(plI[:, :3] / g["plI_legacy"][:, :3] - 1)) < (1e-13 if strict else 1e-12)
    # the dummies bayeslib passes (shape (S, 2, L), bayeslib.py:141-143) do not match len(pT): ignored
    junk = np.full((S, 2, L), 7.0)
    gpu.pvSim(np.empty((S, 17)), junk, junk.copy(), np.full((S, 2, L + 1), 7.0), X[:, :12],
              [Length, 16 * 0.025, L, 16, 1, pT, 7, 10000], (float(g["a_nm3"]), float(g["l_nm"])), init_mode="exp")
    assert (junk == 7.0).all()


def test_snapshots_nonconvergence_small_grids_and_device_entry(gpu, oracle):
    """A system flagged at step t gets NaN from that step's slot on (like its PL) while its wavefront
    partner's snapshots are complete; L = 16 / 64 (blocked layouts); the device-resident entry point."""
    import torch
    w = gpu.workloads
    ini, lens = w.twothick(128)
    X = w.samples(64, seed=43)[:, :12]
    T, Time = 30, 0.75
    steps = [0, 3, 10, 30]
    for kernel in ("single", "pair"):
        got = {}
        pl, st, it, _ = gpu.solve_pl(X, lens[4], Time, 128, T, ini[4], MAX=60, snap_steps=steps, snapshots=got,
                                     kernel=kernel)
        want = oracle.pvsim(X, lens[4], Time, 128, T, ini[4], MAX=60, snap_steps=steps, nthreads=4)
        assert np.array_equal(st, want["status"]) and 0 < (st > 0).sum() < len(st)
        for k in ("plN", "plP", "plE"):
            assert np.array_equal(np.isnan(got[k]), np.isnan(want[k])), (kernel, k)
        live = st == 0
        assert np.max(np.abs(got["plN"][live] - want["plN"][live]) / want["plN"][live]) < 1e-9
    for L in (16, 64):
        ini_s, lens_s = w.power_scan(L)
        want = oracle.pvsim(X[:5], lens_s[1], 0.5, L, 20, ini_s[1], snap_steps=[20, 0, 5])
        for strict in (True, False):
            got = {}
            gpu.solve_pl(X[:5], lens_s[1], 0.5, L, 20, ini_s[1], snap_steps=[20, 0, 5], snapshots=got, strict=strict)
            for k in ("plN", "plP", "plE"):
                if strict:
                    assert np.array_equal(got[k], want[k]), (L, k)
                else:
                    scale = np.abs(want[k]).max(axis=2, keepdims=True) + 1e-300
                    assert np.max(np.abs(got[k] - want[k]) / scale) < 1e-6, (L, k)
    # PL stored every 4th step only (plT = 4): snapshots are taken on their own steps regardless
    want = oracle.pvsim(X[:6], lens[1], Time, 128, T, ini[1], plT=4, snap_steps=[3, 8, 29])
    for kw in ({"strict": True}, {"kernel": "pair"}):
        got = {}
        pl, st, it, _ = gpu.solve_pl(X[:6], lens[1], Time, 128, T, ini[1], plT=4, snap_steps=[3, 8, 29], snapshots=got, **kw)
        assert pl.shape == (6, T // 4 + 1) and np.max(np.abs(pl / want["plI"] - 1)) < 1e-9
        for k in ("plN", "plP"):
            assert np.max(np.abs(got[k] - want[k]) / want[k]) < (1e-9 if "kernel" in kw else 1e-15), (kw, k)
    # n_snap = 0 / no output arrays: plain solve
    pl0, _, _, _ = gpu.solve_pl(X[:6], lens[1], Time, 128, T, ini[1], snap_steps=[])
    pl1, _, _, _ = gpu.solve_pl(X[:6], lens[1], Time, 128, T, ini[1])
    assert np.array_equal(pl0, pl1)
    # device-resident form, unordered steps, only plP requested
    dev = torch.device("cuda", 0)
    Xd = torch.from_numpy(X[:9].copy()).to(dev)
    pl_d = torch.empty((9, T + 1), dtype=torch.float64, device=dev)
    plP_d = torch.zeros((9, 3, 128), dtype=torch.float64, device=dev)
    gpu.device.solve_pl_snap_device(Xd, lens[1], Time, 128, T, torch.from_numpy(ini[1]).to(dev), pl_d, [10, 0, 3],
                                    plP=plP_d, flags=gpu.FLAG_STRICT)
    torch.cuda.synchronize()
    want = oracle.pvsim(X[:9], lens[1], Time, 128, T, ini[1], snap_steps=[10, 0, 3])
    assert np.array_equal(plP_d.cpu().numpy(), want["plP"])
    with pytest.raises(gpu.TrplError):                                      # not built for the fp32 stepper
        gpu.solve_pl(X[:4], lens[1], Time, 128, T, ini[1], snap_steps=[0], snapshots={}, fp32=True, tol=4)


# ------------------------------------------------------------------ configs[4]: L = 512 at an accuracy worth reporting
@pytest.mark.parametrize("L", [128, 512])
def test_cfg4_fp64_state_paths_against_the_oracle(gpu, oracle, L):
    """The accurate paths for BASELINE configs[4] (L = 512; profiles/r2_cfg4_frontier.json): the fp64 stepper
    and the mixed one (TRPL_FLAG_MIXED: fp64 state / assembly / residuals, fp32 correction solves) against the
    fp64 tol-7 oracle over a 400-step window, gates = the measured frontier with a margin:
      tol 7  fp64 1e-9 (FAST parity);  mixed 1e-7, the SAME iteration counts as fp64 (the fp32 solve resolves
             ~1e-5 of a correction that is itself O(tolerance) by the last iteration)
      tol 6  both: PL <= 2e-5, log-likelihood <= 1e-5 -- the accuracy the frontier table recommends
    (the fp32-STATE stepper's gates stay in test_fp32_stepper_vs_fp64_oracle: 2e-3 at 60 steps, and tens of
    percent over 8000 steps -- measured)."""
    w = gpu.workloads
    X = w.samples(8, seed=61)
    T, length = 400, 2000.0
    Time = T * 0.025
    ini = np.stack([w.beer_lambert(A, length, L) for A in w.POWER_SCAN_A_CM3])
    ref = [oracle.pvsim(X[:, :-1], length, Time, L, T, ini[c], nthreads=nthreads()) for c in range(3)]
    obs = [np.log10(r["plI"][3]) + 0.02 for r in ref]
    want = oracle.simulate_loglik(X, ini, length, Time, L, T, [([np.linspace(0, Time, T + 1)] * 3, obs)],
                                  pl_dtype=np.float64, nthreads=nthreads())[0]
    lib = gpu._abi.lib()
    assert lib.trpl_kernel_variant(10 ** 6, L, T, gpu._abi.FLAG_MIXED) == gpu._abi.KERNEL_MIXED
    for mixed, tol, pl_gate, ll_gate in ((False, 7, 1e-9, 1e-8), (True, 7, 1e-7, 1e-7), (False, 6, 2e-5, 1e-5), (True, 6, 2e-5, 1e-5)):
        for c in range(3):
            pl, st, it, _ = gpu.solve_pl(X[:, :-1], length, Time, L, T, ini[c], tol=tol, mixed=mixed, kernel="single" if not mixed else None)
            assert not st.any()
            ok = above_floor(ref[c]["plI"])
            err = np.max(np.abs(pl[ok] - ref[c]["plI"][ok]) / np.abs(ref[c]["plI"][ok]))
            assert err < pl_gate, (mixed, tol, c, err)
            if tol == 7:
                assert abs(it.sum() / ref[c]["iters_total"].sum() - 1) < 0.01, (mixed, c)
            else:
                assert np.all(it <= ref[c]["iters_total"])
        info = {}
        P = gpu.loglik(X, ini, length, Time, L, T, obs, tol=tol, mixed=mixed, info=info)
        assert not info["status"].any()
        assert np.max(np.abs(P - want) / np.abs(want)) < ll_gate, (mixed, tol)
    with pytest.raises(gpu.TrplError):
        gpu.solve_pl(X[:, :-1], length, Time, L, T, ini[0], mixed=True, strict=True)
    with pytest.raises(gpu.TrplError):
        gpu.solve_pl(X[:, :-1], length, Time, L, T, ini[0], mixed=True, fp32=True)
    with pytest.raises(gpu.TrplError):
        gpu.solve_pl(X[:, :-1], length, Time, 64, T, w.beer_lambert(1e17, length, 64), mixed=True)


# ------------------------------------------------------------------ device-resident multi-GPU (RCCL)
@pytest.mark.parametrize("force_pad", [False, True])
def test_multi_device_resident_allgather_on_a_one_rank_communicator(gpu, force_pad):
    """trpl_multi_create (ncclCommInitAll, RCCL bound at first use) + trpl_loglik_multi_dev with the one device
    of this box: the all-gathered P[S] left in device memory equals trpl_loglik_dev's, per-shard outputs
    included; with TRPL_MULTI_FORCE_PAD the padded exchange + unpadding pass runs instead of the direct one.
    (The force_pad case is a child process: the knob is read once per process.)"""
    if force_pad:
        env = dict(os.environ, TRPL_MULTI_FORCE_PAD="1", PYTHONPATH=ROOT)
        code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_round2 as t; "
                "import trpl_amd; t._multi_dev_check(trpl_amd); print('PAD-OK')") % (ROOT, os.path.join(ROOT, "tests"))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "PAD-OK" in out.stdout, out.stderr[-2000:]
        return
    _multi_dev_check(gpu)


def _multi_dev_check(gpu):
    import torch
    w = gpu.workloads
    dev = torch.device("cuda", 0)
    ini, lens = w.power_scan(128)
    S, T, Time = 777, 60, 1.5
    X = torch.from_numpy(w.samples(S, seed=51)).to(dev)
    ini_d = torch.from_numpy(ini).to(dev)
    obs = torch.full((3, T + 1), 20.0, dtype=torch.float64, device=dev) - 0.01 * torch.arange(T + 1, device=dev)
    P = torch.zeros(S, dtype=torch.float64, device=dev)
    sse = torch.empty((3, S), dtype=torch.float64, device=dev)
    st = torch.empty((3, S), dtype=torch.int32, device=dev)
    it = torch.empty((3, S), dtype=torch.int64, device=dev)
    gpu.device.loglik_device(X, ini_d, lens, Time, 128, T, obs, [T + 1] * 3, P, sse, st, it)
    torch.cuda.synchronize()
    with gpu.device.MultiDevice([0]) as md:
        assert md.n == 1
        Pf = torch.full((S,), 123.0, dtype=torch.float64, device=dev)
        sse2, st2, it2 = torch.empty_like(sse), torch.empty_like(st), torch.empty_like(it)
        for _ in range(2):                                                  # the handle is reusable
            md.loglik([X], [ini_d], lens, Time, 128, T, [obs], [T + 1] * 3, [Pf], sse=[sse2], status=[st2],
                      iters_total=[it2])
            md.synchronize()
            assert torch.equal(Pf, P) and torch.equal(sse2, sse) and torch.equal(st2, st) and torch.equal(it2, it)
        Pg = torch.zeros(S, dtype=torch.float64, device=dev)               # optional outputs left out
        md.loglik([X], [ini_d], lens, Time, 128, T, [obs], [T + 1] * 3, [Pg])
        md.synchronize()
        assert torch.equal(Pg, P)
        # off-grid observations through the same entry point
        times = np.sort(np.random.default_rng(3).uniform(0, Time, 25))
        hi, dx, h = gpu.bracket_times(np.linspace(0, Time, T + 1), times)
        rep = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.tile(a, (3, 1)))).to(dev).to(dt)
        o2 = torch.full((3, 25), 19.5, dtype=torch.float64, device=dev)
        hi_d, dx_d, h_d = rep(hi, torch.int32), rep(dx, torch.float64), rep(h, torch.float64)
        P2 = torch.zeros(S, dtype=torch.float64, device=dev)
        sse3 = torch.empty((3, S), dtype=torch.float64, device=dev)
        gpu.device.loglik_obs_device(X, ini_d, lens, Time, 128, T, o2, hi_d, dx_d, h_d, [25] * 3, P2, sse3)
        md.loglik([X], [ini_d], lens, Time, 128, T, [o2], [25] * 3, [Pg], obs_hi=[hi_d], obs_dx=[dx_d], obs_h=[h_d])
        md.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(Pg, P2)
    with pytest.raises(gpu.TrplError):                                      # one RCCL rank per device
        gpu.device.MultiDevice([0, 0])


# ------------------------------------------------------------------ bench.py, N = 2 control flow
def test_bench_two_rank_rehearsal_gathers_the_single_rank_likelihoods(gpu, tmp_path):
    """bench.py --gpus 2 --backend gloo as fresh child processes sharing this box's GPU (the N > 1 path:
    sample shards, pinned kernel variant, all-gather, max-over-ranks timing) must print one contract line
    and gather exactly the likelihood vector a single rank computes for the same 4 096 samples."""
    common = ["--steps", "1", "--warmup", "0", "--T", "200", "--no-cpu-baseline", "--no-pcr", "--no-full-length", "--no-host-api",
              "--no-other-configs"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TRPL_AUTOBUILD="0")
    p1, p2 = str(tmp_path / "p1.npy"), str(tmp_path / "p2.npy")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--samples-per-gpu", "4096",
                         "--dump-p", p1] + common, env=env, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    # the plain form the driver uses for N = 1, with N = 2: bench.py starts its two ranks itself (fresh children under
    # torch.distributed.run, before the parent has imported torch or touched the GPU) and relays rank 0's line
    env2 = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                         "--samples-per-gpu", "2048", "--dump-p", p2] + common,
                        env=env2, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    line1 = json.loads(r1.stdout.strip().splitlines()[-1])
    lines2 = [ln for ln in r2.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines2) == 1                                              # ONE contract line, rank 0's
    line2 = json.loads(lines2[0])
    assert line2["n_gpus"] == 2 and line2["scaling"] == "weak" and line2["config"]["samples_total"] == 4096
    rc = line2["rccl"]                                                   # the ranks are proven, not assumed
    assert rc["world"] == 2 and [d["rank"] for d in rc["devices"]] == [0, 1] and rc["backend"] == "gloo"
    assert len({d["pid"] for d in rc["devices"]}) == 2 and rc["allgather_bytes"] == 4096 * 8 and rc["allgather_us"] > 0
    assert abs(line2["value_n1_equiv"] * 2 - line2["value"]) < 1e-6 * line2["value"] and "cpu_baseline" not in line2
    assert line1["config"]["arithmetic"] == "fast" and line1["config"]["precision"] == "fp64" and "rccl" not in line1
    # a launcher that has already set WORLD_SIZE is honoured as before (the driver's N > 1 form)
    r2b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--samples-per-gpu", "2048"] + common,
                         env=env, capture_output=True, text=True, timeout=900)
    assert r2b.returncode == 0, r2b.stderr[-2000:]
    assert json.loads(r2b.stdout.strip().splitlines()[-1])["rccl"]["world"] == 2
    assert line1["config"]["samples_total"] == 4096
    assert line2["nonconverged_systems"] == line1["nonconverged_systems"] == 0
    a, b = np.load(p1), np.load(p2)
    assert a.shape == b.shape == (1, 4096) and np.array_equal(a, b)
    # ... and the single-process form (one process, trpl_loglik_multi_dev + RCCL) on this box's one device
    p3 = str(tmp_path / "p3.npy")
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "1",
                         "--samples-per-gpu", "4096", "--steps", "1", "--warmup", "0", "--T", "200", "--dump-p", p3],
                        env=env, capture_output=True, text=True, timeout=900)
    assert r3.returncode == 0, r3.stderr[-2000:]
    line3 = json.loads(r3.stdout.strip().splitlines()[-1])
    assert line3["n_gpus"] == 1 and "ncclAllGather" in line3["config"]["collective"]
    assert np.array_equal(np.load(p3), a)


def test_rank_driver_gathers_over_rccl_on_a_one_rank_group(gpu, tmp_path):
    """The one-process-per-GPU driver with the REAL collective backend: a child process joins a 1-rank
    torch.distributed group on the `nccl` backend (= RCCL on ROCm), computes its shard with the fused call and
    gathers with dist.gather_likelihoods on the device; the gathered vector equals the direct call's."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import trpl_amd
from trpl_amd import device as tdev, workloads as wl
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
S, T, L = 301, 40, 128
ini, lens = wl.power_scan(L)
X = torch.from_numpy(wl.samples(S, seed=71)).to(dev)
ini_d = torch.from_numpy(ini).to(dev)
obs = torch.full((3, T + 1), 20.0, dtype=torch.float64, device=dev)
P = torch.zeros(S, dtype=torch.float64, device=dev)
sse = torch.empty((3, S), dtype=torch.float64, device=dev)
flags = trpl_amd._abi.pin_variant(0, 3 * S, L, T)
tdev.loglik_device(X, ini_d, lens, T * 0.025, L, T, obs, [T + 1] * 3, P, sse, flags=flags)
full = trpl_amd.dist.gather_likelihoods(P[None, :], S)
torch.cuda.synchronize()
assert full.is_cuda and tuple(full.shape) == (1, S) and torch.equal(full[0], P)
np.save(%r, full.cpu().numpy())
dist.barrier(); dist.destroy_process_group()
print("RCCL-OK", dist.is_nccl_available())
''' % (ROOT, str(tmp_path / "p.npy"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, TRPL_AUTOBUILD="0"))
    assert out.returncode == 0 and "RCCL-OK True" in out.stdout, out.stderr[-2000:]
    w = gpu.workloads
    ini, lens = w.power_scan(128)
    want = gpu.loglik(w.samples(301, seed=71), ini, lens, 1.0, 128, 40, [np.full(41, 20.0)] * 3)
    assert np.array_equal(np.load(tmp_path / "p.npy")[0], want)


def test_host_buffer_solve_writes_pl_straight_into_the_callers_memory(gpu):
    """A PL block above 8 MB is written by the kernel directly into the caller's (page-locked and mapped for the
    call) numpy buffer -- also a row-strided view of a larger array, float32 and float64 -- and equals the
    device-resident solve bit for bit; the bytes between the rows of the view are untouched."""
    import torch
    w = gpu.workloads
    ini, lens = w.power_scan(128)
    S, T = 1024, 2200
    Time = T * 0.025
    X = w.samples(S, seed=81)[:, :12]
    dev = torch.device("cuda", 0)
    Xd = torch.from_numpy(X.copy()).to(dev)
    ini_d = torch.from_numpy(ini[1]).to(dev)
    for dtype, tdt in ((np.float32, torch.float32), (np.float64, torch.float64)):
        ref = torch.empty((S, T + 1), dtype=tdt, device=dev)
        gpu.device.solve_pl_device(Xd, lens[1], Time, 128, T, ini_d, ref)
        torch.cuda.synchronize()
        ref = ref.cpu().numpy()
        plain = np.empty((S, T + 1), dtype=dtype)
        assert plain.nbytes > (8 << 20)
        _, st, it, sec = gpu.solve_pl(X, lens[1], Time, 128, T, ini[1], out=plain)
        assert sec > 0 and not st.any() and np.array_equal(plain, ref)
        big = np.full((S, T + 1 + 37), -5.0, dtype=dtype)
        view = big[:, 5:5 + T + 1]
        gpu.solve_pl(X, lens[1], Time, 128, T, ini[1], out=view)
        assert np.array_equal(view, ref)
        assert (big[:, :5] == -5.0).all() and (big[:, 5 + T + 1:] == -5.0).all()
    # the same buffer again right away (registration is per call), and from two threads at once
    from concurrent.futures import ThreadPoolExecutor
    bufs = [np.empty((S, T + 1), dtype=np.float32) for _ in range(2)]
    with ThreadPoolExecutor(2) as ex:
        list(ex.map(lambda b: gpu.solve_pl(X, lens[1], Time, 128, T, ini[1], out=b), bufs))
    assert np.array_equal(bufs[0], bufs[1])


def test_multi_rank_logic_with_a_stand_in_collective_library(gpu, tmp_path):
    """The N > 1 logic of trpl_loglik_multi_dev on a one-GPU box: three and four "ranks" on device 0
    (TRPL_MULTI_ALLOW_DUP=1) with the six RCCL entry points bound to tests/mock_rccl (stream-ordered
    device-to-device copies) instead of librccl -- uneven shards (padded exchange + unpadding with every rank
    index), equal shards (direct exchange), more ranks than samples, per-shard outputs.  Every rank's P[S]
    must equal the single launch bit for bit.  RCCL itself is exercised by the one-rank tests above."""
    so = str(tmp_path / "libmock_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", so,
                           os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp")])
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
import trpl_amd
from trpl_amd import device as tdev, workloads as wl
dev = torch.device("cuda", 0)
ini, lens = wl.power_scan(128)
ini_d = torch.from_numpy(ini).to(dev)
T, Time = 40, 1.0
obs = torch.full((3, T + 1), 20.0, dtype=torch.float64, device=dev) - 0.01 * torch.arange(T + 1, device=dev)
for n, S in ((3, 1000), (4, 1000), (3, 999), (4, 2), (2, 5121)):
    Xh = wl.samples(S, seed=91)
    X = torch.from_numpy(Xh).to(dev)
    P = torch.zeros(S, dtype=torch.float64, device=dev)
    sse = torch.empty((3, S), dtype=torch.float64, device=dev)
    st = torch.empty((3, S), dtype=torch.int32, device=dev)
    it = torch.empty((3, S), dtype=torch.int64, device=dev)
    flags = trpl_amd._abi.pin_variant(0, 3 * S, 128, T)
    tdev.loglik_device(X, ini_d, lens, Time, 128, T, obs, [T + 1] * 3, P, sse, st, it, flags=flags)
    torch.cuda.synchronize()
    with tdev.MultiDevice([0] * n) as md:
        b = md.shard_bounds(S)
        Xs = [X[lo:hi].contiguous() for lo, hi in b]
        Pf = [torch.full((S,), -7.0, dtype=torch.float64, device=dev) for _ in range(n)]
        ss = [torch.empty((3, hi - lo), dtype=torch.float64, device=dev) for lo, hi in b]
        sts = [torch.empty((3, hi - lo), dtype=torch.int32, device=dev) for lo, hi in b]
        its = [torch.empty((3, hi - lo), dtype=torch.int64, device=dev) for lo, hi in b]
        for _ in range(2):
            md.loglik(Xs, [ini_d] * n, lens, Time, 128, T, [obs] * n, [T + 1] * 3, Pf, sse=ss, status=sts, iters_total=its)
            md.synchronize()
            for r, (lo, hi) in enumerate(b):
                assert torch.equal(Pf[r], P), (n, S, r)
                assert torch.equal(ss[r], sse[:, lo:hi]) and torch.equal(sts[r], st[:, lo:hi]) and torch.equal(its[r], it[:, lo:hi])
print("MOCK-OK")
''' % ROOT
    env = dict(os.environ, TRPL_RCCL_LIBRARY=so, TRPL_MULTI_ALLOW_DUP="1", TRPL_AUTOBUILD="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "MOCK-OK" in out.stdout, (out.stdout[-500:], out.stderr[-2500:])


def test_strict_mode_is_bit_identical_to_the_oracle_on_hundreds_of_systems(gpu, oracle):
    """STRICT against the pinned oracle on a wider draw than the goldens hold: 512 Power_scan samples and 128
    Twothick samples (all six curves, the stiff 311 nm ones included) x 300 steps -- PL bit for bit (compared as
    float64 bit patterns), per-system iteration totals and status equal."""
    w = gpu.workloads
    T = 300
    Time = T * 0.025
    for name, (ini, lens), S in (("power_scan", w.power_scan(128), 512), ("twothick", w.twothick(128), 128)):
        X = w.samples(S, seed=101)[:, :12]
        for c in range(len(lens)):
            r = oracle.pvsim(X, lens[c], Time, 128, T, ini[c], nthreads=nthreads())
            pl, st, it, _ = gpu.solve_pl(X, lens[c], Time, 128, T, ini[c], strict=True)
            assert np.array_equal(st, r["status"]) and np.array_equal(it, r["iters_total"]), (name, c)
            assert np.array_equal(pl.view(np.uint64), r["plI"].view(np.uint64)), (name, c)


def test_fast_kernels_vs_oracle_over_a_longer_window(gpu, oracle):
    """Both FAST steppers against the oracle over 1200 steps (30 ns: well past the stiff start, deep into the
    two-iterations-per-step regime that dominates a production run), 256 samples x 3 curves: PL to 1e-9 above the
    floor, > 99 % of the systems with exactly the oracle's iteration total, none flagged."""
    w = gpu.workloads
    ini, lens = w.power_scan(128)
    S, T = 256, 1200
    Time = T * 0.025
    X = w.samples(S, seed=111)[:, :12]
    for c in range(3):
        r = oracle.pvsim(X, lens[c], Time, 128, T, ini[c], nthreads=nthreads())
        assert not r["status"].any()
        for kernel in ("single", "pair"):
            err, same = _check_pl_against(gpu, X, lens[c], Time, 128, T, ini[c], r["plI"], r["iters_total"], kernel)
            assert same > 0.99, (c, kernel, same)


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
