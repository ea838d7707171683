"""State snapshots plN / plP / plE (f-4; pvSimPCR.py:283-288, Legacy/pvSim.py:121-126,:169-171; trpl_solve_pl_snap[_dev])
against the oracle (STRICT bit for bit) and Legacy/pvSim.py's own output, through the drop-in signature, with non-convergence,
small grids and the device entry point."""
import numpy as np
import pytest

from gpu_common import nthreads

pytestmark = pytest.mark.gpu


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
