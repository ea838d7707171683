"""U1, the stand-alone batched tridiagonal solve (a-4 `pcreduce`, pvSimPCR.py:42-81; trpl_pcr_solve_batched[_dev]) against the
reference's golden vectors and the oracle: STRICT bit-identical, FAST to rounding, fp32, many systems."""
import numpy as np
import pytest


pytestmark = pytest.mark.gpu


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
