"""Host-buffer entry points under concurrency and in place: calls from several host threads overlap and agree; a large PL
matrix is written by the kernel straight into the caller's mapped buffer."""
import numpy as np
import pytest


pytestmark = pytest.mark.gpu


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
