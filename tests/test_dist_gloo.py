"""The N>1 path on CPU: world_size-2 `gloo` processes shard the samples contiguously, compute
their shard (here with the CPU oracle standing in for the device call) and all-gather P."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, S, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import trpl_amd
    g = np.load(os.path.join(GOLDEN, "bayes_e2e.npz"))
    X = np.concatenate([g["X"]] * (1 + S // len(g["X"])))[:S]
    T, tg = 12, g["tgrid"]
    e_data = [([tg[:T + 1]] * 3, [o[:T + 1] for o in g["obs0"]]), ([tg[:5]] * 3, [o[:5] for o in g["obs1"]])]

    def compute(Xs):
        P = oracle.simulate_loglik(Xs, g["ini"], 2000.0, T * 0.025, 128, T, e_data, sims_per_gpu=4) \
            if len(Xs) else np.zeros((2, 0))
        return torch.from_numpy(P)

    full = trpl_amd.dist.loglik_sharded(compute, X)
    assert full.shape == (2, S)
    np.save(os.path.join(out_dir, "P_rank%d.npy" % rank), full.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "P_single.npy"), compute(X).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_equals_single_process(tmp_path):
    S = 7                                   # odd: ranks get 4 and 3 samples
    mp.spawn(_worker, args=(2, _free_port(), S, str(tmp_path)), nprocs=2, join=True)
    single = np.load(tmp_path / "P_single.npy")
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / ("P_rank%d.npy" % r)), single)
    assert np.isfinite(single).all() and (single < 0).all()


def _worker_small(rank, world, port, S):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, ROOT)
    import trpl_amd
    lo, hi = trpl_amd.dist.shard_bounds(S, world, rank)
    local = torch.arange(lo, hi, dtype=torch.float64).repeat(3, 1) + torch.arange(3, dtype=torch.float64)[:, None] * 100
    full = trpl_amd.dist.gather_likelihoods(local, S)
    want = torch.arange(S, dtype=torch.float64).repeat(3, 1) + torch.arange(3, dtype=torch.float64)[:, None] * 100
    assert torch.equal(full, want)
    dist.destroy_process_group()


def test_gather_with_fewer_samples_than_ranks():
    mp.spawn(_worker_small, args=(2, _free_port(), 1), nprocs=2, join=True)
    mp.spawn(_worker_small, args=(2, _free_port(), 5), nprocs=2, join=True)


def test_eight_ranks_shard_and_gather_like_the_node(tmp_path):
    """configs[3]'s rank count on the CPU: world size 8 over gloo, the oracle standing in for the device call.  S = 19 (not
    divisible by 8: three ranks hold 3 samples, five hold 2 -- trpl_shard_bounds) gathers the single-process likelihoods bit
    for bit on every rank; S = 5 leaves three ranks with an empty shard.  (Eight ranks sharing one GPU is not run anywhere:
    the GPU pool admits six GPU processes per box, launcher and test runner included; tests/test_gpu_multi.py rehearses
    three ranks on the GPU and eight "ranks" inside one process.)"""
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), 19, str(tmp_path)), nprocs=world, join=True)
    single = np.load(tmp_path / "P_single.npy")
    assert single.shape == (2, 19) and np.isfinite(single).all()
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("P_rank%d.npy" % r)), single), r
    import sys
    sys.path.insert(0, ROOT)
    import trpl_amd
    sizes = [np.subtract(*trpl_amd.dist.shard_bounds(19, world, r)[::-1]) for r in range(world)]
    assert sizes == [3, 3, 3, 2, 2, 2, 2, 2]
    mp.spawn(_worker_small, args=(world, _free_port(), 5), nprocs=world, join=True)
    mp.spawn(_worker_small, args=(world, _free_port(), 524288 // 4096), nprocs=world, join=True)


# ---- posterior core over shards (dist.posterior_*_sharded); the CPU oracle stands in for the device calls ----
def _posterior_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import posterior as op
    import trpl_amd
    g = np.load(os.path.join(GOLDEN, "posterior.npz"))
    X, LL = op.filter_nan(g["X"], g["LL"])
    tf = float(g["tf"])
    cols = np.stack([np.log10(X[:, i]) if lg else X[:, i] for i, lg in zip(g["col_index"], g["col_log"])])
    lo, hi = trpl_amd.dist.shard_bounds(len(LL), world, rank)
    if rank == 1:
        lo = hi                                     # an empty shard must not disturb the others
    if rank == 2:
        lo = trpl_amd.dist.shard_bounds(len(LL), world, 1)[0]

    def local_weights(ll, tf_):
        q = ll / tf_
        mx = np.nanmax(q)
        w = np.exp(q - mx + 1000 * np.log(2) - np.log(q.size))
        return w / np.nansum(w), mx, np.nansum(w)

    def local_moments(V, W, mean_in):
        sw = W.sum()
        sums = np.concatenate([[sw, (W ** 2).sum()], V @ W])
        m = sums[2:] / sw if mean_in is None else mean_in
        Vc = V - m[:, None]
        central = np.concatenate([(Vc * W) @ Vc.T, ((Vc ** 3) @ W)[:, None], ((Vc ** 4) @ W)[:, None]], axis=1)
        return sums, central

    W = trpl_amd.dist.posterior_weights_sharded(LL[lo:hi], tf, local_weights=local_weights)
    sums, central = trpl_amd.dist.posterior_summary_sharded(cols[:, lo:hi], W, local_moments=local_moments)
    e = op.edges(*g["limits"][0], int(g["bins"]))
    k = op.bin_index(cols[0, lo:hi], e)
    h = trpl_amd.dist.posterior_hist_sharded(np.bincount(k[k >= 0], weights=W[k >= 0], minlength=int(g["bins"])))
    np.savez(os.path.join(out_dir, "post_rank%d.npz" % rank), W=W, lo=lo, hi=hi, sums=sums, central=central, h=h)
    dist.barrier()
    dist.destroy_process_group()


def test_posterior_shards_combine_to_the_single_process_result(tmp_path):
    """3 ranks (one with an empty shard): renormalised weights, all-reduced moments and histogram equal
    the reference's single-array results stored in the golden."""
    world = 3
    mp.spawn(_posterior_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, "posterior.npz"))
    parts = [np.load(os.path.join(str(tmp_path), "post_rank%d.npz" % r)) for r in range(world)]
    W = np.concatenate([p["W"] for p in parts])
    assert W.shape == g["P"].shape and np.allclose(W, g["P"], rtol=1e-12, atol=0)
    D = len(g["mean"])
    for p in parts:
        sw = p["sums"][0]
        assert abs(sw - 1) < 1e-12
        assert np.allclose(p["sums"][2:] / sw, g["mean"], rtol=1e-12)
        assert np.allclose(p["central"][:, :D] / sw, g["cov"], rtol=1e-9, atol=1e-18)
        var = np.diag(p["central"][:, :D]) / sw
        assert np.allclose(np.sqrt(p["sums"][1] * var), g["sstd"], rtol=1e-10)
        assert np.allclose((p["central"][:, D] / sw) / var ** 1.5, g["skew"], rtol=1e-9)
        assert np.allclose((p["central"][:, D + 1] / sw) / var ** 2, g["kurt"], rtol=1e-9)
        dens = p["h"] / (np.diff(g["edges"][0]) * p["h"].sum())
        assert np.allclose(dens, g["h1"][0], rtol=1e-10, atol=1e-14)
