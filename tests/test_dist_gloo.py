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
    X = np.concatenate([g["X"]] * 2)[:S]
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
