"""Multi-GPU sharding of the sample batch: one process per GPU, contiguous sample ranges, no
data-path collective during the solve, and a single all-gather of the per-sample likelihoods at
the end (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for tests).

The reference has no communication at all: each SLURM array task takes every num_gpus-th block
of 1024 samples and leaves the rest of P zero (bayeslib.py:131,:231).  Contiguous shards give
the same P after the gather.
"""
import numpy as np


def shard_bounds(S, world, rank):
    """[lo, hi) of rank's contiguous share of S samples; the first S % world ranks get one more
    (the same rule as trpl_shard_bounds in the C ABI, which trpl_loglik_multi shards by)."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, rem = divmod(int(S), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_likelihoods(P_local, S, group=None):
    """All-gather the per-rank likelihood shards into the full (n_exp, S) array on every rank.

    P_local: torch tensor (n_exp, hi-lo) on the device the backend communicates from.  Shards are
    padded to the largest shard so that one all_gather_into_tensor moves everything.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n_exp = P_local.shape[0]
    widest = -(-int(S) // world)
    pad = torch.zeros((n_exp, widest), dtype=P_local.dtype, device=P_local.device)
    pad[:, :P_local.shape[1]] = P_local
    flat = torch.empty((world * n_exp, widest), dtype=P_local.dtype, device=P_local.device)
    dist.all_gather_into_tensor(flat, pad.contiguous(), group=group)     # concatenates along dim 0
    out = flat.view(world, n_exp, widest)
    full = torch.empty((n_exp, int(S)), dtype=P_local.dtype, device=P_local.device)
    for r in range(world):
        lo, hi = shard_bounds(S, world, r)
        full[:, lo:hi] = out[r, :, :hi - lo]
    return full


def loglik_sharded(compute, X, S=None, group=None):
    """Run `compute(X[lo:hi]) -> torch tensor (n_exp, hi-lo)` on this rank's shard and gather."""
    import torch.distributed as dist
    S = len(X) if S is None else S
    lo, hi = shard_bounds(S, dist.get_world_size(group), dist.get_rank(group))
    return gather_likelihoods(compute(X[lo:hi]), S, group=group)
