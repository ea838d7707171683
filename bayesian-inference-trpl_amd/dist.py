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


# ---- posterior core over sample shards ("a reduction over S that also shards across GPUs") ----
def _allreduce_np(a, op, group, device):
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=device)
    dist.all_reduce(t, op=op, group=group)
    return t.cpu().numpy()


def posterior_weights_sharded(LL_local, tf, group=None, comm_device="cpu", local_weights=None):
    """Globally normalised posterior weights of this rank's shard of the likelihoods.

    Each rank normalises its own shard on its GPU (trpl_posterior_weights, which also returns the shard's
    max and raw sum); three scalars per rank are all-gathered and every shard is rescaled by
    exp(log z_r - logsumexp(log z)), z_r = raw_sum_r * S_r * exp(max_r): the same weights as one
    normalize() over the concatenated likelihoods, up to rounding.  local_weights(LL, tf) ->
    (W, max, raw_sum) defaults to the GPU path (tests substitute a CPU checker).
    """
    import torch
    import torch.distributed as dist
    if local_weights is None:
        from . import posterior

        def local_weights(ll, tf_):
            info = {}
            w = posterior.weights(ll, tf_, info=info)
            return w, info["max"], info["raw_sum"]
    LL_local = np.asarray(LL_local, dtype=np.float64)
    n = LL_local.size
    if n:
        W, mx, raw = local_weights(LL_local, tf)
        logz = np.log(raw) + np.log(float(n)) + mx if raw > 0 and np.isfinite(mx) else -np.inf
    else:
        W, logz = np.zeros(0), -np.inf
    world = dist.get_world_size(group)
    allz = torch.empty(world, dtype=torch.float64, device=comm_device)
    dist.all_gather_into_tensor(allz, torch.tensor([logz], dtype=torch.float64, device=comm_device), group=group)
    allz = allz.cpu().numpy()
    top = np.max(allz)
    lse = top + np.log(np.sum(np.exp(allz - top)))
    return W * np.exp(logz - lse) if n else W


def posterior_summary_sharded(V_local, W_local, group=None, comm_device="cpu", local_moments=None):
    """Weighted means / covariance / higher moments of columns V_local (D, S_r) under globally normalised
    weights W_local (S_r,): pass 1 sums are all-reduced, the second pass is centred about the global
    means (mean_in) and all-reduced again.  Returns (sums[2+D], central[D][D+2]) of the whole sample set.
    local_moments(V, W, mean_in) -> (sums, central) defaults to the GPU path."""
    import torch.distributed as dist
    if local_moments is None:
        from . import posterior
        local_moments = posterior.moments
    V_local = np.asarray(V_local, dtype=np.float64)
    D = V_local.shape[0]
    if V_local.shape[1]:
        sums, _ = local_moments(V_local, W_local, None)
    else:
        sums = np.zeros(2 + D)
    sums = _allreduce_np(sums, dist.ReduceOp.SUM, group, comm_device)
    mean = sums[2:] / sums[0]
    if V_local.shape[1]:
        _, central = local_moments(V_local, W_local, mean)
    else:
        central = np.zeros((D, D + 2))
    central = _allreduce_np(central, dist.ReduceOp.SUM, group, comm_device)
    return sums, central


def posterior_hist_sharded(hist_local, group=None, comm_device="cpu"):
    """Weighted histograms add across shards (weights already globally normalised)."""
    import torch.distributed as dist
    return _allreduce_np(hist_local, dist.ReduceOp.SUM, group, comm_device)
