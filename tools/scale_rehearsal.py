"""configs[3] (Power_scan x 524 288 samples over 8 GPUs) rehearsed on ONE GPU: the eight shards bench.py --gpus 8 would
give its ranks (same seeded draw of 524 288 samples, same contiguous shard bounds, same pinned kernel variant) are run
one after the other and timed with events.  What this gives: every shard of the configuration executed and checked
(no non-converged system), and the load-balance bound on weak-scaling efficiency -- mean / max of the shard times --
that the per-sample iteration counts imply.  What it does NOT give: the RCCL all-gather (4 MiB, one per pass), the
barrier, or any effect of eight GPUs sharing a node.  A prediction, not a measurement.
    python tools/scale_rehearsal.py [T] [out.json] [cfg4 [tol]]
cfg4: the eight shards of configs[4] instead -- L = 512 x 262 144 samples, 32 768 per GPU, fp64, tol 7 or 6."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import trpl_amd
from trpl_amd import device as tdev, workloads as wl

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
out_path = sys.argv[2] if len(sys.argv) > 2 else None
cfg4 = len(sys.argv) > 3 and sys.argv[3] == "cfg4"
tol = int(sys.argv[4]) if len(sys.argv) > 4 else 7
world, per_gpu, L, C = (8, 32768, 512, 3) if cfg4 else (8, 65536, 128, 3)
S_total = world * per_gpu
Time = T * 0.025
dev = torch.device("cuda", 0)
ini, lens = wl.power_scan(L)
ini_d = torch.from_numpy(ini).to(dev)
X_all = wl.samples(S_total)
mark = torch.from_numpy((wl.MARKED_POINT * trpl_amd.UNIT_CONVERSIONS)[None, :-1].copy()).to(dev)
obs = torch.empty((C, T + 1), dtype=torch.float64, device=dev)
for c in range(C):
    pl = torch.empty((1, T + 1), dtype=torch.float64, device=dev)
    tdev.solve_pl_device(mark, lens[c], Time, L, T, ini_d[c].contiguous(), pl, flags=trpl_amd.FLAG_STRICT)
    obs[c] = torch.log10(pl[0])
flags = trpl_amd._abi.pin_variant(0, S_total * C, L, T)
rows = []
for rank in range(world):
    lo, hi = trpl_amd.dist.shard_bounds(S_total, world, rank)
    X = torch.from_numpy(np.ascontiguousarray(X_all[lo:hi])).to(dev)
    S = hi - lo
    P = torch.zeros(S, dtype=torch.float64, device=dev)
    sse = torch.empty((C, S), dtype=torch.float64, device=dev)
    st = torch.empty((C, S), dtype=torch.int32, device=dev)
    it = torch.empty((C, S), dtype=torch.int64, device=dev)
    best = None
    for rep in range(2):                       # the first pass of the process warms the clocks
        P.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tdev.loglik_device(X, ini_d, lens, Time, L, T, obs, [T + 1] * C, P, sse, st, it, flags=flags, tol=tol)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    rows.append({"rank": rank, "samples": S, "ms": best, "inner_iterations": int(it.sum().item()),
                 "nonconverged": int((st != 0).sum().item()), "finite_likelihoods": int(torch.isfinite(P).sum().item())})
    print("shard %d [%d, %d): %.1f ms, %d iterations, %d non-converged" % (rank, lo, hi, best, rows[-1]["inner_iterations"],
                                                                           rows[-1]["nonconverged"]), flush=True)
ms = np.array([r["ms"] for r in rows])
its = np.array([r["inner_iterations"] for r in rows], dtype=float)
summary = {"T": T, "config": "configs[4] (L = 512, fp64, tol 1e-%d)" % tol if cfg4 else "configs[3]", "L": L, "samples_total": S_total, "shards": rows, "mean_ms": float(ms.mean()), "max_ms": float(ms.max()),
           "load_balance_bound_on_weak_scaling_efficiency": float(ms.mean() / ms.max()),
           "iteration_imbalance_max_over_mean": float(its.max() / its.mean()),
           "predicted_8gpu_system_timesteps_per_s_upper_bound": S_total * C * (T + 1) / (ms.max() * 1e-3),
           "note": "eight shards of the configuration run sequentially on one GPU: a load-balance bound, no collective, no node effects"}
print(json.dumps({k: v for k, v in summary.items() if k != "shards"}, indent=1))
if out_path:
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(summary, open(out_path, "w"), indent=1)
