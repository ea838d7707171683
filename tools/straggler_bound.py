"""How much does the ORDER in which a launch's systems are dispatched cost?  (open direction "stragglers")
A launch ends when its slowest chain of systems does; per-system iteration totals spread by 3x over the prior box.
One pass of the bench workload measures every sample's iteration total; the same batch is then timed in four orders:
as drawn, heaviest samples first (longest-processing-time-first with PERFECT knowledge: the bound for any predictor),
lightest first (the worst case) and sorted by the best single-parameter predictor found so far (the smaller
diffusivity, ascending).  Likelihoods are permuted back and compared bit for bit (a sample's bits do not depend on
its position).      python tools/straggler_bound.py [S=65536] [T=8000] [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import trpl_amd
from trpl_amd import device as tdev, workloads as wl

S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
out_path = sys.argv[3] if len(sys.argv) > 3 else None
L, C, Time = 128, 3, T * 0.025
dev = torch.device("cuda", 0)
ini, lens = wl.power_scan(L)
ini_d = torch.from_numpy(ini).to(dev)
Xh = wl.samples(S)
mark = torch.from_numpy((wl.MARKED_POINT * trpl_amd.UNIT_CONVERSIONS)[None, :-1].copy()).to(dev)
obs = torch.empty((C, T + 1), dtype=torch.float64, device=dev)
for c in range(C):
    pl = torch.empty((1, T + 1), dtype=torch.float64, device=dev)
    tdev.solve_pl_device(mark, lens[c], Time, L, T, ini_d[c].contiguous(), pl, flags=trpl_amd.FLAG_STRICT)
    obs[c] = torch.log10(pl[0])
flags = trpl_amd._abi.pin_variant(0, S * C, L, T)


def run(order, reps=2):
    X = torch.from_numpy(np.ascontiguousarray(Xh[order])).to(dev)
    P = torch.zeros(S, dtype=torch.float64, device=dev); sse = torch.empty((C, S), dtype=torch.float64, device=dev)
    st = torch.empty((C, S), dtype=torch.int32, device=dev); it = torch.empty((C, S), dtype=torch.int64, device=dev)
    best = 1e9
    for _ in range(reps):
        P.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tdev.loglik_device(X, ini_d, lens, Time, L, T, obs, [T + 1] * C, P, sse, st, it, flags=flags)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    back = np.empty(S); back[order] = P.cpu().numpy()
    tot = np.empty(S, dtype=np.int64); tot[order] = it.sum(0).cpu().numpy()
    return best, back, tot


ident = np.arange(S)
ms0, P0, tot = run(ident)
# periods of two samples stay together (the pairing rule works on periods): sort periods by their heavier sample
per = tot.reshape(-1, 2).max(axis=1)
heavy = np.argsort(-per, kind="stable")
orders = {"as drawn": ident,
          "heaviest first (perfect knowledge)": np.stack([2 * heavy, 2 * heavy + 1], 1).ravel(),
          "lightest first": np.stack([2 * heavy[::-1], 2 * heavy[::-1] + 1], 1).ravel()}
dmin = np.minimum(Xh[:, 2], Xh[:, 3]).reshape(-1, 2).min(axis=1)
pred = np.argsort(dmin, kind="stable")
orders["smaller diffusivity ascending (a predictor)"] = np.stack([2 * pred, 2 * pred + 1], 1).ravel()
# round 4: SAMPLES sorted individually (adjacent positions = similar samples: the leftover curve of an odd group pairs
# two adjacent samples in one wavefront, so this order also shrinks that pair's divergence)
orders["samples sorted individually, heaviest first (perfect knowledge)"] = np.argsort(-tot, kind="stable")
dmin1 = np.minimum(Xh[:, 2], Xh[:, 3])
orders["samples sorted individually by the smaller diffusivity, ascending"] = np.argsort(dmin1, kind="stable")
orders["samples sorted individually by the hole diffusivity, ascending"] = np.argsort(Xh[:, 3], kind="stable")
res = {"S": S, "T": T, "iterations_per_sample": {"min": int(tot.min()), "median": float(np.median(tot)), "max": int(tot.max())}, "orders": {}}
for name, o in orders.items():
    ms, P, _ = run(o)
    res["orders"][name] = {"ms": ms, "vs_as_drawn": ms / ms0, "likelihoods_bit_identical": bool(np.array_equal(P, P0))}
    print("%-45s %.1f ms  (%.4f of as-drawn)  bits identical: %s" % (name, ms, ms / ms0, np.array_equal(P, P0)), flush=True)
work_ms = ms0 * (tot.sum() / tot.sum())
print(json.dumps(res))
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
