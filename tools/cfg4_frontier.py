"""configs[4] (L = 512 nodes, fp32): the accuracy / throughput frontier instead of one point.

For every (arithmetic, tol) pair: system-timesteps/s and inner iterations per step on the config's single-GPU
share (Power_scan x 32 768 samples x T = 8000, fused likelihood), and the error against the fp64 tol-7 solve
(which the oracle pins at L = 512, tests/test_gpu_l512.py::test_pvsim_fine_grids_vs_oracle) on a 192-sample
subset over the same window: max relative PL error over points above the cancellation floor and max relative
log-likelihood error.

    python tools/cfg4_frontier.py [out.json] [S] [T]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import trpl_amd as tp
from trpl_amd import device as tdev
from trpl_amd import workloads as wl

out_path = sys.argv[1] if len(sys.argv) > 1 else None
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
L, dt_ns = 512, 0.025
Time = T * dt_ns
dev = torch.device("cuda", 0)
ini, lens = wl.power_scan(L)
C = len(lens)
A = tp._abi
MODES = [("fp64", 0, 7), ("fp64", 0, 6), ("fp64", 0, 5), ("fp64", 0, 4),
         ("fp32 state", A.FLAG_FP32 | A.FLAG_FP32_LONG, 3), ("fp32 state", A.FLAG_FP32 | A.FLAG_FP32_LONG, 4)]
if hasattr(A, "FLAG_MIXED"):
    MODES += [("fp64 state + fp32 solve", A.FLAG_MIXED, t) for t in (7, 6, 5, 4)]

X_host = wl.samples(S)
X = torch.from_numpy(X_host).to(dev)
ini_d = torch.from_numpy(ini).to(dev)
mark = torch.from_numpy((wl.MARKED_POINT * tp.UNIT_CONVERSIONS)[None, :-1].copy()).to(dev)
obs = torch.empty((C, T + 1), dtype=torch.float64, device=dev)
for c in range(C):
    pl = torch.empty((1, T + 1), dtype=torch.float64, device=dev)
    tdev.solve_pl_device(mark, lens[c], Time, L, T, ini_d[c].contiguous(), pl, tol=7)
    obs[c] = torch.log10(pl[0])

# reference on the accuracy subset: fp64, tol 7
NS = 192
Xs = X[:NS].contiguous()
ref_pl = torch.empty((C, NS, T + 1), dtype=torch.float64, device=dev)
for c in range(C):
    tdev.solve_pl_device(Xs[:, :12].contiguous(), lens[c], Time, L, T, ini_d[c].contiguous(), ref_pl[c], tol=7)
ref_P = torch.zeros(NS, dtype=torch.float64, device=dev)
tdev.loglik_device(Xs, ini_d, lens, Time, L, T, obs, [T + 1] * C, ref_P, torch.empty((C, NS), dtype=torch.float64, device=dev),
                   tol=7)
torch.cuda.synchronize()
ref_pl_h, ref_P_h = ref_pl.cpu().numpy(), ref_P.cpu().numpy()
ok = np.abs(ref_pl_h) >= 1e-12 * np.abs(ref_pl_h[:, :, :1])

rows = []
for name, flags, tol in MODES:
    P = torch.zeros(S, dtype=torch.float64, device=dev)
    sse = torch.empty((C, S), dtype=torch.float64, device=dev)
    st = torch.empty((C, S), dtype=torch.int32, device=dev)
    it = torch.empty((C, S), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    tdev.loglik_device(X, ini_d, lens, Time, L, T, obs, [T + 1] * C, P, sse, st, it, flags=flags, tol=tol)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    nfail = int((st != 0).sum().item())
    itn = int(it.sum().item())
    # accuracy subset
    pl = torch.empty((C, NS, T + 1), dtype=torch.float64, device=dev)
    sts = torch.empty((C, NS), dtype=torch.int32, device=dev)
    for c in range(C):
        tdev.solve_pl_device(Xs[:, :12].contiguous(), lens[c], Time, L, T, ini_d[c].contiguous(), pl[c], status=sts[c],
                             flags=flags, tol=tol)
    Pa = torch.zeros(NS, dtype=torch.float64, device=dev)
    tdev.loglik_device(Xs, ini_d, lens, Time, L, T, obs, [T + 1] * C, Pa, torch.empty((C, NS), dtype=torch.float64, device=dev),
                       flags=flags, tol=tol)
    torch.cuda.synchronize()
    pl_h, Pa_h = pl.cpu().numpy(), Pa.cpu().numpy()
    good = ok & np.isfinite(pl_h)
    pl_err = float(np.max(np.abs(pl_h[good] - ref_pl_h[good]) / np.abs(ref_pl_h[good])))
    pl_err_med = float(np.median(np.max(np.where(good, np.abs(pl_h - ref_pl_h) / np.abs(ref_pl_h), 0), axis=2)))
    fin = np.isfinite(Pa_h)
    ll_rel = np.abs(Pa_h[fin] - ref_P_h[fin]) / np.abs(ref_P_h[fin])
    ll_err = float(np.max(ll_rel))
    # the same PL error over the part of every decay a measurement resolves (>= 1e-6 of its start)
    top = good & (np.abs(ref_pl_h) >= 1e-6 * np.abs(ref_pl_h[:, :, :1]))
    pl_err_top = float(np.max(np.abs(pl_h[top] - ref_pl_h[top]) / np.abs(ref_pl_h[top])))
    row = {"arithmetic": name, "tol": tol, "L": L, "S": S, "T": T, "ms": ms,
           "system_timesteps_per_s": S * C * (T + 1) / (ms * 1e-3), "inner_iterations_per_step": itn / (S * C * (T + 1)),
           "nonconverged": nfail, "subset_nonconverged": int((sts != 0).sum().item()),
           "pl_max_rel_err": pl_err, "pl_median_of_row_max_rel_err": pl_err_med, "loglik_max_rel_err": ll_err,
           "pl_max_rel_err_above_1e-6_of_start": pl_err_top, "loglik_median_rel_err": float(np.median(ll_rel))}
    rows.append(row)
    print("%-24s tol %d: %.3e system-timesteps/s  %.2f it/step  nonconv %d  PL err max %.2e (median row max %.2e)  "
          "loglik err %.2e (median %.2e)  PL err where PL >= 1e-6 PL(0): %.2e"
          % (name, tol, row["system_timesteps_per_s"], row["inner_iterations_per_step"], nfail, pl_err, pl_err_med, ll_err,
             row["loglik_median_rel_err"], pl_err_top), flush=True)
if out_path:
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(rows, open(out_path, "w"), indent=1)
