"""End-to-end on one GPU, everything device-resident: draw the parameter box (trpl_sample_box_dev), solve
and score every sample against observations synthesised at the reference's marked point
(Visualization/config.txt:57-68) with the fused kernel (trpl_loglik_dev), then the posterior core
(weights, moments, marginals).  Prints one JSON line.  Usage: python tools/e2e_inference.py [S] [T] [c]
"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
import trpl_amd
from trpl_amd import device as tdev, sampler as sm, workloads as wl

S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
c_val = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0e-4
L, dt = 128, 0.025
dev = torch.device("cuda", 0)
ini, lens = wl.power_scan(L)
C = len(lens)
lo, hi, lg = sm.DEFAULT_MINX * sm.UNIT_CONVERSIONS, sm.DEFAULT_MAXX * sm.UNIT_CONVERSIONS, sm.DEFAULT_DO_LOG


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


t0 = sync()
X = torch.empty((S, 13), dtype=torch.float64, device=dev)
tdev.sample_box_device(X, lo, hi, lg, seed=42)
t1 = sync()
ini_d = torch.from_numpy(ini).to(dev)
mark = torch.from_numpy((wl.MARKED_POINT * sm.UNIT_CONVERSIONS)[None, :-1].copy()).to(dev)
obs = torch.empty((C, T + 1), dtype=torch.float64, device=dev)
for c in range(C):
    pl = torch.empty((1, T + 1), dtype=torch.float64, device=dev)
    tdev.solve_pl_device(mark, lens[c], T * dt, L, T, ini_d[c].contiguous(), pl, flags=trpl_amd.FLAG_STRICT)
    obs[c] = torch.log10(pl[0])
P = torch.zeros(S, dtype=torch.float64, device=dev)
sse = torch.empty((C, S), dtype=torch.float64, device=dev)
status = torch.empty((C, S), dtype=torch.int32, device=dev)
iters = torch.empty((C, S), dtype=torch.int64, device=dev)
t2 = sync()
tdev.loglik_device(X, ini_d, lens, T * dt, L, T, obs, [T + 1] * C, P, sse, status, iters)
t3 = sync()
# posterior: temper by n_obs * c (marginalization_visual.py:589), weights, moments of the free parameters
n_obs = C * (T + 1)
cols = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
names = [sm.PARAM_NAMES[i] for i in cols]
Xc = X / torch.from_numpy(sm.UNIT_CONVERSIONS).to(dev)                    # back to the GUI's units
V = torch.stack([torch.log10(Xc[:, i]) if lg[i] else Xc[:, i] for i in cols]).contiguous()
W = torch.empty_like(P)
ws = tdev.posterior_workspace(len(cols))
sums = torch.zeros(2 + len(cols), dtype=torch.float64, device=dev)
central = torch.zeros((len(cols), len(cols) + 2), dtype=torch.float64, device=dev)
h = torch.zeros(32, dtype=torch.float64, device=dev)
t4 = sync()
tdev.posterior_weights_device(P, n_obs * c_val, W, ws)
tdev.posterior_moments_device(V, W, sums, central, ws)
tdev.posterior_hist_device(V[names.index("taun")], W, 1.0, 1000.0, h)
t5 = sync()
s, cen = sums.cpu().numpy(), central.cpu().numpy()
mean = s[2:] / s[0]
std = np.sqrt(np.diag(cen[:, :len(cols)]) / s[0])
truth = np.array([np.log10(wl.MARKED_POINT[i]) if lg[i] else wl.MARKED_POINT[i] for i in cols])
best = int(torch.argmax(P).item())
out = {
    "workload": "power_scan x %d samples, T=%d, observations synthesised at the marked point, c=%g" % (S, T, c_val),
    "seconds": {"sampler": t1 - t0, "solve_and_likelihood": t3 - t2, "posterior": t5 - t4},
    "nonconverged_systems": int((status != 0).sum().item()),
    "effective_sample_size": float(1.0 / s[1]),
    "max_loglik": float(P[best].item()), "median_loglik": float(torch.median(P).item()),
    "parameters": {n: {"truth": float(tr), "posterior_mean": float(m), "posterior_std": float(sd),
                       "best_sample": float(V[k, best].item())}
                   for k, (n, tr, m, sd) in enumerate(zip(names, truth, mean, std))},
    "taun_marginal_32_bins_1_to_1000": [float(v) for v in (h / h.sum()).cpu().numpy()],
}
print(json.dumps(out))
