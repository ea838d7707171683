"""Upper bound on what a two-wavefronts-per-system stepper could give the reference's 1024-sample blocks
(round-2 review item 7; call shape parallel_bayes_gpu.py:104, bayeslib.py:131-146).

Such a kernel would put one row per lane on 128 lanes: each of a system's two wavefronts does the work of a 64-node
system -- one-row assembly, the cross-lane elimination levels on one row per lane -- PLUS what joins them: one more
elimination level (stride 64) staged through LDS and two workgroup barriers per solve.  The existing one-system
kernel at L = 64 (one row per lane, the same cross-lane levels, no barrier) is therefore a strict lower bound on the
time of one of those wavefronts, and with 2 x 1024 of them resident (two per SIMD) the launch cannot finish sooner
than 2048 independent L = 64 systems do.  Measured here, same parameters, same window:
    t(L = 128, S = 1024, one wave per system)   vs   t(L = 64, S = 2048)
If the ratio is below 1.3 the variant cannot reach the adoption bar whatever the quality of its implementation.
    python tools/two_wave_bound.py [T]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import trpl_amd
from trpl_amd import device as tdev, workloads as wl

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
Time = T * 0.025
dev = torch.device("cuda", 0)
A = trpl_amd._abi


def best_of(S, L, flag, reps=3):
    ini, lens = wl.power_scan(L)
    X = torch.from_numpy(np.ascontiguousarray(np.tile(wl.samples(1024)[:, :12], (S // 1024 or 1, 1))[:S])).to(dev)
    ini_d = torch.from_numpy(ini[1]).to(dev)
    pl = torch.empty((S, T + 1), dtype=torch.float32, device=dev)
    it = torch.zeros(S, dtype=torch.int64, device=dev)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tdev.solve_pl_device(X, lens[1], Time, L, T, ini_d, pl, iters_total=it, flags=flag)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, int(it.sum().item())


t128, it128 = best_of(1024, 128, A.FLAG_KERNEL_SINGLE)
t128p, _ = best_of(1024, 128, A.FLAG_KERNEL_PAIR)
t64, it64 = best_of(2048, 64, 0)
t64_1k, _ = best_of(1024, 64, 0)
out = {"T": T, "L128_S1024_single_s": t128, "L128_S1024_pair_s": t128p, "L64_S2048_s": t64, "L64_S1024_s": t64_1k,
       "iterations_L128": it128, "iterations_L64_2048": it64,
       # per inner iteration, so that the different iteration counts of the two grids drop out
       "bound_speedup_per_iteration": (t128 / it128) / (t64 / (it64 / 2)),
       "bound_speedup_wall": t128 / t64}
print(json.dumps(out))
