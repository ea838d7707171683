"""What honouring max_sims_per_block costs: device-resident solve of one Power_scan curve, every sample on its own
(the paired kernel) against bundles of 2, 3 and 4 (one-system kernel, one workgroup per bundle).
python tools/bench_bundle.py [S] [T]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, trpl_amd
from trpl_amd import device as tdev, workloads as wl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device("cuda", 0); L = 128
ini, lens = wl.power_scan(L)
X = torch.from_numpy(wl.samples(S)[:, :12].copy()).to(dev)
ini_c = torch.from_numpy(np.ascontiguousarray(ini[1])).to(dev)
pl = torch.empty((S, T // 8 + 1), dtype=torch.float64, device=dev)
it = torch.zeros(S, dtype=torch.int64, device=dev)
for m in (1, 2, 3, 4):
    fl = trpl_amd._abi.flag_bundle(m)
    for rep in range(2):
        torch.cuda.synchronize(); a = time.perf_counter()
        tdev.solve_pl_device(X, lens[1], T * 0.025, L, T, ini_c, pl, iters_total=it, plT=8, flags=fl)
        torch.cuda.synchronize(); dt = time.perf_counter() - a
    print(f"max_sims_per_block {m}: {dt:.3f} s, {S * (T + 1) / dt:.3e} system-timesteps/s, "
          f"{it.sum().item() / (S * (T + 1)):.3f} iterations per system-step", flush=True)
