"""How much does each further resident wavefront add?  Launches of S identical-work systems (the same 64 samples tiled),
S from one wave per CU upwards, time per launch.  python tools/occupancy_probe.py L [pair]
Measured (MI355X, T = 400): paired kernel 5.2 ms up to one wave per SIMD (2048 systems), 8.6 ms at two waves per SIMD
(4096): the second wave adds 21 % throughput -- one wave alone keeps its SIMD 83 % as busy as two do.  L = 512: flat
7.8-8.4 ms up to 4 systems per CU (one wave per SIMD, the most that fits), two rounds from the fifth."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch, trpl_amd
from trpl_amd import device as tdev, workloads as wl
L = int(sys.argv[1]); T = 400; Time = T * 0.025
dev = torch.device("cuda", 0)
ini = torch.from_numpy(wl.beer_lambert(wl.POWER_SCAN_A_CM3[1], 2000.0, L)).to(dev)
flag = trpl_amd._abi.FLAG_KERNEL_SINGLE if len(sys.argv) < 3 else trpl_amd._abi.FLAG_KERNEL_PAIR
for S in (256, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096, 6144, 8192):
    X = torch.from_numpy(np.tile(wl.samples(64)[:, :12], (S // 64, 1)).copy()).to(dev)   # identical work per system
    pl = torch.empty((S, T + 1), dtype=torch.float32, device=dev)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tdev.solve_pl_device(X, 2000.0, Time, L, T, ini, pl, flags=flag)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("L=%d S=%5d (%.2f systems per CU): %.4f s" % (L, S, S / 256, best), flush=True)
