"""pvSim mode (PL stored to HBM) vs fused likelihood mode, device-resident, same systems."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, trpl_amd
from trpl_amd import device as tdev, workloads as wl
dev = torch.device("cuda", 0)
S, T, L = 65536, 400, 128
ini, lens = wl.power_scan(L)
X = torch.from_numpy(wl.samples(S)).to(dev)
ini_d = torch.from_numpy(ini).to(dev)
for dt in (torch.float32, torch.float64):
    pl = torch.empty((S, T + 1), dtype=dt, device=dev)
    st = torch.empty(S, dtype=torch.int32, device=dev); it = torch.empty(S, dtype=torch.int64, device=dev)
    m12 = X[:, :12].contiguous()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tdev.solve_pl_device(m12, lens[2], T * 0.025, L, T, ini_d[2].contiguous(), pl, st, it)
        torch.cuda.synchronize(); dtm = time.perf_counter() - t0
    print(f"solve_pl {dt}: {dtm*1e3:.1f} ms  {S*(T+1)/dtm:.3e} system-steps/s  iters/s {it.sum().item()/dtm:.3e}")
obs = torch.log10(pl[:1].double()).expand(1, T + 1).contiguous()
P = torch.zeros(S, dtype=torch.float64, device=dev); sse = torch.empty((1, S), dtype=torch.float64, device=dev)
it2 = torch.empty((1, S), dtype=torch.int64, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tdev.loglik_device(X, ini_d[2:3].contiguous(), lens[2:3], T * 0.025, L, T, obs, [T + 1], P, sse, None, it2)
    torch.cuda.synchronize(); dtm = time.perf_counter() - t0
print(f"fused loglik: {dtm*1e3:.1f} ms  {S*(T+1)/dtm:.3e} system-steps/s  iters/s {it2.sum().item()/dtm:.3e}")
