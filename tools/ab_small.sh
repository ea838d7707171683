#!/bin/bash
# Same-box comparison of the builds under tools/ab/*.so on SMALL launches (the reference's 1024-sample blocks):
#   bash tools/ab_small.sh [T]    -> seconds per solve of 512 / 1024 / 2048 systems, one-system kernel forced
R=$GRAFT_REPO_ROOT
export TRPL_AUTOBUILD=0
T=${1:-8000}
for rep in 1 2; do
for lib in $R/tools/ab/*.so; do
  TRPL_LIBRARY=$lib timeout -k 10 300 python3 - $T <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch, trpl_amd
from trpl_amd import device as tdev, workloads as wl
T = int(sys.argv[1]); L = 128; dev = torch.device("cuda", 0)
ini, lens = wl.power_scan(L); ini_d = torch.from_numpy(ini[1]).to(dev)
row = []
for S in (512, 1024, 2048):
    X = torch.from_numpy(wl.samples(S)[:, :12].copy()).to(dev)
    pl = torch.empty((S, T + 1), dtype=torch.float32, device=dev)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tdev.solve_pl_device(X, lens[1], T * 0.025, L, T, ini_d, pl, flags=trpl_amd._abi.FLAG_KERNEL_SINGLE)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    row.append("%d: %.4f s" % (S, best))
print(os.path.basename(os.environ["TRPL_LIBRARY"]), "  ".join(row), flush=True)
PY
done
done
