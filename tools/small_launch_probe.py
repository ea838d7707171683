"""Small launches (the reference's 1024-sample blocks and below): seconds per solve of S systems for the two FAST
steppers forced per call -- which one a launch that cannot fill the chip should run.
    python tools/small_launch_probe.py [T]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import trpl_amd
from trpl_amd import device as tdev, workloads as wl

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
L, Time = 128, T * 0.025
dev = torch.device("cuda", 0)
ini, lens = wl.power_scan(L)
ini_d = torch.from_numpy(ini[1]).to(dev)
A = trpl_amd._abi
for S in (256, 512, 1024, 2048, 3072, 4096, 6144, 8192, 12288):
    X = torch.from_numpy(wl.samples(S)[:, :12].copy()).to(dev)
    pl = torch.empty((S, T + 1), dtype=torch.float32, device=dev)
    row = []
    for name, flag in (("single", A.FLAG_KERNEL_SINGLE), ("pair", A.FLAG_KERNEL_PAIR)):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tdev.solve_pl_device(X, lens[1], Time, L, T, ini_d, pl, flags=flag)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        row.append(best)
    auto = A.lib().trpl_kernel_variant(S, L, T, 0)
    print("S=%6d  single %.4f s  pair %.4f s  pair/single %.3f   library picks %s" %
          (S, row[0], row[1], row[1] / row[0], "pair" if auto == A.KERNEL_FAST_PAIR else "single"), flush=True)
