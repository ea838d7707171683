"""U1 (batched tridiagonal solve, operands rotated through 1.34 GB) for the library named by TRPL_LIBRARY (default:
in-tree) and the launch shape named by TRPL_PCRB_BLOCKS_PER_CU: one line  `<GB/s> <us>`  per call.
    python tools/bench_pcr_ab.py [L] [fp32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import trpl_amd
from trpl_amd import device as tdev

L = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dt = torch.float32 if len(sys.argv) > 2 else torch.float64
dev = torch.device("cuda", 0)
r = bench.bench_pcr(torch, tdev, dev, 0, L=L, dtype=dt, reps=96)
print("%.0f GB/s  %.2f us  frac %.3f  residual %.1e" % (r["achieved"], r["avg_launch_ms"] * 1e3, r["frac"], r["max_abs_residual"]))
