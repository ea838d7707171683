#!/usr/bin/env python3
"""Iteration totals / status of a few samples of a dumped batch, window by window, for the library in TRPL_LIBRARY.
    TRPL_LIBRARY=x.so python tools/diag_sample.py X.npy 6598 [6599 ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import trpl_amd

w = trpl_amd.workloads
X = np.load(sys.argv[1])
rows = [int(r) for r in sys.argv[2:]]
ini, lens = w.twothick(128)
Xs = X[rows]
print(os.path.basename(os.environ.get("TRPL_LIBRARY", "tree")), "samples", rows)
for T in (1, 2, 4, 8, 12, 13):
    obs = [np.full(T + 1, 18.0)] * 6
    info = {}
    trpl_amd.loglik(Xs, ini, lens, T * 0.025, 128, T, obs, info=info, MAX=1000, kernel="pair")
    print("T=%3d status %s iters %s sse %s" % (T, info["status"].T.tolist(), info["iters_total"].T.tolist(),
                                               [["%016x" % v for v in r] for r in info["sse"].T.view(np.uint64).tolist()]))
