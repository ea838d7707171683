#!/usr/bin/env python3
"""Do pvSim launches from several host threads overlap on the GPU?  N threads each solve 1024-sample blocks (T = 80 000, float32
PL written into a host buffer) back to back; wall time per launch and aggregate rate for 1 / 2 / 3 / 6 threads, with a fresh
pageable buffer per launch (what driver.simulate does, like the reference) and with one pre-pinned buffer per thread."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import trpl_amd  # noqa: E402

w = trpl_amd.workloads
ini, lens = w.power_scan(128)
S, T, Time = 1024, 80000, 2000.0
X = w.samples(S)
par = [2000.0, Time, 128, T, 1, (0,), 7, 10000]
trpl_amd.pvSim(np.empty((8, T + 1), np.float32), None, None, None, X[:8, :-1], par, ini[0], init_mode="points")
REPS = 6


def worker(kind, out, k):
    buf = torch.empty((S, T + 1), dtype=torch.float32, pin_memory=True).numpy() if kind == "pinned" else (
        np.empty((S, T + 1), dtype=np.float32) if kind == "reused pageable" else None)
    secs = []
    for i in range(REPS):
        b = buf if buf is not None else np.empty((S, T + 1), dtype=np.float32)
        a = time.perf_counter()
        trpl_amd.pvSim(b, None, None, None, X[:, :-1], par, ini[(i + k) % 3], init_mode="points")
        secs.append(time.perf_counter() - a)
    out[k] = secs


for kind in ("fresh pageable", "reused pageable", "pinned"):
    for n in (1, 3, 6):
        out = {}
        th = [threading.Thread(target=worker, args=(kind, out, k)) for k in range(n)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
        per = np.median([s for v in out.values() for s in v[1:]])
        print("%-15s threads %d: wall %.2f s for %d launches = %.3f s per launch aggregate, %.3f s median per call, %.2e system-timesteps/s"
              % (kind, n, wall, n * REPS, wall / (n * REPS), per, n * REPS * S * (T + 1) / wall), flush=True)
