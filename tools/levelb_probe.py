#!/usr/bin/env python3
"""Where a (block, curve) task of the unfused call sequence spends its wall time (single thread, sims_per_gpu = 1024,
T = 80 000, float32 PL): buffer allocation, pvSim (wall against the kernel seconds it returns), fastlog, the host's
interpolation, prob -- with a fresh buffer per task (what driver.simulate does, like the reference) and with one reused
buffer, pageable and pinned."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import trpl_amd  # noqa: E402

w = trpl_amd.workloads
ini, lens = w.power_scan(128)
S, T, Time = 1024, 80000, 2000.0
X = w.samples(S)
par = [2000.0, Time, 128, T, 1, (0,), 7, 10000]
sim_t = np.linspace(0, Time, T + 1)
times = sim_t[:12801]
vals = np.full(12801, -3.0)
mag = np.ascontiguousarray(X[:, -1])
trpl_amd.pvSim(np.empty((8, T + 1), np.float32), None, None, None, X[:8, :-1], par, ini[0], init_mode="points")


def task(buf, c):
    t = {}
    a = time.perf_counter()
    if buf is None:
        buf = np.empty((S, T + 1), dtype=np.float32)
    t["alloc"] = time.perf_counter() - a
    a = time.perf_counter()
    t["pvSim_kernel_s"] = trpl_amd.pvSim(buf, None, None, None, X[:, :-1], par, ini[c], init_mode="points")
    t["pvSim_wall"] = time.perf_counter() - a
    a = time.perf_counter()
    trpl_amd.fastlog(buf, sys.float_info.min)
    t["fastlog_wall"] = time.perf_counter() - a
    a = time.perf_counter()
    lg = trpl_amd.interp_rows(sim_t, buf, times)
    t["interp_wall"] = time.perf_counter() - a
    a = time.perf_counter()
    P = np.zeros(S)
    trpl_amd.prob(P, lg, vals, None, mag)
    t["prob_wall"] = time.perf_counter() - a
    return t


def run(label, make):
    rows = []
    for i in range(6):
        rows.append(task(make(), i % 3))
    keys = list(rows[0])
    med = {k: float(np.median([r[k] for r in rows[1:]])) for k in keys}
    print("%-34s" % label, "  ".join("%s %.3f" % (k, med[k]) for k in keys), "  total %.3f" % sum(v for k, v in med.items() if k != "pvSim_kernel_s"), flush=True)


run("fresh pageable buffer per task", lambda: None)
one = np.empty((S, T + 1), dtype=np.float32)
run("one reused pageable buffer", lambda: one)
try:
    import torch
    pinned = torch.empty((S, T + 1), dtype=torch.float32, pin_memory=True).numpy()
    run("one reused PINNED buffer", lambda: pinned)
except Exception as e:                                                    # noqa: BLE001
    print("pinned run skipped:", e)
