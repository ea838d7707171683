#!/usr/bin/env python3
"""Inputs that are flagged at a step t > 0 (every iteration-capped system of the reference's box is flagged at step 0:
tools/flag_step_probe.py): a NEGATIVE radiative coefficient makes dn/dt = +|B| n p blow up in finite time."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import trpl_amd as gpu  # noqa: E402

w = gpu.workloads
DT = 0.025
ini, lens = w.power_scan(128)
for T, fac in ((400, -30.0), (400, -100.0), (1000, -10.0), (400, -300.0)):
    S = 24
    X = w.samples(S, seed=23)
    X[::2, 4] *= fac                      # every other sample: B < 0
    rng = np.random.default_rng(3)
    Time = T * DT
    times = [np.sort(rng.uniform(0.0, Time, 300)) for _ in range(3)]
    obs = [np.full(len(t), 18.0) - 0.2 * t for t in times]
    res = {}
    for name, kw in (("strict", dict(strict=True)), ("single", dict(kernel="single")), ("pair", dict(kernel="pair"))):
        info = {}
        P = gpu.loglik(X, ini, lens, Time, 128, T, obs, times=times, info=info, MAX=200, **kw)
        res[name] = (P, info)
    st = res["strict"][1]["status"]
    print("T", T, "fac", fac, "strict status:", st.tolist())
    for name in ("single", "pair"):
        i = res[name][1]
        print("  ", name, "status equal:", bool(np.array_equal(i["status"], st)), "iters equal:",
              bool(np.array_equal(i["iters_total"], res["strict"][1]["iters_total"])),
              "floor_col flagged:", sorted(set(i["floor_col"][st != 0].tolist())), "sse flagged inf:",
              bool(np.isinf(i["sse"][st != 0]).all()), "P -inf where flagged:", bool(np.isneginf(res[name][0][(st != 0).any(axis=0)]).all()),
              flush=True)
