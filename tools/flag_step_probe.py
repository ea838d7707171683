#!/usr/bin/env python3
"""At which step do iteration-capped / hostile systems get flagged?  (Choosing inputs for the off-grid test with flagged
systems: a flag at a step t > 0 exercises the partial flush of the batched off-grid emission.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import trpl_amd as gpu  # noqa: E402

sm, w = gpu.sampler, gpu.workloads
DT = 0.025
T = 200
lo = np.array([1e8, 1e12, 0.01, 0.01, 1e-13, 1e-3, 1e-3, 1e-32, 1e-32, 0.1, 0.1, 0.1, 0])
hi = np.array([1e8, 1e18, 500, 500, 1e-8, 1e5, 1e5, 1e-26, 1e-26, 1e4, 1e4, 0.1, 0])
lg = np.array([1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0])
for wl_name in ("power_scan", "twothick"):
    ini, lens = getattr(w, wl_name)(128)
    obs = [np.full(T + 1, 18.0) - 0.01 * np.arange(T + 1)] * len(lens)
    for S, seed in ((600, 123), (600, 7)):
        X = sm.random_grid(lo * sm.UNIT_CONVERSIONS, hi * sm.UNIT_CONVERSIONS, lg, S, rng=np.random.RandomState(seed))
        for MAX in (30, 100, 400):
            info = {}
            gpu.loglik(X, ini, lens, T * DT, 128, T, obs, info=info, MAX=MAX, kernel="single")
            st = info["status"]
            fl = st[st > 0]
            vals, cnt = np.unique(fl, return_counts=True)
            print(wl_name, "seed", seed, "MAX", MAX, "flagged", len(fl), "of", st.size, "status histogram (1 + step):",
                  dict(zip(vals.tolist()[:12], cnt.tolist()[:12])), "later than step 0:", int((fl > 1).sum()), flush=True)
