#!/usr/bin/env python3
"""Observation times that are a PREFIX of the simulation grid (the shipped example data: 0.025 ns spacing, 140 - 320 ns of a
2000 ns window): the fused likelihood through the on-grid entry point (trpl_loglik: batched emission, curve-pair table) against
the off-grid one (trpl_loglik_obs: per-step emission with the interp1d arithmetic, what a shape mismatch sends the reference
into, bayeslib.py:173-191).  Same observations, same samples; prints both times and the largest difference of the likelihoods.
    python tools/bench_prefix_vs_interp.py [S]"""
import gzip
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import trpl_amd  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
gold = os.path.join(ROOT, "tests", "golden")
work = tempfile.mkdtemp()
obs_csv = os.path.join(work, "obs.csv")
with gzip.open(os.path.join(gold, "obs_balanced_full.csv.gz"), "rb") as fh, open(obs_csv, "wb") as out:
    out.write(fh.read())
ic = {"time_cutoff": 2000, "select_obs_sets": None, "noise_level": None}
e = trpl_amd.get_data([obs_csv], ic, {"log_pl": True, "self_normalize": False})[0]
ini = trpl_amd.get_initpoints(os.path.join(gold, "exc_power_scan.csv"), ic)
times, obs = [np.asarray(t) for t in e[0]], [np.asarray(v) for v in e[1]]
X = trpl_amd.workloads.samples(S)
L, T, Time = 128, 80000, 2000.0
for pl_f32 in (False, True):
    res = {}
    for name, kw in (("on-grid prefix (trpl_loglik)", {}), ("off-grid (trpl_loglik_obs)", {"times": times})):
        trpl_amd.loglik(X[:256], ini, 2000.0, Time, L, T, obs, pl_f32=pl_f32, **kw)          # warm-up
        info = {}
        t0 = time.perf_counter()
        P = trpl_amd.loglik(X, ini, 2000.0, Time, L, T, obs, info=info, pl_f32=pl_f32, **kw)
        dt = time.perf_counter() - t0
        res[name] = (P, info)
        steps = sum(len(o) for o in obs)
        print("pl_f32=%s  %-32s %.3f s  (%.3e system-timesteps/s, kernel seconds %.3f)" % (pl_f32, name, dt, S * steps / dt, info["seconds"]), flush=True)
    (Pa, ia), (Pb, ib) = res.values()
    both = np.isfinite(Pa) & np.isfinite(Pb)
    clear = both & (ia["floor_col"] == -1).all(axis=0) & (ib["floor_col"] == -1).all(axis=0)
    print("   likelihoods: max rel diff among floor-free samples %.3e (all: %.3e); iteration totals equal: %s; finite in one only: %d" % (
        np.max(np.abs(Pa[clear] / Pb[clear] - 1)), np.max(np.abs(Pa[both] / Pb[both] - 1)), np.array_equal(ia["iters_total"], ib["iters_total"]),
        int((np.isfinite(Pa) != np.isfinite(Pb)).sum())), flush=True)
