#!/usr/bin/env python3
"""How often does the sign vote of the residual tests (crosslane.hpp: wave_sum_negative, residual_below2) decide, and how
often must the paired kernel still reduce?  Needs a measurement build (tools/build_variants.sh stats="-DTRPL_VOTE_STATS=1"),
which packs the wave's count of reductions into iters_total of its first system:
    TRPL_LIBRARY=tools/ab/stats.so python tools/vote_stats.py [--S 8192] [--T 8000] [--workload power_scan]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=8192)
    ap.add_argument("--T", type=int, default=8000)
    ap.add_argument("--workload", default="power_scan", choices=["power_scan", "twothick"])
    a = ap.parse_args()
    import trpl_amd
    w = trpl_amd.workloads
    L, T = 128, a.T
    ini, lens = w.twothick(L) if a.workload == "twothick" else w.power_scan(L)
    C = len(lens)
    X = w.samples(a.S)
    Time = T * 0.025
    obs = [np.full(T + 1, 20.0)] * C
    info = {}
    trpl_amd.loglik(X, ini, lens, Time, L, T, obs, info=info, kernel="pair")
    it = info["iters_total"].astype(np.int64)
    iters = it & ((1 << 24) - 1)
    redN = (it >> 24) & ((1 << 20) - 1)
    redP = it >> 44
    waves = int((redN + redP > 0).sum()) or 1
    # a wave runs max(itA, itB) iterations per step; its N tests = its iterations, bounded below by the larger total
    out = {"workload": a.workload, "S": a.S, "T": T, "systems": int(it.size), "inner_iterations": int(iters.sum()),
           "iterations_per_system_step": float(iters.sum() / it.size / T),
           "N_test_reductions": int(redN.sum()), "P_test_reductions": int(redP.sum()),
           "N_reductions_per_wave_step": float(redN.sum() / (it.size / 2) / T),
           "P_reductions_per_wave_step": float(redP.sum() / (it.size / 2) / T),
           "wave_iterations_lower_bound_per_step": float(iters.sum() / it.size / T)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
