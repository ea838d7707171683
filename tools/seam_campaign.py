#!/usr/bin/env python3
"""Differential campaign: the paired kernel's optimistic seam against its always-isolating form (both in the library;
TRPL_FLAG_PAIR_ALWAYS_SEAM selects per call) on hostile batches of varied size, window, iteration cap, workload and seed.
    python tools/seam_campaign.py [first_seed] [n] [offgrid]     -> one line per batch, exit status 1 on any difference
With `offgrid` the observations sit at irregular times off the simulation grid (trpl_loglik_obs: the batched cross-lane
emission of round 5, dense clusters and long gaps included)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(trpl_amd, workload, seed, S, T, MAX, extra, offgrid=False):
    w = trpl_amd.workloads
    ini, lens = w.twothick(128) if workload == "twothick" else w.power_scan(128)
    rng = np.random.RandomState(seed)
    X = w.samples(max(S, 8), seed=7)[:S]
    X[:, :12] *= 10.0 ** rng.uniform(-20, 20, size=(S, 12))
    special = np.array([0.0, -1.0, np.inf, -np.inf, np.nan, 1e-310, 1e300, -1e-300])
    rows = rng.choice(S, size=max(1, S // 8), replace=False)
    X[rows, rng.randint(0, 12, size=rows.size)] = special[rng.randint(0, special.size, size=rows.size)]
    info = {}
    kw = {}
    obs = [np.full(T + 1, 18.0)] * len(lens)
    if offgrid:
        Time = T * 0.025
        tt = [np.sort(np.concatenate([[0.0, Time], rng.uniform(0, Time, 3 * T), rng.uniform(0, Time / 7, 200)])) for _ in lens]
        kw["times"] = tt
        obs = [np.full(len(t), 18.0) for t in tt]
    P = trpl_amd.loglik(X, ini, lens, T * 0.025, 128, T, obs, info=info, MAX=MAX, kernel="pair", extra_flags=extra, **kw)
    return dict(P=P, sse=info["sse"], it=info["iters_total"], st=info["status"], fc=info["floor_col"])


def main():
    import trpl_amd
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    offgrid = len(sys.argv) > 3 and sys.argv[3] == "offgrid"
    bad = 0
    for seed in range(first, first + n):
        rng = np.random.RandomState(seed)
        workload = ("power_scan", "twothick")[seed % 2]
        S = int(rng.choice([1, 2, 3, 5, 64, 1023, 4097, 12001, 20000]))
        T = int(rng.choice([5, 40, 120, 300]))
        MAX = int(rng.choice([3, 50, 300, 1000]))
        res = [run(trpl_amd, workload, seed, S, T, MAX, extra, offgrid) for extra in (0, trpl_amd._abi.FLAG_PAIR_ALWAYS_SEAM)]
        same = all(res[0][k].tobytes() == res[1][k].tobytes() for k in res[0])
        bad += not same
        print("seed %d %s%s S=%d T=%d MAX=%d: systems %d flagged %d -> %s" % (
            seed, "off-grid " if offgrid else "", workload, S, T, MAX, res[0]["st"].size, int((res[0]["st"] != 0).sum()), "identical" if same else "DIFFERENT"), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
