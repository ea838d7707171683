#!/usr/bin/env python3
"""The snapshot / resume instantiation of the paired kernel, optimistic against always-isolating (TRPL_FLAG_PAIR_ALWAYS_SEAM,
per call), on hostile batches: PL(t), status, iteration totals and the recorded states (raw snapshots of N, P, E at three
steps, the last of them a checkpoint) must be the same bits; so must a run resumed from that checkpoint.
    python tools/seam_campaign_snap.py [first_seed] [n]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(trpl_amd, seed, S, T, MAX, extra):
    w = trpl_amd.workloads
    ini, lens = w.power_scan(128)
    rng = np.random.RandomState(seed)
    X = w.samples(max(S, 8), seed=7)[:S, :12].copy()
    X *= 10.0 ** rng.uniform(-12, 12, size=(S, 12))
    special = np.array([0.0, -1.0, np.inf, np.nan, 1e300])
    rows = rng.choice(S, size=max(1, S // 8), replace=False)
    X[rows, rng.randint(0, 12, size=rows.size)] = special[rng.randint(0, special.size, size=rows.size)]
    t0 = T // 2
    steps = sorted(set([3, T // 4] + list(trpl_amd.checkpoint_steps(t0))))
    snaps = {}
    pl, st, it, _ = trpl_amd.solve_pl(X, lens[1], T * 0.025, 128, T, ini[1], MAX=MAX, kernel="pair", snap_steps=steps,
                                      snapshots=snaps, snap_raw=True, extra_flags=extra)
    k = [steps.index(s) for s in trpl_amd.checkpoint_steps(t0)]
    res = (t0, snaps["plN"][:, k], snaps["plP"][:, k], snaps["plE"][:, k])
    pl2 = pl.copy()
    pl2[:, t0:] = 0
    pl2, st2, it2, _ = trpl_amd.solve_pl(X, lens[1], T * 0.025, 128, T, None, MAX=MAX, kernel="pair", resume=res, out=pl2,
                                         extra_flags=extra)
    return dict(pl=pl, st=st, it=it, N=snaps["plN"], P=snaps["plP"], E=snaps["plE"], pl2=pl2, st2=st2, it2=it2)


def main():
    import trpl_amd
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    bad = 0
    for seed in range(first, first + n):
        rng = np.random.RandomState(seed)
        S = int(rng.choice([2, 5, 1023, 8192, 12001]))
        T = int(rng.choice([24, 60, 120]))
        MAX = int(rng.choice([30, 300, 1000]))
        res = [run(trpl_amd, seed, S, T, MAX, extra) for extra in (0, trpl_amd._abi.FLAG_PAIR_ALWAYS_SEAM)]
        same = all(res[0][k].tobytes() == res[1][k].tobytes() for k in res[0])
        bad += not same
        print("seed %d S=%d T=%d MAX=%d: flagged %d of %d, resumed flagged %d -> %s" % (
            seed, S, T, MAX, int((res[0]["st"] != 0).sum()), S, int((res[0]["st2"] != 0).sum()), "identical" if same else "DIFFERENT"), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
