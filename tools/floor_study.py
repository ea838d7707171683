#!/usr/bin/env python3
"""Where does PL = B (sum N P - L n0p0) stop being determined by the physics?  (round-3 review item 1)

Solves short-lifetime samples (they decay to the cancellation floor inside the window) with STRICT (the reference
evaluation, bit for bit), the two FAST kernels and -- when present -- the oracle, and tabulates the largest
relative PL deviation from STRICT per decade of
    r(t) = PL(t) / (B L n0p0)     mean excess product per node over the equilibrium product
    q(t) = PL(t) / PL(0)
so that the floor criterion of include/trpl.h (floor_col) can be set from data.
    python tools/floor_study.py [--S 256] [--T 8000] [--tau 0.3 3]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=256)
    ap.add_argument("--T", type=int, default=8000)
    ap.add_argument("--tau", type=float, nargs=2, default=[0.3, 3.0], help="tauN = tauP range, ns (log-uniform)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--workload", default="power_scan", choices=["power_scan", "twothick"])
    a = ap.parse_args()
    import trpl_amd
    w = trpl_amd.workloads
    ini, lens = w.power_scan(128) if a.workload == "power_scan" else w.twothick(128)
    X = w.samples(a.S, seed=7)
    rng = np.random.default_rng(3)
    tau = 10 ** rng.uniform(np.log10(a.tau[0]), np.log10(a.tau[1]), a.S)
    X[:, 9] = tau
    X[:, 10] = tau * 10 ** rng.uniform(-0.3, 0.3, a.S)
    Time = a.T * 0.025
    rows = []
    for c in range(len(lens)):
        ref, st, it_ref, _ = trpl_amd.solve_pl(X[:, :12], lens[c], Time, 128, a.T, ini[c], strict=True)
        dx, dt = lens[c] / 128, Time / a.T
        # non-dimensional B L n0p0, re-dimensionalised like PL (pvSimPCR.py:327-331,:393)
        base = (X[:, 4] * dt / dx ** 3) * 128 * (X[:, 0] * dx ** 3) * (X[:, 1] * dx ** 3) / (dx ** 2 * dt)
        r = ref / base[:, None]
        q = ref / ref[:, :1]
        for kern in ("single", "pair"):
            pl, st2, it, _ = trpl_amd.solve_pl(X[:, :12], lens[c], Time, 128, a.T, ini[c], kernel=kern)
            dev = np.abs(pl / ref - 1)
            dev[~np.isfinite(dev)] = np.inf
            row = {"curve": c, "length_nm": float(lens[c]), "kernel": kern, "iters_equal": int((it == it_ref).sum()), "systems": a.S, "by_r": {}, "by_q": {}}
            for name, v in (("by_r", r), ("by_q", q)):
                for d in range(2, -17, -1):
                    m = (v >= 10.0 ** d) & (v < 10.0 ** (d + 1)) & (ref > 0)
                    if m.any():
                        row[name]["1e%d" % d] = {"points": int(m.sum()), "max_dev": float(dev[m].max()),
                                                 "p99_dev": float(np.quantile(dev[m], 0.99))}
            row["ref_nonpositive_points"] = int((ref <= 0).sum())
            row["fast_nonpositive_points"] = int((pl <= 0).sum())
            rows.append(row)
            print(json.dumps(row), flush=True)
    if a.out:
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
