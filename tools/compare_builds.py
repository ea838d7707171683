#!/usr/bin/env python3
"""Do two builds of the library produce the same BITS?  (same-box, one child process per library)

Each library under test (paths given on the command line, e.g. tools/ab/*.so from tools/build_variants.sh) runs the fused
likelihood of one workload in its own process (TRPL_LIBRARY selects it) and stores P, sse, status, iters_total and
floor_col; the parent compares every array of every build with the first build's, bit for bit.  Used for changes that
must not move a result: the optimistic seam and the sign vote of the residual tests (round 4).

    python tools/compare_builds.py tools/ab/a_base.so tools/ab/b_new.so [--S 65536] [--T 8000] [--workload power_scan]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(a):
    sys.path.insert(0, ROOT)
    import trpl_amd
    w = trpl_amd.workloads
    L, T = a.L, a.T
    if a.workload == "twothick":
        ini, lens = w.twothick(L)
    else:
        ini, lens = w.power_scan(L)
        if L != 128:
            ini = np.stack([w.beer_lambert(A, 2000.0, L) for A in w.POWER_SCAN_A_CM3])
    C = len(lens)
    X = w.samples(a.S)
    if a.wide:                         # a parameter box 2-4 decades wider than the reference's (tools/fuzz_pair.py): many systems
        from trpl_amd import sampler as sm        # stop converging or turn non-finite at some step, beside partners that do not
        lo = np.array([1e8, 1e12, 0.01, 0.01, 1e-13, 1e-3, 1e-3, 1e-32, 1e-32, 0.1, 0.1, 0.1, 0])
        hi = np.array([1e8, 1e18, 500, 500, 1e-8, 1e5, 1e5, 1e-26, 1e-26, 1e4, 1e4, 0.1, 0])
        lg = np.array([1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0])
        X = sm.random_grid(lo * sm.UNIT_CONVERSIONS, hi * sm.UNIT_CONVERSIONS, lg, a.S, rng=np.random.RandomState(a.seed))
    if a.extreme:                      # hostile inputs: every parameter log-uniform over 40 decades around its box, with zeros,
        rng = np.random.RandomState(a.seed)       # negative values, infinities and NaNs sprinkled in (one sample in eight)
        X = w.samples(a.S, seed=7)
        mag = 10.0 ** rng.uniform(-20, 20, size=(a.S, 12))
        X[:, :12] *= mag
        special = np.array([0.0, -1.0, np.inf, -np.inf, np.nan, 1e-310, 1e300, -1e-300])
        rows = rng.choice(a.S, size=a.S // 8, replace=False)
        X[rows, rng.randint(0, 12, size=rows.size)] = special[rng.randint(0, special.size, size=rows.size)]
    if a.broken:                       # samples that are flagged: the repeated-step path of the paired kernel
        X[5, 9] = np.nan
        X[1000, 4] = np.inf
        X[2001, 2] = -1e9
    Time = T * 0.025
    mark = (w.MARKED_POINT * trpl_amd.UNIT_CONVERSIONS)[None, :-1]
    obs = [np.log10(trpl_amd.solve_pl(mark, lens[c], Time, L, T, ini[c], strict=True, tol=a.tol)[0][0]) for c in range(C)]
    info = {}
    kw = {}
    if a.MAX:
        kw["MAX"] = a.MAX
    if a.kernel:
        kw["kernel"] = a.kernel
    P = trpl_amd.loglik(X, ini, lens, Time, L, T, obs, tol=a.tol, info=info, **kw)
    np.savez(a.out, P=P, **{k: np.asarray(v) for k, v in info.items() if isinstance(v, np.ndarray)})
    if os.environ.get('TRPL_DUMP_X'):
        np.save(os.environ['TRPL_DUMP_X'], X)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--S", type=int, default=65536)
    ap.add_argument("--T", type=int, default=8000)
    ap.add_argument("--L", type=int, default=128)
    ap.add_argument("--tol", type=int, default=7)
    ap.add_argument("--MAX", type=int, default=0)
    ap.add_argument("--broken", action="store_true")
    ap.add_argument("--wide", action="store_true")
    ap.add_argument("--extreme", action="store_true")
    ap.add_argument("--kernel", default=None, choices=["pair", "single"], help="force the L = 128 FAST stepper")
    ap.add_argument("--seed", type=int, default=123, help="seed of the wide box's draw")
    ap.add_argument("--workload", default="power_scan", choices=["power_scan", "twothick"])
    ap.add_argument("--out", default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.out:
        return child(a)
    res = []
    with tempfile.TemporaryDirectory() as d:
        for i, lib in enumerate(a.libs):
            out = os.path.join(d, "b%d.npz" % i)
            env = dict(os.environ, TRPL_LIBRARY=os.path.abspath(lib), TRPL_AUTOBUILD="0")
            cmd = [sys.executable, os.path.abspath(__file__), "--out", out, "--S", str(a.S), "--T", str(a.T), "--L", str(a.L),
                   "--tol", str(a.tol), "--MAX", str(a.MAX), "--workload", a.workload] + (["--broken"] if a.broken else []) + (["--wide", "--seed", str(a.seed)] if a.wide else []) + (["--extreme", "--seed", str(a.seed)] if a.extreme else []) + (["--kernel", a.kernel] if a.kernel else [])
            subprocess.run(cmd, env=env, check=True)
            res.append(dict(np.load(out)))
    ref = res[0]
    report = {"workload": a.workload, "S": a.S, "T": a.T, "L": a.L, "tol": a.tol, "MAX": a.MAX, "broken": a.broken, "wide_box": a.wide, "extreme_inputs": a.extreme, "kernel": a.kernel, "seed": a.seed if (a.wide or a.extreme) else None,
              "reference": os.path.basename(a.libs[0]), "arrays": sorted(ref.keys()),
              "flagged_systems": int((ref["status"] != 0).sum()) if "status" in ref else None, "builds": {}}
    ok = True
    for lib, r in zip(a.libs[1:], res[1:]):
        same = {k: bool(ref[k].shape == r[k].shape and np.array_equal(ref[k].view(np.uint8), r[k].view(np.uint8))) for k in ref}
        report["builds"][os.path.basename(lib)] = {"identical": all(same.values()), "per_array": same}
        ok = ok and all(same.values())
    print(json.dumps(report))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
