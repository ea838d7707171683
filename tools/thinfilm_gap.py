#!/usr/bin/env python3
"""How far is the default (FAST) arithmetic from the reference evaluation (STRICT = the pinned oracle, bit for bit) on a
given workload, and which of FAST's choices sets the distance?  (round-3 review, "Next" 1b)

For every curve: PL(t) of S samples over T steps from STRICT and from a FAST kernel of the library named by TRPL_LIBRARY
(tools/build_variants.sh makes the variants: reciprocals with two Newton steps / IEEE divides in the solver's quotients,
in the pointwise reciprocals, unshared reciprocals), then
  * the deviation |PL_fast / PL_strict - 1| on the points ABOVE the cancellation floor (r = PL / (B L n0p0) >= 1e-4):
    largest, 99.9th percentile, median -- and the prefactor k of the envelope k / r, max over points of dev * r;
  * systems whose iteration total differs;
  * the squared-error sums (fused likelihood of the same samples) of the floor-free systems.
One JSON line per curve group (thickness).   python tools/thinfilm_gap.py --S 2048 --T 8000 --workload twothick"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=2048)
    ap.add_argument("--T", type=int, default=8000)
    ap.add_argument("--L", type=int, default=128)
    ap.add_argument("--tol", type=int, default=7)
    ap.add_argument("--workload", default="twothick", choices=["twothick", "power_scan"])
    ap.add_argument("--kernel", default="pair", choices=["pair", "single"])
    ap.add_argument("--flags", default="", help="extra keyword for the FAST solve, e.g. hist32")
    a = ap.parse_args()
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
    Time = T * 0.025
    kw = dict(kernel=a.kernel) if L == 128 else dict(kernel="single")
    for f in a.flags.split(","):
        if f:
            kw[f] = True
    groups = {}
    t_fast = 0.0
    for c in range(C):
        ref, st, it_ref, _ = trpl_amd.solve_pl(X[:, :12], lens[c], Time, L, T, ini[c], strict=True, tol=a.tol)
        t0 = time.perf_counter()
        pl, st2, it, sec = trpl_amd.solve_pl(X[:, :12], lens[c], Time, L, T, ini[c], tol=a.tol, **kw)
        t_fast += sec
        dx = lens[c] / L
        scale = X[:, 4] * L * X[:, 0] * X[:, 1] * dx                 # B L n0p0 in the units of PL
        r = ref / scale[:, None]
        ok = ~(st.astype(bool) | st2.astype(bool))
        above = (r >= 1e-4) & ok[:, None]
        dev = np.abs(pl / ref - 1)
        g = groups.setdefault(float(lens[c]), dict(dev=[], k=[], itdiff=0, systems=0, flagged=0, worst_by_r={}))
        g["dev"].append(dev[above])
        g["k"].append((dev * r)[above & (r < 1e-1)])
        g["itdiff"] += int((it[ok] != it_ref[ok]).sum())
        g["systems"] += int(ok.sum())
        g["flagged"] += int((~ok).sum())
        for d in range(2, -5, -1):
            m = (r >= 10.0 ** d) & (r < 10.0 ** (d + 1)) & ok[:, None]
            if m.any():
                key = "1e%d" % d
                g["worst_by_r"][key] = max(g["worst_by_r"].get(key, 0.0), float(dev[m].max()))
    # the fused likelihood of the same samples, FAST against STRICT, floor-free systems
    obs = []
    mark = (w.MARKED_POINT * trpl_amd.UNIT_CONVERSIONS)[None, :-1]
    for c in range(C):
        obs.append(np.log10(trpl_amd.solve_pl(mark, lens[c], Time, L, T, ini[c], strict=True, tol=a.tol)[0][0]))
    info_s, info_f = {}, {}
    trpl_amd.loglik(X, ini, lens, Time, L, T, obs, strict=True, info=info_s, tol=a.tol)
    lkw = dict(kw)
    trpl_amd.loglik(X, ini, lens, Time, L, T, obs, info=info_f, tol=a.tol, **lkw)
    out = dict(library=os.path.basename(os.environ.get("TRPL_LIBRARY", "libtrpl_hip.so")), workload=a.workload, S=a.S, T=T, L=L,
               tol=a.tol, kernel=kw.get("kernel"), extra=a.flags, fast_solve_seconds=round(t_fast, 4),
               fast_loglik_seconds=round(info_f["seconds"], 4), groups={})
    for length, g in groups.items():
        dev = np.concatenate(g["dev"]); k = np.concatenate(g["k"])
        cs = [c for c in range(C) if float(lens[c]) == length]
        clear = (info_s["floor_col"][cs] == -1) & (info_f["floor_col"][cs] == -1) & (info_s["status"][cs] == 0) & (info_f["status"][cs] == 0)
        gap = np.abs(info_f["sse"][cs] - info_s["sse"][cs]) / info_s["sse"][cs]
        out["groups"]["%g nm" % length] = dict(
            systems=g["systems"], flagged=g["flagged"], iteration_totals_differ=g["itdiff"],
            pl_dev_above_floor=dict(max=float(dev.max()), p999=float(np.quantile(dev, 0.999)), median=float(np.median(dev))),
            envelope_k=dict(max=float(k.max()) if k.size else None, p999=float(np.quantile(k, 0.999)) if k.size else None),
            worst_dev_by_decade_of_r=g["worst_by_r"],
            floor_free_systems=int(clear.sum()),
            sse_gap_floor_free=dict(max=float(gap[clear].max()), p999=float(np.quantile(gap[clear], 0.999)), median=float(np.median(gap[clear])),
                                    above_1e8=int((gap[clear] > 1e-8).sum())),
            floor_col_equal=int((info_s["floor_col"][cs] == info_f["floor_col"][cs]).sum()))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
