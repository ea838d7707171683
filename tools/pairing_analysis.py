#!/usr/bin/env python3
"""Which two systems should share a wavefront of the paired stepper?  A pair pays max(itA, itB) inner iterations per time
step.  From the ORACLE's per-step iteration traces (test infrastructure; CPU only, ~90 s on 8 cores for the default
size) this prints the wave-iterations lost to the partner under several pairing rules -- the analysis behind
build_pair_table() in csrc/trpl_api.hip and DESIGN.md section 8.
    python tools/pairing_analysis.py [S=1024] [T=8000] > profiles/r3_pairing_analysis.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle      # noqa: E402
import trpl_amd    # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
w = trpl_amd.workloads
ini, lens = w.power_scan(128)
X = w.samples(65536)[:S]
st = [oracle.pvsim(X[:, :12], lens[c], T * 0.025, 128, T, ini[c], nthreads=os.cpu_count() or 1, want_step_iters=True)["step_iters"].astype(np.int64)
      for c in range(3)]
T1 = T + 1


def loss(a, b):
    return np.maximum(a, b).sum(), (a + b).sum() / 2


print("# wave-iterations lost to the wavefront partner, Power_scan x %d seeded samples x 3 curves x T = %d (oracle traces)" % (S, T))
for name, probe in (("adjacent samples of one curve (round 2)", None), ("sorted by the total of a 64-step probe", 64),
                    ("sorted by a 256-step probe", 256), ("sorted by a 1024-step probe", 1024),
                    ("sorted by the whole window's total (unknowable bound)", T1)):
    wave = work = 0
    for c in range(3):
        s = st[c]
        order = np.arange(S) if probe is None else np.argsort(s[:, :probe].sum(axis=1), kind="stable")
        a, b = s[order[0::2]], s[order[1::2]]
        if probe is not None and probe < T1:       # the probe phase itself runs in the default pairing
            w0, k0 = loss(s[0::2, :probe], s[1::2, :probe]); w1, k1 = loss(a[:, probe:], b[:, probe:])
            wave += w0 + w1; work += k0 + k1
        else:
            w_, k_ = loss(a, b); wave += w_; work += k_
    print("%-58s lost %.2f %%" % (name, 100 * (wave / work - 1)))
for (i, j, k) in ((1, 2, 0), (0, 1, 2), (0, 2, 1)):
    w1, k1 = loss(st[i], st[j]); w2, k2 = loss(st[k][0::2], st[k][1::2])
    print("curves %d,%d of ONE sample + curve %d across adjacent samples%s lost %.2f %%  (same-sample pairs alone %.2f %%, the leftover alone %.2f %%)"
          % (i, j, k, "  <- build_pair_table" if (i, j, k) == (1, 2, 0) else "                    ", 100 * ((w1 + w2) / (k1 + k2) - 1),
             100 * (w1 / k1 - 1), 100 * (w2 / k2 - 1)))
for T_ in (1000, 4000, T1):
    wv = k = 0
    for c in range(3):
        a, b = loss(st[c][0::2, :T_], st[c][1::2, :T_]); wv += a; k += b
    w1, k1 = loss(st[1][:, :T_], st[2][:, :T_]); w2, k2 = loss(st[0][0::2, :T_], st[0][1::2, :T_])
    print("first %5d steps: adjacent samples %.2f %%, build_pair_table %.2f %%" % (T_, 100 * (wv / k - 1), 100 * ((w1 + w2) / (k1 + k2) - 1)))
allst = np.concatenate(st)
tot = allst.sum(axis=1)
print("inner iterations per step: mean %.3f, after step 1000: %.3f; per-system totals: min %d, median %d, max %d (max / mean %.2f: the straggler)"
      % (allst.mean(), allst[:, 1000:].mean(), tot.min(), np.median(tot), tot.max(), tot.max() / tot.mean()))
