#!/usr/bin/env python3
"""What a marker trace of an unmodified caller shows (include/trpl.h ABI 5: optional ROCTx ranges around the host-buffer
entry points): the reference's call sequence pvSim -> fastlog -> prob on one small block, then one fused call.
    cd /tmp && rocprofv3 --marker-trace --kernel-trace --output-format csv -d <dir> -- python3 tools/marker_probe.py
The marker CSV then lists "trpl_solve_pl (pvSim)", "trpl_log10_clamp (fastlog)", "trpl_sse_accumulate (prob)" and
"trpl_loglik (...)" ranges (tools/marker_report.py condenses it into profiles/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import trpl_amd  # noqa: E402

w = trpl_amd.workloads
ini, lens = w.power_scan(128)
S, T = 1024, 400
X = w.samples(S)
Time = T * 0.025
par = [float(lens[0]), Time, 128, T, 1, (0,), 7, 10000]
pl = np.empty((S, T + 1), dtype=np.float32)
P = np.zeros(S)
vals = np.full(T + 1, -3.0)
mag = np.ascontiguousarray(X[:, -1])
for c in range(3):                                       # bayeslib.py:117-205, one block
    par[0] = float(lens[c])
    trpl_amd.pvSim(pl, None, None, None, X[:, :-1], par, ini[c], init_mode="points")
    trpl_amd.fastlog(pl, sys.float_info.min)
    trpl_amd.prob(P, pl, vals, None, mag)
P2 = trpl_amd.loglik(X, ini, lens, Time, 128, T, [vals] * 3, pl_f32=True)
print("unfused vs fused likelihoods: max rel diff %.2e" % float(np.max(np.abs(P2 / P - 1))))
