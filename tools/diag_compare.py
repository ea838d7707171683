#!/usr/bin/env python3
"""Which systems differ between two builds on a compare_builds.py batch, and who shares their wavefront?
    python tools/diag_compare.py OLD.so NEW.so [compare_builds.py options]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs, rest = sys.argv[1:3], sys.argv[3:]
res = []
with tempfile.TemporaryDirectory() as d:
    for i, lib in enumerate(libs):
        out = os.path.join(d, "b%d.npz" % i)
        env = dict(os.environ, TRPL_LIBRARY=os.path.abspath(lib), TRPL_AUTOBUILD="0")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_builds.py"), "--out", out] + rest, env=env, check=True)
        res.append(dict(np.load(out)))
a, b = res
C, S = a["status"].shape
diff = (a["status"] != b["status"]) | (a["iters_total"] != b["iters_total"]) | (a["floor_col"] != b["floor_col"]) \
    | (a["sse"].view(np.uint64) != b["sse"].view(np.uint64))
print("systems that differ:", int(diff.sum()), "of", diff.size)
for c, s in zip(*np.nonzero(diff)):
    print("curve %d sample %d: status %d -> %d, iters %d -> %d, floor_col %d -> %d, sse %r -> %r" % (
        c, s, a["status"][c, s], b["status"][c, s], a["iters_total"][c, s], b["iters_total"][c, s],
        a["floor_col"][c, s], b["floor_col"][c, s], a["sse"][c, s], b["sse"][c, s]))
    for cc in range(C):
        print("      same sample, curve %d: status %d / %d iters %d / %d" % (cc, a["status"][cc, s], b["status"][cc, s], a["iters_total"][cc, s], b["iters_total"][cc, s]))
    for ds in (-1, 1):
        if 0 <= s + ds < S:
            print("      sample %d, curve %d: status %d / %d iters %d / %d" % (s + ds, c, a["status"][c, s + ds], b["status"][c, s + ds], a["iters_total"][c, s + ds], b["iters_total"][c, s + ds]))
sys.path.insert(0, ROOT)
