#!/usr/bin/env python3
"""Condense a `rocprofv3 --marker-trace --kernel-trace` run of tools/marker_probe.py (gpurun_out/<tag>/markers) into
profiles/<tag>_marker_trace.txt: every ROCTx range the library opened around a host-buffer entry point, its duration, and
the kernels dispatched inside it -- what a marker trace of an unmodified caller of the three callables shows."""
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag, "markers")
markers = sorted(glob.glob(os.path.join(src, "*", "*_marker_api_trace.csv")), key=os.path.getmtime)[-1]
kernels = sorted(glob.glob(os.path.join(src, "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(kernels))]
lines = ["rocprofv3 --marker-trace --kernel-trace -- python3 tools/marker_probe.py   (ranges opened by libtrpl_hip.so, ABI 5)",
         "%-62s %10s  kernels dispatched while the range was open" % ("range", "ms")]
for r in csv.DictReader(open(markers)):
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    inside = [k for s, e, k in ks if a <= s <= b]
    names = []
    for k in inside:
        k = k.replace("void ", "")
        if not names or names[-1][0] != k:
            names.append([k, 0])
        names[-1][1] += 1
    lines.append("%-62s %10.3f  %s" % (r["Function"], (b - a) * 1e-6, "; ".join("%s x%d" % (k, n) for k, n in names)))
out = os.path.join(root, "profiles", tag + "_marker_trace.txt")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
