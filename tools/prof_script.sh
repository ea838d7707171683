#!/bin/bash
# rocprofv3 --kernel-trace --stats of one of the tools/*.py scripts: bash tools/prof_script.sh <tag> <script.py> [args]; prints our kernels
R=$GRAFT_REPO_ROOT
TAG=$1; shift
export TRPL_AUTOBUILD=0 PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/"$@" > $R/gpurun_out/$TAG.log 2>&1 || { echo "profiled run failed"; tail -5 $R/gpurun_out/$TAG.log; exit 1; }
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$R/gpurun_out/$TAG/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "trpl" in r["Name"]:
        print("%-90s calls %5s avg %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
