#!/bin/bash
# Same-box A/B of two builds of libtrpl_hip.so: tools/ab/libtrpl_prev.so (copy of an earlier build) against
# the in-tree library, alternating, default bench workload.  Usage on the GPU box: bash tools/ab_bench.sh [reps] [bench args]
R=$GRAFT_REPO_ROOT
export TRPL_AUTOBUILD=0        # the library travels with the snapshot: never start a build under the profiler or between A/B runs
REPS=${1:-3}
shift
for i in $(seq $REPS); do
  for which in prev cur; do
    if [ $which = prev ]; then export TRPL_LIBRARY=$R/tools/ab/libtrpl_prev.so; else unset TRPL_LIBRARY; fi
    v=$(timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcr --no-full-length --no-host-api "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e it/s %.4e' % (d['value'], d['inner_iterations_per_s']))")
    echo "$which $v"
  done
done
