#!/bin/bash
# Same-box comparison of several library builds under tools/ab/*.so (alternating): bash tools/ab_multi.sh [reps] [bench args]
R=$GRAFT_REPO_ROOT
export TRPL_AUTOBUILD=0        # the library travels with the snapshot: never start a build under the profiler or between A/B runs
REPS=${1:-2}
shift
for i in $(seq $REPS); do
  for lib in $R/tools/ab/*.so; do
    v=$(TRPL_LIBRARY=$lib timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcr --no-full-length --no-host-api --no-other-configs --no-e2e "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e' % d['value'])")
    echo "$(basename $lib) $v"
  done
done
