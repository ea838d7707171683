#!/bin/bash
# Same-box A/B of one library under two settings of an environment switch (alternating):
#   bash tools/ab_env.sh VAR "val_a val_b" [reps] [bench args]
R=$GRAFT_REPO_ROOT
export TRPL_AUTOBUILD=0
VAR=$1; VALS=$2; REPS=${3:-2}
shift 3
for i in $(seq $REPS); do
  for v in $VALS; do
    out=$(env $VAR=$v timeout -k 10 400 python3 $R/bench.py --no-cpu-baseline --no-pcr --no-full-length --no-host-api --no-other-configs "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e it/s %.4e fail %d' % (d['value'], d['inner_iterations_per_s'], d['nonconverged_systems']))")
    echo "$VAR=$v $out"
  done
done
