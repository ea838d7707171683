#!/bin/bash
# Round profile on the GPU box (bash tools/profile_round.sh <tag>): the default bench line, a
# rocprofv3 --kernel-trace --stats pass of the same command, and the HBM-traffic PMC passes
# (FETCH_SIZE and WRITE_SIZE in separate runs, kernel-trace only -- never with other trace domains).
# Everything lands under gpurun_out/<tag>/; tools/parse_rocprof.py turns it into profiles/.
TAG=${1:-rX}
R=$GRAFT_REPO_ROOT
export TRPL_AUTOBUILD=0        # the library travels with the snapshot: never start a build under the profiler or between A/B runs
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
# the same GPU work as the K timed steps of the bench line above (default T, steps, warmup); the CPU legs and
# the single full-length pass are skipped so that the kernel's average over these launches is the timed one
BENCH_ARGS="--no-cpu-baseline --no-full-length --no-host-api --no-other-configs --no-e2e"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $BENCH_ARGS > $OUT/stats.log 2>&1 || echo "stats pass failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py $BENCH_ARGS > $OUT/pmc_$c.log 2>&1 || echo "pmc $c failed"
done
ls $OUT
