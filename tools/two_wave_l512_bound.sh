#!/bin/bash
# Upper bound for "two wavefronts per L = 512 system"  by measurement instead of instruction counts: each
# half of such a system would run the L = 256 stepper's shape (4 rows per lane, 2 waves per SIMD), so the L = 256 kernel's
# NODE throughput -- no interface solve, no barriers -- bounds what the split kernel could reach.  Same box, same tol.
#   bash tools/two_wave_l512_bound.sh [reps]      -> node-steps/s of L = 256 x 65 536 and L = 512 x 32 768 (same node count)
R=$GRAFT_REPO_ROOT
export TRPL_AUTOBUILD=0
COMMON="--steps 2 --warmup 1 --no-cpu-baseline --no-pcr --no-full-length --no-host-api --no-other-configs --no-e2e"
for i in $(seq ${1:-2}); do
  for cfg in "256 65536" "512 32768"; do
    set -- $cfg
    timeout -k 10 300 python3 $R/bench.py --L $1 --samples-per-gpu $2 $COMMON 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
L = d['config']['L']
print('L=%d  %.4e system-timesteps/s  %.4e node-steps/s  %.3f iterations/step  frac %.3f  %s' % (L, d['value'], d['value'] * L, d['mean_inner_iterations_per_step'], d['roofline']['frac'], d['roofline']['rocprof_name']))"
  done
done
