#!/bin/bash
# SQ-counter passes for the stepper kernel IN THE REGIME THE BENCH TIMES (separate rocprofv3 runs, kernel-trace only,
# the program straight after `--`, as the pool requires).  One bench pass per counter set (--steps 1 --warmup 0).
# On the GPU box:   TAG=pair [PMC_T=8000] [BENCH_EXTRA="--L 512 --samples-per-gpu 32768 --tol 6"] bash tools/pmc_profile.sh
# Results under gpurun_out/pmc_<TAG>_<set>/; summarise with tools/pmc_report.py <TAG> [kernel-name substring].
# (Rounds 1-3 profiled --T 100, the first 100 steps after the excitation: ~8 iterations per step and the worst pair
# divergence -- not the 2.2 iterations per step of the T = 8000 window; round-3 review, "What's weak" 2.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export TRPL_AUTOBUILD=0        # the library travels with the snapshot: never start a build under the profiler or between A/B runs
TAG=${TAG:-v}
PMC_T=${PMC_T:-8000}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_SCA SQ_WAVES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INSTS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  if [ -n "$PMC_SETS" ] && ! echo " $PMC_SETS " | grep -q " $i "; then continue; fi      # PMC_SETS="1 4": only those passes
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/bench.py --steps 1 --warmup 0 --T $PMC_T --no-cpu-baseline --no-pcr --no-full-length --no-host-api --no-other-configs --no-e2e $BENCH_EXTRA > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || { echo "pmc set $i failed"; tail -3 $R/gpurun_out/pmc_${TAG}_$i.log; exit 1; }
  echo "pmc set $i done"
done
