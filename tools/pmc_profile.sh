#!/bin/bash
# PMC counter passes for the stepper kernel (separate rocprofv3 runs, kernel-trace only, as the
# pool requires).  Usage on the GPU box: TAG=v3 [BENCH_EXTRA="--L 512 --samples-per-gpu 32768"] bash tools/pmc_profile.sh ;
# results under gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export TRPL_AUTOBUILD=0        # the library travels with the snapshot: never start a build under the profiler or between A/B runs
TAG=${TAG:-v}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_SCA SQ_WAVES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INSTS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/bench.py --steps 1 --warmup 0 --T 100 --no-cpu-baseline --no-pcr --no-full-length --no-host-api --no-other-configs $BENCH_EXTRA > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || echo "pmc set $i failed"
done
