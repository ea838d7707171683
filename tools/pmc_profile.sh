#!/bin/bash
# PMC counter passes for the stepper kernel (separate rocprofv3 runs, kernel-trace only, as the
# pool requires).  Usage on the GPU box: bash tools/pmc_profile.sh ; results under gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_${TAG:-v}_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --T 100 --no-cpu-baseline --no-pcr > $R/gpurun_out/pmc_${TAG:-v}_$tag.log 2>&1 || echo "pmc set $tag failed"
done
ls $R/gpurun_out/ | head -30
