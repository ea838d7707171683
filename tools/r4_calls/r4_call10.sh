set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -m gpu -q --no-header -k "hist32 or sentinel" > gpurun_out/r4/c10_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4/c10_tests.log
bash tools/profile_round.sh r4_v2
timeout -k 10 200 python3 tools/scale_rehearsal.py 8000 gpurun_out/r4/configs3_shard_rehearsal.json > gpurun_out/r4/scale_rehearsal.log 2>&1 || echo rehearsal failed
timeout -k 10 200 python3 tools/e2e_inference.py > gpurun_out/r4/e2e_inference.json 2> gpurun_out/r4/e2e.err || echo e2e failed
tail -2 gpurun_out/r4/scale_rehearsal.log | cut -c1-400
