set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header > gpurun_out/r4/c41_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4/c41_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
bash tools/profile_round.sh r4_v5 || exit 1
TAG=r4v5pair bash tools/pmc_profile.sh || exit 1
timeout -k 10 200 python3 tools/scale_rehearsal.py 8000 gpurun_out/r4/configs3_shard_rehearsal_v5.json > gpurun_out/r4/scale_rehearsal_v5.log 2>&1 || echo rehearsal failed
timeout -k 10 200 python3 tools/e2e_inference.py > gpurun_out/r4/e2e_inference_v5.json 2> gpurun_out/r4/e2e_v5.err || echo e2e failed
