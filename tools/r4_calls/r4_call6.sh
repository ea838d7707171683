set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -m gpu -q --no-header -k "sixteen or l512_bench" > gpurun_out/r4/c6_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4/c6_tests.log
bash tools/profile_round.sh r4_v1 && TAG=r4pair bash tools/pmc_profile.sh
