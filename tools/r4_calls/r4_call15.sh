set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 400 python -m pytest tests/test_gpu_round4.py -m gpu -q --no-header -k "offgrid" > gpurun_out/r4/c15_tests.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r4/c15_tests.log
