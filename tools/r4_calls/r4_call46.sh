set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py -m gpu -q --no-header -k "hostile or nonfinite or repeated" 2>&1 | tail -5
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header > gpurun_out/r4/c46_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4/c46_tests.log
