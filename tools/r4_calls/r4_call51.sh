set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header > gpurun_out/r4/c51_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4/c51_tests.log
