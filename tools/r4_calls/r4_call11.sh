set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -m gpu -q --no-header -k "hist32_at_256" > gpurun_out/r4/c11_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4/c11_tests.log
timeout -k 10 900 python3 tools/validate_fast_vs_strict.py 65536 80000 twothick > gpurun_out/r4/validate_twothick_65536_T80000.txt 2>&1 || echo validate twothick full failed
tail -8 gpurun_out/r4/validate_twothick_65536_T80000.txt
