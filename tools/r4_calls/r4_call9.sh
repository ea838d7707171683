set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/r4/c9_tests.log 2>&1; rc=$?
echo "pytest rc=$rc" >> gpurun_out/r4/c9_tests.log
tail -8 gpurun_out/r4/c9_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 500 python3 tools/validate_fast_vs_strict.py 65536 80000 > gpurun_out/r4/validate_full_config1_T80000.txt 2>&1 || echo validate config1 failed
tail -8 gpurun_out/r4/validate_full_config1_T80000.txt
