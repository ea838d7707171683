set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/r4/c5_tests.log 2>&1; rc=$?
echo "pytest rc=$rc" >> gpurun_out/r4/c5_tests.log
tail -15 gpurun_out/r4/c5_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 120 tools/bin/ubench_f32_f64 > gpurun_out/r4/ubench_f32_f64.txt 2>&1 && cat gpurun_out/r4/ubench_f32_f64.txt
