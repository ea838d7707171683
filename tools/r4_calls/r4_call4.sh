set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/r4/c4_tests.log 2>&1; rc=$?
echo "pytest rc=$rc" >> gpurun_out/r4/c4_tests.log
tail -40 gpurun_out/r4/c4_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python3 tools/floor_study.py --S 256 --T 8000 --out gpurun_out/r4/floor_study_power.json > gpurun_out/r4/floor_study_power.log 2>&1 || echo floor power failed
timeout -k 10 300 python3 tools/floor_study.py --S 256 --T 8000 --workload twothick --out gpurun_out/r4/floor_study_twothick.json > gpurun_out/r4/floor_study_twothick.log 2>&1 || echo floor twothick failed
