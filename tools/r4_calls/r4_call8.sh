set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
R=$PWD
TRPL_LIBRARY=$R/tools/ab/fb.so timeout -k 10 200 python -m pytest tests/test_gpu_round4.py -m gpu -q --no-header -k "l512_bench and hist32" > gpurun_out/r4/c8_hist32_fb.log 2>&1; echo "hist32 fb rc=$?"
cp gpurun_out/r4/test_l512_T8000_hist32.json gpurun_out/r4/hist32_accuracy_fb.json
TRPL_LIBRARY=$R/tools/ab/w2e.so TAG=r4L512w2e PMC_SETS="1 4" BENCH_EXTRA="--L 512 --samples-per-gpu 32768 --tol 7 --hist32" bash tools/pmc_profile.sh
TAG=r4L512t7 PMC_SETS="1 4" BENCH_EXTRA="--L 512 --samples-per-gpu 32768 --tol 7" bash tools/pmc_profile.sh
timeout -k 10 300 python3 tools/floor_study.py --S 256 --T 8000 --out gpurun_out/r4/floor_study_power.json > gpurun_out/r4/floor_study_power.log 2>&1 || echo floor power failed
timeout -k 10 300 python3 tools/floor_study.py --S 256 --T 8000 --workload twothick --out gpurun_out/r4/floor_study_twothick.json > gpurun_out/r4/floor_study_twothick.log 2>&1 || echo floor twothick failed
timeout -k 10 500 python3 tools/validate_fast_vs_strict.py 32768 80000 twothick > gpurun_out/r4/validate_twothick_32768_T80000.txt 2>&1 || echo validate twothick failed
tail -12 gpurun_out/r4/validate_twothick_32768_T80000.txt
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -m gpu -q --no-header -k "sentinel" > gpurun_out/r4/c8_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4/c8_tests.log
