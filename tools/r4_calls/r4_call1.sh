set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 400 python -m pytest tests/test_gpu_round4.py -m gpu -q -x --no-header -rA > gpurun_out/r4/c1_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/c1_tests.log
tail -15 gpurun_out/r4/c1_tests.log
timeout -k 10 120 tools/bin/ubench_f32_f64 > gpurun_out/r4/ubench_f32_f64.txt 2>&1 && cat gpurun_out/r4/ubench_f32_f64.txt
for lib in base solve2 rows2 both2 unshared ieee; do
  TRPL_LIBRARY=$PWD/tools/ab/$lib.so timeout -k 10 200 python3 tools/thinfilm_gap.py --S 2048 --T 8000 --workload twothick >> gpurun_out/r4/thinfilm_gap_variants.jsonl 2>gpurun_out/r4/gap_$lib.err || { echo "gap $lib failed"; tail -3 gpurun_out/r4/gap_$lib.err; }
  echo "gap $lib done"
done
TAG=r4pair bash tools/pmc_profile.sh
