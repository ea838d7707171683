set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
R=$PWD
for lib in base w2 w2e; do
  TRPL_LIBRARY=$R/tools/ab/$lib.so timeout -k 10 200 python -m pytest tests/test_gpu_round4.py -m gpu -q --no-header -k "l512_bench and hist32" > gpurun_out/r4/c7_hist32_$lib.log 2>&1; echo "hist32 test $lib rc=$?"
  cp gpurun_out/r4/test_l512_T8000_hist32.json gpurun_out/r4/hist32_accuracy_$lib.json 2>/dev/null
  grep -E "AssertionError|passed|failed" gpurun_out/r4/c7_hist32_$lib.log | head -3
done
for rep in 1 2; do
  for lib in base w2 w2e; do
    for tol in 6 7; do
      for mode in "" "--hist32"; do
        v=$(TRPL_LIBRARY=$R/tools/ab/$lib.so timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --L 512 --samples-per-gpu 32768 --tol $tol $mode --no-cpu-baseline --no-pcr --no-full-length --no-host-api --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e' % d['value'])")
        echo "$lib tol$tol ${mode:-fp64hist} $v" | tee -a gpurun_out/r4/speed_hist32.txt
      done
    done
  done
done
TAG=r4twothick BENCH_EXTRA="--workload twothick" bash tools/pmc_profile.sh
TAG=r4L512 BENCH_EXTRA="--L 512 --samples-per-gpu 32768 --tol 6" bash tools/pmc_profile.sh
