set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
R=$PWD
# one-system kernel: hybrid (base, both2) against pure PCR in the reference's elimination order, all with accurate reciprocals
for lib in base both2 purepcr2; do
  TRPL_LIBRARY=$R/tools/ab/$lib.so timeout -k 10 200 python3 tools/thinfilm_gap.py --S 2048 --T 8000 --workload twothick --kernel single >> gpurun_out/r4/thinfilm_gap_single.jsonl 2>gpurun_out/r4/gap_single_$lib.err || { echo "gap $lib failed"; tail -3 gpurun_out/r4/gap_single_$lib.err; }
  echo "gap single $lib done"
done
# same-box speed of the reciprocal variants on Twothick and Power_scan at the bench's window
for rep in 1 2; do
  for lib in base solve2 both2; do
    for wl in twothick power_scan; do
      v=$(TRPL_LIBRARY=$R/tools/ab/$lib.so timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --workload $wl --no-cpu-baseline --no-pcr --no-full-length --no-host-api --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e' % d['value'])")
      echo "$lib $wl $v" | tee -a gpurun_out/r4/speed_variants.txt
    done
  done
done
TAG=r4twothick BENCH_EXTRA="--workload twothick" bash tools/pmc_profile.sh
TAG=r4L512 BENCH_EXTRA="--L 512 --samples-per-gpu 32768 --tol 6" bash tools/pmc_profile.sh
