set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
R=$PWD
for lib in fix fix_solve2 fix_cubic; do
  for wl in twothick power_scan; do
    TRPL_LIBRARY=$R/tools/ab/$lib.so timeout -k 10 200 python3 tools/thinfilm_gap.py --S 2048 --T 8000 --workload $wl --kernel pair >> gpurun_out/r4/gap_fix_pair.jsonl 2>gpurun_out/r4/gap_fix_$lib.err || { echo "gap $lib failed"; tail -3 gpurun_out/r4/gap_fix_$lib.err; }
  done
  TRPL_LIBRARY=$R/tools/ab/$lib.so timeout -k 10 200 python3 tools/thinfilm_gap.py --S 2048 --T 8000 --workload twothick --kernel single >> gpurun_out/r4/gap_fix_single.jsonl 2>>gpurun_out/r4/gap_fix_$lib.err
  echo "gap $lib done"
done
for rep in 1 2; do
  for lib in base fix fix_cubic fix_solve2; do
    for wl in power_scan twothick; do
      v=$(TRPL_LIBRARY=$R/tools/ab/$lib.so timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --workload $wl --no-cpu-baseline --no-pcr --no-full-length --no-host-api --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e' % d['value'])")
      echo "$lib $wl $v" | tee -a gpurun_out/r4/speed_fix.txt
    done
  done
done
for lib in base fix fix_cubic; do
  v=$(TRPL_LIBRARY=$R/tools/ab/$lib.so timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --L 512 --samples-per-gpu 32768 --tol 6 --no-cpu-baseline --no-pcr --no-full-length --no-host-api --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e' % d['value'])")
  echo "$lib L512tol6 $v" | tee -a gpurun_out/r4/speed_fix.txt
done
