set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header > gpurun_out/r4/c26_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r4/c26_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
( echo "# same-box A/B: a_prev = commit da711b5, b_new = commit 781ed03 (steps beside a parked system run with the seam selects), c_benign = a parked system is replaced by a benign one (no per-step choice)"; bash tools/ab_multi.sh 3 --steps 3 --warmup 1 ) | tee gpurun_out/r4/c26_ab.txt
timeout -k 10 300 python tools/compare_builds.py tools/ab/a_prev.so tools/ab/c_benign.so --S 20000 --T 300 --MAX 2000 --wide --workload twothick | tee gpurun_out/r4/c26_compare.txt
