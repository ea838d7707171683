set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header > gpurun_out/r4/c17_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r4/c17_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
( echo "# same-box A/B: a_novote = residual tests always reduce, b_vote = sign vote first (both with the optimistic seam)"; echo "## power_scan x 65536 x 3, T = 8000"; bash tools/ab_multi.sh 3 --steps 3 --warmup 1; echo "## twothick x 65536 x 6"; bash tools/ab_multi.sh 2 --workload twothick --steps 2 --warmup 1;  echo "## L = 512 x 32768 x 3, tol 6"; bash tools/ab_multi.sh 2 --L 512 --samples-per-gpu 32768 --tol 6 --steps 2 --warmup 1 ) | tee gpurun_out/r4/c17_ab_vote.txt
