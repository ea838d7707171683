set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
( echo "# same-box A/B, round-3 library (commit cd023a7) against the round-4 tree; bench.py default window T = 8000"; echo "## power_scan x 65536 x 3"; bash tools/ab_multi.sh 3; echo "## twothick x 65536 x 6"; bash tools/ab_multi.sh 2 --workload twothick --steps 2; echo "## L = 512 x 32768 x 3, tol 6"; bash tools/ab_multi.sh 2 --L 512 --samples-per-gpu 32768 --tol 6 --steps 2 ) | tee gpurun_out/r4/ab_r3_vs_r4.txt
