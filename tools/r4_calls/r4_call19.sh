set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
( echo "# same-box A/B: a_vote = verdict of a residual test taken where its terms are formed, b_defer = where it is first used"; echo "## power_scan x 65536 x 3, T = 8000"; bash tools/ab_multi.sh 3 --steps 3 --warmup 1; echo "## twothick x 65536 x 6"; bash tools/ab_multi.sh 2 --workload twothick --steps 2 --warmup 1 ) | tee gpurun_out/r4/c19_ab_defer.txt
timeout -k 10 300 python tools/compare_builds.py tools/ab/a_vote.so tools/ab/b_defer.so --S 16384 --T 2000 | tee gpurun_out/r4/c19_compare.txt
