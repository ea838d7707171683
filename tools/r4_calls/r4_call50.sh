set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
( echo "# same-box A/B: a_tree = the shipped verdict (per-half scalar logic), b_fastpath = whole-wave outcomes tested first"; bash tools/ab_multi.sh 3 --steps 3 --warmup 1 ) | tee gpurun_out/r4/c50_ab.txt
timeout -k 10 300 python tools/compare_builds.py tools/ab/a_tree.so tools/ab/b_fastpath.so --S 20001 --T 120 --MAX 1000 --extreme --seed 12 --workload twothick | tail -1 | cut -c1-400
