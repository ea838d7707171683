set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
( echo "# same-box A/B: a_prev = commit da711b5, b_new = steps beside a parked system run with the seam selects from the start, c/d = b with -falign-loops=64 / 128"; bash tools/ab_multi.sh 3 --steps 3 --warmup 1 ) | tee gpurun_out/r4/c25_ab.txt
