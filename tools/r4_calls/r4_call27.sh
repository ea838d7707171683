set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
( echo "# small launches, one-system kernel forced: a_stage = PCR strides >= 2 staged through LDS (shipped), b_bperm = ds_bpermute"; bash tools/ab_small.sh 8000 ) | tee gpurun_out/r4/c27_ab_small.txt
( echo "## L = 512 x 32768 x 3, tol 6 (full launch)"; bash tools/ab_multi.sh 2 --L 512 --samples-per-gpu 32768 --tol 6 --steps 2 --warmup 1 ) | tee -a gpurun_out/r4/c27_ab_small.txt
