set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
( echo "# same-box A/B, net effect of this round's kernel changes after the accuracy fix: a_commit_54434c7 = the library of commit 54434c7 (always-isolating seam, reductions in every residual test), b_final_tree = optimistic seam with both repeat conditions, benign park, sign vote, pair-step add"; echo "## power_scan x 65536 x 3, T = 8000"; bash tools/ab_multi.sh 3 --steps 3 --warmup 1; echo "## twothick x 65536 x 6"; bash tools/ab_multi.sh 2 --workload twothick --steps 2 --warmup 1; echo "## L = 512 x 32768 x 3, tol 6"; bash tools/ab_multi.sh 2 --L 512 --samples-per-gpu 32768 --tol 6 --steps 2 --warmup 1 ) | tee gpurun_out/r4/c42_ab_net.txt
