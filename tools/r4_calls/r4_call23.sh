set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
mkdir -p /tmp/hold && mv tools/ab/0_old.so /tmp/hold/
( echo "# same-box A/B on the round-4 paired kernel (optimistic seam, sign vote): s_setprio level around the cross-lane PCR levels (a_prio2 = shipped, 0 = off, 1, 3) and e_xm3 = stride-1 PCR level on ds_swizzle too"; echo "## power_scan x 65536 x 3, T = 8000"; bash tools/ab_multi.sh 3 --steps 3 --warmup 1 ) | tee gpurun_out/r4/c23_ab_prio_xm.txt
timeout -k 10 300 python tools/compare_builds.py /tmp/hold/0_old.so bayesian-inference-trpl_amd/libtrpl_hip.so --S 65536 --T 80000 | tee gpurun_out/r4/c23_compare_T80000.txt
