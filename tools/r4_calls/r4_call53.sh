set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
( echo "# same-box A/B of compiler scheduling options on the shipped sources (all translation units): b_bias0 = -amdgpu-schedule-metric-bias=0, c_nounclust = -amdgpu-disable-unclustered-high-rp-reschedule, d_trackers = -amdgpu-use-amdgpu-trackers"; bash tools/ab_multi.sh 3 --steps 3 --warmup 1 ) | tee gpurun_out/r4/c53_ab.txt
