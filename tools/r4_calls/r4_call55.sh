set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 800 python tools/seam_campaign_snap.py 300 24 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/c55_seam_campaign_snap.txt
