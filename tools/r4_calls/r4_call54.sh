set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 1000 python tools/seam_campaign.py 100 80 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/c54_seam_campaign.txt
