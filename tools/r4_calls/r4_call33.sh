set -o pipefail
export TRPL_AUTOBUILD=0
mkdir -p gpurun_out/r4
for lib in tools/ab/commit_54434c7.so bayesian-inference-trpl_amd/libtrpl_hip.so; do
  TRPL_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/diag_sample.py gpurun_out_in/c31_X.npy 6598
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/c33_diag.txt
