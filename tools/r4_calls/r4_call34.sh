set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
for lib in tools/ab/v_opt2.so tools/ab/v_opt3.so; do
  TRPL_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/diag_sample.py gpurun_out_in/c31_X.npy 6598
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/c34_diag.txt
timeout -k 10 300 python tools/compare_builds.py tools/ab/commit_54434c7.so tools/ab/v_opt2.so --S 16384 --T 600 | tee -a gpurun_out/r4/c34_diag.txt
