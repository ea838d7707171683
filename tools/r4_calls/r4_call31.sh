set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
TRPL_DUMP_X=gpurun_out/r4/c31_X.npy timeout -k 10 300 python tools/diag_compare.py tools/ab/commit_54434c7.so bayesian-inference-trpl_amd/libtrpl_hip.so --S 20001 --T 120 --MAX 1000 --extreme --seed 12 --workload twothick | tee gpurun_out/r4/c31_diag.txt
