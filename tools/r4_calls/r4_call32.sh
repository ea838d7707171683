set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 300 python tools/compare_builds.py tools/ab/commit_54434c7.so bayesian-inference-trpl_amd/libtrpl_hip.so tools/ab/v_noadd.so tools/ab/v_novote.so tools/ab/v_noopt.so --S 20001 --T 120 --MAX 1000 --extreme --seed 12 --workload twothick | tee gpurun_out/r4/c32_which.txt
