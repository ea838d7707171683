set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
OLD=tools/ab/commit_54434c7.so
NEW=bayesian-inference-trpl_amd/libtrpl_hip.so
(
timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 8192 --T 3000 --MAX 2000 --wide --seed 5 --workload twothick || echo MISMATCH
timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 8192 --T 3000 --MAX 2000 --wide --seed 6 || echo MISMATCH
timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 8192 --T 3000 --MAX 500 --extreme --seed 7 --workload twothick || echo MISMATCH
timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 8192 --T 20000 || echo MISMATCH
timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 8192 --T 20000 --kernel single || echo MISMATCH
timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 4099 --T 8000 --workload twothick --kernel pair || echo MISMATCH
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/c57_compare_long.txt
