set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
OLD=tools/ab/commit_54434c7.so
NEW=bayesian-inference-trpl_amd/libtrpl_hip.so
(
for seed in 11 12 13 14; do
  timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 20000 --T 200 --MAX 300 --extreme --seed $seed || echo "MISMATCH extreme seed $seed"
  timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 20001 --T 120 --MAX 1000 --extreme --seed $seed --workload twothick || echo "MISMATCH extreme twothick seed $seed"
done
) | tee gpurun_out/r4/c30_compare_extreme.txt
