set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
OLD=tools/ab/commit_54434c7.so
NEW=bayesian-inference-trpl_amd/libtrpl_hip.so
mkdir -p /tmp/hold && mv $OLD /tmp/hold/
( echo "# same-box A/B: a_head = the tree before the finiteness witness, b_witness = excess sums at the end of a step + repeat when they are not finite"; bash tools/ab_multi.sh 3 --steps 3 --warmup 1 ) | tee gpurun_out/r4/c36_ab.txt
mv /tmp/hold/commit_54434c7.so $OLD
(
for seed in 11 12 13 14 15 16; do
  timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 20000 --T 200 --MAX 300 --extreme --seed $seed || echo "MISMATCH extreme seed $seed"
  timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 20001 --T 120 --MAX 1000 --extreme --seed $seed --workload twothick || echo "MISMATCH extreme twothick seed $seed"
done
for seed in 1 2; do
  timeout -k 10 200 python tools/compare_builds.py $OLD $NEW --S 20000 --T 300 --MAX 2000 --wide --seed $seed --workload twothick || echo "MISMATCH seed $seed"
  timeout -k 10 200 python tools/compare_builds.py $OLD $NEW --S 20001 --T 200 --MAX 50 --wide --seed $seed || echo "MISMATCH seed $seed capped"
done
timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 65536 --T 8000 || echo "MISMATCH bench batch"
timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 32768 --T 8000 --workload twothick || echo "MISMATCH twothick"
) | tee gpurun_out/r4/c36_compare.txt
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header > gpurun_out/r4/c36_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4/c36_tests.log
