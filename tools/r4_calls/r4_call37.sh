set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
OLD=tools/ab/commit_54434c7.so
NEW=bayesian-inference-trpl_amd/libtrpl_hip.so
mkdir -p /tmp/hold && mv $OLD /tmp/hold/
( echo "# same-box A/B: a_head = the tree before the finiteness witness, c_witness_lane = witness on every lane's excess term (PL sums moved to the end of the step), d_witness_e = witness on the last row's new field (PL where it was)"; bash tools/ab_multi.sh 3 --steps 3 --warmup 1 ) | tee gpurun_out/r4/c38_ab.txt
mv /tmp/hold/commit_54434c7.so $OLD
(
for seed in 11 12 13 14 15 16; do
  timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 20000 --T 200 --MAX 300 --extreme --seed $seed || echo "MISMATCH extreme seed $seed"
  timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 20001 --T 120 --MAX 1000 --extreme --seed $seed --workload twothick || echo "MISMATCH extreme twothick seed $seed"
done
timeout -k 10 200 python tools/compare_builds.py $OLD $NEW --S 20000 --T 300 --MAX 2000 --wide --seed 1 --workload twothick || echo "MISMATCH wide"
timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 65536 --T 8000 || echo "MISMATCH bench batch"
) | tee gpurun_out/r4/c38_compare.txt
