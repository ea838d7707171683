set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
OLD=tools/ab/commit_54434c7.so
mkdir -p /tmp/hold && mv $OLD /tmp/hold/
( echo "# same-box A/B of the finiteness witness: a_witness_e = the last row's new field (one compare), b_witness_p = the sum of the lane's four new P (ready before the field update)"; bash tools/ab_multi.sh 3 --steps 3 --warmup 1 ) | tee gpurun_out/r4/c43_ab.txt
mv /tmp/hold/commit_54434c7.so $OLD
(
for seed in 11 12 13 14; do
  timeout -k 10 300 python tools/compare_builds.py $OLD tools/ab/b_witness_p.so --S 20000 --T 200 --MAX 300 --extreme --seed $seed || echo "MISMATCH extreme seed $seed"
  timeout -k 10 300 python tools/compare_builds.py $OLD tools/ab/b_witness_p.so --S 20001 --T 120 --MAX 1000 --extreme --seed $seed --workload twothick || echo "MISMATCH extreme twothick seed $seed"
done
) | tee gpurun_out/r4/c43_compare.txt
