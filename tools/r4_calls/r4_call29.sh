set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
OLD=tools/ab/commit_54434c7.so       # the library of the commit this session started from (round-4 tree before the optimistic seam, the vote, the pair-step add and the benign park)
NEW=bayesian-inference-trpl_amd/libtrpl_hip.so
(
for seed in 1 2 3; do
  timeout -k 10 200 python tools/compare_builds.py $OLD $NEW --S 20000 --T 300 --MAX 2000 --wide --seed $seed --workload twothick || echo "MISMATCH seed $seed"
  timeout -k 10 200 python tools/compare_builds.py $OLD $NEW --S 20001 --T 200 --MAX 50 --wide --seed $seed || echo "MISMATCH seed $seed capped"
done
timeout -k 10 200 python tools/compare_builds.py $OLD $NEW --S 16385 --T 400 --MAX 40 --broken || echo "MISMATCH broken"
timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 65536 --T 8000 || echo "MISMATCH bench batch"
timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 32768 --T 8000 --workload twothick || echo "MISMATCH twothick"
timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 4096 --T 4000 --L 512 --tol 6 || echo "MISMATCH L512"
) | tee gpurun_out/r4/c29_compare_commit.txt
