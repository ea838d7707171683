set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
OLD=tools/ab/commit_54434c7.so
NEW=bayesian-inference-trpl_amd/libtrpl_hip.so
(
for seed in 51 52 53 54 55 56; do
  timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 8000 --T 150 --MAX 200 --extreme --seed $seed --kernel single > /tmp/o.txt || echo "MISMATCH single seed $seed"; python -c "import json;d=json.load(open('/tmp/o.txt'));print(d['workload'],d['L'],d['kernel'],d['seed'],d['flagged_systems'],d['builds'])"
  timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 4000 --T 100 --MAX 300 --extreme --seed $seed --L 512 --tol 6 > /tmp/o.txt || echo "MISMATCH L512 seed $seed"; python -c "import json;d=json.load(open('/tmp/o.txt'));print(d['workload'],d['L'],d['kernel'],d['seed'],d['flagged_systems'],d['builds'])"
done
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/c44_compare_single.txt
