set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
OLD=tools/ab/commit_54434c7.so
NEW=bayesian-inference-trpl_amd/libtrpl_hip.so
(
for seed in $(seq 21 44); do
  timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 20000 --T 150 --MAX 200 --extreme --seed $seed > /tmp/o.txt || echo "MISMATCH extreme seed $seed"; python -c "import json;d=json.load(open('/tmp/o.txt'));print(d['workload'],d['seed'],d['flagged_systems'],d['builds'])"
  timeout -k 10 300 python tools/compare_builds.py $OLD $NEW --S 20001 --T 100 --MAX 500 --extreme --seed $seed --workload twothick > /tmp/o.txt || echo "MISMATCH extreme twothick seed $seed"; python -c "import json;d=json.load(open('/tmp/o.txt'));print(d['workload'],d['seed'],d['flagged_systems'],d['builds'])"
done
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/c40_compare_extreme.txt
