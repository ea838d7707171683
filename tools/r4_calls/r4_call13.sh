set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 600 python3 tools/straggler_bound.py 65536 8000 gpurun_out/r4/straggler_bound.json 2>&1 | grep -v "^{" | tee gpurun_out/r4/straggler_bound.txt
