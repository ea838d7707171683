set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
echo "== the new test against the tree before the finiteness witness (expected: FAIL)"
TRPL_LIBRARY=$PWD/tools/ab/a_head.so timeout -k 10 200 python -m pytest tests/test_gpu_round4.py -m gpu -q --no-header -k "nonfinite_in_its_last_iteration" 2>&1 | tail -4
echo "== and against the tree (expected: pass)"
timeout -k 10 200 python -m pytest tests/test_gpu_round4.py -m gpu -q --no-header -k "nonfinite_in_its_last_iteration or repeated_steps" 2>&1 | tail -3
