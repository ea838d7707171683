set -o pipefail
export TRPL_AUTOBUILD=0
echo "== differential tests against a build WITHOUT the finiteness witness (expected: the seed-12 case fails)"
TRPL_LIBRARY=$PWD/tools/ab/x_nowitness.so timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -m gpu -q --no-header -k "hostile or nonfinite" 2>&1 | tail -6
echo "== the tree"
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -m gpu -q --no-header -k "hostile or nonfinite" 2>&1 | tail -3
