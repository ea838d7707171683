set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/r4/c14_tests.log 2>&1; rc=$?
echo "pytest rc=$rc" >> gpurun_out/r4/c14_tests.log
tail -5 gpurun_out/r4/c14_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
