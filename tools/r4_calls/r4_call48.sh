set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header > gpurun_out/r4/c48_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4/c48_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
bash tools/profile_round.sh r4_v6 || exit 1
TAG=r4v6pair bash tools/pmc_profile.sh || exit 1
