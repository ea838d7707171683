set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
TAG=r4v7two BENCH_EXTRA="--workload twothick" bash tools/pmc_profile.sh || exit 1
TAG=r4v7L512 BENCH_EXTRA="--L 512 --samples-per-gpu 32768 --tol 6" bash tools/pmc_profile.sh
