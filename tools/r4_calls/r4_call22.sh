set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
bash tools/profile_round.sh r4_v3 || exit 1
tail -c 600 gpurun_out/r4_v3/bench.json
TAG=r4v3pair bash tools/pmc_profile.sh || exit 1
TAG=r4v3L512 PMC_SETS="1 2 4" BENCH_EXTRA="--L 512 --samples-per-gpu 32768 --tol 6" bash tools/pmc_profile.sh
