set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
bash tools/profile_round.sh r4_v7 || exit 1
TAG=r4v7pair bash tools/pmc_profile.sh || exit 1
