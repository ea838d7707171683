#!/bin/bash
# round 6, call f: SQ counters at the timed window on the shipped library -- stepper_kernel<512> (tol 7 and tol 6) and the paired kernel
R=$GRAFT_REPO_ROOT
cd $R
TAG=r6L512tol7 BENCH_EXTRA="--L 512 --samples-per-gpu 32768 --tol 7" bash tools/pmc_profile.sh || exit 1
TAG=r6L512tol6 BENCH_EXTRA="--L 512 --samples-per-gpu 32768 --tol 6" bash tools/pmc_profile.sh || exit 1
TAG=r6pair bash tools/pmc_profile.sh || exit 1
