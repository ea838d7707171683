#!/bin/bash
# round 6, call e: the whole -m gpu suite on the default library, then the round profile (bench line, rocprofv3 kernel stats,
# HBM-traffic PMC passes) of the same library
export TRPL_AUTOBUILD=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/e_tests.log 2>&1; rc=$?; tail -6 $O/e_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile_round.sh r6_v1
