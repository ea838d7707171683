#!/bin/bash
# round 6, call n: the whole -m gpu suite on the final library (U1 wide-row stores changed), then the round profile r6_v2
export TRPL_AUTOBUILD=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/n_tests.log 2>&1; rc=$?; tail -4 $O/n_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile_round.sh r6_v2
