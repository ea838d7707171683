#!/bin/bash
# round 6, call r: the -m gpu suite on the final tree, the production shape through the reference's call sequence (PL matrices
# reused, host interpolation off the interpreter lock), then the round profile r6_v3 of the final library
export TRPL_AUTOBUILD=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/r_tests.log 2>&1; rc=$?; tail -4 $O/r_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/e2e_production.py --levels B1024,B16384 --no-strict --oracle-samples 0 --out $O/e2e_production_d.json > $O/e2e_production_d.log 2>&1
grep "^level\|Error\|error" $O/e2e_production_d.log
bash tools/profile_round.sh r6_v3
