#!/bin/bash
# round 6, call q: the -m gpu suite on the library with trpl_interp_rows, the per-phase probe of an unfused task, and the
# production shape through the reference's call sequence with the host interpolation off the interpreter lock
export TRPL_AUTOBUILD=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/q_tests.log 2>&1; rc=$?; tail -4 $O/q_tests.log
[ $rc -eq 0 ] || exit $rc
python tools/levelb_probe.py > $O/levelb_probe2.txt 2>&1; tail -3 $O/levelb_probe2.txt
timeout -k 10 600 python tools/e2e_production.py --levels A,B1024,B16384 --no-strict --oracle-samples 0 --out $O/e2e_production_c.json > $O/e2e_production_c.log 2>&1
grep "^level\|Error\|error" $O/e2e_production_c.log
