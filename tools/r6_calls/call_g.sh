#!/bin/bash
# round 6, call g: the production shape through the reference's call sequence -- the overlapped launches on the stepper the
# driver now pins for the systems in flight (default) against the one-system kernel each 1024-block ran before (forced)
export TRPL_AUTOBUILD=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
timeout -k 10 1000 python tools/e2e_production.py --levels A,B1024single,B1024,B16384 --no-strict --oracle-samples 0 --out $O/e2e_production_ab.json > $O/e2e_production_ab.log 2>&1
grep "^level\|Error\|error" $O/e2e_production_ab.log
