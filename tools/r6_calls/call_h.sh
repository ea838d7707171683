#!/bin/bash
# round 6, call h: the production shape through the reference's call sequence with the byte-bounded overlap (PL matrix let go after
# interpolation): default budget (32 GiB) at sims_per_gpu 1024 and 16 384, and 16 384 under a 16 GiB budget
export TRPL_AUTOBUILD=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
timeout -k 10 600 python tools/e2e_production.py --levels B1024,B16384 --no-strict --oracle-samples 0 --out $O/e2e_production_b.json > $O/e2e_production_b.log 2>&1
grep "^level\|Error\|error" $O/e2e_production_b.log
timeout -k 10 400 python tools/e2e_production.py --levels B16384 --max-host-gib 16 --no-strict --oracle-samples 0 --out $O/e2e_production_b16g.json > $O/e2e_production_b16g.log 2>&1
grep "^level\|Error\|error" $O/e2e_production_b16g.log
