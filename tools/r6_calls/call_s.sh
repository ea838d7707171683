#!/bin/bash
# round 6, call s (experiment): B1024 on a quarter of the production batch with the PL-matrix reuse shared / per thread / off
export TRPL_AUTOBUILD=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
for rep in 1 2; do
for mode in off shared thread; do
  TRPL_DRV_POOL=$mode timeout -k 10 200 python tools/e2e_production.py --S 32768 --levels B1024 --no-strict --oracle-samples 0 --out $O/e2e_pool_$mode.json > $O/e2e_pool_$mode.log 2>&1
  echo "$mode: $(grep '^level' $O/e2e_pool_$mode.log)"
done
done
