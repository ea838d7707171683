#!/bin/bash
# round 6, call m: U1, non-temporal stores on the direct (16 bytes per lane) path too
export TRPL_AUTOBUILD=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
for rep in 1 2 3; do
  for v in st0 st2 st3; do
    for cfg in "128" "128 fp32" "256 fp32" "64" "512"; do
      echo "$v L=$cfg : $(TRPL_LIBRARY=$R/tools/ab/$v.so python tools/bench_pcr_ab.py $cfg 2>/dev/null | tail -1)"
    done
  done
done > $O/pcr_store_ab2.txt 2>&1
cat $O/pcr_store_ab2.txt
