#!/bin/bash
# round 6, call l: U1 on wide rows (L = 512 fp32 / fp64, L = 256 fp64) -- direct stores against LDS-transposed coalesced stores
export TRPL_AUTOBUILD=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
for rep in 1 2 3; do
  for v in st0 st1 st2; do
    for cfg in "512 fp32" "512" "256" "256 fp32"; do
      echo "$v L=$cfg : $(TRPL_LIBRARY=$R/tools/ab/$v.so python tools/bench_pcr_ab.py $cfg 2>/dev/null | tail -1)"
    done
  done
done > $O/pcr_store_ab.txt 2>&1
cat $O/pcr_store_ab.txt
