#!/bin/bash
# round 6, call o: full-window evidence refreshed on the shipped library -- FAST vs STRICT over T = 80 000 (configs[1] in full,
# half of configs[2]), the production shape at the fused level with its oracle subsample and STRICT comparison, the eight shards
# of configs[3] one after the other
export TRPL_AUTOBUILD=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
timeout -k 10 400 python tools/validate_fast_vs_strict.py 65536 80000 power_scan > $O/validate_full_config1_T80000.txt 2>&1; tail -4 $O/validate_full_config1_T80000.txt
timeout -k 10 500 python tools/validate_fast_vs_strict.py 32768 80000 twothick > $O/validate_twothick_32768_T80000.txt 2>&1; tail -4 $O/validate_twothick_32768_T80000.txt
timeout -k 10 400 python tools/e2e_production.py --levels A --out $O/e2e_production_A.json > $O/e2e_production_A.log 2>&1; grep "^level\|FAST vs\|oracle sub" $O/e2e_production_A.log
timeout -k 10 200 python tools/scale_rehearsal.py > $O/configs3_shard_rehearsal.json 2> $O/configs3_shard_rehearsal.err; tail -c 600 $O/configs3_shard_rehearsal.json
