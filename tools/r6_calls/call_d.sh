#!/bin/bash
# round 6, call d: flag-step probe, the experimental library through its tests, old-vs-new bit comparison
export TRPL_AUTOBUILD=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
python tools/flag_step_probe.py > $O/flag_probe.txt 2>&1; tail -14 $O/flag_probe.txt
python -m pytest tests -m gpu -q -k "twothick_bench_window" > $O/d_tests_fix.log 2>&1; tail -3 $O/d_tests_fix.log
TRPL_LIBRARY=$R/tools/ab/r6_exp.so python -m pytest tests -m gpu -q -k "l512 or resume or bundle" > $O/d_tests_exp.log 2>&1; tail -5 $O/d_tests_exp.log
for wl in power_scan twothick; do
  TRPL_LIBRARY_ANY_ABI=1 python tools/compare_builds.py tools/ab/r5_final.so tools/ab/r6.so --S 65536 --T 8000 --workload $wl > $O/compare_r5_vs_r6_$wl.json 2>$O/compare_$wl.err; echo "compare $wl rc=$?"
done
TRPL_LIBRARY_ANY_ABI=1 python tools/compare_builds.py tools/ab/r5_final.so tools/ab/r6.so --S 16384 --T 2000 --L 512 > $O/compare_r5_vs_r6_L512.json 2>$O/compare_L512.err; echo "compare L512 rc=$?"
