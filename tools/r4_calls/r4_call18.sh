set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
( TRPL_LIBRARY=tools/ab/stats.so timeout -k 10 200 python tools/vote_stats.py --S 8192 --T 8000 && TRPL_LIBRARY=tools/ab/stats.so timeout -k 10 200 python tools/vote_stats.py --S 8192 --T 8000 --workload twothick && TRPL_LIBRARY=tools/ab/stats.so timeout -k 10 200 python tools/vote_stats.py --S 8192 --T 400 ) | tee gpurun_out/r4/c18_vote_stats.txt
( timeout -k 10 300 python tools/compare_builds.py tools/ab/0_old.so tools/ab/b_vote.so bayesian-inference-trpl_amd/libtrpl_hip.so --S 65536 --T 8000 && timeout -k 10 300 python tools/compare_builds.py tools/ab/0_old.so tools/ab/b_vote.so --S 32768 --T 8000 --workload twothick && timeout -k 10 300 python tools/compare_builds.py tools/ab/0_old.so tools/ab/b_vote.so --S 16384 --T 400 --MAX 40 --broken ) | tee gpurun_out/r4/c18_compare_builds.txt
