set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
mkdir -p /tmp/hold && mv tools/ab/0_old.so /tmp/hold/
( echo "# same-box A/B: a_base = the committed tree (optimistic seam, sign vote), b_add = + pair-step coupling as A + C, c_add_defer1 = b + deferred verdicts in the one-system steppers"; echo "## power_scan x 65536 x 3, T = 8000"; bash tools/ab_multi.sh 3 --steps 3 --warmup 1; echo "## L = 512 x 32768 x 3, tol 6"; bash tools/ab_multi.sh 2 --L 512 --samples-per-gpu 32768 --tol 6 --steps 2 --warmup 1 ) | tee gpurun_out/r4/c20_ab_add_defer1.txt
mv /tmp/hold/0_old.so tools/ab/
( timeout -k 10 300 python tools/compare_builds.py tools/ab/0_old.so tools/ab/b_add.so tools/ab/c_add_defer1.so --S 20000 --T 300 --MAX 2000 --wide --workload twothick && timeout -k 10 300 python tools/compare_builds.py tools/ab/0_old.so tools/ab/b_add.so tools/ab/c_add_defer1.so --S 4096 --T 2000 --L 512 --tol 6 && timeout -k 10 300 python tools/compare_builds.py tools/ab/0_old.so tools/ab/b_add.so --S 32768 --T 4000 ) | tee gpurun_out/r4/c20_compare.txt
