set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header > gpurun_out/r4/c21_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r4/c21_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
( echo "# same-box A/B, lane^32 exchange of the one-system steppers: a_bperm = ds_bpermute (shipped), b_swap = v_permlane32_swap"; echo "## L = 512 x 32768 x 3, tol 6"; bash tools/ab_multi.sh 3 --L 512 --samples-per-gpu 32768 --tol 6 --steps 2 --warmup 1 ) | tee gpurun_out/r4/c21_ab_partner.txt
