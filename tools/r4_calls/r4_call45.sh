set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
timeout -k 10 500 python3 tools/validate_fast_vs_strict.py 65536 80000 > gpurun_out/r4/validate_full_config1_T80000_final.txt 2>&1 || echo validate config1 failed
tail -6 gpurun_out/r4/validate_full_config1_T80000_final.txt
timeout -k 10 900 python3 tools/validate_fast_vs_strict.py 65536 80000 twothick > gpurun_out/r4/validate_full_config2_twothick_T80000_final.txt 2>&1 || echo validate twothick full failed
tail -6 gpurun_out/r4/validate_full_config2_twothick_T80000_final.txt
