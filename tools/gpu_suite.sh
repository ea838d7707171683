#!/bin/bash
# On the GPU box: the -m gpu suite, then (unless the suite was killed at its limit) further commands given as arguments.
#   tools/gpu_suite.sh <tag> [cmd ...]      logs under gpurun_out/r5/<tag>_*.log
tag=$1; shift
mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r5/${tag}_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r5/${tag}_tests.log
tail -3 gpurun_out/r5/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite killed at its limit: no further GPU step"; exit $rc; fi
for cmd in "$@"; do
  echo "== $cmd"
  bash -c "$cmd" || { echo "step failed: $cmd"; exit 1; }
done
