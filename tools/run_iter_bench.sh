#!/bin/bash
# Build and run tools/iter_bench.hip for a list of ablation masks (on the GPU box).
for a in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -DTRPL_ABLATE=$a -Ibayesian-inference-trpl_amd/csrc tools/iter_bench.hip -o /tmp/ib$a > /tmp/ib$a.log 2>&1 || { echo "build $a failed"; tail -5 /tmp/ib$a.log; continue; }
  timeout -k 5 120 /tmp/ib$a
done
