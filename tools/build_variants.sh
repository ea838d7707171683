#!/bin/bash
# Variant builds of the library for same-box A/B runs: each "name=defines" pair is compiled from a scratch copy of the
# package (so the in-tree objects and libtrpl_hip.so stay the shipped build) into tools/ab/<name>.so, which travels to
# the GPU box with the snapshot and is selected there with TRPL_LIBRARY (tools/ab_multi.sh, tools/thinfilm_gap.py).
#   bash tools/build_variants.sh base= solve2="-DTRPL_RCP_SOLVE_STEPS=2" rows2="-DTRPL_RCP_ROWS_STEPS=2"
# Run in the development container (hipcc cross-compiles gfx950), never on the GPU box.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
PKG="$R/bayesian-inference-trpl_amd"
mkdir -p "$R/tools/ab"
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  W=$(mktemp -d /tmp/trpl_variant_XXXX)
  mkdir -p "$W/pkg" "$W/include"
  cp -r "$PKG/csrc" "$PKG/Makefile" "$W/pkg/"; rm -rf "$W/pkg/csrc/build"
  cp "$R/include/trpl.h" "$W/include/"
  ( cd "$W/pkg" && make -s -j8 libtrpl_hip.so EXTRA_DEFS="$defs" > "$W/build.log" 2>&1 ) || { tail -20 "$W/build.log"; exit 1; }
  cp "$W/pkg/libtrpl_hip.so" "$R/tools/ab/$name.so"
  rm -rf "$W"
  echo "built tools/ab/$name.so   [$defs]"
done
