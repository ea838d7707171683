#!/bin/bash
# ISA fingerprint of the device code of every object of the library: sha256 of the gfx950 disassembly (llvm-objdump -d) of
# each csrc/build/*.o.  Two builds whose fingerprints agree run the same machine code (the fat binary itself carries a
# per-compilation id and cannot be compared directly).  Usage: tools/isa_hash.sh [object-dir] > file; diff two files.
set -e
DIR=${1:-bayesian-inference-trpl_amd/csrc/build}
BIN=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
for o in "$DIR"/*.o; do
    n=$(basename "$o" .o)
    objcopy --dump-section .hip_fatbin="$TMP/$n.fatbin" "$o" 2>/dev/null || true
    [ -s "$TMP/$n.fatbin" ] || { echo "$n host-only"; continue; }
    $BIN/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$TMP/$n.fatbin" --output="$TMP/$n.elf" --unbundle
    $BIN/llvm-objdump -d "$TMP/$n.elf" | grep -v "file format" | sha256sum | cut -c1-32 | sed "s/^/$n /"
done
rm -rf "$TMP"
