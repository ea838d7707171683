#!/usr/bin/env python3
"""Register / LDS / occupancy table of the stepper kernels from hipcc's kernel-resource-usage remarks.
    python tools/kernel_resources.py [pair fast strict mixed f32]   (cross-compiles, no GPU needed)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bayesian-inference-trpl_amd", "csrc")
CONTRACT = {"strict": "off"}


def main():
    for n in sys.argv[1:] or ["pair", "fast", "strict", "hist32"]:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950",
                            "-ffp-contract=" + CONTRACT.get(n, "on"), "-c", os.path.join(CSRC, "stepper_%s.hip" % n),
                            "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
            name = b.split("\n")[0].split(" ")[0]
            g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print("%-7s %-78s VGPR %3s AGPR %3s spill %3s scratch %4s occ %s LDS %6s" % (
                n, dem[:78], g("VGPRs"), g("AGPRs"), g("VGPR Spill"), g(r"ScratchSize \[bytes/lane\]"),
                g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main()
