"""Does the relative placement of the five operand arrays matter for the batched solve's HBM rate?  Same kernel,
same data volume, operands rotated over 4 sets; the arrays of a set are views into one buffer at offsets
k * (array bytes + skew).    python tools/pcr_layout_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from trpl_amd import device as tdev

dev = torch.device("cuda", 0)
S, L, nsets, reps = 65536, 128, 4, 96
n = S * L
for skew_bytes in (0, 256, 4096, 4096 + 256, 65536 + 4096, 1 << 20, (1 << 20) + 4096 + 256):
    skew = skew_bytes // 8
    sets = []
    for k in range(nsets):
        buf = torch.empty(5 * (n + skew) + 64, dtype=torch.float64, device=dev)
        v = [buf[i * (n + skew): i * (n + skew) + n].view(S, L) for i in range(5)]
        v[0].uniform_(-1, 1); v[2].uniform_(-1, 1); v[1].uniform_(2.5, 4.0); v[3].normal_()
        v[0][:, 0] = 0; v[2][:, -1] = 0
        sets.append(v)
    for i in range(2 * nsets):
        ld, d, ud, b, x = sets[i % nsets]
        tdev.pcr_solve_device(ld, d, ud, b, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        ld, d, ud, b, x = sets[i % nsets]
        tdev.pcr_solve_device(ld, d, ud, b, x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("skew %8d B: %.2f us  %.0f GB/s" % (skew_bytes, ms * 1e3, 5 * n * 8 / ms / 1e6), flush=True)
    del sets
