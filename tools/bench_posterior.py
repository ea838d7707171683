"""Bandwidth of the posterior-core kernels (device-resident): S samples, D parameter columns."""
import sys
sys.path.insert(0, ".")
import torch
import trpl_amd
from trpl_amd import device as tdev

dev = torch.device("cuda", 0)
S, D = (int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 1024 * 1024), 13
g = torch.Generator(device=dev); g.manual_seed(1)
LL = -1e5 * torch.rand(S, dtype=torch.float64, device=dev, generator=g) ** 2
V = torch.randn((D, S), dtype=torch.float64, device=dev, generator=g)
W = torch.empty_like(LL)
ws = tdev.posterior_workspace(D)
sums = torch.zeros(2 + D, dtype=torch.float64, device=dev)
central = torch.zeros((D, D + 2), dtype=torch.float64, device=dev)
h1 = torch.zeros(64, dtype=torch.float64, device=dev)
h2 = torch.zeros((64, 64), dtype=torch.float64, device=dev)


def timed(name, fn, nbytes, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:34s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s algorithmic ({nbytes / ms / 1e6 / 80:.1f} % of 8 TB/s)")


print(f"S = {S}, D = {D}")
timed("weights (2 reads + write + rescale)", lambda: tdev.posterior_weights_device(LL, 4e3, W, ws), 5 * 8 * S)
timed("moments (means + 13x13 covariance)", lambda: tdev.posterior_moments_device(V, W, sums, central, ws), (D + 1) * 8 * S * 2)         # two passes (means, then centred sums), every column read once in each
timed("hist 1-D, 64 bins", lambda: tdev.posterior_hist_device(V[0], W, -4, 4, h1), 2 * 8 * S)
timed("hist 2-D, 64 x 64 bins", lambda: tdev.posterior_hist_device(V[0], W, -4, 4, h2, y=V[1], ylo=-4, yhi=4), 3 * 8 * S)
