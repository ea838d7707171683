"""Bandwidth of the two stand-alone likelihood kernels (rows a-7 / a-8) at the reference's block shape
(sims_per_gpu = 1024 rows x T+1 = 80 001 columns, float32 buffer), device-resident."""
import sys, ctypes
sys.path.insert(0, ".")
import torch, trpl_amd
from trpl_amd import _abi
lib = _abi.lib()
dev = torch.device("cuda", 0)
rows, cols = 1024, 80001
for dt, eb in ((torch.float32, 4), (torch.float64, 8)):
    x = torch.rand((rows, cols), dtype=dt, device=dev) + 1e-3
    vals = torch.randn(cols, dtype=torch.float64, device=dev)
    mag = torch.randn(rows, dtype=torch.float64, device=dev)
    P = torch.zeros(rows, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for name, fn, nbytes in (("log10_clamp", lambda: lib.trpl_log10_clamp_dev(x.data_ptr(), eb, rows, cols, cols, 1e-300, st), 2 * rows * cols * eb),
                             ("sse_accumulate", lambda: lib.trpl_sse_accumulate_dev(P.data_ptr(), x.data_ptr(), eb, rows, cols, cols, vals.data_ptr(), mag.data_ptr(), st), rows * cols * eb)):
        for _ in range(2): _abi.check(fn())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): _abi.check(fn())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name:16s} elem {eb} B: {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s  ({nbytes / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s)")

# trpl_loglik_from_pl_dev: one pass over resident PL rows (on-grid: every column; off-grid: 8801 observations, two columns each)
from trpl_amd import device as tdev
for dt, eb in ((torch.float32, 4), (torch.float64, 8)):
    x = torch.rand((rows, cols), dtype=dt, device=dev) + 1e-3
    obs = torch.randn(cols, dtype=torch.float64, device=dev)
    mag = torch.randn(rows, dtype=torch.float64, device=dev)
    P = torch.zeros(rows, dtype=torch.float64, device=dev)
    n_off = 8801
    hi = torch.sort(torch.randint(1, cols, (n_off,), device=dev, dtype=torch.int32))[0].contiguous()
    dx = torch.rand(n_off, dtype=torch.float64, device=dev) * 0.025
    h = torch.full((n_off,), 0.025, dtype=torch.float64, device=dev)
    for name, fn, nbytes in (("loglik_from_pl on-grid", lambda: tdev.loglik_from_pl_device(x, obs, mag, P=P), rows * cols * eb),
                             ("loglik_from_pl off-grid", lambda: tdev.loglik_from_pl_device(x, obs[:n_off].contiguous(), mag, P=P, obs_hi=hi, obs_dx=dx, obs_h=h), rows * 2 * n_off * eb)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name:24s} elem {eb} B: {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s  ({nbytes / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s)")
