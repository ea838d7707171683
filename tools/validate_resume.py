"""Checkpoint / continue at benchmark scale: every system of a curve of the default workload is run to T in one launch and
in two segments cut at t0 (five raw time levels checkpointed in device memory), and the PL matrices, iteration totals and
status words are compared bit for bit.  python tools/validate_resume.py [S] [T] [t0] [plT] [pair|single|strict]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, trpl_amd
from trpl_amd import device as tdev, workloads as wl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
t0 = int(sys.argv[3]) if len(sys.argv) > 3 else T // 2 + 37
plT = int(sys.argv[4]) if len(sys.argv) > 4 else 8
kern = sys.argv[5] if len(sys.argv) > 5 else "pair"
fl = {"pair": trpl_amd.FLAG_KERNEL_PAIR, "single": trpl_amd.FLAG_KERNEL_SINGLE, "strict": trpl_amd.FLAG_STRICT}[kern]
dev = torch.device("cuda", 0); L = 128; dt = 2.0 ** -5          # a power of two: both segments have the window's time step exactly
ini, lens = wl.power_scan(L)
X = torch.from_numpy(wl.samples(S)[:, :12].copy()).to(dev)
for c in range(len(lens)):
    ini_c = torch.from_numpy(np.ascontiguousarray(ini[c])).to(dev)
    ncol = T // plT + 1
    full = torch.empty((S, ncol), dtype=torch.float64, device=dev)
    itf = torch.zeros(S, dtype=torch.int64, device=dev); stf = torch.zeros(S, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(); a = time.perf_counter()
    tdev.solve_pl_snap_device(X, lens[c], T * dt, L, T, ini_c, full, [], status=stf, iters_total=itf, plT=plT, flags=fl)
    torch.cuda.synchronize(); t_full = time.perf_counter() - a
    cN = torch.zeros((S, 5, L), dtype=torch.float64, device=dev); cP = torch.zeros_like(cN)
    cE = torch.zeros((S, 5, L + 1), dtype=torch.float64, device=dev)
    first = torch.empty((S, t0 // plT + 1), dtype=torch.float64, device=dev)
    ita = torch.zeros_like(itf); itb = torch.zeros_like(itf); itc = torch.zeros_like(itf)
    sta = torch.zeros_like(stf); stb = torch.zeros_like(stf)
    a = time.perf_counter()
    tdev.solve_pl_snap_device(X, lens[c], t0 * dt, L, t0, ini_c, first, trpl_amd.checkpoint_steps(t0), cN, cP, cE, status=sta,
                              iters_total=ita, plT=plT, flags=fl | trpl_amd.FLAG_SNAP_RAW)
    out = torch.full((S, ncol), float("nan"), dtype=torch.float64, device=dev)
    out[:, :first.shape[1]] = first
    tdev.solve_pl_resume_device(X, lens[c], T * dt, L, T, t0, cN, cP, cE, out, status=stb, iters_total=itb, plT=plT, flags=fl)
    torch.cuda.synchronize(); t_split = time.perf_counter() - a
    tail = torch.empty_like(first)
    tdev.solve_pl_resume_device(X, lens[c], t0 * dt, L, t0, t0, cN, cP, cE, tail, iters_total=itc, plT=plT, flags=fl)
    torch.cuda.synchronize()
    ok = (stf == 0) & (sta == 0)
    same_pl = torch.equal(out[ok].view(torch.int64), full[ok].view(torch.int64))
    same_it = torch.equal((ita + itb - itc)[ok], itf[ok])
    same_st = torch.equal(stb[ok], stf[ok])
    print(f"curve {c} ({kern}, S={S}, T={T}, cut at {t0}, plT={plT}): PL bit-identical {same_pl}, iteration totals equal {same_it}, "
          f"status equal {same_st}; systems compared {int(ok.sum())} (non-converged before the cut: {int((~ok).sum())}); "
          f"one launch {t_full:.2f} s, two segments {t_split:.2f} s; checkpoint {3 * cN.numel() * 8 / 2**30:.2f} GiB", flush=True)
