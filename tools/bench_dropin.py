"""Host-buffer drop-in loop (pvSim -> fastlog -> [interp] -> prob per curve, the reference's control flow and dtypes)
vs the fused call, on one reference-shaped block: sims_per_gpu = 1024 samples, 3 curves, observations on a prefix of
the grid (the shipped example files' shape).  python tools/bench_dropin.py [T]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, trpl_amd as tp
from trpl_amd import workloads as wl
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
S, L, Time = 1024, 128, T * 0.025
ini, lens = wl.power_scan(L)
X = wl.samples(S)
n_obs = int(0.07 * T) + 1                                     # Balancedhighsurf curve 0: 5601 of 80001 points
sim_t = np.linspace(0, Time, T + 1)
mark = (wl.MARKED_POINT * tp.UNIT_CONVERSIONS)[None, :-1]
obs = [np.log10(tp.solve_pl(mark, lens[c], Time, L, T, ini[c], strict=True)[0][0][:n_obs]) for c in range(3)]
e_data = [([sim_t[:n_obs]] * 3, obs, [None] * 3)]
flags = {"load_PL_from_file": False, "log_pl": True, "self_normalize": False}
for name, fused, overlap in (("unfused, one curve at a time", False, False), ("unfused, curves overlapped ", False, True),
                             ("fused                      ", True, True)):
    P = np.zeros((1, S)); st, et, mt = np.zeros(1), np.zeros(1), np.zeros(1)
    t0 = time.perf_counter()
    tp.simulate(tp.pvSim, e_data, P, X, [None], [None], 3, [2000.0, Time, L, T, 1, (0,), 7, 10000], ini, flags,
                {"sims_per_gpu": 1024, "num_gpus": 1, "fused": fused, "overlap_curves": overlap}, 0, st, et, mt)
    dt = time.perf_counter() - t0
    print(f"{name} T={T}: wall {dt:.3f} s  (solver {st[0]:.3f} s, fastlog+interp {mt[0]:.3f} s, prob {et[0]:.3f} s)"
          f"  -> {S * 3 * (T + 1) / dt:.3e} system-timesteps/s incl. PCIe and host work;  P[0]={P[0,0]:.6f}")
