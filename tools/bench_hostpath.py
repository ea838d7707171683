"""The reference-API (host-buffer) path at the reference's production block shape (parallel_bayes_gpu.py:72-81,
:104: sims_per_gpu = 1024 samples, 3 curves, T = 80 000, float32 PL buffer = 328 MB per curve): the three
callables one after the other as bayeslib.simulate issues them -- pvSim -> fastlog -> prob -- per curve, and the
fused single call, with the PL block's PCIe traffic priced.

    python tools/bench_hostpath.py [T] [out.json]

Each host-memory mode runs in its own child process (the thresholds are read once per process):
  pageable   TRPL_HOST_DIRECT_MIN=-1 TRPL_HOST_PIN_MIN=-1: device PL buffer + copies from/to pageable memory
             (round 1's path)
  pinned     TRPL_HOST_DIRECT_MIN=-1: the caller's buffer is page-locked for the duration of each call, copies
             are asynchronous DMA
  direct     defaults: pvSim's kernel writes PL straight into the caller's (mapped) buffer; fastlog / prob pin
  prepinned  defaults, but the PL buffer is allocated page-locked once (what driver.simulate does for its own
             buffer): no per-call registration at all
"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(T, mode):
    import numpy as np
    import trpl_amd as tp
    from trpl_amd import workloads as wl
    S, L, Time = 1024, 128, T * 0.025
    ini, lens = wl.power_scan(L)
    X = wl.samples(S)
    n_obs = int(0.07 * T) + 1
    if mode == "prepinned":
        import torch
        keep = torch.empty((S, T + 1), dtype=torch.float32, pin_memory=True)
        pl = keep.numpy()
    else:
        pl = np.empty((S, T + 1), dtype=np.float32)
    pl[:] = 1.0                                               # touch every page before anything is timed
    vals = np.full(n_obs, -3.0)
    mag = np.ascontiguousarray(X[:, -1])
    par = [2000.0, Time, L, T, 1, (0,), 7, 10000]
    rec = {"mode": mode, "T": T, "S": S, "pl_bytes": int(pl.nbytes)}
    tp.pvSim(pl[:8], None, None, None, X[:8, :-1], par, ini[0], init_mode="points")      # warm the context
    walls = {"pvSim": [], "fastlog": [], "prob": []}
    kern = []
    P = np.zeros(S)
    for c in range(3):
        t0 = time.perf_counter()
        sec = tp.pvSim(pl, None, None, None, X[:, :-1], par, ini[c], init_mode="points")
        t1 = time.perf_counter()
        tp.fastlog(pl, sys.float_info.min)
        t2 = time.perf_counter()
        tp.prob(P, pl[:, :n_obs], vals, None, mag)
        t3 = time.perf_counter()
        walls["pvSim"].append(t1 - t0); walls["fastlog"].append(t2 - t1); walls["prob"].append(t3 - t2)
        kern.append(sec)
    rec["pvSim_wall_s"] = min(walls["pvSim"]); rec["pvSim_kernel_s"] = min(kern)
    rec["pvSim_outside_kernel_s"] = min(w - k for w, k in zip(walls["pvSim"], kern))
    rec["fastlog_wall_s"] = min(walls["fastlog"]); rec["prob_wall_s"] = min(walls["prob"])
    # PCIe rate of the PL block: pvSim moves it once (D2H), fastlog twice (H2D + D2H), prob moves the observed prefix
    rec["pvSim_pl_GBps_outside_kernel"] = pl.nbytes / max(rec["pvSim_outside_kernel_s"], 1e-9) / 1e9
    rec["fastlog_pl_GBps"] = 2 * pl.nbytes / rec["fastlog_wall_s"] / 1e9
    rec["prob_pl_GBps"] = S * n_obs * 4 / rec["prob_wall_s"] / 1e9
    per_curve = rec["pvSim_wall_s"] + rec["fastlog_wall_s"] + rec["prob_wall_s"]
    rec["per_curve_wall_s"] = per_curve
    rec["system_timesteps_per_s_unfused"] = S * (T + 1) / per_curve
    # the fused call on the same block (all three curves, stops at the last observation)
    obs = [vals] * 3
    info = {}
    t0 = time.perf_counter()
    tp.loglik(X, ini, lens, Time, L, T, obs, pl_f32=True, info=info)
    rec["fused_wall_s_3_curves"] = time.perf_counter() - t0
    rec["checksum"] = float(P.sum())
    print(json.dumps(rec), flush=True)


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
    out = sys.argv[2] if len(sys.argv) > 2 else None
    modes = {"pageable": {"TRPL_HOST_DIRECT_MIN": "-1", "TRPL_HOST_PIN_MIN": "-1"},
             "pinned": {"TRPL_HOST_DIRECT_MIN": "-1"}, "direct": {}, "prepinned": {}}
    recs = []
    for mode, env in modes.items():
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(T), mode], env=e,
                           capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print(mode, "FAILED", r.stderr[-1500:], flush=True)
            continue
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        recs.append(rec)
        print("%-9s pvSim %.3f s (kernel %.3f, outside %.3f = %.1f GB/s)  fastlog %.3f s (%.1f GB/s)  prob %.3f s  "
              "per curve %.3f s = %.3e system-timesteps/s   fused(3 curves) %.3f s"
              % (mode, rec["pvSim_wall_s"], rec["pvSim_kernel_s"], rec["pvSim_outside_kernel_s"],
                 rec["pvSim_pl_GBps_outside_kernel"], rec["fastlog_wall_s"], rec["fastlog_pl_GBps"], rec["prob_wall_s"],
                 rec["per_curve_wall_s"], rec["system_timesteps_per_s_unfused"], rec["fused_wall_s_3_curves"]), flush=True)
    if recs:
        assert len({round(r["checksum"], 6) for r in recs}) == 1, "modes disagree"
    if out:
        json.dump(recs, open(out, "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3])
    else:
        main()
