"""Differential fuzz of the two FAST L = 128 kernels over a parameter box much wider than the reference's:
   python tools/fuzz_pair.py run gpurun_out/fz0.npz single ; python tools/fuzz_pair.py run gpurun_out/fz1.npz pair   (or: strict)
   python tools/fuzz_pair.py cmp gpurun_out/fz0.npz gpurun_out/fz1.npz"""
import sys
sys.path.insert(0, ".")
import numpy as np

if sys.argv[1] == "run":
    import trpl_amd
    from trpl_amd import sampler as sm, workloads as wl
    S, T = 20000, 300
    lo = np.array([1e8, 1e12, 0.01, 0.01, 1e-13, 1e-3, 1e-3, 1e-32, 1e-32, 0.1, 0.1, 0.1, 0])
    hi = np.array([1e8, 1e18, 500, 500, 1e-8, 1e5, 1e5, 1e-26, 1e-26, 1e4, 1e4, 0.1, 0])
    lg = np.array([1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0])
    X = sm.random_grid(lo * sm.UNIT_CONVERSIONS, hi * sm.UNIT_CONVERSIONS, lg, S, rng=np.random.RandomState(123))
    ini, lens = wl.twothick(128)
    obs = [np.full(T + 1, 18.0) - 0.01 * np.arange(T + 1)] * len(lens)
    info = {}
    leg = sys.argv[3] if len(sys.argv) > 3 else None
    strict = leg == "strict"                                      # the reference's arithmetic as the third leg
    P = trpl_amd.loglik(X, ini, lens, T * 0.025, 128, T, obs, info=info, MAX=2000, strict=strict,
                        kernel=leg if leg in ("single", "pair") else None)
    v = trpl_amd._abi.lib().trpl_kernel_variant(S * len(lens), 128, T, (trpl_amd.FLAG_STRICT if strict else 0) | trpl_amd._abi.kernel_flag(leg if leg in ('single', 'pair') else None))
    np.savez(sys.argv[2], P=P, variant=v, **{k: info[k] for k in ("sse", "status", "iters_total")})
    print("variant", v, "non-converged", int((info["status"] != 0).sum()), "of", info["status"].size,
          "iterations", int(info["iters_total"].sum()), "seconds", info["seconds"])
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    print("variants", int(a["variant"]), int(b["variant"]))
    same_status = a["status"] == b["status"]
    print("status mismatches:", int((~same_status).sum()), "of", same_status.size)
    ok = (a["status"] == 0) & (b["status"] == 0)
    dit = np.abs(a["iters_total"][ok] - b["iters_total"][ok])
    print("converged in both:", int(ok.sum()), " iteration-total mismatches:", int((dit != 0).sum()), "max", int(dit.max()))
    rel = np.abs(a["sse"][ok] - b["sse"][ok]) / np.abs(a["sse"][ok])
    print("sse rel diff: median %.2e  99.9%% %.2e  max %.2e" % (np.median(rel), np.quantile(rel, 0.999), rel.max()))
    print("non-finite sse among converged:", int((~np.isfinite(a["sse"][ok])).sum()), int((~np.isfinite(b["sse"][ok])).sum()))
