"""Offline validation at benchmark scale: FAST vs STRICT (bit-identical-to-reference arithmetic) likelihoods and
iteration counts over the whole sampled parameter box.  python tools/validate_fast_vs_strict.py [S] [T] [power_scan|twothick]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, trpl_amd
from trpl_amd import device as tdev, workloads as wl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda", 0); L = 128; Time = T * 0.025
ini, lens = (wl.twothick(L) if len(sys.argv) > 3 and sys.argv[3] == "twothick" else wl.power_scan(L)); C = len(lens)
X = torch.from_numpy(wl.samples(S)).to(dev); ini_d = torch.from_numpy(ini).to(dev)
mark = torch.from_numpy((wl.MARKED_POINT * trpl_amd.UNIT_CONVERSIONS)[None, :-1].copy()).to(dev)
obs = torch.empty((C, T + 1), dtype=torch.float64, device=dev)
for c in range(C):
    pl = torch.empty((1, T + 1), dtype=torch.float64, device=dev)
    tdev.solve_pl_device(mark, lens[c], Time, L, T, ini_d[c].contiguous(), pl, flags=trpl_amd.FLAG_STRICT)
    obs[c] = torch.log10(pl[0])
res = {}
for name, flags in (("fast", 0), ("strict", trpl_amd.FLAG_STRICT)):
    P = torch.zeros(S, dtype=torch.float64, device=dev); sse = torch.empty((C, S), dtype=torch.float64, device=dev)
    st = torch.empty((C, S), dtype=torch.int32, device=dev); it = torch.empty((C, S), dtype=torch.int64, device=dev)
    fc = torch.empty((C, S), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tdev.loglik_device(X, ini_d, lens, Time, L, T, obs, [T + 1] * C, P, sse, st, it, flags=flags, floor_col=fc)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[name] = (P.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), dt, fc.cpu().numpy(), sse.cpu().numpy())
    print(f"{name}: {dt:.2f} s, non-converged {int((st != 0).sum())}, iterations {int(it.sum())}")
Pf, sf, itf, _, fcf, ssef = res["fast"]; Ps, ss, its, _, fcs, sses = res["strict"]
ok = ~(sf.any(0) | ss.any(0))
rel = np.abs(Pf[ok] - Ps[ok]) / np.abs(Ps[ok])
print(f"S={S} T={T}: max |P_fast-P_strict|/|P_strict| = {rel.max():.3e} (median {np.median(rel):.1e}); "
      f"systems with different iteration totals: {int((itf != its).sum())} of {itf.size} "
      f"(max |diff| {int(np.abs(itf - its).max())}); total iterations ratio {itf.sum() / its.sum():.6f}")
bad = rel > 1e-6
print(f"samples with rel diff > 1e-6: {int(bad.sum())} of {int(ok.sum())}")
if bad.any():
    idx = np.where(ok)[0][bad]
    order = np.argsort(-rel[bad])[:8]
    Xh = X.cpu().numpy() / trpl_amd.UNIT_CONVERSIONS
    for k in order:
        i = idx[k]
        print(f"  sample {i}: P_fast {Pf[i]:.6e} P_strict {Ps[i]:.6e}  tau_n {Xh[i,9]:.1f} tau_p {Xh[i,10]:.1f} Sf {Xh[i,5]:.2f} Sb {Xh[i,6]:.2f} mu_n {Xh[i,2]:.1f} p0 {Xh[i,1]:.2e}")
    print(f"  |P| of affected samples: min {np.abs(Ps[idx]).min():.3e}, median {np.median(np.abs(Ps[idx])):.3e};  |P| of unaffected: median {np.median(np.abs(Ps[ok][~bad])):.3e}")

# ---- the cancellation floor as the library reports it (include/trpl.h: floor_col)
print(f"floor_col: identical in both arithmetics for {int((fcf == fcs).sum())} of {fcf.size} systems; "
      f"systems that reach the floor {int((fcs >= 0).sum())} ({100 * (fcs >= 0).mean():.2f} %), earliest column {int(fcs[fcs >= 0].min()) if (fcs >= 0).any() else -1}")
clear = (fcs == -1).all(0) & (fcf == -1).all(0) & ok
relc = np.abs(Pf[clear] - Ps[clear]) / np.abs(Ps[clear])
print(f"samples that never reach the floor: {int(clear.sum())} of {S} ({100 * clear.mean():.2f} %): max |P_fast-P_strict|/|P_strict| = {relc.max():.3e}, "
      f"99.9th percentile {np.quantile(relc, 0.999):.2e}, above 1e-8: {int((relc > 1e-8).sum())}")
hit = ~clear & ok
relh = np.abs(Pf[hit] - Ps[hit]) / np.abs(Ps[hit])
print(f"samples on the floor: {int(hit.sum())}: median gap {np.median(relh):.2e}, above 1e-6: {int((relh > 1e-6).sum())}, max {relh.max():.2e}")
print(f"every sample with a gap above 1e-6 is flagged by floor_col: {bool(((rel > 1e-6) <= (~clear[ok])).all())}")
