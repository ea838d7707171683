"""Diagnostic: FAST vs STRICT status / iteration agreement under a small iteration cap (which kernel: argv[1] = single | pair, default the library's choice)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trpl_amd as trpl
S, T, Time = 2048, 30, 0.75
X = trpl.workloads.samples(S, seed=11)
ini, lengths = trpl.workloads.power_scan(128)
obs = [np.full(T + 1, 20.0)] * 3
fi, si = {}, {}
pf = trpl.loglik(X, ini, lengths, Time, 128, T, obs, info=fi, MAX=60, kernel=sys.argv[1] if len(sys.argv) > 1 else None)
ps = trpl.loglik(X, ini, lengths, Time, 128, T, obs, info=si, MAX=60, strict=True)
d = fi["status"] != si["status"]
print("kernel variant", trpl._abi.lib().trpl_kernel_variant(3 * S, 128, 0), "status mismatches", d.sum(), "of", d.size)
idx = np.argwhere(d)
for c, s in idx[:10]:
    print(" curve", c, "sample", s, "fast", fi["status"][c, s], fi["iters_total"][c, s], "strict", si["status"][c, s], si["iters_total"][c, s])
same = ~d
print("iteration-total mismatches among equal status:", (fi["iters_total"][same] != si["iters_total"][same]).sum())
