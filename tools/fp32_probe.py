"""Development probe: accuracy / convergence of the fp32 stepper vs the fp64 oracle at several tolerances."""
import sys, numpy as np
sys.path.insert(0, ".")
import trpl_amd as gpu, oracle
w = gpu.workloads
X = w.samples(6)
for L in (128, 512):
    T, Time, length = 60, 60 * 0.025, 2000.0
    ini = np.stack([w.beer_lambert(A, length, L) for A in w.POWER_SCAN_A_CM3])
    for c in (0, 2):
        ref = oracle.pvsim(X[:, :-1], length, Time, L, T, ini[c], nthreads=6)
        for tol in (3, 4, 5):
            pl, st, it, _ = gpu.solve_pl(X[:, :-1], length, Time, L, T, ini[c], tol=tol, fp32="long", MAX=2000)     # "long": also valid should T be raised beyond TRPL_FP32_MAX_STEPS
            ok = st == 0
            err = np.max(np.abs(pl[ok] / ref["plI"][ok] - 1)) if ok.any() else float("nan")
            print(f"L={L} curve={c} tol={tol}: status={st.tolist()} max rel PL err (converged) {err:.2e} iters {it.tolist()} (fp64 tol7: {ref['iters_total'].tolist()})")
