#!/usr/bin/env python3
"""Which of the default arithmetic's choices moves the state away from the reference evaluation on stiff grids?
Runs tools/forensics/fast_emul.c (CPU, exact divisions) with ONE choice swapped in at a time against the pinned oracle:
    python tools/forensics/run_emul.py [--S 48] [--T 8000] [--curve 4]      (Twothick curves: 0,2,4 = 311 nm; 1,3,5 = 2000 nm)"""
import argparse, ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def build():
    so = os.path.join(HERE, "libfastemul.so")
    src = os.path.join(HERE, "fast_emul.c")
    if not os.path.isfile(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "oracle", "trpl_oracle.c"))):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-ffp-contract=off", "-mfma", "-fopenmp", "-shared", "-w", "-o", so, src, "-lm"])
    return C.CDLL(so)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=48)
    ap.add_argument("--T", type=int, default=8000)
    ap.add_argument("--curve", type=int, default=4)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--configs", default=None, help="comma list of solver:assembly:field:quad[:history]")
    ap.add_argument("--L", type=int, default=128)
    ap.add_argument("--tol", type=int, default=7)
    ap.add_argument("--workload", default="twothick")
    a = ap.parse_args()
    import oracle, trpl_amd
    oracle.load()
    lib = build()
    w = trpl_amd.workloads
    L, T = a.L, a.T
    if a.workload == "twothick":
        ini, lens = w.twothick(L)
    else:
        ini = np.stack([w.beer_lambert(A, 2000.0, L) for A in w.POWER_SCAN_A_CM3]); lens = np.full(3, 2000.0)
    c = a.curve
    X = np.ascontiguousarray(w.samples(a.S)[:, :12])
    Time = T * 0.025
    ref = oracle.pvsim(X, lens[c], Time, L, T, ini[c], tol=a.tol, nthreads=a.threads)
    dx = lens[c] / L
    sig = np.maximum(X[:, 2], X[:, 3]) * 0.025 / dx ** 2
    print("curve %d, %g nm, %d samples, T = %d; stiffness D dt/dx^2: median %.0f max %.0f" % (c, lens[c], a.S, T, np.median(sig), sig.max()))
    names = {(0, 0, 0, 0): "reference pieces only (must be 0)", (1, 0, 0, 0): "S1 PCR on normalised rows + Cramer",
             (2, 0, 0, 0): "S2 CR x1 + PCR64 + Cramer (one-system kernel)", (3, 0, 0, 0): "S3 CR x2 + PCR32 + Cramer (paired kernel)",
             (4, 0, 0, 0): "S4 CR x2 + PCR32 + reference 2x2", (0, 1, 0, 0): "A  rewritten assembly", (0, 0, 1, 0): "F  folded field update",
             (0, 0, 0, 1): "Q  per-node excess quadrature", (3, 1, 1, 1): "all of FAST (paired)"}
    cfgs = list(names) if a.configs is None else [tuple(int(v) for v in s.split(":")) for s in a.configs.split(",")]
    ini_c = np.ascontiguousarray(ini[c])
    scale = X[:, 4] * L * X[:, 0] * X[:, 1] * dx
    above = ref["plI"] >= 1e-4 * scale[:, None]
    for cf in cfgs:
        lib.emul_set_history(C.c_int(cf[4] if len(cf) > 4 else 0))
        pl = np.empty((a.S, T + 1))
        it = np.zeros(a.S, dtype=np.int64)
        lib.emul_pvsim(X.ctypes.data_as(C.c_void_p), C.c_long(a.S), C.c_double(lens[c]), C.c_double(Time), C.c_int(L), C.c_long(T),
                       C.c_int(a.tol), C.c_int(10000), ini_c.ctypes.data_as(C.c_void_p), pl.ctypes.data_as(C.c_void_p),
                       it.ctypes.data_as(C.c_void_p), C.c_int(cf[0]), C.c_int(cf[1]), C.c_int(cf[2]), C.c_int(cf[3]), C.c_int(a.threads))
        dev = np.abs(pl / ref["plI"] - 1)
        dev[~above] = 0.0
        print("  %-52s max %.2e  p99 %.2e  median %.2e   final-step median %.2e   iteration totals differ on %d (max %d)" % (
            names.get(cf, str(cf)), dev.max(), np.quantile(dev, 0.99), np.median(dev), np.median(dev[:, -1]), int((it != ref["iters_total"]).sum()), int(np.abs(it - ref["iters_total"]).max())), flush=True)


if __name__ == "__main__":
    main()
