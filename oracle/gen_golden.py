#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE'S OWN code in this container.

TEST INFRASTRUCTURE.  Development-container only: it needs /root/reference (read-only)
and refuses to run without it.  Nothing from the reference is copied: the script imports
the reference modules where they lie, with oracle/refshim (a sequential stand-in for the
absent `numba`) first on sys.path, calls the reference functions, and stores *data only*
(inputs and the outputs they produced).

Reference entry points exercised (file:line):
  pvSimPCR.pcreduce :42-81, pvSimPCR.norm2 :14-40          -> pcr_norm.npz
  pvSimPCR.pvSim :309-401 (tEvol :227-306, iterate :93-225) -> pvsim_*.npz (pvsim_bundle: max_sims_per_block 2 and 3)
  probs.fastlog :78-85, probs.prob :49-62                   -> probs.npz
  bayeslib.random_grid :18-32, bayeslib.bayes :207-252      -> bayes_e2e.npz, sampler.npz
  pvSim_fallback.pvSim_cpu_fallback :80-117 (as shipped)    -> fallback.npz
  bayeslib.bayes(pvSim_cpu_fallback) CPU branch, 64 samples x 3 curves, 8 array tasks -> fallback64.npz (configs[0])
  bayes_io.get_initpoints :106-119, get_data :15-104 + bayes -> bayes_realdata.npz
  Legacy/pvSim.pvSim :129-173; Testing/PV_tester2.dydt :13-49 + odeint -> legacy_odeint.npz
  Legacy/pvSim.pvSim :129-173 over whole curves (2400 steps, 4 films x 9 samples)  -> legacy_full.npz
  Testing/PV_tester2.dydt :13-49 + odeint (:91-93) at rtol 1e-10, 800 steps, 4 films x 9 samples -> tester_refine.npz
  Visualization/utils.py normalize :157-166, w_* :185-226, covariance :222-227, credible_interval :185-196,
  marginalize_1D :239-262, marginalize_2D :264-285 (tempering: marginalization_visual.py:589-591) -> posterior.npz

Usage:  python oracle/gen_golden.py [case ...]     (default: all cases)
"""
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("TRPL_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
if not os.path.isfile(os.path.join(REF, "pvSimPCR.py")):
    sys.exit("gen_golden.py: reference checkout not found at %s -- refusing to run" % REF)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "refshim"))
os.environ.setdefault("SLURM_ARRAY_TASK_ID", "0")          # bayeslib.py:231

import numpy as np  # noqa: E402

import bayeslib  # noqa: E402  (reference)
import probs  # noqa: E402  (reference)
import pvSimPCR  # noqa: E402  (reference)
from bayes_io import get_initpoints  # noqa: E402  (reference)

EXC_POWER = os.path.join(REF, "Example Data", "Power_scan_Excitations.csv")
EXC_TWO = os.path.join(REF, "Example Data", "Twothick_Excitations.csv")

# parallel_bayes_gpu.py:27-33 (values restated; they are physical constants of the config)
UNIT = np.array([(1e7) ** -3, (1e7) ** -3, (1e7) ** 2 / (1e9) * .02569257,
                 (1e7) ** 2 / (1e9) * .02569257, (1e7) ** 3 / (1e9), (1e7) / (1e9), (1e7) / (1e9),
                 (1e7) ** 6 / (1e9), (1e7) ** 6 / (1e9), 1, 1, 704.3, 1])
DO_LOG = np.array([1, 1, 0, 0, 1, 1, 1, 1, 1, 0, 0, 1, 0])              # :86
MINX = np.array([1e8, 1e14, 0, 0, 1e-11, 0.1, 0.1, 1e-30, 1e-30, 1, 1, 10 ** -1, 0])   # :91
MAXX = np.array([1e8, 1e16, 50, 50, 1e-9, 100, 100, 1e-28, 1e-28, 1000, 2000, 10 ** -1, 0])  # :92
# marked point of Visualization/config.txt:57-68
MARK = np.array([1e8, 3e15, 20, 20, 4.8e-11, 2, 2, 4.4e-29, 4.4e-29, 511, 871, 0.1, 0])
PT = (0, 1, 3, 10, 30, 100)


def draw(S, seed=42):
    np.random.seed(seed)                                                 # parallel_bayes_gpu.py:35
    return bayeslib.random_grid(MINX * UNIT, MAXX * UNIT, DO_LOG, S)


class IterRecorder:
    """Wrap pvSimPCR.iterate (module global looked up at call time) to log its returns."""

    def __enter__(self):
        self.log = []
        self.orig = pvSimPCR.iterate

        def rec(N, P, E, matPar, par, p, t):
            r = self.orig(N, P, E, matPar, par, p, t)
            self.log.append((int(p), int(t), int(r)))
            return r
        pvSimPCR.iterate = rec
        return self

    def __exit__(self, *a):
        pvSimPCR.iterate = self.orig


def run_pvsim(mat12, length, time_ns, L, T, ini, dtype, tol=7, MAX=10000, plT=1, mspb=1):
    """mspb > 1: max_sims_per_block consecutive samples share one convergence test (pvSimPCR.py:213-216,:258-266);
    the iteration count of a bundle is recorded for each of its samples."""
    S = len(mat12)
    simPar = [length, time_ns, L, T, plT, PT, tol, MAX]
    plI = np.empty((S, T // plT + 1), dtype=dtype)
    dummyN = np.empty((S, 2, L)); dummyE = np.empty((S, 2, L + 1))
    with IterRecorder() as rec:
        pvSimPCR.pvSim(plI, dummyN, dummyN.copy(), dummyE, mat12, simPar, ini, (1,), max((S + mspb - 1) // mspb, 1), mspb,
                       init_mode="points")
    it = np.zeros((S, T + 1), dtype=np.int32)
    for p, t, r in rec.log:
        it[p:p + mspb, t] = r
    return plI, it


def case_pcr_norm():
    rng = np.random.default_rng(1234)
    out = {}
    for N in (4, 8, 32, 128, 512):
        nsys = 6
        ld = rng.uniform(-1, 1, (nsys, N)); ud = rng.uniform(-1, 1, (nsys, N))
        d = rng.uniform(2.5, 4, (nsys, N)) * rng.choice([-1, 1], (nsys, 1))
        ld[:, 0] = 0; ud[:, -1] = 0
        B = rng.normal(size=(nsys, N)); cprev = rng.normal(size=(nsys, N))
        x = np.zeros((nsys, N)); err = np.zeros(nsys)
        for s in range(nsys):
            a2 = ld[s].copy()[:, None]; a1 = d[s].copy()[:, None]; a0 = ud[s].copy()[:, None]
            b = B[s].copy()[:, None]; c = cprev[s].copy()[:, None]
            buf = np.zeros((4 * N, 1)); e = np.zeros(1)
            pvSimPCR.norm2(a0, a1, a2, b, c, buf, e, 1, 1)               # pvSimPCR.py:172 call form
            err[s] = e[0]
            pvSimPCR.pcreduce(a2, a1, a0, b, c, buf, 1, 1)               # pvSimPCR.py:175 call form
            x[s] = c[:, 0]
        out.update({f"ld{N}": ld, f"d{N}": d, f"ud{N}": ud, f"B{N}": B, f"c{N}": cprev,
                    f"x{N}": x, f"err{N}": err})
    np.savez_compressed(os.path.join(OUT, "pcr_norm.npz"), **out)


def case_pvsim_power():
    """Power_scan: 3 excitations, Length 2000 nm, L=128, dt=0.025 ns."""
    ini = get_initpoints(EXC_POWER, {"select_obs_sets": None})
    X = np.vstack([draw(4), MARK * UNIT])
    T = 160
    pls, its = [], []
    for c in range(3):
        p, i = run_pvsim(X[:, :-1], 2000, T * 0.025, 128, T, ini[c], np.float64)
        pls.append(p); its.append(i)
    p32, _ = run_pvsim(X[:2, :-1], 2000, 40 * 0.025, 128, 40, ini[2], np.float32)
    np.savez_compressed(os.path.join(OUT, "pvsim_power.npz"), X=X, ini=ini, length=2000.0,
                        time=T * 0.025, L=128, T=T, tol=7, MAX=10000, plI=np.array(pls),
                        iters=np.array(its), plI32=p32, T32=40)


def case_pvsim_twothick():
    """Twothick rows 0 (311 nm) and 1 (2000 nm); 311 nm / high power stresses the iteration."""
    ini = get_initpoints(EXC_TWO, {"select_obs_sets": None})
    X = np.vstack([draw(2), MARK * UNIT])
    T = 60
    lengths = [311, 2000, 311, 2000, 311, 2000]                         # parallel_bayes_gpu.py:71
    sel = [0, 1, 4]
    pls, its = [], []
    for c in sel:
        p, i = run_pvsim(X[:, :-1], lengths[c], T * 0.025, 128, T, ini[c], np.float64)
        pls.append(p); its.append(i)
    np.savez_compressed(os.path.join(OUT, "pvsim_twothick.npz"), X=X, ini=ini[sel],
                        lengths=np.array([lengths[c] for c in sel], dtype=float), time=T * 0.025,
                        L=128, T=T, tol=7, MAX=10000, plI=np.array(pls), iters=np.array(its))


def case_pvsim_small():
    """Other grid sizes / tolerances / plT, and a forced non-convergence (MAX small)."""
    out = {}
    X = np.vstack([draw(2), MARK * UNIT])
    for L in (8, 32, 64):
        x = (np.arange(L) + 0.5) * (500.0 / L)
        ini = 1e17 * 1e-21 * np.exp(-6e-3 * x)
        p, i = run_pvsim(X[:, :-1], 500, 30 * 0.05, L, 30, ini, np.float64, tol=6)
        out[f"plI_L{L}"] = p; out[f"it_L{L}"] = i; out[f"ini_L{L}"] = ini
    x = (np.arange(32) + 0.5) * (500.0 / 32)
    ini = 1e17 * 1e-21 * np.exp(-6e-3 * x)
    p, i = run_pvsim(X[:, :-1], 500, 40 * 0.05, 32, 40, ini, np.float64, tol=6, plT=4)
    out["plI_plT4"] = p; out["it_plT4"] = i
    # forced non-convergence: one sample, MAX = 3, high injection on a thin film
    x = (np.arange(32) + 0.5) * (311.0 / 32)
    ini_hi = 1.6e18 * 1e-21 * np.exp(-6e-3 * x)
    pl = np.full((1, 11), -777.0)
    simPar = [311, 10 * 0.025, 32, 10, 1, PT, 7, 3]
    d = np.empty((1, 2, 32))
    with IterRecorder() as rec:
        pvSimPCR.pvSim(pl, d, d.copy(), np.empty((1, 2, 33)), X[2:3, :-1], simPar, ini_hi, (1,), 1, 1,
                       init_mode="points")
    out["nc_plI"] = pl; out["nc_log"] = np.array(rec.log); out["nc_ini"] = ini_hi
    np.savez_compressed(os.path.join(OUT, "pvsim_small.npz"), X=X, **out)


def case_pvsim_bundle():
    """max_sims_per_block = 3 and 2 (bayes_validate.connect_to_gpu's default is 3): 7 samples = bundles of 3, 3, 1 /
    2, 2, 2, 1; the high-power Power_scan curve and the stiff 311 nm Twothick curve."""
    iniP = get_initpoints(EXC_POWER, {"select_obs_sets": None})
    iniT = get_initpoints(EXC_TWO, {"select_obs_sets": None})
    X = np.vstack([draw(6, seed=7), MARK * UNIT])
    T = 80
    out = {"X": X, "iniP": iniP[2], "iniT": iniT[0], "T": T, "time": T * 0.025, "L": 128, "lengthP": 2000.0, "lengthT": 311.0}
    for m in (3, 2):
        p, i = run_pvsim(X[:, :-1], 2000, T * 0.025, 128, T, iniP[2], np.float64, mspb=m)
        out["plP%d" % m] = p; out["itP%d" % m] = i
        p, i = run_pvsim(X[:, :-1], 311, T * 0.025, 128, T, iniT[0], np.float64, mspb=m)
        out["plT%d" % m] = p; out["itT%d" % m] = i
    np.savez_compressed(os.path.join(OUT, "pvsim_bundle.npz"), **out)


def case_probs():
    rng = np.random.default_rng(7)
    rows, cols = 5, 37
    pl64 = rng.lognormal(-8, 3, (rows, cols)); pl64[1, 3] = 0.0; pl64[2, 5] = -1e-9
    pl32 = pl64.astype(np.float32); pl32[1, 3] = 1e-30; pl32[2, 5] = 1e-38      # keep > 0 (see note)
    MIN = sys.float_info.min                                                    # bayeslib.py:157
    l64 = pl64.copy(); probs.fastlog(l64, MIN, 1, 3)
    l32 = pl32.copy(); probs.fastlog(l32, MIN, 1, 3)
    values = rng.uniform(-12, -4, cols); mag = rng.uniform(-1, 1, rows)
    P64 = rng.normal(size=rows); P64_in = P64.copy()
    probs.prob(P64, l64, values, np.ones(cols), mag, 1, 3)
    P32 = np.zeros(rows)
    probs.prob(P32, l32, values, np.ones(cols), mag, 1, 3)
    np.savez_compressed(os.path.join(OUT, "probs.npz"), pl64=pl64, pl32=pl32, log64=l64, log32=l32,
                        values=values, mag=mag, P64_in=P64_in, P64=P64, P32=P32, MIN=MIN)


def case_sampler():
    out = {}
    for S in (4, 64):
        out[f"X{S}"] = draw(S)
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), minX=MINX, maxX=MAXX, do_log=DO_LOG, unit=UNIT,
                        **out)


def case_bayes_e2e():
    """bayeslib.bayes(pvSim, ...) GPU branch end to end: 3 Power_scan curves, two experiments:
    exp 0 on the simulation grid (bypass, bayeslib.py:182-183), exp 1 on a prefix of it
    (per-row griddata, :184-191)."""
    ini = get_initpoints(EXC_POWER, {"select_obs_sets": None})
    S, T, dt = 6, 48, 0.025
    simPar = [2000, T * dt, 128, T, 1, PT, 7, 10000]
    tgrid = np.linspace(0, T * dt, T + 1)
    # synthetic observations: the reference's own solver at the marked point, + fixed offsets
    plm = []
    for c in range(3):
        p, _ = run_pvsim((MARK * UNIT)[None, :-1], 2000, T * dt, 128, T, ini[c], np.float64)
        plm.append(np.log10(p[0]))
    rng = np.random.default_rng(99)
    obs0 = [v + rng.normal(0, 0.02, v.shape) for v in plm]
    npre = 31
    obs1 = [v[:npre] + 0.1 for v in plm]
    e_data = [([tgrid] * 3, obs0, [np.ones(T + 1)] * 3),
              ([tgrid[:npre]] * 3, obs1, [np.ones(npre)] * 3)]
    sim_flags = {"load_PL_from_file": False, "override_equal_auger": False, "override_equal_mu": False,
                 "override_equal_s": False, "log_pl": True, "self_normalize": False,
                 "random_sample": True, "num_points": S}
    gpu_info = {"sims_per_gpu": 4, "num_gpus": 1, "has_GPU": True, "threads_per_block": (1,),
                "max_sims_per_block": 1}
    minX = MINX * UNIT; maxX = MAXX * UNIT
    minX[-1] = -0.5; maxX[-1] = 0.5                  # exercise the mag_offset column (linear)
    np.random.seed(42)
    N, P, X = bayeslib.bayes(pvSimPCR.pvSim, np.array([0]), None, minX, maxX, DO_LOG, ini, simPar,
                             e_data, sim_flags, gpu_info, logger=None)
    np.savez_compressed(os.path.join(OUT, "bayes_e2e.npz"), X=X, P=P, ini=ini, T=T, time=T * dt,
                        length=2000.0, L=128, tol=7, MAX=10000, obs0=np.array(obs0), obs1=np.array(obs1),
                        tgrid=tgrid, npre=npre, minX=minX, maxX=maxX, do_log=DO_LOG, sims_per_gpu=4)


def case_fallback():
    """The reference CPU model exactly as shipped (no stand-in involved): scipy BDF +
    Simpson PL (pvSim_fallback.py:80-117).  Timing-baseline fixture, not a parity target
    for the GPU path (different quadrature)."""
    from pvSim_fallback import pvSim_cpu_fallback
    ini = get_initpoints(EXC_POWER, {"select_obs_sets": None})
    X = np.vstack([draw(2), MARK * UNIT])
    T = 400
    simPar = [2000, T * 0.025, 128, T, 1, PT, 7, 10000]
    pl = np.empty((3, len(X), T + 1)); secs = []
    for c in range(3):
        secs.append(pvSim_cpu_fallback(pl[c], X, simPar, ini[c]))
    np.savez_compressed(os.path.join(OUT, "fallback.npz"), X=X, ini=ini, T=T, time=T * 0.025,
                        length=2000.0, L=128, plI=pl, seconds=np.array(secs))


OBS_BALANCED = os.path.join(REF, "Example Data", "Balancedhighsurf_Power_scan_Observations.csv")


def _fallback64_task(job):
    """One SLURM array task of the reference's own distribution (bayeslib.py:131,:231): bayes() with the CPU model,
    has_GPU False, num_gpus = the number of tasks, sims_per_gpu = its block of samples.  The model handed to bayes() is
    pvSim_fallback.pvSim_cpu_fallback as shipped behind a recorder that keeps what each call wrote (in fp64; bayes's own
    buffer is float32, bayeslib.py:137) and the seconds it returned."""
    from pvSim_fallback import pvSim_cpu_fallback
    from bayes_io import get_data
    task, ntasks, S, T, Time, cutoff = job
    os.environ["SLURM_ARRAY_TASK_ID"] = str(task)
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)                            # noqa: F841  (one BLAS thread per task)
    except ImportError:
        pass
    ic_flags = {"time_cutoff": cutoff, "select_obs_sets": None, "noise_level": None}
    sim_flags = {"load_PL_from_file": False, "override_equal_auger": False, "override_equal_mu": False,
                 "override_equal_s": False, "log_pl": True, "self_normalize": False,
                 "random_sample": True, "num_points": S}
    ini = get_initpoints(EXC_POWER, ic_flags)
    e_data = get_data([OBS_BALANCED], ic_flags, sim_flags, scale_f=1e-23)
    calls = []

    def model(plI, matPar, simPar, init_dN):                             # bayeslib.py:148 call form
        pl64 = np.empty(plI.shape)
        sec = pvSim_cpu_fallback(pl64, matPar, simPar, init_dN)
        plI[:] = pl64                                                    # the cast pvSim_fallback.py:113 makes into bayes's buffer
        calls.append((pl64, sec))
        return sec

    gpu_info = {"sims_per_gpu": S // ntasks, "num_gpus": ntasks, "has_GPU": False}
    simPar = [2000, Time, 128, T, 1, PT, 7, 10000]
    np.random.seed(42)                                                   # parallel_bayes_gpu.py:35
    t0 = time.time()
    N, P, X = bayeslib.bayes(model, np.array([0]), None, MINX * UNIT, MAXX * UNIT, DO_LOG, ini, simPar, e_data,
                             sim_flags, gpu_info, logger=None)
    wall = time.time() - t0
    return task, P, X, [c[0] for c in calls], [c[1] for c in calls], wall, e_data


def _run_fallback64(T, Time, cutoff, ntasks=8, S=64):
    import multiprocessing as mp
    with mp.get_context("fork").Pool(ntasks) as pool:
        t0 = time.time()
        res = pool.map(_fallback64_task, [(k, ntasks, S, T, Time, cutoff) for k in range(ntasks)], chunksize=1)
        wall = time.time() - t0
    res.sort(key=lambda r: r[0])
    X = res[0][2]
    assert all(np.array_equal(r[2], X) for r in res)
    P = sum(r[1] for r in res)                                           # every task leaves the other blocks zero (bayeslib.py:131)
    blk = S // ntasks
    pl = np.empty((3, S, T + 1))
    secs = np.empty((3, ntasks))
    for k, r in enumerate(res):
        assert len(r[3]) == 3                                            # one model call per curve (one block per task)
        for c in range(3):
            pl[c, k * blk:(k + 1) * blk] = r[3][c]
            secs[c, k] = r[4][c]
    return X, P, pl, secs, np.array([r[5] for r in res]), wall, res[0][6]


def case_fallback64():
    """BASELINE.json configs[0] as worded: Power_scan (3 excitations, 128 nodes) x 64 random parameter samples on the CPU
    through the reference's own CPU path -- bayeslib.bayes(pvSim_fallback.pvSim_cpu_fallback, ...) with has_GPU False
    (bayeslib.py:148,:158-161,:198-201), observations = the shipped Balancedhighsurf_Power_scan_Observations.csv read by
    bayes_io.get_data -- run the way the reference distributes work: one SLURM array task per block of samples
    (bayeslib.py:131,:231; here 8 tasks x 8 samples on the container's 8 cores, their P arrays summed).  Two windows: the
    bench's (T = 8000 steps = 200 ns; observations cut at 200 ns) and the reference's full one (T = 80 000, 2000 ns).
    Stored: X, the CPU-branch likelihoods P, PL(t) of all 192 systems (float32, every 8th / 80th column; the first two
    samples' first 401 columns in fp64), the seconds every model call returned, wall time, the core count, and the
    observation values the likelihood was taken against (log10, bayes_io.py:67-77)."""
    ncores = os.cpu_count()
    out = {"ntasks": 8, "cores": ncores, "L": 128, "length": 2000.0}
    with open("/proc/cpuinfo") as fh:
        models = [ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")]
    out["cpu_model"] = np.array(models[0] if models else "unknown")
    for tag, T, Time, cutoff, dec in (("w8k", 8000, 200.0, 200, 8), ("full", 80000, 2000.0, 2000, 80)):
        X, P, pl, secs, task_wall, wall, e_data = _run_fallback64(T, Time, cutoff)
        out.update({"X": X, f"{tag}_T": T, f"{tag}_time": Time, f"{tag}_P": P, f"{tag}_dec": dec,
                    f"{tag}_pl32": pl[:, :, ::dec].astype(np.float32), f"{tag}_pl_head": pl[:, :2, :401].copy(),
                    f"{tag}_model_seconds": secs, f"{tag}_task_wall": task_wall, f"{tag}_wall": wall})
        for c in range(3):
            t, v = np.asarray(e_data[0][0][c]), np.asarray(e_data[0][1][c])
            assert np.allclose(t, 0.025 * np.arange(len(t)), rtol=0, atol=1e-9)     # a prefix of the simulation grid
            out[f"{tag}_obs_{c}"] = v
        print("fallback64 %s: wall %.1f s on %d tasks; model seconds per (sample x curve): %.3f" %
              (tag, wall, 8, secs.sum() / (64 * 3)), flush=True)
    out["ini"] = get_initpoints(EXC_POWER, {"select_obs_sets": None})
    np.savez_compressed(os.path.join(OUT, "fallback64.npz"), **out)


def case_bayes_realdata():
    """The reference's own ingestion (bayes_io.get_initpoints / get_data on the shipped example
    files, bayes_io.py:15-119) feeding bayeslib.bayes: experiment 0 = the shipped
    Balancedhighsurf observations cut at 5 ns (201 on-grid points -> bypass path), experiment 1 =
    off-grid, irregular observation times (per-row griddata, bayeslib.py:184-191)."""
    from bayes_io import get_data
    obs_file = os.path.join(REF, "Example Data", "Balancedhighsurf_Power_scan_Observations.csv")
    ic_flags = {"time_cutoff": 5, "select_obs_sets": None, "noise_level": None}
    sim_flags = {"load_PL_from_file": False, "override_equal_auger": False, "override_equal_mu": False,
                 "override_equal_s": False, "log_pl": True, "self_normalize": False,
                 "random_sample": True, "num_points": 4}
    ini = get_initpoints(EXC_POWER, ic_flags)
    e_data = get_data([obs_file], ic_flags, sim_flags, scale_f=1e-23)
    T, Time = 200, 5.0
    rng = np.random.default_rng(5)
    t_off, v_off = [], []
    for c in range(3):
        tt = np.sort(np.concatenate([[0.0, Time], rng.uniform(0, Time, 55), [0.025 * 17, 0.025 * 118]]))
        t_off.append(tt)
        v_off.append(np.interp(tt, e_data[0][0][c], e_data[0][1][c]) + 0.05)
    e_data.append((t_off, v_off, [np.ones_like(t) for t in t_off]))
    simPar = [2000, Time, 128, T, 1, PT, 7, 10000]
    gpu_info = {"sims_per_gpu": 3, "num_gpus": 1, "has_GPU": True, "threads_per_block": (1,),
                "max_sims_per_block": 1}
    np.random.seed(42)
    N, P, X = bayeslib.bayes(pvSimPCR.pvSim, np.array([0]), None, MINX * UNIT, MAXX * UNIT, DO_LOG, ini, simPar,
                             e_data, sim_flags, gpu_info, logger=None)
    out = {"X": X, "P": P, "ini": ini, "T": T, "time": Time, "length": 2000.0, "L": 128}
    for e in range(2):
        for c in range(3):
            out[f"t_{e}_{c}"] = np.asarray(e_data[e][0][c]); out[f"v_{e}_{c}"] = np.asarray(e_data[e][1][c])
            out[f"u_{e}_{c}"] = np.asarray(e_data[e][2][c])
    np.savez_compressed(os.path.join(OUT, "bayes_realdata.npz"), **out)


def case_csv_fixture():
    """Small DATA fixtures in the reference's two file formats, cut from its shipped example files:
    the three excitation rows, and the observation rows with t <= 6 ns of each of the three
    Balancedhighsurf curves (the tests apply time_cutoff = 5 on top)."""
    import csv
    obs_file = os.path.join(REF, "Example Data", "Balancedhighsurf_Power_scan_Observations.csv")
    with open(EXC_POWER, newline="") as fh, open(os.path.join(OUT, "exc_power_scan.csv"), "w", newline="") as out:
        w = csv.writer(out)
        for row in csv.reader(fh):
            if len(row):
                w.writerow(row)
    with open(obs_file, newline="") as fh, open(os.path.join(OUT, "obs_balanced_6ns.csv"), "w", newline="") as out:
        w = csv.writer(out)
        for row in csv.reader(fh):
            if row[0] == "END" or float(row[0]) <= 6.0:
                w.writerow(row)
    # the whole observation file (5601 / 8801 / 12801 points, 140 / 220 / 320 ns at 0.025 ns), gzip-compressed: the input of the
    # production-shape run on the GPU box (tools/e2e_production.py), where the reference checkout does not exist
    import gzip
    with open(obs_file, "rb") as fh, gzip.GzipFile(os.path.join(OUT, "obs_balanced_full.csv.gz"), "wb", mtime=0) as out:
        out.write(fh.read())


def case_legacy_odeint():
    """The two independent solvers the north star names as parity references, run on the same inputs
    (Length 311 nm, L = 128, exponential excitation a = 1e18 cm^-3, l = 100 nm like Testing/pvSetup.py
    :49-90, no Auger):
      Legacy/pvSim.pvSim (:129-173): same discretisation, BDF2 only, Thomas solve -- identical to the
        PCR path for PL[0..2], ~3e-4 apart afterwards (BDF order), SURVEY 8c T-C;
      Testing/PV_tester2.dydt (:13-49) + scipy odeint as in its __main__ (:91-99) with tight
        tolerances -- the time-converged solution of the same spatial scheme, SURVEY 8c T-D."""
    sys.path.insert(0, os.path.join(REF, "Legacy"))
    sys.path.insert(0, os.path.join(REF, "Testing"))
    import io
    import contextlib
    import pvSim as legacy
    import PV_tester2 as tester
    from scipy.integrate import odeint
    X = np.vstack([draw(2), MARK * UNIT])
    m10 = X[:, [0, 1, 2, 3, 4, 5, 6, 9, 10, 11]]                        # no CN, CP
    Length, L, T, dt = 311.0, 128, 240, 0.025
    Time = T * dt
    a_nm3, l_nm = 1e18 * 1e-21, 100.0
    # state snapshots (Legacy/pvSim.py:121-126,:169-171): the steps on which BDF1/BDF2 and the BDF ramp
    # coincide (0, 1, 2) and 3 / 10 / 30 / 100 % of the window (Testing/compare.py samples N, P, E there)
    pT = (0, 1, 2, 7, 24, 72, 240)
    simPar = [Length, Time, L, T, 1, pT, 7, 10000]
    with contextlib.redirect_stdout(io.StringIO()):
        itrs, (plN, plP, plE, plI) = legacy.pvSim(m10.copy(), simPar, (a_nm3, l_nm))
    # PV_tester2's recipe, non-dimensional (its __main__ :55-99)
    dx = Length / L
    dx3 = dx ** 3; dtdx = dt / dx; dtdx2 = dtdx / dx
    scales = np.array([dx3, dx3, dtdx2, dtdx2, dtdx2 / dx, dtdx, dtdx, 1 / dt, 1 / dt, 1 / dx])
    mp = m10 * scales
    xg = np.arange(L) + 0.5
    dN = (a_nm3 * dx3) * np.exp(-xg / (l_nm / dx))
    tSteps = np.linspace(0, T, T + 1)
    pl_ode = np.zeros((len(mp), T + 1))
    for thr in range(len(mp)):
        y0 = np.concatenate([mp[thr, 0] + dN, mp[thr, 1] + dN, np.zeros(L + 1)])
        # the tester halves hmax while a density is negative (:101-118), treating negatives as
        # integration artefacts; with random parameter samples the spatial scheme itself can
        # undershoot slightly, so time-convergence is established by two tolerance levels instead
        sols = []
        for rtol, atol in ((1e-8, 1e-12), (1e-10, 1e-14)):
            data, info = odeint(tester.dydt, y0, tSteps, args=(L, *mp[thr]), tfirst=True, rtol=rtol, atol=atol, hmax=0.5,
                                mxstep=50000, full_output=True)
            assert info["message"] == "Integration successful."
            sols.append(data)
        N, P = sols[1][:, :L], sols[1][:, L:2 * L]
        N9, P9 = sols[0][:, :L], sols[0][:, L:2 * L]
        conv = np.max(np.abs(np.sum(N * P, axis=1) / np.sum(N9 * P9, axis=1) - 1))
        assert conv < 1e-7, (thr, conv)
        pl_ode[thr] = mp[thr, 4] * np.sum(N * P - mp[thr, 0] * mp[thr, 1], axis=1) / (dx ** 2 * dt)   # pvSim's units
    np.savez_compressed(os.path.join(OUT, "legacy_odeint.npz"), X=X, length=Length, L=L, T=T, time=Time,
                        a_nm3=a_nm3, l_nm=l_nm, plI_legacy=plI, iters_legacy=np.array(itrs), plI_odeint=pl_ode,
                        pT=np.array(pT), plN_legacy=plN, plP_legacy=plP, plE_legacy=plE)


LEGACY_FULL_FILMS = ((2000.0, 1.273836e16), (2000.0, 1.153946e17), (2000.0, 1.648494e18), (311.0, 1.648494e18))


def _legacy_full_job(job):
    """One film of case_legacy_full in a worker process: Legacy/pvSim.pvSim as shipped."""
    import contextlib
    import io
    sys.path.insert(0, os.path.join(REF, "Legacy"))
    import pvSim as legacy
    m10, simPar, a_nm3, l_nm = job
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        itrs, (plN, plP, plE, plI) = legacy.pvSim(m10.copy(), simPar, (a_nm3, l_nm))
    return np.array(itrs), plN, plP, plE, plI, time.time() - t0


def case_legacy_full():
    """WHOLE-CURVE fixture of the second solver the north star names: Legacy/pvSim.pvSim (:129-173, iterate :36-88, Thomas
    solve :14-34) -- Euler at t = 0, BDF2 from t = 1 on (:94-97), no Auger terms -- over 2400 steps (60 ns at the
    reference's dt = 0.025 ns) on the three Power_scan excitations (Beer-Lambert profiles A exp(-x / l), A as fitted to
    Example Data/Power_scan_Excitations.csv; Legacy's own "exp" initialisation) of a 2000 nm film and the
    strongest one on a 311 nm film (Twothick's thin film), x 8 random samples of the box + the marked point.  With the
    BDF order capped at 2 and CN = CP = 0 pvSimPCR.py discretises the same equations the same way, so
    this pins the discretisation through an implementation that shares no code with pvSimPCR.py."""
    import multiprocessing as mp
    X = np.vstack([draw(8), MARK * UNIT])
    m10 = X[:, [0, 1, 2, 3, 4, 5, 6, 9, 10, 11]]                        # no CN, CP
    L, T, dt = 128, 2400, 0.025
    Time = T * dt
    l_nm = 1.0 / 6.000e-3
    pT = (2, 240, 2400)                                                  # 0.1 %, 10 %, 100 % of the window
    jobs = [(m10, [length, Time, L, T, 1, pT, 7, 10000], a * 1e-21, l_nm) for length, a in LEGACY_FULL_FILMS]
    with mp.get_context("fork").Pool(len(jobs)) as pool:
        res = pool.map(_legacy_full_job, jobs, chunksize=1)
    # PL columns kept: every step of the first 200 (BDF start-up, the stiff first steps), every 5th afterwards
    cols = np.concatenate([np.arange(0, 200), np.arange(200, T + 1, 5)])
    np.savez_compressed(os.path.join(OUT, "legacy_full.npz"), X=X, L=L, T=T, time=Time, l_nm=l_nm,
                        lengths=np.array([f[0] for f in LEGACY_FULL_FILMS]),
                        a_nm3=np.array([f[1] * 1e-21 for f in LEGACY_FULL_FILMS]), pT=np.array(pT), cols=cols,
                        iters_max=np.array([r[0] for r in res]), plN=np.array([r[1] for r in res]),
                        plP=np.array([r[2] for r in res]), plE=np.array([r[3] for r in res]),
                        plI=np.array([r[4][:, cols] for r in res]), seconds=np.array([r[5] for r in res]))


def _tester_refine_job(job):
    """One (film, sample) of case_tester_refine in a worker process: Testing/PV_tester2.dydt under scipy odeint."""
    sys.path.insert(0, os.path.join(REF, "Testing"))
    import PV_tester2 as tester
    from scipy.integrate import odeint
    mp_row, dN, L, T = job
    y0 = np.concatenate([mp_row[0] + dN, mp_row[1] + dN, np.zeros(L + 1)])      # PV_tester2.py:85-89
    tSteps = np.linspace(0, T, T + 1)                                            # :91
    t0 = time.time()
    sols = []
    from threadpoolctl import threadpool_limits
    for rtol, atol in ((1e-8, 1e-12), (1e-10, 1e-14)):
        with threadpool_limits(1):       # LSODA factorises a 385 x 385 Jacobian: 8 workers x 8 BLAS threads would spin
            data, info = odeint(tester.dydt, y0, tSteps, args=(L, *mp_row), tfirst=True, rtol=rtol, atol=atol, hmax=0.5,
                                mxstep=200000, full_output=True)
        assert info["message"] == "Integration successful."
        sols.append(mp_row[4] * np.sum(data[:, :L] * data[:, L:2 * L] - mp_row[0] * mp_row[1], axis=1))   # :120
    neg = bool((data[:, :2 * L] < 0).any())                                      # the tester's own acceptance test, :101
    return sols[1], float(np.max(np.abs(sols[1] / sols[0] - 1))), neg, time.time() - t0


def case_tester_refine():
    """The THIRD solver the north star names, as the target of a time-step refinement study: Testing/PV_tester2.dydt
    (:13-49: the semi-discrete equations of the same spatial scheme, no Auger) integrated by scipy odeint the way its
    __main__ does (:91-93, non-dimensional variables of :55-76), at rtol 1e-10 / atol 1e-14 -- the TIME-CONVERGED solution
    of the scheme pvSimPCR.py steps with BDF at fixed dt (each curve is checked against a second odeint run at rtol 1e-8:
    the two agree far below the refinement errors the tests measure, stored as `ode_conv`).  8 random samples of the box +
    the marked point x the three Power_scan excitations on a 2000 nm film + the strongest on a 311 nm film (the films of
    case_legacy_full), 800 steps of the reference's dt = 0.025 ns (20 ns), PL stored at every step in pvSim's units.
    tests/: the oracle / the GPU at T * k steps with plT = k, k = 1 .. 16, must approach these curves at second order
    (the Euler start of pvSimPCR.py:241-242 dominates the error of the early columns)."""
    import multiprocessing as mp
    X = np.vstack([draw(8), MARK * UNIT])
    m10 = X[:, [0, 1, 2, 3, 4, 5, 6, 9, 10, 11]]                        # no CN, CP
    L, T, dt = 128, 800, 0.025
    Time = T * dt
    l_nm = 1.0 / 6.000e-3
    jobs, unit = [], []
    for length, a in LEGACY_FULL_FILMS:
        dx = length / L
        dx3 = dx ** 3; dtdx = dt / dx; dtdx2 = dtdx / dx
        scales = np.array([dx3, dx3, dtdx2, dtdx2, dtdx2 / dx, dtdx, dtdx, 1 / dt, 1 / dt, 1 / dx])   # PV_tester2.py:62-65
        dN = (a * 1e-21 * dx3) * np.exp(-(np.arange(L) + 0.5) / (l_nm / dx))                              # :67-73
        jobs += [(row, dN, L, T) for row in m10 * scales]
        unit.append(dx ** 2 * dt)                                        # pvSimPCR.py:393 (the tester's own `dx**4/dt` is
        #                                                                  another unit; the curves are compared as ratios)
    with mp.get_context("fork").Pool(8) as pool:
        res = pool.map(_tester_refine_job, jobs, chunksize=1)
    F, S = len(LEGACY_FULL_FILMS), len(X)
    pl = np.array([r[0] for r in res]).reshape(F, S, T + 1) / np.array(unit)[:, None, None]
    conv = np.array([r[1] for r in res]).reshape(F, S)
    assert conv.max() < 1e-7, conv
    np.savez_compressed(os.path.join(OUT, "tester_refine.npz"), X=X, L=L, T=T, time=Time, l_nm=l_nm,
                        lengths=np.array([f[0] for f in LEGACY_FULL_FILMS]),
                        a_nm3=np.array([f[1] * 1e-21 for f in LEGACY_FULL_FILMS]), plI_odeint=pl, ode_conv=conv,
                        negative=np.array([r[2] for r in res]).reshape(F, S),
                        seconds=np.array([r[3] for r in res]).reshape(F, S))


def case_posterior():
    """The numeric core of the GUI that consumes *_BAYRAN_{P,X}.npy, run as shipped (statsmodels stand-in:
    refshim/statsmodels, only needed for the module-level import)."""
    sys.path.insert(0, os.path.join(REF, "Visualization"))
    import contextlib
    import io
    import utils as vis                                                      # (reference)
    S = 3000
    Xs = draw(S, seed=7)                                                     # solver units
    X = Xs / UNIT                                                            # the GUI works on *_BAYRAN_X.npy (raw units)
    rng = np.random.RandomState(3)
    # a likelihood surface with the scale of a real run (sum of ~1e4 squared log errors): peaked near
    # the marked point in (p0, B, tau_n, tau_p), long tails elsewhere, some NaN and -inf entries
    z = np.stack([np.log10(X[:, 1] / MARK[1]), np.log10(X[:, 4] / MARK[4]), (X[:, 9] - MARK[9]) / 400.0,
                  (X[:, 10] - MARK[10]) / 800.0, (X[:, 2] - X[:, 3]) / 60.0], axis=1)
    LL = -2.0e4 * np.sum(z ** 2, axis=1) - 50.0 * rng.rand(S)
    LL[rng.choice(S, 25, replace=False)] = np.nan
    LL[rng.choice(S, 15, replace=False)] = -np.inf
    keep = ~np.isnan(LL)                                                     # LikelihoodData.filter_nan :33-38
    LLk, Xk = LL[keep], X[keep]
    n_obs, c = 3 * 8000.0, 2.0
    tf = n_obs * c                                                           # marginalization_visual.py:589
    P = vis.normalize(LLk / tf)                                              # :590-591
    cols = {"p0": np.log10(Xk[:, 1]), "mu_n": Xk[:, 2], "mu_p": Xk[:, 3], "B": np.log10(Xk[:, 4]),
            "tau_n": Xk[:, 9], "tau_p": Xk[:, 10]}
    names = list(cols)
    mean = np.array([vis.w_mean(cols[k], P) for k in names])
    var = np.array([vis.w_variance(cols[k], P) for k in names])
    ws = np.sum(P ** 2)
    sstd = np.array([vis.w_sample_var(cols[k], P, ws) for k in names])
    skew = np.array([vis.w_skew(cols[k], P) for k in names])
    kurt = np.array([vis.w_kurtosis(cols[k], P) for k in names])
    cov = np.array([[vis.covariance(cols[a], cols[b], P) for b in names] for a in names])
    with contextlib.redirect_stdout(io.StringIO()):
        ci = np.array([vis.credible_interval(cols[k], P) for k in names])
    limits = {"p0": (14.0, 16.0), "mu_n": (0.0, 50.0), "mu_p": (0.0, 50.0), "B": (-11.0, -9.0),
              "tau_n": (1.0, 1000.0), "tau_p": (100.0, 1500.0)}             # tau_p: narrower than the data
    secondary = {k: False for k in names}
    bins = 32
    h1 = {}
    for k in names:                                                          # "mu" in the name -> sampling correction
        marP, edges = vis.marginalize_1D(P, limits, bins, secondary, k, cols[k])
        h1[k] = (marP, edges)
    pairs = [("p0", "B"), ("tau_n", "tau_p"), ("mu_n", "p0")]
    h2 = [vis.marginalize_2D(P, limits, bins, secondary, pr, cols[pr[0]], cols[pr[1]])[0] for pr in pairs]
    np.savez_compressed(os.path.join(OUT, "posterior.npz"), X=X, LL=LL, tf=tf, names=np.array(names), P=P,
                        col_index=np.array([1, 2, 3, 4, 9, 10]), col_log=np.array([1, 0, 0, 1, 0, 0]), mean=mean, var=var, ws=ws, sstd=sstd,
                        skew=skew, kurt=kurt, cov=cov, ci=ci, bins=bins,
                        limits=np.array([limits[k] for k in names]),
                        h1=np.stack([h1[k][0] for k in names]), edges=np.stack([h1[k][1] for k in names]),
                        pairs=np.array([[names.index(a), names.index(b)] for a, b in pairs]), h2=np.stack(h2))


CASES = {"posterior": case_posterior, "legacy_odeint": case_legacy_odeint, "tester_refine": case_tester_refine, "legacy_full": case_legacy_full, "csv_fixture": case_csv_fixture, "bayes_realdata": case_bayes_realdata, "pcr_norm": case_pcr_norm, "probs": case_probs, "sampler": case_sampler,
         "pvsim_small": case_pvsim_small, "pvsim_power": case_pvsim_power, "pvsim_bundle": case_pvsim_bundle,
         "pvsim_twothick": case_pvsim_twothick, "bayes_e2e": case_bayes_e2e,
         "fallback": case_fallback, "fallback64": case_fallback64}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    names = sys.argv[1:] or list(CASES)
    for n in names:
        t0 = time.time()
        CASES[n]()
        print("golden %-16s %.1f s" % (n, time.time() - t0), flush=True)
