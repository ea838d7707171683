#!/usr/bin/env python3
"""The reference's PRODUCTION shape end to end, both integration levels (round-4 review, "What's missing" 4).

The entry script's configuration (parallel_bayes_gpu.py:72-131, :183-198): S = 2^17 random samples of the shipped 13-column
box (seed 42), L = 128, Time = 2000 ns in T = 80 000 steps, tol 7, MAX 10 000, log-PL likelihood, real observation files read
by get_data (here: Example Data/Balancedhighsurf_Power_scan_Observations.csv, 5601 / 8801 / 12801 points, carried as
tests/golden/obs_balanced_full.csv.gz; excitations tests/golden/exc_power_scan.csv), export of <name>_BAYRAN_{P,X}.npy.
(Thickness: the Power_scan excitations belong to the 2000 nm film; the script's literal Length = 311 goes with an
input file that is not shipped.)

  level A  trpl_amd.driver.bayes with gpu_info["fused"] = True: one fused launch per block (observation times are a prefix of
           the simulation grid: the in-kernel griddata of trpl_loglik_obs); the window ends at the last observation.
  level B  the reference's own call sequence -- pvSim -> fastlog -> griddata -> prob per (curve, block of sims_per_gpu samples,
           experiment), float32 PL staged through host memory, all T + 1 steps solved (bayeslib.py:117-201) -- with the three
           drop-in callables, at the reference's sims_per_gpu = 1024 (parallel_bayes_gpu.py:104) and at the value
           INTEGRATION.md recommends.

For each: wall time from bayes() to the two .npy files on disk, the reference's three timers (bayeslib.py:248-251),
likelihoods/s.  Validation: a 256-sample subsample against the CPU oracle's restatement of bayeslib.simulate, level B against
level A, and FAST against STRICT (floor_col of every system, likelihood gap of the floor-free samples).

    python tools/e2e_production.py [--S 131072] [--levels A,B1024,B16384] [--out gpurun_out/r5/e2e_production.json]
"""
import argparse
import gzip
import json
import logging
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


class ListHandler(logging.Handler):
    def __init__(self):
        super().__init__()
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def timers_from(lines):
    out = {}
    for ln in lines:
        for key, tag in (("Total tEvol time", "solver_s"), ("Total err_sq time", "err_sq_s"), ("Total misc time", "misc_s")):
            if ln.startswith(key):
                out[tag] = float(ln.split("[")[1].split("]")[0].split()[0])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=2 ** 17)                      # parallel_bayes_gpu.py:123
    ap.add_argument("--T", type=int, default=80000)                        # :75
    ap.add_argument("--time", type=float, default=2000.0)                  # :74
    ap.add_argument("--levels", default="A,B1024,B16384")
    ap.add_argument("--oracle-samples", type=int, default=256)
    ap.add_argument("--no-strict", action="store_true")
    ap.add_argument("--max-host-gib", type=float, default=None, help="gpu_info['max_host_bytes'] of the unfused levels (default: the driver's)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r6", "e2e_production.json"))
    a = ap.parse_args()

    import trpl_amd
    from trpl_amd import sampler as sm

    work = tempfile.mkdtemp(prefix="trpl_e2e_")
    obs_csv = os.path.join(work, "Balancedhighsurf_Power_scan_Observations.csv")
    with gzip.open(os.path.join(GOLDEN, "obs_balanced_full.csv.gz"), "rb") as fh, open(obs_csv, "wb") as out:
        out.write(fh.read())
    # ---- the entry script's configuration (parallel_bayes_gpu.py:72-131) ----
    Length, L, T, Time = 2000.0, 128, a.T, a.time
    simPar = [Length, Time, L, T, 1, (0, 1, 3, 10, 30, 100), 7, 10000]
    ic_flags = {"time_cutoff": Time, "select_obs_sets": None, "noise_level": None}
    sim_flags = {"load_PL_from_file": False, "override_equal_auger": False, "override_equal_mu": False,
                 "override_equal_s": False, "log_pl": True, "self_normalize": False, "random_sample": True,
                 "num_points": a.S}
    t0 = time.perf_counter()
    iniPar = trpl_amd.get_initpoints(os.path.join(GOLDEN, "exc_power_scan.csv"), ic_flags)
    e_data = trpl_amd.get_data([obs_csv], ic_flags, sim_flags, scale_f=1e-23)
    ingest_s = time.perf_counter() - t0
    n_obs = [len(t) for t in e_data[0][0]]
    minX, maxX = sm.DEFAULT_MINX * sm.UNIT_CONVERSIONS, sm.DEFAULT_MAXX * sm.UNIT_CONVERSIONS      # :183-184
    report = {"config": {"S": a.S, "L": L, "T": T, "time_ns": Time, "length_nm": Length, "tol": 7, "MAX": 10000, "curves": 3,
                         "observations": "Balancedhighsurf_Power_scan_Observations.csv via dataio.get_data", "n_obs": n_obs,
                         "seed": 42, "box": "parallel_bayes_gpu.py:86-92"},
              "ingest_s": ingest_s, "levels": {}}
    print("ingested %s observation points in %.2f s" % (n_obs, ingest_s), flush=True)

    results = {}
    for level in a.levels.split(","):
        fused = level == "A"
        digits = "".join(ch for ch in level[1:] if ch.isdigit())
        forced = level[1 + len(digits):] or None                            # "B1024single" / "B1024pair": gpu_info["kernel"]
        group = a.S if fused else int(digits)
        gpu_info = {"sims_per_gpu": group, "num_gpus": 1, "has_GPU": True, "threads_per_block": (128,), "max_sims_per_block": 1,
                    "fused": fused}
        if forced:
            gpu_info["kernel"] = forced
        if a.max_host_gib is not None and not fused:
            gpu_info["max_host_bytes"] = int(a.max_host_gib * 2 ** 30)
        h = ListHandler()
        logger = logging.getLogger("e2e_" + level)
        logger.setLevel(logging.INFO)
        logger.handlers = [h]

        class Progress(logging.Handler):                                     # a line a minute while the blocks go by
            last = time.perf_counter()

            def emit(self, record):
                now = time.perf_counter()
                if now - Progress.last > 30:
                    Progress.last = now
                    print("  [%s] %s" % (level, record.getMessage()), flush=True)
        logger.addHandler(Progress())
        np.random.seed(42)                                                   # parallel_bayes_gpu.py:35
        out_dir = os.path.join(work, "out_" + level)
        t0 = time.perf_counter()
        N, P, X = trpl_amd.bayes(trpl_amd.pvSim, None, None, minX, maxX, sm.DEFAULT_DO_LOG, iniPar, list(simPar), e_data,
                                 sim_flags, gpu_info, logger=logger)
        t1 = time.perf_counter()
        Xc = X / sm.UNIT_CONVERSIONS                                         # :194
        trpl_amd.export(out_dir, P[0], Xc)                                   # :197-198
        t2 = time.perf_counter()
        files = sorted(os.listdir(out_dir))
        rec = {"integration": "driver.bayes, fused launch per block" if fused else
               "reference call sequence pvSim -> fastlog -> griddata -> prob (drop-in callables)",
               "sims_per_gpu": group, "kernel": forced or "library's choice per launch", "max_host_bytes": gpu_info.get("max_host_bytes", trpl_amd.driver.DEFAULT_MAX_HOST_BYTES), "bayes_wall_s": t1 - t0, "export_s": t2 - t1, "wall_to_npy_s": t2 - t0,
               "likelihoods_per_s": a.S / (t2 - t0), "files": files, "finite_likelihoods": int(np.isfinite(P[0]).sum()),
               "steps_solved_per_system": [n - 1 for n in n_obs] if fused else [T] * 3}
        rec.update(timers_from(h.lines))
        steps = sum(rec["steps_solved_per_system"]) + 3
        rec["system_timesteps_per_s"] = a.S * steps / (t1 - t0)
        report["levels"][level] = rec
        results[level] = P[0].copy()
        print("level %s: %.1f s to %s (%.0f likelihoods/s; solver %.1f s)" % (level, t2 - t0, files, rec["likelihoods_per_s"],
                                                                              rec.get("solver_s", float("nan"))), flush=True)
        json.dump(report, open(a.out, "w"), indent=1) if os.path.isdir(os.path.dirname(a.out)) else None

    # ---- validation ----
    val = {}
    ref_level = "A" if "A" in results else sorted(results)[0]
    np.random.seed(42)
    _, _, X = sm.make_grid(1, minX, maxX, sm.DEFAULT_DO_LOG, sim_flags)
    times = [np.asarray(t) for t in e_data[0][0]]
    obs = [np.asarray(v) for v in e_data[0][1]]
    t0 = time.perf_counter()
    fi, si = {}, {}
    Pf = trpl_amd.loglik(X, iniPar, Length, Time, L, T, obs, times=times, info=fi)
    t1 = time.perf_counter()
    clear_f = (fi["floor_col"] == -1).all(axis=0)                            # floor-free samples (include/trpl.h)
    for level, Pl in results.items():
        if level == ref_level:
            continue
        both = np.isfinite(Pl) & np.isfinite(results[ref_level])
        rel = np.abs(Pl[both] / results[ref_level][both] - 1)
        relc = np.abs(Pl[both & clear_f] / results[ref_level][both & clear_f] - 1)
        val["%s_vs_%s" % (level, ref_level)] = {
            "max_rel_floor_free": float(relc.max()), "max_rel_all": float(rel.max()), "median_rel": float(np.median(rel)),
            "finite_in_one_only": int((np.isfinite(Pl) != np.isfinite(results[ref_level])).sum()),
            "note": "both levels stage PL in float32 like the reference (bayeslib.py:137); a launch of <= 3072 systems runs the "
                    "one-system FAST kernel, larger ones the paired kernel: the two agree to rounding except below the "
                    "cancellation floor (floor_col >= 0), where PL is set by rounding in any evaluation"}
    if not a.no_strict:
        print("FAST pass %.1f s; STRICT pass running ..." % (t1 - t0), flush=True)
        Ps = trpl_amd.loglik(X, iniPar, Length, Time, L, T, obs, times=times, info=si, strict=True)
        t2 = time.perf_counter()
        clear = clear_f & (si["floor_col"] == -1).all(axis=0)
        gap = np.abs(Pf[clear] / Ps[clear] - 1)
        big = np.abs(Pf / Ps - 1) > 1e-6
        val["fast_vs_strict"] = {"fast_s": t1 - t0, "strict_s": t2 - t1, "floor_col_identical": bool(np.array_equal(fi["floor_col"], si["floor_col"])),
                                 "status_identical": bool(np.array_equal(fi["status"], si["status"])),
                                 "floor_free_fraction": float(clear.mean()), "max_rel_gap_floor_free": float(gap.max()),
                                 "p999_rel_gap_floor_free": float(np.quantile(gap, 0.999)),
                                 "samples_above_1e-6": int(big.sum()), "of_which_flagged_by_floor_col": int((big & ~clear).sum()),
                                 "iteration_totals_differ_on": int((fi["iters_total"] != si["iters_total"]).sum()),
                                 "systems": int(fi["iters_total"].size)}
        print("FAST vs STRICT: %s" % val["fast_vs_strict"], flush=True)
    if a.oracle_samples > 0:
        import oracle
        n = a.oracle_samples
        idx = np.linspace(0, a.S - 1, n).astype(int)
        threads = max(1, min(len(os.sched_getaffinity(0)), 64))
        T_o = max(n_obs) - 1                                                 # the oracle steps to the last observation
        Time_o = T_o * (Time / T)
        t0 = time.perf_counter()
        want = oracle.simulate_loglik(X[idx], iniPar, Length, Time_o, L, T_o, [(times, obs)], sims_per_gpu=n,
                                      pl_dtype=np.float64, nthreads=threads)[0]
        t1 = time.perf_counter()
        got = results[ref_level][idx]
        rel = np.abs(got / want - 1)
        val["oracle_subsample"] = {"samples": n, "oracle_s": t1 - t0, "oracle_threads": threads, "window_steps": T_o,
                                   "max_rel": float(rel.max()), "median_rel": float(np.median(rel)),
                                   "samples_within_1e-8": int((rel < 1e-8).sum())}
        print("oracle subsample: %s" % val["oracle_subsample"], flush=True)
    report["validation"] = val
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
