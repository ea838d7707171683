"""File formats either side of the hot path: the observation / excitation CSV files
the reference reads (bayes_io.get_data :15-104, get_initpoints :106-119) and the two .npy files
it writes (bayes_io.export :121-140).  Host-side, tiny; restated so that real data can be fed to
the fused likelihood without the reference on the path.

Observation file: rows `t, PL, uncertainty`; a row whose t is 0 starts the next curve; a final
row `END`.  Excitation file: one row of L values per curve.
"""
import csv
import os
import sys

import numpy as np


def get_initpoints(init_file, ic_flags=None, scale_f=1e-21):
    """Excitation profiles, (C, L), cm^-3 -> nm^-3 (bayes_io.py:106-119)."""
    select = (ic_flags or {}).get("select_obs_sets")
    with open(init_file, newline="") as fh:
        rows = [r for r in csv.reader(fh) if len(r)]
    arr = np.array(rows, dtype=float)
    if select is not None:
        arr = arr[select]
    return arr * scale_f


def _finish_curve(t, pl, unc, scale_f, noise, normalize, log_pl, cutoff):
    t = np.array(t)
    pl = np.array(pl)
    if noise is not None:                                   # bayes_io.py:51-52 (global numpy RNG)
        pl = (pl + noise * np.random.normal(0, 1, len(pl))) * scale_f
    else:
        pl = pl * scale_f
    unc = np.array(unc) * scale_f
    if normalize:                                           # :59-60
        pl = pl / max(pl)
    if log_pl:                                              # :67-77
        pl = np.abs(pl)
        pl[pl < cutoff] = cutoff
        unc = unc / pl
        unc = unc / 2.3
        pl = np.log10(pl)
    return t, pl, unc


def get_data(exp_files, ic_flags, sim_flags, logger=None, scale_f=1e-23):
    """Observations: list (one per file) of (times[c], values[c], uncertainty[c]) with the
    reference's preprocessing: scale, optional noise / self-normalisation, |.|, clamp to
    DBL_MIN, log10, sigma -> sigma/PL/2.3 (bayes_io.py:15-104)."""
    cutoff = sys.float_info.min
    early = ic_flags.get("time_cutoff")
    select = ic_flags.get("select_obs_sets")
    noise = ic_flags.get("noise_level")
    log_pl = sim_flags["log_pl"]
    normalize = sim_flags["self_normalize"]
    out = []
    for path in exp_files:
        ts, pls, uncs = [], [], []
        cur = ([], [], [])
        with open(path, newline="") as fh:
            for row in csv.reader(fh):
                end = row[0] == "END"
                if end or (float(row[0]) == 0 and len(cur[0])):     # a curve is complete
                    t, p, u = _finish_curve(*cur, scale_f, noise, normalize, log_pl, cutoff)
                    ts.append(t); pls.append(p); uncs.append(u)
                    if logger is not None:
                        logger.info("PL curve #{} finished reading: {} points".format(len(ts), len(t)))
                    cur = ([], [], [])
                if end:
                    break
                tv = float(row[0])
                if early is None or tv <= early:                    # :90-95
                    cur[0].append(tv); cur[1].append(float(row[1])); cur[2].append(float(row[2]))
        if select is not None:
            ts = [ts[i] for i in select]; pls = [pls[i] for i in select]; uncs = [uncs[i] for i in select]
        out.append((ts, pls, uncs))
    return out


def export(out_filename, P, X, logger=None):
    """<dir>/<base>_BAYRAN_P.npy and _BAYRAN_X.npy (bayes_io.py:121-140)."""
    os.makedirs(out_filename, exist_ok=True)
    base = os.path.basename(out_filename)
    np.save(os.path.join(out_filename, "{}_BAYRAN_P.npy".format(base)), P)
    np.save(os.path.join(out_filename, "{}_BAYRAN_X.npy".format(base)), X)
    if logger is not None:
        logger.info("Wrote {}".format(out_filename))
