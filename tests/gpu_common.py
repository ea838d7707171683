"""Helpers and constants shared by the -m gpu test files (tests/test_gpu_*.py; fixtures shared between files are in conftest.py).
The envelope constants are those of include/trpl.h; tests/test_abi.py checks that header, binding and this file agree."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DT = 0.025


T_BENCH = 8000


FLOOR = 1e-4                       # TRPL_PL_FLOOR_EXCESS


KERNELS = [dict(strict=True), dict(kernel="single"), dict(kernel="pair")]


IDS = ["strict", "single", "pair"]


RTOL_STRICT = 0.0          # bit-identical: relerr(...) <= RTOL_STRICT


RTOL_FAST = 1e-9


ENVELOPE_K_THICK = 5e-13          # TRPL_PL_ENVELOPE_K_THICK (include/trpl.h): the 2000 nm films at L = 128


# prefactor k of the envelope |dPL / PL| <= 1e-9 + k / r between FAST and the reference evaluation, per film
# (include/trpl.h: the state gap that 1 / r amplifies grows with the stencil's stiffness D dt / dx^2)
ENVELOPE_K = {2000.0: 5e-13, 311.0: 1e-11}       # TRPL_PL_ENVELOPE_K_THICK / _THIN of include/trpl.h


ENVELOPE_K_L512 = 2e-12       # TRPL_PL_ENVELOPE_K_L512: the 2000 nm film at L = 512 (dx = 3.9 nm)
SSE_GATE = {2000.0: 1e-9, 311.0: 1e-9}           # floor-free squared-error sums over 8000 steps (measured 4e-12)


def needs_experimental(gpu, what):
    """TRPL_FLAG_MIXED / TRPL_FLAG_HIST32 are the measured-and-rejected steppers of DESIGN.md section 7: compiled only into
    `make EXPERIMENTAL=1` (libtrpl_hip_exp.so, loaded with TRPL_LIBRARY=...).  Under the default library their tests skip --
    after checking that the flag is REFUSED with a message, not ignored."""
    import pytest
    if gpu._abi.has_experimental():
        return
    w = gpu.workloads
    L = 256
    with pytest.raises(gpu.TrplError) as e:
        gpu.solve_pl(w.samples(2)[:, :12], 2000.0, 4 * DT, L, 4, w.beer_lambert(w.POWER_SCAN_A_CM3[0], 2000.0, L), **what)
    assert e.value.code == gpu._abi.ERR_UNSUPPORTED and "EXPERIMENTAL=1" in str(e.value), str(e.value)
    pytest.skip("default library: %s needs `make EXPERIMENTAL=1`" % ", ".join(what))


def nthreads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 32))


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.abs(b)))


def above_floor(pl, rel=1e-12):
    """PL points above the cancellation floor: >= rel * PL(0)."""
    return np.abs(pl) >= rel * np.abs(pl[:, :1])


def excess_scale(X, length, L=128):
    """B L n0p0 in the units of PL (nm^-2 ns^-1): the non-dimensional rate * L * N0 * P0 of pvSimPCR.py:327-331 divided
    by dx^2 dt (:393) -- the time step cancels."""
    dx = length / L
    return X[:, 4] * L * X[:, 0] * X[:, 1] * dx


def deviation_bound(pl_ref, scale):
    """The header's envelope, 1e-9 + TRPL_PL_ENVELOPE_K_THICK / r per point (measured prefactor 2e-13, tools/floor_study.py);
    inf where the reference PL is not positive."""
    r = pl_ref / scale[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        b = 1e-9 + ENVELOPE_K_THICK / r
    b[~(pl_ref > 0)] = np.inf
    return b


def first_below(pl, thr):
    """first column with pl < thr (or non-positive / NaN), -1 if none"""
    bad = ~(pl >= thr[:, None])
    return np.where(bad.any(axis=1), bad.argmax(axis=1), -1).astype(np.int32)


def follows_iteration_path(it, want, label=""):
    """THE rule for "FAST follows the reference's iteration path" (include/trpl.h, TRPL_FLAG_STRICT paragraph), one for the
    whole suite: every system's iteration total EQUALS the reference evaluation's (reference golden, oracle or STRICT); at
    most ONE system of the batch may differ, by ONE iteration (a knife-edge decision of the convergence test: measured 0 on
    every golden and oracle batch of the suite, 8 systems of 196 608 over T = 80 000).  Returns the number of differing
    systems (0 or 1)."""
    d = np.abs(np.asarray(it, dtype=np.int64) - np.asarray(want, dtype=np.int64))
    n = int((d != 0).sum())
    if n:
        where = np.argwhere(d != 0)[:4].tolist()
        assert d.max() <= 1 and n <= 1, "%s: %d system(s) leave the reference's iteration path (first at %s, largest gap %d)" % (
            label, n, where, int(d.max()))
    return n


# ------------------------------------------------------------------ paired kernel vs oracle / goldens
def _check_pl_against(gpu, X12, length, Time, L, T, ini, want_pl, want_iters, kernel, rtol=1e-9):
    pl, st, it, _ = gpu.solve_pl(X12, length, Time, L, T, ini, kernel=kernel)
    assert not st.any()
    ok = above_floor(want_pl)
    assert ok.mean() > 0.95
    err = np.max(np.abs(pl[ok] - want_pl[ok]) / np.abs(want_pl[ok]))
    assert err < rtol, err
    follows_iteration_path(it, want_iters, "kernel=%s" % kernel)
    return err, float((it == want_iters).mean())


def record(name, payload):
    """measured figures of a run, kept beside the logs (gpurun_out/ is merged back from the GPU box)"""
    d = os.path.join(ROOT, "gpurun_out", "r6")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "test_%s.json" % name), "w") as f:
            json.dump(payload, f, indent=1)
    except OSError:
        pass
