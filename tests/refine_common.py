"""Time-step refinement against Testing/PV_tester2.py + odeint (tests/golden/tester_refine.npz), shared by the CPU oracle
test and the -m gpu test: the inputs of a film as the oracle / the product take them, and ONE set of bounds.

The fixture holds the time-converged solution of the spatial scheme (PV_tester2.dydt :13-49 under scipy odeint at rtol
1e-10, the way its __main__ :91-93 integrates it) at the 801 output times t = j * 0.025 ns of a 20 ns window.  pvSimPCR's
scheme on the same grid with T * k steps of dt / k (plT = k stores the same 801 columns) must approach it at SECOND order:
the Euler start of pvSimPCR.py:241-242 leaves an O(dt^2) error in the first columns, which the BDF ramp (:243-250) carries
on at its own higher order.  Measured on the CPU oracle (the worst of 9 samples, per film and k = 1, 2, 4, 8, 16):
    worst column (step 1):   6.7e-3 .. 8.8e-3 -> 2.3e-3 .. 3.0e-3 -> 5.6e-4 .. 7.6e-4 -> 1.5e-4 .. 2.0e-4 -> 3.8e-5 .. 5.1e-5
                             = 2.9x, 4.0x, 3.8x, 3.9x per halving
    last column (20 ns):     1.2e-6 .. 1.6e-4 -> ... -> 8.7e-9 .. 1.0e-6 = 3.1x .. 3.7x per halving
"""
import numpy as np

REFINE_K = (1, 2, 4, 8, 16)
# per halving of dt, the film's worst deviation from the odeint curve shrinks by at least ...
SHRINK_FIRST = 2.8          # dt -> dt / 2 (measured 2.92 .. 2.96: the second-order regime is not fully reached at dt)
SHRINK_LATER = 3.5          # every later halving (measured 3.82 .. 4.0)
SHRINK_END = 3.0            # the last column (measured 3.1 .. 3.7)
# ... and at dt / 16 it is at most
WORST_AT_16 = 1e-4          # any column (measured 5.1e-5)
END_AT_16 = 3e-6            # the last column (measured 1.04e-6)
TAIL_AT_16 = 5e-6           # every column from 1 ns on (measured 2.8e-6)
# the odeint curves themselves are time-converged to (rtol 1e-8 against rtol 1e-10 run): the fixture's `ode_conv`
ODE_CONV = 1e-7


def film_inputs(g, f):
    """X (Auger off: PV_tester2.dydt has no Auger terms), thickness and the "points" excitation of film f."""
    X = g["X"].copy()
    X[:, 7] = 0.0; X[:, 8] = 0.0
    L, length = int(g["L"]), float(g["lengths"][f])
    x = np.arange(L) + 0.5
    dN = float(g["a_nm3"][f]) * np.exp(-x / (float(g["l_nm"]) / (length / L)))      # PV_tester2.py:67-73 (pvSimPCR.py:347-353)
    return X, length, dN


def deviation(pl, ode):
    """|PL / PL_odeint - 1| per (sample, column)."""
    return np.abs(pl / ode - 1)


def check_refinement(devs, label=""):
    """devs: {k: deviation array (samples, T + 1)} of ONE film for k in REFINE_K.  Asserts second-order convergence to the
    odeint curves and the bounds at dt / 16; returns the film's (worst, last-column) deviations per k for the record."""
    worst = {k: float(devs[k].max()) for k in REFINE_K}
    end = {k: float(devs[k][:, -1].max()) for k in REFINE_K}
    for a, b in zip(REFINE_K[:-1], REFINE_K[1:]):
        need = SHRINK_FIRST if a == 1 else SHRINK_LATER
        assert worst[a] / worst[b] >= need, (label, "worst column", a, b, worst[a], worst[b])
        assert end[a] / end[b] >= SHRINK_END, (label, "last column", a, b, end[a], end[b])
        # the worst column is an early one (the start-up error), never the tail
        assert int(devs[b].max(axis=0).argmax()) <= 3, (label, b, int(devs[b].max(axis=0).argmax()))
    assert worst[16] <= WORST_AT_16, (label, worst[16])
    assert end[16] <= END_AT_16, (label, end[16])
    assert float(devs[16][:, 40:].max()) <= TAIL_AT_16, (label, float(devs[16][:, 40:].max()))
    return worst, end
