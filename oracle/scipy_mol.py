"""Port of the reference's CPU `model`, pvSim_fallback.pvSim_cpu_fallback (pvSim_fallback.py:80-117):
method of lines on the same finite-volume grid (RHS: dydt2, :18-78), scipy solve_ivp(method='BDF',
max_step=1, rtol=1e-5, atol=1e-8) sampled on the T+1 output times (:105-107), Simpson PL (:112).

TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT: it is the "scipy CPU path" timing baseline named by
the north star, usable on the GPU box where the reference is absent.  Pinned to
tests/golden/fallback.npz, which was produced by the reference's own module as shipped
(tests/test_oracle_golden.py::test_scipy_port_matches_reference_fallback).  It is NOT a parity
target of the GPU path: Simpson over cell centres omits the two half end-cells, so log10 PL differs
from the midpoint rule by up to 0.04 dex at t = 0.
"""
import time

import numpy as np
from scipy.integrate import simpson, solve_ivp

EPS0 = 8.854 * 1e-12 * 1e-9      # C / (V nm)                    pvSim_fallback.py:12-16
Q = 1.0
Q_C = 1.602e-19
KBT = .02569257
LAMBDA0 = 704.3


def rhs(t, y, m, dx, Sf, Sb, mu_n, mu_p, n0, p0, CN, CP, tauN, tauP, B, eps):
    """d/dt of (N[m], P[m], E[m+1]) -- drift-diffusion-recombination, pvSim_fallback.py:18-78."""
    N, P, E = y[:m], y[m:2 * m], y[2 * m:]
    excess = N * P - n0 * p0
    Jn = np.empty(m + 1)
    Jp = np.empty(m + 1)
    front = Sf * excess[0] / (N[0] + P[0])                 # surface recombination currents (:40-45)
    back = Sb * excess[-1] / (N[-1] + P[-1])
    Jn[0], Jn[m], Jp[0], Jp[m] = front, -back, -front, back
    Jn[1:-1] = mu_n * ((N[:-1] + N[1:]) / 2) * (Q * E[1:-1]) + (mu_n * KBT) * ((N[1:] - N[:-1]) / dx)   # :50-51
    Jp[1:-1] = mu_p * ((P[:-1] + P[1:]) / 2) * (Q * E[1:-1]) - (mu_p * KBT) * ((P[1:] - P[:-1]) / dx)   # :54-55
    dE = -(Jn + Jp) * (Q_C / (eps * EPS0))                                                              # :58
    # the three loss terms are subtracted one after the other, in the reference's order (:60-74): with the same association
    # every step-size decision of the adaptive integrator is the reference's, and the port reproduces its PL to rounding
    # instead of to the integrator's accumulated tolerance (a summed `loss` measured up to 4e-4 apart on 3 of 192 curves)
    rad = B * excess                                                                                    # :60
    nonrad = excess / ((tauN * P) + (tauP * N))                                                         # :61
    auger = (CN * N + CP * P) * excess                                                                  # :63
    dN = (1 / Q) * ((Jn[1:] - Jn[:-1]) / dx) - rad - nonrad - auger                                     # :65-68
    dP = (-1 / Q) * ((Jp[1:] - Jp[:-1]) / dx) - rad - nonrad - auger                                    # :71-74
    return np.concatenate([dN, dP, dE])


def solve_one(mp, length, Time, L, T, init_dN):
    """PL(t_0..t_T) of one parameter row mp[13] (solver units), one excitation."""
    n0, p0, DN, DP, B, Sf, Sb, CN, CP, tauN, tauP, lam, _mag = mp
    dx = length / L
    args = (L, dx, Sf, Sb, DN / KBT, DP / KBT, n0, p0, CN, CP, tauN, tauP, B, (lam / LAMBDA0) ** -1)
    y0 = np.concatenate([init_dN + n0, init_dN + p0, np.zeros(L + 1)])
    sol = solve_ivp(rhs, [0, Time], y0, args=args, t_eval=np.linspace(0, Time, T + 1), method="BDF", max_step=1,
                    rtol=1e-5, atol=1e-8)
    N, P = sol.y[:L], sol.y[L:2 * L]
    return simpson(B * (N * P - n0 * p0), dx=dx, axis=0)


def pvsim_cpu(plI, matPar, simPar, init_dN):
    """Same call form as the reference: fills plI (S, T+1) in place, returns seconds."""
    length, Time, L, T = simPar[0], simPar[1], simPar[2], simPar[3]
    t0 = time.perf_counter()
    for i, mp in enumerate(matPar):
        plI[i] = solve_one(mp, length, Time, L, T, init_dN)
    return time.perf_counter() - t0


def _job(a):
    mp, length, Time, L, T, ini = a
    pl = solve_one(mp, length, Time, L, T, ini)
    return float(np.sum((np.log10(np.abs(pl) + np.finfo(float).tiny)) ** 2))     # bayeslib.py:160-161,200


def cpu_branch_loglik(pl, obs, Time, T):
    """The likelihood bayeslib.simulate's CPU branch forms from one curve's PL rows (bayeslib.py:137,:158-161,:173-191,:198-201):
    PL staged in float32, log10(|PL| + MIN) in that dtype (MIN = DBL_MIN is 0 in float32), per-row griddata onto the
    observation times when they are not the whole simulation grid, then -sum (log10 PL - log10 obs)^2 in float64 -- no
    mag_offset on this branch.  pl (S, T+1); obs (n_obs,) log10 observations on the first n_obs grid points."""
    from scipy.interpolate import griddata
    pl32 = np.asarray(pl, dtype=np.float32)
    with np.errstate(divide="ignore"):
        lg = np.log10(np.abs(pl32) + np.float32(np.finfo(float).tiny))
    sim_t = np.linspace(0, Time, T + 1)
    times = sim_t[:len(obs)]
    if len(obs) != T + 1:                                                   # bayeslib.almost_equal fails on the shapes (:78-81)
        lg = np.stack([griddata(sim_t, row, times) for row in lg])
    return -np.sum((lg - np.asarray(obs)[None, :]) ** 2, axis=1)


def _pl_job(a):
    mp, length, Time, L, T, ini = a
    return solve_one(mp, length, Time, L, T, ini)


def pl_batch(X, ini, lengths, Time, L, T, processes):
    """PL(t) of len(X) x len(ini) systems on `processes` single-threaded workers: array (C, S, T+1)."""
    import multiprocessing as mp
    jobs = [(X[s], lengths[c], Time, L, T, ini[c]) for c in range(len(ini)) for s in range(len(X))]
    with mp.get_context("fork").Pool(processes, initializer=_one_blas_thread) as pool:
        out = pool.map(_pl_job, jobs, chunksize=1)
    return np.array(out).reshape(len(ini), len(X), T + 1)


_limiter = None


def _one_blas_thread():
    """Worker initialiser: one BLAS thread per process (BDF's dense LU would otherwise start a
    thread pool in every worker and oversubscribe the cores by orders of magnitude)."""
    global _limiter
    try:
        from threadpoolctl import threadpool_limits
        _limiter = threadpool_limits(limits=1)
    except ImportError:
        pass


def timed_batch(X, ini, lengths, Time, L, T, processes):
    """Solve len(X) x len(ini) systems on `processes` single-threaded worker processes; returns
    (seconds, n_systems).  Pool start-up is outside the timed region."""
    import multiprocessing as mp
    jobs = [(X[s], lengths[c], Time, L, T, ini[c]) for s in range(len(X)) for c in range(len(ini))]
    ctx = mp.get_context("fork")
    with ctx.Pool(processes, initializer=_one_blas_thread) as pool:
        pool.map(_job, jobs[:processes], chunksize=1)              # warm the workers (imports, LU setup)
        t0 = time.perf_counter()
        pool.map(_job, jobs, chunksize=1)
        sec = time.perf_counter() - t0
    return sec, len(jobs)
