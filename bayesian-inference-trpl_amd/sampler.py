"""Parameter-box sampler: restatement of bayeslib.random_grid / make_grid (bayeslib.py:18-76) for
the random-sample mode.  Host-side, microseconds of work; it defines the synthetic benchmark
inputs, so the draw order matches the reference exactly (column by column, fixed columns draw
nothing, legacy numpy global RNG unless an explicit RandomState is given)."""
import numpy as np

# parallel_bayes_gpu.py:24-33 -- parameter order and the conversion from common units
# (cm, s, V) to the solver's (nm, ns, V)
PARAM_NAMES = ("n0", "p0", "mun", "mup", "B", "Sf", "Sb", "CN", "CP", "taun", "taup", "lambda", "mag_offset")
_KT = .02569257
UNIT_CONVERSIONS = np.array([1e-21, 1e-21, 1e14 / 1e9 * _KT, 1e14 / 1e9 * _KT, 1e21 / 1e9, 1e7 / 1e9, 1e7 / 1e9,
                             1e42 / 1e9, 1e42 / 1e9, 1, 1, 704.3, 1])
# the reference's shipped search box (parallel_bayes_gpu.py:86-92), common units
DEFAULT_DO_LOG = np.array([1, 1, 0, 0, 1, 1, 1, 1, 1, 0, 0, 1, 0])
DEFAULT_MINX = np.array([1e8, 1e14, 0, 0, 1e-11, 0.1, 0.1, 1e-30, 1e-30, 1, 1, 0.1, 0])
DEFAULT_MAXX = np.array([1e8, 1e16, 50, 50, 1e-9, 100, 100, 1e-28, 1e-28, 1000, 2000, 0.1, 0])


def random_grid(minX, maxX, do_log, num_points, rng=None):
    """Uniform / log-uniform sample of the box (bayeslib.py:18-32)."""
    draw = (np.random if rng is None else rng).uniform
    cols = []
    for lo, hi, logscale in zip(minX, maxX, do_log):
        if lo == hi:
            cols.append(np.full(num_points, lo, dtype=float))
        elif logscale:
            cols.append(10 ** draw(np.log10(lo), np.log10(hi), (num_points,)))
        else:
            cols.append(draw(lo, hi, (num_points,)))
    return np.stack(cols, axis=1)


def make_grid(num_exp, minX, maxX, do_log, sim_flags, rng=None):
    """Random-sample branch of bayeslib.make_grid (bayeslib.py:34-76) with its three equality
    overrides.  Returns (N, P, X)."""
    if not sim_flags.get("random_sample", True):
        raise NotImplementedError("only random sampling is supported (the coarse-grid sampler is "
                                  "deprecated in the reference, bayeslib.py:46-63)")
    n = int(sim_flags["num_points"])
    X = random_grid(minX, maxX, do_log, n, rng=rng)
    if sim_flags.get("override_equal_mu"):
        X[:, 2] = X[:, 3]
    if sim_flags.get("override_equal_s"):
        X[:, 6] = X[:, 5]
    if sim_flags.get("override_equal_auger"):
        X[:, 8] = X[:, 7]
    return np.arange(n), np.zeros((num_exp, n)), X


def default_box(seed=42, num_points=4):
    """The shipped box in solver units and a seeded sample of it (parallel_bayes_gpu.py:35,:183-184)."""
    rng = np.random.RandomState(seed)
    return random_grid(DEFAULT_MINX * UNIT_CONVERSIONS, DEFAULT_MAXX * UNIT_CONVERSIONS, DEFAULT_DO_LOG,
                       num_points, rng=rng)


def random_grid_device(minX, maxX, do_log, num_points, seed=42, sim_flags=None, device=0):
    """random_grid (+ make_grid's overrides when sim_flags is given) drawn ON THE GPU by
    trpl_sample_box: the reference's MT19937 stream after numpy.random.seed(seed).  Linear columns are
    bit-identical to random_grid's, log-uniform columns agree to the last ulp or two of pow().  Returns
    the (num_points, len(minX)) array (host copy; trpl_amd.device.sample_box_device keeps it in HBM)."""
    from . import _abi
    lo = np.ascontiguousarray(minX, dtype=np.float64)
    hi = np.ascontiguousarray(maxX, dtype=np.float64)
    lg = np.ascontiguousarray(do_log, dtype=np.int32)
    if not (lo.shape == hi.shape == lg.shape and lo.ndim == 1):
        raise ValueError("minX, maxX and do_log must be one-dimensional and of equal length")
    X = np.empty((int(num_points), lo.size))
    _abi.check(_abi.lib().trpl_sample_box(int(seed) & 0xFFFFFFFF, int(num_points), lo.size, _abi.ptr(lo), _abi.ptr(hi),
                                          _abi.ptr(lg), box_flags(sim_flags), _abi.ptr(X), int(device), None))
    return X


def box_flags(sim_flags):
    """TRPL_BOX_* bits of make_grid's three overrides (bayeslib.py:36-38)."""
    f = sim_flags or {}
    return (1 if f.get("override_equal_mu") else 0) | (2 if f.get("override_equal_s") else 0) \
        | (4 if f.get("override_equal_auger") else 0)
