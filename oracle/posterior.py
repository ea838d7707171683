"""CPU restatement of the reference's posterior core (TEST INFRASTRUCTURE ONLY -- never imported by the
product package).  numpy restatements of Visualization/utils.py, each citing the lines it follows; pinned
to tests/golden/posterior.npz, which oracle/gen_golden.py produced by running the reference's own
functions (tests/test_oracle_golden.py).
"""
import numpy as np


def filter_nan(X, LL):
    """LikelihoodData.filter_nan, utils.py:33-38."""
    keep = ~np.isnan(LL)
    return X[keep], LL[keep]


def normalize(lnP):
    """utils.py:157-166: shift by the max, lift by 2^1000 / size, exponentiate, divide by the sum."""
    w = np.exp(lnP - np.nanmax(lnP) + 1000 * np.log(2) - np.log(lnP.size))
    return w / np.nansum(w)


def weights(LL, tf):
    """marginalization_visual.py:589-591."""
    return normalize(LL / tf)


def w_mean(v, w):
    """utils.py:197-199."""
    return np.sum(v * w) / np.sum(w)


def w_central(v, w, k):
    return np.sum((v - w_mean(v, w)) ** k * w) / np.sum(w)


def w_variance(v, w):
    """utils.py:202-204."""
    return w_central(v, w, 2)


def w_sample_std(v, w):
    """w_sample_var, utils.py:168-170: sqrt(sum(w^2) * weighted variance)."""
    return np.sqrt(np.sum(w ** 2) * w_variance(v, w))


def w_skew(v, w):
    """utils.py:207-210."""
    return w_central(v, w, 3) / w_variance(v, w) ** 1.5


def w_kurtosis(v, w):
    """utils.py:212-215."""
    return w_central(v, w, 4) / w_variance(v, w) ** 2


def covariance(x, y, w):
    """utils.py:222-227."""
    return np.sum((x - w_mean(x, w)) * (y - w_mean(y, w)) * w) / np.sum(w)


def credible_interval(x, w):
    """utils.py:185-196: 2.5 % / 97.5 % points of the weight-cumulated sorted samples."""
    order = np.argsort(x)
    xs, cs = x[order], np.cumsum(w[order])
    return xs[np.where(cs < 0.025)[0][-1]], xs[np.where(cs > 0.975)[0][0]]


def edges(lo, hi, bins):
    """utils.py:243-244."""
    return lo + (hi - lo) * np.arange(bins + 1) / bins


def bin_index(x, e):
    """numpy.histogram's rule for explicit edges: left-closed bins, the last one closed; -1 = dropped."""
    k = np.searchsorted(e, x, side="right") - 1
    k[x == e[-1]] = len(e) - 2
    k[(x < e[0]) | (x > e[-1]) | np.isnan(x)] = -1
    return k


def marginalize_1D(w, lo, hi, bins, x, correct_sampling=False):
    """utils.py:239-262: weighted density histogram; with correct_sampling each bin is divided by its
    sample count and the result renormalised to unit area."""
    e = edges(lo, hi, bins)
    k = bin_index(x, e)
    ok = k >= 0
    raw = np.bincount(k[ok], weights=w[ok], minlength=bins)
    dens = raw / (np.diff(e) * raw.sum())
    if correct_sampling:
        cnt = np.bincount(k[ok], minlength=bins)
        corr = np.where(cnt != 0, dens / np.where(cnt != 0, cnt, 1), 0.0)
        dens = corr / np.sum(np.diff(e) * corr)
    return dens, e


def marginalize_2D(w, xlim, ylim, bins, x, y):
    """utils.py:264-285: weighted 2-D density histogram, [x bin][y bin]."""
    ex, ey = edges(xlim[0], xlim[1], bins), edges(ylim[0], ylim[1], bins)
    kx, ky = bin_index(x, ex), bin_index(y, ey)
    ok = (kx >= 0) & (ky >= 0)
    raw = np.bincount(kx[ok] * bins + ky[ok], weights=w[ok], minlength=bins * bins).reshape(bins, bins)
    return raw / (np.outer(np.diff(ex), np.diff(ey)) * raw.sum())
