"""Posterior core: what consumes the likelihood vector.

Host-side mirror of the numeric functions of the reference's Visualization/utils.py, same names and
argument meaning, every reduction over the samples running on the GPU through libtrpl_hip.so
(trpl_posterior_weights / _moments / _hist; csrc/posterior.hip).  No CPU fallback: without the library
or a device these raise TrplError / ImportError.

    normalize(lnP)                               utils.py:157-166
    temper(LL, num_observations, c)              marginalization_visual.py:589-591
    w_mean, w_variance, w_sample_var, w_skew,
    w_kurtosis, covariance                       utils.py:168-170, :197-227
    marginalize_1D, marginalize_2D               utils.py:239-285
    filter_nan                                   utils.py:33-38
    summarize(...)                               LikelihoodData.stats_summarize / calc_covariance, utils.py:117-143
"""
import numpy as np

from . import _abi


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def filter_nan(X, LL):
    """Drop the samples whose likelihood is NaN (LikelihoodData.filter_nan, utils.py:33-38)."""
    LL = np.asarray(LL)
    keep = ~np.isnan(LL)
    return np.asarray(X)[keep], LL[keep]


def weights(LL, tf=1.0, device=0, info=None):
    """normalize(LL / tf): posterior weights that sum to 1 (NaN stays NaN, -inf gives 0)."""
    LL = _f64(LL)
    if LL.ndim != 1:
        raise ValueError("LL must be one-dimensional")
    W = np.empty_like(LL)
    stats = np.zeros(2)
    sec = _abi.C.c_double(0.0)
    _abi.check(_abi.lib().trpl_posterior_weights(_abi.ptr(LL), LL.size, float(tf), _abi.ptr(W), _abi.ptr(stats),
                                                 int(device), _abi.C.byref(sec)))
    if info is not None:
        info.update(max=stats[0], raw_sum=stats[1], seconds=sec.value)
    return W


def normalize(lnP, device=0):
    """utils.py:157-166."""
    return weights(lnP, 1.0, device=device)


def temper(LL, num_observations, c, device=0):
    """P = normalize(LL / (num_observations * c)), marginalization_visual.py:589-591."""
    return weights(LL, float(num_observations) * float(c), device=device)


def moments(V, W, mean_in=None, device=0):
    """Raw weighted sums of the columns V (D, S) under weights W (S,):
    returns (sums[2+D] = [sum w, sum w^2, sum w v_d], central[D][D+2]) as trpl_posterior_moments defines them."""
    V = _f64(V)
    W = _f64(W)
    if V.ndim == 1:
        V = V[None, :]
    D, S = V.shape
    if W.shape != (S,):
        raise ValueError("W must have one weight per sample")
    sums = np.zeros(2 + D)
    central = np.zeros((D, D + 2))
    m = None if mean_in is None else _f64(mean_in)
    if m is not None and m.shape != (D,):
        raise ValueError("mean_in must have one entry per column")
    _abi.check(_abi.lib().trpl_posterior_moments(_abi.ptr(V), S, D, _abi.ptr(W), _abi.ptr(m), _abi.ptr(sums),
                                                 _abi.ptr(central), int(device), None))
    return sums, central


def w_mean(var, wts, device=0):
    """utils.py:197-199."""
    s, _ = moments(var, wts, device=device)
    return s[2] / s[0]


def w_variance(var, wts, device=0):
    """utils.py:202-204."""
    s, c = moments(var, wts, device=device)
    return c[0, 0] / s[0]


def w_sample_var(val, wts, ws, device=0):
    """utils.py:168-170 (the reference's name; it is a standard deviation): sqrt(ws * weighted variance)."""
    return np.sqrt(ws * w_variance(val, wts, device=device))


def w_skew(var, wts, device=0):
    """utils.py:207-210."""
    s, c = moments(var, wts, device=device)
    return (c[0, 1] / s[0]) / (c[0, 0] / s[0]) ** 1.5


def w_kurtosis(var, wts, device=0):
    """utils.py:212-215."""
    s, c = moments(var, wts, device=device)
    return (c[0, 2] / s[0]) / (c[0, 0] / s[0]) ** 2


def covariance(X, Y, weights, device=0):
    """utils.py:222-227."""
    s, c = moments(np.stack([_f64(X), _f64(Y)]), weights, device=device)
    return c[0, 1] / s[0]


def hist(x, W, lo, hi, bins, y=None, ylo=0.0, yhi=1.0, ybins=1, device=0):
    """Weighted counts (W None: plain counts) in `bins` equal bins of [lo, hi] (numpy.histogram's edge rules);
    with y, a (bins, ybins) array like numpy.histogram2d."""
    x = _f64(x)
    S = x.size
    yy = None if y is None else _f64(y)
    ww = None if W is None else _f64(W)
    if (yy is not None and yy.shape != x.shape) or (ww is not None and ww.shape != x.shape):
        raise ValueError("x, y and W must have the same length")
    out = np.zeros((int(bins), int(ybins)) if yy is not None else (int(bins),))
    _abi.check(_abi.lib().trpl_posterior_hist(_abi.ptr(x), _abi.ptr(yy), _abi.ptr(ww), S, float(lo), float(hi), int(bins),
                                              float(ylo), float(yhi), int(ybins), _abi.ptr(out), int(device), None))
    return out


def bin_edges(lo, hi, bins):
    """utils.py:243-244."""
    return lo + (hi - lo) * np.arange(int(bins) + 1) / int(bins)


def marginalize_1D(P, axis_overrides, bin_count, SECONDARY_PARAMS, param, X, device=0):
    """utils.py:239-262, same arguments: (density, edges); secondary parameters and mobilities are corrected
    for non-uniform sampling (each bin divided by its sample count, then renormalised to unit area)."""
    minX, maxX = axis_overrides[param]
    bins = int(bin_count)
    e = bin_edges(minX, maxX, bins)
    raw = hist(X, P, minX, maxX, bins, device=device)
    marP = raw / (np.diff(e) * raw.sum())                                  # numpy's density=True
    if SECONDARY_PARAMS[param] or "mu" in param:
        cnt = hist(X, None, minX, maxX, bins, device=device)
        corr = np.zeros_like(marP)
        nz = cnt != 0
        corr[nz] = marP[nz] / cnt[nz]
        marP = corr / np.sum(np.diff(e) * corr)
    return marP, e


def marginalize_2D(P, axis_overrides, bin_count, SECONDARY_PARAMS, param_names, X, Y, device=0):
    """utils.py:264-285, same arguments and return value (density[x bin][y bin], X_corr, Y_corr)."""
    px, py = param_names
    (minX, maxX), (minY, maxY) = axis_overrides[px], axis_overrides[py]
    bins = int(bin_count)
    ex, ey = bin_edges(minX, maxX, bins), bin_edges(minY, maxY, bins)
    raw = hist(X, P, minX, maxX, bins, y=Y, ylo=minY, yhi=maxY, ybins=bins, device=device)
    dens = raw / (np.outer(np.diff(ex), np.diff(ey)) * raw.sum())
    Y_corr, X_corr = np.meshgrid(ex, ey)                                   # :282, as the reference names them
    return dens, X_corr, Y_corr


def credible_interval(X, P):
    """utils.py:185-196 (host: one argsort, like the reference; trpl_amd.device.credible_interval_device
    does the same with torch.sort on the GPU)."""
    X = np.asarray(X)
    order = np.argsort(X)
    xs, cs = X[order], np.cumsum(np.asarray(P)[order])
    return xs[np.where(cs < 0.025)[0][-1]], xs[np.where(cs > 0.975)[0][0]]


def summarize(columns, P, device=0):
    """stats_summarize + calc_covariance (utils.py:117-143) in one device pass: `columns` maps a name to its
    (S,) values.  Returns dict(names, mean, variance, sample_std, skew, kurtosis, covariance (D, D), w2)."""
    names = list(columns)
    V = np.stack([_f64(columns[k]) for k in names])
    s, c = moments(V, P, device=device)
    D = len(names)
    mean = s[2:] / s[0]
    cov = c[:, :D] / s[0]
    var = np.diag(cov).copy()
    return {"names": names, "mean": mean, "variance": var, "sample_std": np.sqrt(s[1] * var),
            "skew": (c[:, D] / s[0]) / var ** 1.5, "kurtosis": (c[:, D + 1] / s[0]) / var ** 2,
            "covariance": cov, "w2": s[1]}
