"""The steps either side of the solver: the sampler (f-2; bayeslib.random_grid, csrc/sampler.hip -- the reference's MT19937
stream, bit-identical draws) and the posterior core (f-3; Visualization/utils.py, csrc/posterior.hip)."""
import numpy as np
import pytest


pytestmark = pytest.mark.gpu


# ---- posterior core (csrc/posterior.hip) against the reference's own outputs and the oracle ----
def test_posterior_core_matches_the_reference(trpl, gpu, golden):
    """trpl_amd.posterior (same names / arguments as Visualization/utils.py) on the GPU vs the golden made
    by the reference's functions: weights to 1e-13 relative, moments to 1e-11, histograms to 1e-10."""
    po = trpl.posterior
    g = golden("posterior")
    X, LL = po.filter_nan(g["X"], g["LL"])
    P = po.temper(LL, float(g["tf"]) / 2.0, 2.0)
    assert P.shape == g["P"].shape and np.allclose(P, g["P"], rtol=1e-13, atol=0) and abs(P.sum() - 1) < 1e-13
    assert (P[np.isinf(LL)] == 0).all()
    names = [str(n) for n in g["names"]]
    cols = {n: (np.log10(X[:, i]) if lg else X[:, i]) for n, i, lg in zip(names, g["col_index"], g["col_log"])}
    ws = float(np.sum(P ** 2))
    for k, n in enumerate(names):
        assert np.isclose(po.w_mean(cols[n], P), g["mean"][k], rtol=1e-12)
        assert np.isclose(po.w_variance(cols[n], P), g["var"][k], rtol=1e-10)
        assert np.isclose(po.w_sample_var(cols[n], P, ws), g["sstd"][k], rtol=1e-10)
        assert np.isclose(po.w_skew(cols[n], P), g["skew"][k], rtol=1e-9)
        assert np.isclose(po.w_kurtosis(cols[n], P), g["kurt"][k], rtol=1e-9)
        assert po.credible_interval(cols[n], g["P"]) == tuple(g["ci"][k])
    assert np.isclose(po.covariance(cols["p0"], cols["B"], P), g["cov"][0, 3], rtol=1e-9)
    summ = po.summarize(cols, P)
    assert np.allclose(summ["mean"], g["mean"], rtol=1e-12) and np.allclose(summ["covariance"], g["cov"], rtol=1e-9, atol=1e-18)
    assert np.allclose(summ["sample_std"], g["sstd"], rtol=1e-10) and np.isclose(summ["w2"], g["ws"], rtol=1e-12)
    limits = {n: tuple(g["limits"][k]) for k, n in enumerate(names)}
    secondary = {n: False for n in names}
    for k, n in enumerate(names):
        dens, e = po.marginalize_1D(P, limits, int(g["bins"]), secondary, n, cols[n])
        assert np.array_equal(e, g["edges"][k]) and np.allclose(dens, g["h1"][k], rtol=1e-10, atol=1e-14), n
    for (a, b), h in zip(g["pairs"], g["h2"]):
        dens, xc, yc = po.marginalize_2D(P, limits, int(g["bins"]), secondary, (names[a], names[b]), cols[names[a]], cols[names[b]])
        assert dens.shape == h.shape and xc.shape == yc.shape == (h.shape[1] + 1, h.shape[0] + 1)
        assert np.allclose(dens, h, rtol=1e-10, atol=1e-15)


def test_posterior_core_at_scale_and_edges(trpl, gpu, oracle):
    """1e6 samples (more than one pass of the fixed grid) against the CPU oracle; bin-edge rules on values
    that sit exactly on edges, outside the range and NaN; a 2-D histogram too large for the LDS bins."""
    from oracle import posterior as op
    po = trpl.posterior
    rng = np.random.default_rng(5)
    S = 1_000_003
    LL = -1e5 * rng.random(S) ** 2
    LL[::1000] = -np.inf
    V = np.stack([rng.normal(3.0, 2.0, S), rng.uniform(-1, 1, S), rng.lognormal(0, 1, S)])
    W = po.weights(LL, 4.0e3)
    Wo = op.weights(LL, 4.0e3)
    assert np.allclose(W, Wo, rtol=1e-12, atol=0)
    s, c = po.moments(V, W)
    assert abs(s[0] - 1) < 1e-12
    for d in range(3):
        assert np.isclose(s[2 + d] / s[0], op.w_mean(V[d], Wo), rtol=1e-11)
        assert np.isclose(c[d, d] / s[0], op.w_variance(V[d], Wo), rtol=1e-10)
        for e_ in range(3):
            assert np.isclose(c[d, e_] / s[0], op.covariance(V[d], V[e_], Wo), rtol=1e-8, atol=1e-12)
    # centring about given means (the sharded second pass)
    m = np.array([3.0, 0.0, 1.5])
    _, c2 = po.moments(V, W, mean_in=m)
    assert np.isclose(c2[0, 1], np.sum((V[0] - 3.0) * (V[1] - 0.0) * Wo), rtol=1e-8, atol=1e-12)
    # edges: numpy's own histogram on the reference's edge array is the checker here
    bins, lo, hi = 10, 0.1, 0.9
    e = po.bin_edges(lo, hi, bins)
    x = np.concatenate([e, e[:-1] + 1e-17, np.nextafter(e, -1), np.nextafter(e, 2), [np.nan, -1.0, 5.0], rng.uniform(0, 1, 5000)])
    w = rng.random(x.size)
    ok = ~np.isnan(x)
    want = np.histogram(x[ok], bins=e, weights=w[ok])[0]
    assert np.allclose(po.hist(x, w, lo, hi, bins), want, rtol=1e-12, atol=1e-15)
    assert np.array_equal(po.hist(x, None, lo, hi, bins), np.histogram(x[ok], bins=e)[0])
    xb = yb = 100                                                             # 10 000 bins: global-atomic path
    h2 = po.hist(V[1], W, -1, 1, xb, y=V[0], ylo=-3, yhi=9, ybins=yb)
    want2 = np.histogram2d(V[1], V[0], bins=[po.bin_edges(-1, 1, xb), po.bin_edges(-3, 9, yb)], weights=Wo)[0]
    assert np.allclose(h2, want2, rtol=1e-9, atol=1e-15)
    with pytest.raises(trpl.TrplError):
        po.hist(x, w, 1.0, 1.0, bins)
    assert po.weights(np.zeros(0)).shape == (0,)


# ---- device sampler (csrc/sampler.hip) against the reference's own draws ----
def test_device_sampler_draws_the_reference_stream(trpl, gpu, golden):
    """trpl_sample_box vs the reference's random_grid after numpy.random.seed(42) (sampler.npz holds its
    output): fixed and linear columns bit-identical, log-uniform columns to 2 ulp (device pow vs host pow);
    sizes that end inside / exactly on a 624-word block; the make_grid overrides."""
    sm = trpl.sampler
    lo, hi, lg = sm.DEFAULT_MINX * sm.UNIT_CONVERSIONS, sm.DEFAULT_MAXX * sm.UNIT_CONVERSIONS, sm.DEFAULT_DO_LOG
    g = golden("sampler")
    lin = np.array([not l for l in lg])
    for key in ("X4", "X64"):
        want = g[key]
        got = sm.random_grid_device(g["minX"] * g["unit"], g["maxX"] * g["unit"], g["do_log"], len(want), seed=42)
        assert np.array_equal(got[:, lin], want[:, lin]), key
        assert np.allclose(got[:, ~lin], want[:, ~lin], rtol=5e-16, atol=0), key
    for S in (1, 311, 312, 313, 624, 5000):                     # 312 doubles per regenerated block
        want = sm.default_box(42, S)
        got = sm.random_grid_device(lo, hi, lg, S, seed=42)
        assert np.array_equal(got[:, lin], want[:, lin]) and np.allclose(got, want, rtol=5e-16, atol=0), S
    want = sm.default_box(7, 1000)
    flags = {"override_equal_mu": True, "override_equal_s": True, "override_equal_auger": True}
    got = sm.random_grid_device(lo, hi, lg, 1000, seed=7, sim_flags=flags)
    assert np.array_equal(got[:, 2], want[:, 3]) and np.array_equal(got[:, 3], want[:, 3])
    assert np.allclose(got[:, 6], want[:, 5], rtol=5e-16) and np.allclose(got[:, 8], want[:, 7], rtol=5e-16)
    # the samples it produces drive the solver like the host-drawn ones
    import torch
    X = torch.empty((4096, 13), dtype=torch.float64, device="cuda")
    trpl.device.sample_box_device(X, lo, hi, lg, seed=42)
    assert np.allclose(X.cpu().numpy(), sm.default_box(42, 4096), rtol=5e-16, atol=0)
    with pytest.raises(trpl.TrplError):
        sm.random_grid_device(hi, lo, lg, 4)
