"""vp.sample, Monte-Carlo moments and kl_div.

tests/golden/vpmc.npz: the reference's own sample / moments / kl_div / kl_div_mvn under a
seeded NumPy stream (oracle/make_golden.py vpmc).  CPU: the mirror's NumPy-stream sampling
reproduces the reference draw for draw; kl_div_mvn.  GPU: the mirror's kl_div / moments with
the reference stream (density on the device) against the reference; the device generator
against the oracle's restatement of it (oracle/sample_ref.py), bit for bit on the labels and
to rounding on the samples; the device-side Monte-Carlo KL against the oracle on identical
draws; statistical checks of the device samples against the closed-form moments.
"""
import numpy as np
import pytest
from helpers import oracle_mix, rel_err

from oracle import mixture_ref, sample_ref
from pyvbmc_amd import synthetic

CASES = {"c1": (1, {}), "c2s": (2, dict(Ns_total=20 * 100))}


def workload(name):
    cfg, shrink = CASES[name]
    wl = synthetic.make_workload(cfg, S=1, **shrink)
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta)
    return wl, wd


def host_vp(wd, ctx=None):
    from pyvbmc_amd import VariationalPosterior

    vp = VariationalPosterior(int(wd["D"]), int(wd["K"]))
    vp.mu = wd["mu"].copy()
    vp.sigma = wd["sigma"].reshape(1, -1).copy()
    vp.lambd = wd["lambd"].reshape(-1, 1).copy()
    vp.w = wd["w"].reshape(1, -1).copy()
    vp.eta = wd["eta"].reshape(1, -1).copy()
    if ctx is not None:
        vp.ctx = ctx
    return vp


def second(g, name, wd):
    d2 = dict(wd)
    d2["mu"], d2["sigma"], d2["w"] = g[f"{name}_mu2"], g[f"{name}_sigma2"], g[f"{name}_w2"]
    return d2


# ---------------------------------------------------------------- CPU
@pytest.mark.parametrize("name", list(CASES))
def test_numpy_stream_sampling_matches_reference(golden, name):
    g = golden("vpmc")
    wl, wd = workload(name)
    for bal in (0, 1):
        vp = host_vp(wd)  # the constructor consumes np.random: build first, seed after
        np.random.seed(7)
        x, i = vp.sample(500, orig_flag=False, balance_flag=bool(bal))
        assert np.array_equal(i, g[f"{name}_sample_i_{bal}"])
        assert np.array_equal(x, g[f"{name}_sample_x_{bal}"])
    x, i = host_vp(wd).sample(0)
    assert x.shape == (0, wl.D) and i.shape == (0, 1)


@pytest.mark.parametrize("name", list(CASES))
def test_kl_div_mvn_matches_reference(golden, name):
    from pyvbmc_amd.variational_posterior import kl_div_mvn

    g = golden("vpmc")
    wl, wd = workload(name)
    vp, vp2 = host_vp(wd), host_vp(second(g, name, wd))
    m1, c1 = vp.moments(orig_flag=False, cov_flag=True)
    m2, c2 = vp2.moments(orig_flag=False, cov_flag=True)
    assert rel_err(kl_div_mvn(m1, c1, m2, c2), g[f"{name}_kl_mvn"]) < 1e-12
    assert np.all(np.isinf(kl_div_mvn(m1, np.zeros_like(c1), m2, c2)))


def test_oracle_sampler_balanced_counts():
    wl, wd = workload("c2s")
    mix = oracle_mix(wd)
    N = 4001
    x, lab = sample_ref.sample(mix, N, 11, balance_flag=True)
    cnt = np.bincount(lab, minlength=wl.K)
    assert np.all(cnt >= np.floor(wl.w * N)) and cnt.sum() == N
    assert np.all(np.diff(lab[: int(np.floor(wl.w * N).sum())]) >= 0)  # exact part grouped by component
    x2, lab2 = sample_ref.sample(mix, N, 11, balance_flag=False)
    assert not np.array_equal(lab, lab2) and np.array_equal(x.shape, x2.shape)


def test_kl_div_argument_errors():
    _, wd = workload("c1")
    vp = host_vp(wd)
    with pytest.raises(ValueError):
        vp.kl_div()
    with pytest.raises(ValueError):
        vp.kl_div(samples=np.zeros((5, 2)))  # not gaussianised and no vp2
    with pytest.raises(ValueError):
        vp.kl_div(vp2=vp, N=0, gauss_flag=True)


# ---------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def ctx():
    from pyvbmc_amd import _lib

    c = _lib.Context(0)
    _lib.set_default_context(c)
    yield c
    _lib.set_default_context(None)
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_reference_stream_kl_and_moments(ctx, golden, name):
    """NumPy-stream sampling on the host + densities on the device == the reference."""
    g = golden("vpmc")
    wl, wd = workload(name)
    vp, vp2 = host_vp(wd, ctx), host_vp(second(g, name, wd), ctx)
    np.random.seed(8)
    m, c = vp.moments(20000, orig_flag=True, cov_flag=True)
    assert np.array_equal(m, g[f"{name}_mom_mc_mean"]) and rel_err(c, g[f"{name}_mom_mc_cov"]) < 1e-13
    np.random.seed(9)
    assert rel_err(vp.kl_div(vp2, N=20000), g[f"{name}_kl_mc"]) < 1e-10
    np.random.seed(10)
    assert rel_err(vp.kl_div(vp2, N=20000, gauss_flag=True), g[f"{name}_kl_gauss"]) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("balance", [False, True])
def test_device_sampler_vs_oracle(ctx, name, balance):
    wl, wd = workload(name)
    vp, mix = host_vp(wd, ctx), oracle_mix(wd)
    N = 20003
    x, i = vp.sample(N, orig_flag=False, balance_flag=balance, rng="philox", seed=77, shuffle=False)
    xo, io = sample_ref.sample(mix, N, 77, balance_flag=balance)
    assert np.array_equal(i, io)
    # log / sqrt / sincos of the generator agree to a few ulp between device and NumPy
    assert np.max(np.abs(x - xo)) <= 1e-13 * max(1.0, np.max(np.abs(xo)))
    # reproducible per index: a shorter request is a prefix of the unbalanced stream
    if not balance:
        xs, _ = vp.sample(1000, orig_flag=False, rng="philox", seed=77)
        assert np.array_equal(xs, x[:1000])
    # the shuffled variant is a permutation of the same rows
    if balance:
        xp, ip = vp.sample(N, orig_flag=False, balance_flag=True, rng="philox", seed=77)
        assert np.array_equal(np.sort(xp[:, 0]), np.sort(x[:, 0])) and np.array_equal(np.bincount(ip), np.bincount(i))


def test_oracle_gamma_variates_are_gamma():
    """The restated device gamma stream (Marsaglia-Tsang on Philox): moments and a KS test against
    scipy's gamma, for a shape above and one below 1."""
    from scipy import stats

    for shape in (2.5, 0.4):
        g = sample_ref.gamma_variates(np.arange(200000, dtype=np.uint64), shape, 99)
        assert abs(g.mean() - shape) < 6 * np.sqrt(shape / g.size)
        assert abs(g.var() - shape) < 0.05 * shape
        assert stats.kstest(g[:20000], "gamma", args=(shape,)).pvalue > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name,df", [("c1", 3.0), ("c2s", 7.5), ("c2s", 0.8)])
def test_device_student_t_sampler_vs_oracle(ctx, name, df):
    """Heavy-tailed sampling (reference variational_posterior.py:329-353) on the device generator:
    identical component labels, samples to rounding against the oracle on the restated streams,
    and the radial law of a multivariate t."""
    wl, wd = workload(name)
    vp, mix = host_vp(wd, ctx), oracle_mix(wd)
    N = 20001
    for balance in (False, True):
        x, i = vp.sample(N, orig_flag=False, balance_flag=balance, df=df, rng="philox", seed=41, shuffle=False)
        xo, io = sample_ref.sample(mix, N, 41, balance_flag=balance, df=df)
        assert np.array_equal(i, io)
        scale = np.maximum(1.0, np.abs(xo))
        assert np.max(np.abs(x - xo) / scale) <= 1e-12
    # K = 1 takes the reference's other association (lam * t * z)
    one = dict(wd, K=1, mu=wd["mu"][:, :1], sigma=wd["sigma"][:1], w=np.ones(1), eta=np.zeros(1))
    v1, m1 = host_vp(one, ctx), oracle_mix(one)
    x, _ = v1.sample(5000, orig_flag=False, df=df, rng="philox", seed=5)
    xo, _ = sample_ref.sample(m1, 5000, 5, df=df)
    assert np.max(np.abs(x - xo) / np.maximum(1.0, np.abs(xo))) <= 1e-12
    # statistical: |(x - mu)/(lam sigma)|^2 / D ~ F(D, df)
    if df > 2:
        from scipy import stats

        r2 = np.sum(((x - m1.mu.T) / (m1.lambd.reshape(1, -1) * m1.sigma[0])) ** 2, axis=1) / wl.D
        assert stats.kstest(r2, "f", args=(wl.D, df)).pvalue > 1e-3
    with pytest.raises(ValueError):
        vp.sample(10, orig_flag=False, df=-3.0, rng="philox", seed=1)


@pytest.mark.gpu
def test_device_sampler_single_component_and_moments(ctx):
    wl, wd = workload("c2s")
    one = dict(wd, K=1, mu=wd["mu"][:, :1], sigma=wd["sigma"][:1], w=np.ones(1), eta=np.zeros(1))
    vp = host_vp(one, ctx)
    x, i = vp.sample(5000, orig_flag=False, rng="philox", seed=3)
    assert np.all(i == 0) and x.shape == (5000, wl.D)
    # statistical: Monte-Carlo moments from device samples vs the closed form, 6 standard errors
    vp = host_vp(wd, ctx)
    N = 400000
    m_mc, c_mc = vp.moments(N, orig_flag=True, cov_flag=True, rng="philox", seed=5)
    m, c = vp.moments(orig_flag=False, cov_flag=True)
    se = np.sqrt(np.diag(c) / N)
    assert np.all(np.abs(m_mc - m).ravel() < 6 * se)
    assert np.max(np.abs(c_mc - c)) < 6 * np.max(np.diag(c)) * np.sqrt(2.0 / N) * 3


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_device_kl_div_vs_oracle(ctx, golden, name):
    g = golden("vpmc")
    wl, wd = workload(name)
    vp, vp2 = host_vp(wd, ctx), host_vp(second(g, name, wd), ctx)
    mix1, mix2 = oracle_mix(wd), oracle_mix(second(g, name, wd))
    N = 30000
    kl = vp.kl_div(vp2, N=N, rng="philox", seed=1234)
    ref = sample_ref.kl_div_mc(mix1, mix2, N, 1234)
    assert rel_err(kl, ref) < 1e-10, (kl, ref)
    # and statistically the same quantity the reference estimated from its own stream
    assert np.all(np.abs(kl - g[f"{name}_kl_mc"]) < 0.15 * np.maximum(g[f"{name}_kl_mc"], 0.05))
    assert np.allclose(vp.kl_div(vp, N=2000, rng="philox", seed=1), 0.0)
    # the composed path (device samples, densities through pdf) agrees with the fused call
    class Other:  # a transformer the fused path cannot assume to cancel
        lb_orig = vp.parameter_transformer.lb_orig
        ub_orig = vp.parameter_transformer.ub_orig

        def __call__(self, x):
            return x

        def inverse(self, u):
            return u

        def log_abs_det_jacobian(self, u):
            return np.zeros(np.atleast_2d(u).shape[0])

    vp2.parameter_transformer = Other()
    assert rel_err(vp.kl_div(vp2, N=N, rng="philox", seed=1234), ref) < 1e-10
