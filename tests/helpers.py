"""Shared builders: golden arrays -> oracle objects (test side only)."""
import numpy as np

from oracle import gp_ref, mixture_ref


def oracle_mix(g):
    return mixture_ref.Mixture.make(g["mu"], g["sigma"], g["lambd"], g["w"], g["eta"])


def oracle_gp(g, hyp=None):
    s2 = g["s2"] if g["s2"].size else None
    hyp = g["hyp"] if hyp is None else hyp
    return gp_ref.make_gp(g["X"], g["y"], hyp, gp_ref.MEAN_NEGQUAD, s2=s2, noise_user=s2 is not None)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)) / den)


class PlainTransformer:
    """Identity parameter transformer with the reference's member names."""

    def __init__(self, D):
        self.lb_orig = np.full((1, D), -np.inf)
        self.ub_orig = np.full((1, D), np.inf)

    def __call__(self, x):
        return x

    def inverse(self, u):
        return u

    def log_abs_det_jacobian(self, u):
        return np.zeros(np.atleast_2d(u).shape[0])


class PlainVP:
    """Attribute-only stand-in for the REFERENCE's VariationalPosterior: the public attributes
    of variational_posterior.py:106-138 with the reference's shapes and nothing else.
    ``__slots__`` makes any access to a member the reference class does not have (``_ctx``,
    ``_upload``, ``optimize_mask`` ...) an AttributeError, so a hot-path function that accepts
    a PlainVP relies on public attributes only.  ``set_parameters`` / ``get_parameters`` are the
    oracle's restatement."""

    __slots__ = ("D", "K", "mu", "sigma", "lambd", "w", "eta", "optimize_mu", "optimize_sigma",
                 "optimize_lambd", "optimize_weights", "parameter_transformer", "bounds", "stats")

    def __init__(self, g_or_mix):
        m = g_or_mix if isinstance(g_or_mix, mixture_ref.Mixture) else oracle_mix(g_or_mix)
        self.D, self.K = m.D, m.K
        plain_vp_take(self, m)
        self.optimize_mu = self.optimize_sigma = self.optimize_lambd = self.optimize_weights = True
        self.parameter_transformer = PlainTransformer(m.D)
        self.bounds = self.stats = None

    def set_parameters(self, theta, raw_flag=True):
        m = plain_vp_mix(self)
        mixture_ref.set_parameters(m, theta, raw_flag)
        eta = self.eta
        plain_vp_take(self, m)
        self.eta = eta  # set_parameters leaves eta alone (variational_posterior.py:680-759)

    def get_parameters(self, raw_flag=True):
        m = plain_vp_mix(self)
        th = mixture_ref.get_parameters(m, raw_flag)
        eta = self.eta
        plain_vp_take(self, m)
        self.eta = eta
        return th


def plain_vp_take(vp, m):
    vp.mu = m.mu.copy()
    vp.sigma = m.sigma.reshape(1, -1).copy()
    vp.lambd = m.lambd.reshape(-1, 1).copy()
    vp.w = m.w.reshape(1, -1).copy()
    vp.eta = m.eta.reshape(1, -1).copy()


def plain_vp_mix(vp):
    m = mixture_ref.Mixture.make(vp.mu, vp.sigma, vp.lambd, vp.w, vp.eta)
    m.optimize_mu, m.optimize_sigma = vp.optimize_mu, vp.optimize_sigma
    m.optimize_lambd, m.optimize_weights = vp.optimize_lambd, vp.optimize_weights
    return m


class PlainGP:
    """Attribute-only stand-in for ``gpyreg.GP`` on this path: X, y, posteriors (object array of
    records), mean (class name is what the path reads), temporary_data."""

    def __init__(self, ogp, mean_name="NegativeQuadratic"):
        from types import SimpleNamespace

        self.D, self.X, self.y, self.s2 = ogp.D, ogp.X.copy(), ogp.y.copy(), ogp.s2
        self.posteriors = np.empty(len(ogp.posteriors), dtype=object)
        for i, p in enumerate(ogp.posteriors):
            self.posteriors[i] = SimpleNamespace(hyp=p.hyp.copy(), alpha=p.alpha.copy(), sW=p.sW.copy(),
                                                 L=p.L.copy(), sn2_mult=p.sn2_mult, L_chol=p.L_chol)
        self.mean = type(mean_name, (), {})()
        self.temporary_data = {}


def entmc_extended(mix, NsK, eps_half, jacobian_flag=True):
    """The oracle's Monte-Carlo entropy formulas (entropy_ref.entmc_partial: the reference's loops,
    entmc_vbmc.py:64-112) evaluated in extended precision (np.longdouble: 64-bit mantissa on x86-64),
    finalised in float64.  For mixtures whose far components nearly underflow, the reference's own
    float64 arithmetic (lsum / q with q ~ 1e-300 relative weights) loses up to ~4e-5 on single gradient
    entries; the device's per-component-centred form does not (tests/test_gpu_multibatch.py)."""
    from types import SimpleNamespace

    from oracle import entropy_ref

    ld = np.longdouble
    m = SimpleNamespace(D=mix.D, K=mix.K, mu=mix.mu.astype(ld), sigma=mix.sigma.astype(ld), lambd=mix.lambd.astype(ld),
                        w=mix.w.astype(ld))
    p = entropy_ref.entmc_partial(m, np.asarray(eps_half, dtype=ld), NsK, (True,) * 4)
    p = {k: (float(v) if np.ndim(v) == 0 else np.asarray(v, dtype=np.float64)) for k, v in p.items()}
    return entropy_ref.entmc_finalize(mix, p, (True,) * 4, jacobian_flag)
