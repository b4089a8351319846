"""Oracle: a bounded parameter transformer and the orig-space density.  TEST INFRASTRUCTURE.

The reference's ``ParameterTransformer`` (/root/reference/pyvbmc/parameter_transformer/
parameter_transformer.py) is out of the accelerated path: ``VariationalPosterior.pdf`` only
calls three of its members.  The parity tests need SOME object with finite bounds to drive
``pdf(orig_flag=True)`` (variational_posterior.py:429-439, 543-559) on the GPU box, where the
reference cannot travel, so the logit transform is written out here from its formulas and
pinned against what the reference's class returned (tests/golden/variants.npz:
``pdfo_u``, ``pdfo_ladj``, ``pdfo_inv``, ``pt_mu``, ``pt_delta``):

    bounded dimension d:   z = (x - lb) / (ub - lb),  y = log(z / (1 - z)),  u = (y - mu) / delta
    unbounded dimension:   u = (x - mu) / delta
    log |det J| (u -> x):  sum over bounded d of  log(ub - lb) - y - 2 log(1 + exp(-y)) + log(delta)
                           + sum over unbounded d of  log(delta)

with (mu, delta) the midpoint and width of the transformed plausible box (mu = 0, delta = 1 when no
plausible bounds are given; parameter_transformer.py:116-133, :284-322, :476-522).
"""
import numpy as np

from . import mixture_ref


class BoundedLogit:
    """Duck type of the reference transformer for the members ``pdf`` / ``sample`` touch:
    ``lb_orig``, ``ub_orig``, ``__call__``, ``inverse``, ``log_abs_det_jacobian``."""

    def __init__(self, D, lb=None, ub=None, plb=None, pub=None):
        self.lb_orig = np.full((1, D), -np.inf) if lb is None else np.asarray(lb, dtype=np.float64).reshape(1, D)
        self.ub_orig = np.full((1, D), np.inf) if ub is None else np.asarray(ub, dtype=np.float64).reshape(1, D)
        self.bounded = (np.isfinite(self.lb_orig) & np.isfinite(self.ub_orig) & (self.lb_orig < self.ub_orig)).ravel()
        self.mu, self.delta = np.zeros(D), np.ones(D)
        if plb is not None and pub is not None:
            plb = np.asarray(plb, dtype=np.float64).reshape(1, D)
            pub = np.asarray(pub, dtype=np.float64).reshape(1, D)
            if not (np.array_equal(plb, self.lb_orig) and np.array_equal(pub, self.ub_orig)):
                lo, hi = self(plb)[0], self(pub)[0]
                ok = np.isfinite(lo) & np.isfinite(hi)
                self.mu[ok] = 0.5 * (lo[ok] + hi[ok])
                self.delta[ok] = hi[ok] - lo[ok]

    def _span(self):
        b = self.bounded
        return self.lb_orig[0, b], self.ub_orig[0, b]

    def __call__(self, x):
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        u = x.copy()
        b = self.bounded
        lo, hi = self._span()
        z = (x[:, b] - lo) / (hi - lo)
        with np.errstate(divide="ignore"):
            y = np.where(z == 0, -np.inf, np.where(z == 1, np.inf, np.log(z / (1 - z))))
        u[:, b] = y
        return (u - self.mu) / self.delta

    def inverse(self, u):
        u = np.atleast_2d(np.asarray(u, dtype=np.float64))
        y = u * self.delta + self.mu
        x = y.copy()
        b = self.bounded
        lo, hi = self._span()
        z = 1.0 / (1.0 + np.exp(-y[:, b]))
        xb = z * (hi - lo) + lo
        x[:, b] = np.minimum(np.maximum(xb, np.nextafter(lo, np.inf)), np.nextafter(hi, -np.inf))
        return x

    def log_abs_det_jacobian(self, u):
        u = np.atleast_2d(np.asarray(u, dtype=np.float64))
        y = u * self.delta + self.mu
        b = self.bounded
        lo, hi = self._span()
        yb = y[:, b]
        terms = np.log(hi - lo) - yb - 2.0 * np.log1p(np.exp(-yb)) + np.log(self.delta[b])
        return np.sum(terms, axis=1) + np.sum(np.log(self.delta[~b]))


def pdf_orig(mix, pt, x, log_flag=False, grad_flag=False, df=np.inf):
    """``VariationalPosterior.pdf(x, orig_flag=True, ...)`` restated (variational_posterior.py:425-564):
    strict-inequality bounds mask, density of the transformed points, 0 / -inf outside the
    bounds, Jacobian divided out (or subtracted in the log domain)."""
    x = np.array(np.atleast_2d(x), dtype=np.float64)
    n = x.shape[0]
    if grad_flag and log_flag:
        raise NotImplementedError("vbmc_pdf:NoOriginalGrad")
    mask = np.logical_and(np.all(x > pt.lb_orig, axis=1), np.all(x < pt.ub_orig, axis=1))
    xt = x.copy()
    xt[mask] = pt(x[mask])
    # rows outside the bounds stay in original coordinates and still go through the density: their
    # value is overwritten below, their GRADIENT rows are what the reference returns (:464-469)
    r = mixture_ref.pdf(mix, xt, log_flag=log_flag, grad_flag=grad_flag, df=df)
    y, dy = (r if grad_flag else (r, None))
    y = np.array(y, dtype=np.float64).reshape(n, 1)
    y[~mask] = -np.inf if log_flag else 0.0
    ladj = pt.log_abs_det_jacobian(xt[mask])[:, None]
    if log_flag:
        y[mask] -= ladj
    else:
        y[mask] /= np.exp(ladj)
    return (y, dy) if grad_flag else y
