"""Duck typing of the ``vp`` argument of the hot-path functions.

The reference hands its own ``VariationalPosterior`` to ``_neg_elcbo`` and friends
(/root/reference/pyvbmc/vbmc/vbmc.py:1172-1180, active_sample.py:554-561), so the
mirrors may only rely on what that class has: the public attributes ``D, K, mu (D,K),
sigma (1,K), lambd (D,1), w (1,K), eta (1,K), optimize_{mu,sigma,lambd,weights}`` and
``parameter_transformer`` (variational_posterior.py:106-138).  Everything the device
needs is derived from those here -- exactly what ``gp.upload_gp`` does for the GP.
"""
import numpy as np

from . import _lib


def ctx_of(obj, ctx=None):
    """The device context to use: the explicit one, the one a pyvbmc_amd object was
    given (``vp.ctx = ...``), or the process-wide default."""
    if ctx is not None:
        return ctx
    own = getattr(obj, "_ctx", None)
    return own if own is not None else _lib.default_context()


def upload_vp(vp, ctx):
    """Push the mixture attributes of ``vp`` to the device context (a no-op inside the
    library when they are the values the device already holds)."""
    D, K = int(vp.D), int(vp.K)
    ctx.set_mixture(np.asarray(vp.mu, dtype=np.float64).reshape(D, K), vp.sigma, vp.lambd, vp.w,
                    getattr(vp, "eta", None))
    return ctx


def optimize_mask(vp):
    return ((1 if vp.optimize_mu else 0) | (2 if vp.optimize_sigma else 0) | (4 if vp.optimize_lambd else 0)
            | (8 if vp.optimize_weights else 0))


def store_mixture(vp, mu_KD, sigma, lambd, w, eta=None):
    """Write a mixture back into ``vp`` with the reference's attribute shapes -- the side
    effect of ``vp.set_parameters(theta)`` (variational_posterior.py:680-759)."""
    # (float64 ndarrays in: .copy() is the cheapest way to detach them from the call's buffers --
    # this runs once per ELBO evaluation)
    vp.mu = mu_KD.T.copy()
    vp.sigma = sigma.reshape(1, -1).copy()
    vp.lambd = lambd.reshape(-1, 1).copy()
    vp.w = w.reshape(1, -1).copy()
    if eta is not None:
        vp.eta = eta.reshape(1, -1).copy()
    if hasattr(vp, "_mode"):
        vp._mode = None  # set_parameters drops the cached mode (:759)
