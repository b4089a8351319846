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


class _VpArgs:
    """The ctypes argument tuple of vbmc_set_mixture_dk for one set of attribute ARRAYS (held, so their ids stay
    theirs): built when an attribute is rebound, reused while the same arrays are edited in place or left alone --
    the library compares their contents with what the device holds."""

    __slots__ = ("ids", "held", "args", "D", "K")


def _f64c(a):
    return isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]


def upload_vp(vp, ctx):
    """Push the mixture attributes of ``vp`` to the device context (a no-op inside the
    library when they are the values the device already holds)."""
    mu, sg, lm, w = vp.mu, vp.sigma, vp.lambd, vp.w
    eta = getattr(vp, "eta", None)
    ids = (id(mu), id(sg), id(lm), id(w), id(eta))
    st = ctx.__dict__.get("_vp_args")
    if st is None or st.ids != ids or st.D != vp.D or st.K != vp.K:
        D, K = int(vp.D), int(vp.K)
        if not (_f64c(mu) and _f64c(sg) and _f64c(lm) and _f64c(w) and mu.shape == (D, K) and sg.size == K
                and lm.size == D and w.size == K and (eta is None or (_f64c(eta) and eta.size == K))):
            # anything else (lists, other dtypes, views): through NumPy conversions, every call
            ctx.__dict__["_vp_args"] = None
            ctx.set_mixture(np.asarray(mu, dtype=np.float64).reshape(D, K), sg, lm, w, eta)
            return ctx
        st = _VpArgs()
        st.ids, st.held, st.D, st.K = ids, (mu, sg, lm, w, eta), D, K
        st.args = (ctx._h, D, K, _lib.ptr(mu), _lib.ptr(sg), _lib.ptr(lm), _lib.ptr(w), _lib.ptr(eta))
        ctx.__dict__["_vp_args"] = st
    ctx.check(ctx._lib.vbmc_set_mixture_dk(*st.args))
    ctx.D, ctx.K = st.D, st.K
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
