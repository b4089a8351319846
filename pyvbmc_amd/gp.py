"""GP container for the ELBO path: the subset of ``gpyreg.GP`` PyVBMC touches on it.

gpyreg is a third-party dependency of the reference that is not part of its tree
(`gpyreg >= 0.1.0`, /root/reference/pyproject.toml:13).  What PyVBMC needs from it
on this path is (a) the posterior record per hyper-parameter
sample -- ``posteriors[s].{hyp, alpha, sW, L, sn2_mult, L_chol}`` -- and (b)
``predict``.  This module provides a duck-type with the same attribute/method
names so the hot-path functions accept either it or a real ``gpyreg.GP``:

* ``update(X_new, y_new, s2_new, hyp)`` builds the posterior records on the host.
  That is the O(N^3) Cholesky done ONCE per GP fit -- an input of the path
  (SURVEY 8a row a11), not part of it; hyper-parameter fitting itself (slice
  sampling) is out of scope.
* ``predict`` runs on the MI355X (vbmc_gp_predict).
"""
import ctypes as C
import sys
from types import SimpleNamespace

import numpy as np
import scipy.linalg as sla

from . import _lib


class SquaredExponential:
    """SE-ARD: k(a,b) = sf^2 exp(-1/2 sum_d ((a_d-b_d)/ell_d)^2); hyp = [log ell (D), log sf]."""

    def hyperparameter_count(self, D):
        return D + 1

    def compute(self, hyp, X, X_star=None):
        X = np.atleast_2d(X)
        Xs = X if X_star is None else np.atleast_2d(X_star)
        D = X.shape[1]
        ell = np.exp(hyp[:D])
        d2 = np.zeros((X.shape[0], Xs.shape[0]))
        for d in range(D):
            d2 += ((X[:, d] / ell[d])[:, None] - (Xs[:, d] / ell[d])[None, :]) ** 2
        return np.exp(2 * hyp[D]) * np.exp(-0.5 * d2)


class ZeroMean:
    kind = _lib.MEAN_ZERO

    def hyperparameter_count(self, D):
        return 0

    def compute(self, hyp, X):
        return np.zeros(X.shape[0])


class ConstantMean:
    kind = _lib.MEAN_CONST

    def hyperparameter_count(self, D):
        return 1

    def compute(self, hyp, X):
        return np.full(X.shape[0], hyp[0])


class NegativeQuadratic:
    """m(x) = m0 - 1/2 sum_d ((x_d - xm_d)/omega_d)^2; hyp = [m0, xm (D), log omega (D)]
    (layout used at variational_optimization.py:1383-1392)."""

    kind = _lib.MEAN_NEGQUAD

    def hyperparameter_count(self, D):
        return 1 + 2 * D

    def compute(self, hyp, X):
        D = X.shape[1]
        return hyp[0] - 0.5 * np.sum(((X - hyp[1 : 1 + D]) / np.exp(hyp[1 + D : 1 + 2 * D])) ** 2, 1)


class GaussianNoise:
    def __init__(self, constant_add=False, user_provided_add=False, scale_user_provided=False,
                 rectified_linear_output_dependent_add=False):
        if scale_user_provided or rectified_linear_output_dependent_add:
            raise NotImplementedError("only constant and user-provided additive noise are supported")
        self.constant_add = constant_add
        self.user_provided_add = user_provided_add

    def hyperparameter_count(self):
        return 1

    def compute(self, hyp, X, y=None, s2=None):
        sn2 = np.full((X.shape[0], 1), np.exp(2 * hyp[0]) if self.constant_add else np.spacing(1.0))
        if self.user_provided_add and s2 is not None:
            sn2 = sn2 + np.reshape(s2, (-1, 1))
        return sn2


_MEAN_KINDS = {"ZeroMean": _lib.MEAN_ZERO, "ConstantMean": _lib.MEAN_CONST,
               "NegativeQuadratic": _lib.MEAN_NEGQUAD}


def mean_kind_of(gp):
    """Mean-function kind of a GP duck type (ours or a real gpyreg.GP), by class name."""
    name = type(gp.mean).__name__
    if name not in _MEAN_KINDS:
        raise NotImplementedError(f"mean function {name} is not supported on the ELBO path")
    return _MEAN_KINDS[name]


class _ChecksumPlan:
    """The data pointers of every posterior's ``alpha`` and ``hyp`` (float64, contiguous), laid out
    for ONE library call that checksums all of them (vbmc_host_checksum).  Built when the identities
    of those arrays change and reused while they stay the same -- the objects are held, so an
    unchanged ``id`` is the same array with the same buffer."""

    __slots__ = ("ids", "held", "ptrs", "lens", "n", "out", "slow", "fn")

    def __init__(self, ids, held, arrays):
        self.ids, self.held = ids, held
        fast = [a for a in arrays if isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"] and a.size > 0]
        self.slow = [a for a in arrays if not any(a is f for f in fast)]  # other dtypes / layouts: hashed through a copy
        self.n = len(fast)
        self.ptrs = (C.c_void_p * max(self.n, 1))(*[a.ctypes.data for a in fast])
        self.lens = (C.c_int64 * max(self.n, 1))(*[a.size for a in fast])
        self.out = C.c_uint64()
        self.fn = _lib.load().vbmc_host_checksum

    def checksum(self):
        self.fn(self.ptrs, self.lens, self.n, self.out)
        v = self.out.value
        for a in self.slow:
            v ^= hash(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        return v


def _gp_ids(gp):
    """The identity half of the GP key: ids of X, of the posteriors array, of every record and of its
    alpha / L / hyp arrays, and the end elements of X and L."""
    ps, X = gp.posteriors, gp.X
    ids = [id(ps), id(X), X.shape[0]]
    tail = [X.item(0), X.item(-1)]
    for p in ps:
        L = p.L
        ids += (id(p), id(p.alpha), id(L), id(p.hyp))
        tail += (L.item(-1), p.L_chol)
    return ids, tail


def _gp_plan(gp, ctx, ids):
    plan = getattr(ctx, "_gp_ck", None) if ctx is not None else None
    if plan is None or plan.ids != ids:
        ps = gp.posteriors
        held, arrays = [ps, gp.X], []
        for p in ps:
            held += (p, p.alpha, p.L, p.hyp)
            arrays += (p.alpha, p.hyp)
        plan = _ChecksumPlan(ids, held, arrays)
        if ctx is not None:
            ctx._gp_ck = plan
    return plan


def _gp_fingerprint(gp, ctx=None):
    """State key of a GP duck type: identities of X, of the posteriors array, of every posterior
    record and of its alpha / L / hyp arrays, the end elements of X and L, and a checksum of EVERY
    element of every alpha and hyp.  It catches what the reference does to a GP between two calls
    -- ``gp.posteriors[s] = ...`` (active_importance_sampling.py:207-209), attribute re-binding,
    ``gp.X`` replaced or extended by ``gp.update`` (active_sample.py:584) -- and any in-place edit
    of alpha or hyp (alpha = (K + Sigma)^-1 (y - m) changes whenever anything about the GP does).
    An in-place edit of X or of the interior of L that leaves alpha untouched still needs
    ``invalidate_gp``.  ~2 us at S = 1, ~5 us at S = 8, N = 800 (one library call checksums all
    the arrays: 55 KB there); the optimiser's inner loop does not pay it up front (``upload_gp``
    with ``lazy``)."""
    ids, tail = _gp_ids(gp)
    plan = _gp_plan(gp, ctx, ids)
    return ids + tail + [plan.checksum()], plan.held


def invalidate_gp(ctx=None):
    """Forget the GP the context holds: the next call uploads it again.  Only needed after
    editing GP arrays in place in a way the fingerprint cannot see (see ``_gp_fingerprint``)."""
    ctx = _lib.default_context() if ctx is None else ctx
    ctx._gp_key = ctx._gp_ref = ctx._gp_ck = ctx._gp_quick = None
    vo = sys.modules.get(__package__ + ".variational_optimization")
    if vo is not None:
        vo.clear_fast_path()


def upload_gp(gp, ctx, lazy=False):
    """Ship X and the posterior records of ``gp`` to the context (skipped while the GP's
    fingerprint is the one already uploaded).

    ``lazy`` (the fused objective only): when identities and end elements are unchanged the
    checksum of the arrays' contents is NOT taken here -- ``vbmc_neg_elcbo`` takes it itself after it
    has released its launches, while the device works (vbmc_set_gp_watch), and answers
    ``W_GP_CHANGED`` if an array was edited in place; the caller then uploads and evaluates again.
    The test stays complete, its cost moves off the path between two evaluations."""
    if lazy:
        # identities of what the watch covers (alpha, hyp of every record) and of the containers; L and the
        # end elements are part of the full key only -- L never changes without alpha changing
        ps, X = gp.posteriors, gp.X
        quick = [id(ps), id(X), X.shape[0]]
        for p in ps:
            quick += (id(p), id(p.alpha), id(p.hyp))
        if quick == getattr(ctx, "_gp_quick", None):
            return
    ids, tail = _gp_ids(gp)
    plan = _gp_plan(gp, ctx, ids)
    key = ids + tail + [plan.checksum()]
    if getattr(ctx, "_gp_key", None) == key:
        return
    ctx._gp_quick = None
    X = _lib.f64(gp.X)
    N, D = X.shape
    posts = list(gp.posteriors)
    S = len(posts)
    hyp = _lib.f64(np.stack([np.ravel(p.hyp) for p in posts]))
    alpha = _lib.f64(np.stack([np.ravel(p.alpha) for p in posts]))
    L = _lib.f64(np.stack([np.asarray(p.L) for p in posts]))
    sW = _lib.f64(np.stack([np.ravel(p.sW) * np.ones(N) for p in posts]))
    chol = np.ascontiguousarray([1 if p.L_chol else 0 for p in posts], dtype=np.int32)
    mult = _lib.f64([float(getattr(p, "sn2_mult", 1.0)) for p in posts])
    ctx._gp_key = ctx._gp_quick = None
    ctx.check(
        ctx._lib.vbmc_set_gp(
            ctx._h, N, D, S, hyp.shape[1], mean_kind_of(gp), _lib.ptr(X), _lib.ptr(hyp),
            _lib.ptr(alpha), _lib.ptr(L), chol.ctypes.data_as(C.POINTER(C.c_int32)), _lib.ptr(sW),
            _lib.ptr(mult),
        )
    )
    # the held references keep every fingerprinted object alive, so an id cannot be recycled (and the
    # watched buffers stay allocated)
    ctx._gp_key, ctx._gp_ref = key, plan.held
    if not plan.slow and plan.n > 0:
        ctx.check(ctx._lib.vbmc_set_gp_watch(ctx._h, plan.ptrs, plan.lens, plan.n, C.c_uint64(key[-1])))
        quick = [id(gp.posteriors), id(gp.X), gp.X.shape[0]]
        for p in gp.posteriors:
            quick += (id(p), id(p.alpha), id(p.hyp))
        ctx._gp_quick = quick  # (set only while the library watches the arrays' contents)


class GP:
    def __init__(self, D, covariance, mean, noise):
        if not isinstance(covariance, SquaredExponential):
            raise NotImplementedError("only the squared-exponential ARD kernel is on the ELBO path")
        self.D = D
        self.covariance = covariance
        self.mean = mean
        self.noise = noise
        self.X = None
        self.y = None
        self.s2 = None
        self.posteriors = None
        self.temporary_data = {}
        self._ctx = None

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        return self._ctx

    @ctx.setter
    def ctx(self, v):
        self._ctx = v

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_ctx"] = None
        return st

    def _posterior(self, hyp):
        """alpha, L, sW for one hyper-parameter vector (SURVEY Appendix A 'Posterior')."""
        X, y = self.X, self.y
        N, D = X.shape
        cov_n = self.covariance.hyperparameter_count(D)
        noise_n = self.noise.hyperparameter_count()
        Kxx = self.covariance.compute(hyp[:cov_n], X)
        m = self.mean.compute(hyp[cov_n + noise_n :], X)
        sn2 = self.noise.compute(hyp[cov_n : cov_n + noise_n], X, y, self.s2).ravel()
        sn2_div = np.min(sn2)
        sn2_mult = 1.0
        resid = y.ravel() - m
        if sn2_div * sn2_mult >= 1e-6:
            sl = sn2_div * sn2_mult
            L = sla.cholesky(Kxx / sl + np.diag(sn2 / sn2_div), lower=False)
            alpha = sla.cho_solve((L, False), resid) / sl
            L_chol = True
        else:
            L_chol = False
            L = -np.linalg.inv(Kxx + sn2_mult * np.diag(sn2))
            alpha = -L @ resid
        return SimpleNamespace(
            hyp=np.array(hyp, dtype=np.float64), alpha=alpha.reshape(-1, 1),
            sW=np.ones(N) / np.sqrt(sn2_div * sn2_mult), L=L, sn2_mult=sn2_mult, L_chol=L_chol,
        )

    def update(self, X_new=None, y_new=None, s2_new=None, hyp=None, compute_posterior=True):
        if X_new is not None:
            self.X = np.atleast_2d(np.asarray(X_new, dtype=np.float64))
            self.y = np.asarray(y_new, dtype=np.float64).reshape(-1, 1)
            self.s2 = None if s2_new is None else np.asarray(s2_new, dtype=np.float64).reshape(-1, 1)
        if hyp is None:
            hyp = np.stack([p.hyp for p in self.posteriors])
        hyp = np.atleast_2d(np.asarray(hyp, dtype=np.float64))
        posts = np.empty(hyp.shape[0], dtype=object)
        for i, h in enumerate(hyp):
            posts[i] = self._posterior(h)
        self.posteriors = posts

    def get_hyperparameters(self, as_array=True):
        return np.stack([p.hyp for p in self.posteriors])

    def predict(self, x_star, y_star=None, s2_star=0, add_noise=False, separate_samples=False, *,
                ctx=None):
        """Posterior predictive mean/variance at ``x_star`` (M, D) on the MI355X.
        Returns (M, S) arrays when ``separate_samples`` else (M, 1)."""
        if self.noise.user_provided_add and add_noise and np.any(np.asarray(s2_star) != 0):
            raise NotImplementedError("add_noise with user-provided s2_star is not supported")
        ctx = self.ctx if ctx is None else ctx
        upload_gp(self, ctx)
        xs = _lib.f64(np.atleast_2d(x_star))
        M = xs.shape[0]
        S = len(self.posteriors)
        shape = (M, S) if separate_samples else (M, 1)
        fmu, fs2 = np.empty(shape), np.empty(shape)
        ctx.check(
            ctx._lib.vbmc_gp_predict(
                ctx._h, M, _lib.ptr(xs), int(bool(add_noise)), int(bool(separate_samples)),
                _lib.ptr(fmu), _lib.ptr(fs2),
            )
        )
        return fmu, fs2
