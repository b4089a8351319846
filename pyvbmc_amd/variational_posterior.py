"""``VariationalPosterior`` -- the reference's Gaussian-mixture class, hot-path subset,
with the density evaluated on the MI355X.

Mirrors /root/reference/pyvbmc/variational_posterior/variational_posterior.py
(class at :24): same constructor, attribute names/shapes (:106-138) and the
methods on the ELBO path -- ``get_bounds`` (:140-239), ``sample`` (:241-363),
``pdf`` (:365-564), ``log_pdf`` (:566-621), ``get_parameters`` (:623-678),
``set_parameters`` (:680-759), ``moments`` (:761-808) -- with the same mutation
side effects and exceptions -- plus ``kl_div`` (:1032-1127), the Monte-Carlo consumer
SURVEY.md 8f row 4 names.  ``mode``, ``mtv``, ``plot`` are host-side analysis outside
the path and are not provided.

Where the arithmetic runs: ``pdf``/``log_pdf`` -> HIP kernel (vbmc_mixture_pdf).
State bookkeeping (get/set_parameters, bounds) and the closed-form K*D^2 moments are
plain NumPy object state, as in the reference.  ``sample`` draws from NumPy's global
stream exactly like the reference by default; ``rng="philox"`` (keyword-only, or env
``VBMC_HIP_RNG=philox``) generates the samples on the device instead
(vbmc_mixture_sample), which is what ``moments(orig_flag=True)`` and ``kl_div`` then use;
with ``rng="philox"`` ``kl_div`` runs entirely on the device (vbmc_kl_div_mc).
"""
import ctypes as C
import os
import sys

import numpy as np

from . import _lib
from ._duck import upload_vp
from .entropy import _HOST_RANDN_MIN, host_randn


def _randn(n, d):
    """``np.random.randn(n, d)`` -- the same values and generator state; large requests through the
    library's multi-threaded restatement of NumPy's stream (csrc/host_randn.hip)."""
    if n * d >= _HOST_RANDN_MIN:
        flat = host_randn(n * d)
        if flat is not None:
            return flat.reshape(n, d)
    return np.random.randn(n, d)


class IdentityTransformer:
    """Unbounded-space parameter transformer (the reference's default
    ``ParameterTransformer(D)`` with infinite bounds is the identity map).  Any
    object with the same five members may be passed instead, e.g. the
    reference's own ``ParameterTransformer``."""

    def __init__(self, D):
        self.lb_orig = np.full((1, D), -np.inf)
        self.ub_orig = np.full((1, D), np.inf)

    def __call__(self, x):
        return x

    def inverse(self, u):
        return u

    def log_abs_det_jacobian(self, u):
        return np.zeros(np.atleast_2d(u).shape[0])


class VariationalPosterior:
    def __init__(self, D, K=2, x0=None, parameter_transformer=None):
        self.D = D
        self.K = K
        if x0 is None:
            x0 = np.zeros((D, K))
        elif x0.size == D:
            x0 = np.tile(x0.reshape(-1), (K, 1)).T
        else:
            x0 = x0.T
            x0 = np.tile(x0, int(np.ceil(K / x0.shape[1])))[:, :K]
        self.w = np.ones((1, K)) / K
        self.eta = np.ones((1, K)) / K
        # the reference perturbs the means and consumes the global RNG here (:121)
        self.mu = x0 + 1e-6 * np.random.randn(D, K)
        self.sigma = 1e-3 * np.ones((1, K))
        self.lambd = np.ones((D, 1))
        self.optimize_weights = True
        self.optimize_mu = True
        self.optimize_sigma = True
        self.optimize_lambd = True
        self.parameter_transformer = (
            IdentityTransformer(D) if parameter_transformer is None else parameter_transformer
        )
        self.bounds = None
        self.stats = None
        self._mode = None
        self._ctx = None

    # -- device plumbing (never pickled) -----------------------------------------
    def __getstate__(self):
        st = self.__dict__.copy()
        st["_ctx"] = None
        return st

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        return self._ctx

    @ctx.setter
    def ctx(self, value):
        self._ctx = value

    def _upload(self, ctx=None):
        """Push the current attributes to the device context (pyvbmc_amd._duck.upload_vp;
        the hot-path functions call that helper directly, so they accept any object with
        the reference's public attributes, not just this class)."""
        return upload_vp(self, self.ctx if ctx is None else ctx)

    # -- bounds (:140-239) ------------------------------------------------------------
    def get_bounds(self, X, options, K=None):
        if K is None:
            K = self.K
        D = self.D
        if self.bounds is None:
            self.bounds = {
                "mu_lb": np.full((D,), np.inf),
                "mu_ub": np.full((D,), -np.inf),
                "lnscale_lb": np.full((D,), np.inf),
                "lnscale_ub": np.full((D,), -np.inf),
            }
        xmin, xmax = np.min(X, axis=0), np.max(X, axis=0)
        b = self.bounds
        b["mu_lb"] = np.minimum(xmin, b["mu_lb"])
        b["mu_ub"] = np.maximum(xmax, b["mu_ub"])
        ln_range = np.log(xmax - xmin)
        b["lnscale_lb"] = np.minimum(b["lnscale_lb"], ln_range + np.log(options["tol_length"]))
        b["lnscale_ub"] = np.maximum(b["lnscale_ub"], ln_range)
        if self.optimize_weights:
            b["eta_lb"] = -np.inf if options["tol_weight"] == 0 else np.log(0.5 * options["tol_weight"])
            b["eta_ub"] = 0
        lo, hi = [], []
        if self.optimize_mu:
            lo.append(np.tile(b["mu_lb"], (K,)))
            hi.append(np.tile(b["mu_ub"], (K,)))
        if self.optimize_sigma or self.optimize_lambd:
            lo.append(np.tile(b["lnscale_lb"], (K,)))
            hi.append(np.tile(b["lnscale_ub"], (K,)))
        if self.optimize_weights:
            lo.append(np.tile(b["eta_lb"], (K,)))
            hi.append(np.tile(b["eta_ub"], (K,)))
        theta_bnd = {"lb": np.concatenate(lo), "ub": np.concatenate(hi)}
        theta_bnd["tol_con"] = options["tol_con_loss"]
        if self.optimize_weights:
            theta_bnd["weight_threshold"] = max(1 / (4 * K), options["tol_weight"])
            theta_bnd["weight_penalty"] = options["weight_penalty"]
        return theta_bnd

    # -- sampling (:241-363): RNG-bound, consumes np.random in the reference's order ----
    def sample(self, N, orig_flag=True, balance_flag=False, df=np.inf, *, rng=None, seed=None,
               shuffle=True):
        """Reference signature; keyword-only extras: ``rng`` ("numpy": the reference's global
        MT19937 stream, the default; "philox": the device generator), ``seed`` of the device
        generator (drawn from ``np.random`` when omitted) and ``shuffle`` (the device returns
        balanced samples grouped by component; ``True`` permutes them like the reference)."""
        if N < 1:
            return np.zeros((0, self.D)), np.zeros((0, 1))
        mode = os.environ.get("VBMC_HIP_RNG", "numpy") if rng is None else rng
        if mode == "philox" and not (np.isfinite(df) and df < 0):  # (df < 0: numpy's gamma raises, below)
            N = int(N)
            ctx = self._upload()
            if seed is None:
                seed = int(np.random.randint(0, 2**62, dtype=np.int64))
            x = np.empty((N, self.D))
            i = np.empty(N, dtype=np.int32)
            tdf = float(df) if (np.isfinite(df) and df != 0) else float("inf")
            ctx.check(ctx._lib.vbmc_mixture_sample_t(ctx._h, N, int(seed), int(bool(balance_flag)), tdf, _lib.ptr(x),
                                                     i.ctypes.data_as(C.POINTER(C.c_int32))))
            if balance_flag and shuffle and self.K > 1:
                perm = np.random.permutation(N)
                x, i = x[perm], i[perm]
            if orig_flag:
                x = self.parameter_transformer.inverse(x)
            return x, (i.astype(np.int64) if self.K > 1 else np.zeros(N))
        if mode not in ("numpy", "philox"):
            raise ValueError(f"unknown rng {mode!r}")
        lam = self.lambd.reshape(1, -1)
        heavy = np.isfinite(df) and df != 0
        if self.K > 1:
            if balance_flag:
                reps = np.floor(self.w * N).astype("int")
                i = np.repeat(range(self.K), reps.ravel())
                if N > i.shape[0]:
                    w_extra = self.w * N - reps
                    n_extra = np.ceil(np.sum(w_extra))
                    w_extra += self.w * (n_extra - sum(w_extra))
                    w_extra /= np.sum(w_extra)
                    i = np.append(
                        i, np.random.choice(range(self.K), size=n_extra.astype("int"), p=w_extra.ravel())
                    )
                np.random.shuffle(i)
                i = i[:N]
            else:
                i = np.random.choice(range(self.K), size=N, p=self.w.ravel())
            if heavy:
                t = df / 2 / np.sqrt(np.random.gamma(df / 2, df / 2, (N, 1)))
                x = self.mu.T[i] + lam * _randn(N, self.D) * t * self.sigma[:, i].T
            else:
                x = self.mu.T[i] + lam * _randn(N, self.D) * self.sigma[:, i].T
        else:
            if heavy:
                t = df / 2 / np.sqrt(np.random.gamma(df / 2, df / 2, (N, 1)))
                x = self.mu.T + lam * t * _randn(N, self.D) * self.sigma
            else:
                x = self.mu.T + lam * _randn(N, self.D) * self.sigma
            i = np.zeros(N)
        if orig_flag:
            x = self.parameter_transformer.inverse(x)
        return x, i

    # -- density (:365-621) ---------------------------------------------------------------
    def pdf(self, x, orig_flag=True, log_flag=False, grad_flag=False, df=np.inf):
        # 0-D / 1-D inputs are lifted to 2-D and 1-D results raveled, like the
        # reference's handle_0D_1D_input decorator (decorators/handle_0D_1D_input.py:46-58)
        in_dims = np.ndim(x)
        x = np.array(np.atleast_2d(x), dtype=np.float64)  # copy (:425)
        n, D = x.shape
        finite_df = np.isfinite(df) and df != 0
        if grad_flag and finite_df:
            raise NotImplementedError("Gradient of heavy-tailed pdf not supported yet.")
        if grad_flag and orig_flag and log_flag:
            raise NotImplementedError(
                "vbmc_pdf:NoOriginalGrad: Gradient computation in original space not supported yet."
            )
        if orig_flag:
            pt = self.parameter_transformer
            mask = np.logical_and(np.all(x > pt.lb_orig, axis=1), np.all(x < pt.ub_orig, axis=1))
            x[mask] = pt(x[mask])
        else:
            mask = np.full(n, True)
        ctx = self._upload()
        y = np.empty(n)
        dy = np.empty((n, D)) if grad_flag else None
        xin = np.ascontiguousarray(x)
        if not np.all(mask) and not np.all(np.isfinite(xin)):
            # rows outside the bounds stay in original coordinates and go through the density like
            # the reference's (their value is overwritten below, their gradient rows are returned
            # as computed, :464-469); only non-finite coordinates are kept off the device
            xin = np.where(np.isfinite(xin), xin, 0.0)
        ctx.check(
            ctx._lib.vbmc_mixture_pdf(
                ctx._h, n, _lib.ptr(xin), int(bool(log_flag)), int(bool(grad_flag)), float(df),
                _lib.ptr(y), _lib.ptr(dy),
            )
        )
        y = y.reshape(n, 1)
        if log_flag:
            y[~mask] = -np.inf
        else:
            y[~mask] = 0
        if orig_flag:
            ladj = self.parameter_transformer.log_abs_det_jacobian(x[mask])[:, np.newaxis]
            if log_flag:
                y[mask] -= ladj
            else:
                y[mask] /= np.exp(ladj)
        out = (y, dy) if grad_flag else y
        if in_dims == 1:
            return tuple(o.ravel() for o in out) if grad_flag else out.ravel()
        return out

    def log_pdf(self, *args, **kwargs):
        return self.pdf(*args, **kwargs, log_flag=True)

    # -- parameter vector (:623-759) ----------------------------------------------------------
    def _renormalise(self):
        nl = np.sqrt(np.sum(self.lambd**2) / self.D)
        self.lambd = self.lambd.reshape(-1, 1) / nl
        self.sigma = self.sigma.reshape(1, -1) * nl
        if self.optimize_weights:
            self.w = self.w.reshape(1, -1) / np.sum(self.w)

    def get_parameters(self, raw_flag=True):
        self._renormalise()
        theta = self.mu.ravel(order="F") if self.optimize_mu else np.array([])
        tail = [np.array([])]
        if self.optimize_sigma:
            tail.append(self.sigma.ravel())
        if self.optimize_lambd:
            tail.append(self.lambd.ravel())
        if self.optimize_weights:
            tail.append(self.w.ravel())
        tail = np.concatenate(tail)
        return np.concatenate((theta, np.log(tail) if raw_flag else tail))

    def set_parameters(self, theta, raw_flag=True):
        theta = np.array(theta, dtype=np.float64)
        D, K = self.D, self.K
        if not raw_flag:
            n_con = K * self.optimize_weights + D * self.optimize_lambd + K * self.optimize_sigma
            # same slice the reference checks (theta[-check_idx:] with check_idx = -n_con, :701-710)
            if np.any(theta[n_con:] < 0.0):
                raise ValueError("sigma, lambda and weights must be positive when raw_flag = False")
        pos = 0
        if self.optimize_mu:
            self.mu = np.reshape(theta[: D * K], (D, K), order="F")
            pos = D * K
        if self.optimize_sigma:
            s = theta[pos : pos + K]
            self.sigma = np.exp(s) if raw_flag else s
            pos += K
        if self.optimize_lambd:
            l = theta[pos : pos + D]
            self.lambd = np.exp(l) if raw_flag else l
        if self.optimize_weights:
            eta = theta[-K:]
            self.w = (np.exp(eta - np.amax(eta)) if raw_flag else eta)[np.newaxis, :]
        self._renormalise()
        self._mode = None

    # -- moments (:761-808) ----------------------------------------------------------------------
    def moments(self, N=int(1e6), orig_flag=True, cov_flag=False, *, rng=None, seed=None):
        if orig_flag:
            # mean / covariance do not depend on the order: skip the shuffle on the device path
            x, _ = self.sample(int(N), orig_flag=True, balance_flag=True, rng=rng, seed=seed, shuffle=False)
            mubar = np.mean(x, axis=0)
            if cov_flag:
                cov = np.cov(x.T)
        else:
            mubar = np.sum(self.w * self.mu, axis=1)
            if cov_flag:
                cov = np.sum(self.w * self.sigma**2) * np.eye(len(self.lambd)) * self.lambd**2
                dev = self.mu - mubar[:, np.newaxis]
                cov = cov + (self.w * dev) @ dev.T
        return (mubar.reshape(1, -1), cov) if cov_flag else mubar.reshape(1, -1)

    # -- Kullback-Leibler divergence (:1032-1127) ------------------------------------------------
    def kl_div(self, vp2=None, samples=None, N=int(1e5), gauss_flag=False, *, rng=None, seed=None):
        """Forward and reverse KL divergence between two posteriors, reference signature.
        With ``rng="philox"`` (or ``VBMC_HIP_RNG=philox``) and both posteriors sharing one
        parameter transformer the Monte-Carlo branch runs in one device call."""
        if samples is None and vp2 is None:
            raise ValueError("Either vp2 or samples have to be not None")
        if not gauss_flag and vp2 is None:
            raise ValueError("Unless the KL divergence is gaussianized, VP2 is required.")
        mode = os.environ.get("VBMC_HIP_RNG", "numpy") if rng is None else rng
        if gauss_flag:
            if N == 0:
                raise ValueError("Analytical moments are available only for the transformed space.")
            q1mu, q1sigma = self.moments(N, True, True, rng=rng, seed=seed)
            if vp2 is not None:
                q2mu, q2sigma = vp2.moments(N, True, True, rng=rng, seed=None if seed is None else seed + 1)
            else:
                q2mu = np.mean(samples)  # sic (:1103)
                q2sigma = np.cov(samples.T)
            kls = kl_div_mvn(q1mu, q1sigma, q2mu, q2sigma)
        elif mode == "philox" and _same_transformer(self, vp2) and vp2.D == self.D:
            ctx = self._upload()
            if seed is None:
                seed = int(np.random.randint(0, 2**62, dtype=np.int64))
            mu2 = _lib.f64(np.asarray(vp2.mu, dtype=np.float64).reshape(vp2.D, vp2.K).T)
            sg2, lm2, w2 = _lib.f64(np.ravel(vp2.sigma)), _lib.f64(np.ravel(vp2.lambd)), _lib.f64(np.ravel(vp2.w))
            kls = np.empty(2)
            ctx.check(ctx._lib.vbmc_kl_div_mc(ctx._h, int(N), int(seed), vp2.K, _lib.ptr(mu2), _lib.ptr(sg2),
                                              _lib.ptr(lm2), _lib.ptr(w2), _lib.ptr(kls)))
        else:
            minp = sys.float_info.min
            xx1, _ = self.sample(N, True, True, rng=rng, seed=seed, shuffle=False)
            q1 = self.pdf(xx1, True)
            q2 = vp2.pdf(xx1, True)
            # the reference writes `q == 0 | np.isinf(q)`, which Python parses as
            # q == (0 | isinf(q)): true exactly where q == 0 (:1113-1114)
            q1[q1 == 0] = 1.0
            q2[q2 == 0] = minp
            kl1 = -np.mean(np.log(q2) - np.log(q1))
            xx2, _ = vp2.sample(N, True, True, rng=rng, seed=None if seed is None else seed + 1, shuffle=False)
            q1 = self.pdf(xx2, True)
            q2 = vp2.pdf(xx2, True)
            q1[q1 == 0] = minp
            q2[q2 == 0] = 1.0
            kl2 = -np.mean(np.log(q1) - np.log(q2))
            kls = np.concatenate((kl1, kl2), axis=None)
        return np.maximum(0, kls)  # correct for numerical errors (:1126)


def _same_transformer(a, b):
    ta, tb = a.parameter_transformer, b.parameter_transformer
    return ta is tb or (isinstance(ta, IdentityTransformer) and isinstance(tb, IdentityTransformer))


def kl_div_mvn(mu1, sigma1, mu2, sigma2):
    """Analytical KL divergences between two multivariate normals, both directions
    (reference pyvbmc/stats/kl_div_mvn.py:10-47)."""
    D = mu1.size
    dmu = mu2.reshape(-1, 1) - mu1.reshape(-1, 1)
    det1, det2 = np.linalg.det(sigma1), np.linalg.det(sigma2)
    if det1 == 0 or det2 == 0:
        return np.array([np.inf, np.inf])
    lndet = np.log(det2 / det1)
    out = []
    for S_to, S_from, sign in ((sigma2, sigma1, 1.0), (sigma1, sigma2, -1.0)):
        a = np.linalg.lstsq(S_to, S_from, rcond=None)[0]
        b = np.linalg.lstsq(S_to, dmu, rcond=None)[0]
        out.append(0.5 * (np.trace(a) + dmu.T @ b - D + sign * lndet))
    return np.concatenate(out, axis=None)
