"""Acquisition functions -- mirror of ``pyvbmc/acquisition_functions`` for the
closed-form members (reference ``AcqFcn``, ``AcqFcnLog``, ``AcqFcnVanilla``,
``AcqFcnNoisy``) and of ``AbstractAcqFcn._sq_dist`` (row a13).

Same call signature as the reference, ``acq(Xs, gp, vp, function_logger, optim_state)``
(abstract_acq_fcn.py:36-147).  The GP prediction for every hyper-parameter sample, the
variational density and the acquisition formula run in one library call
(``vbmc_acq_eval``, csrc/api_acq.hip): the points go to the GPU once and only the M
acquisition values come back.  What needs the caller's ``parameter_transformer`` -- rounding
of integer variables and the hard-bound mask -- stays here on the host, as NumPy.

The two importance-sampled members for noisy targets, ``AcqFcnVIQR`` and ``AcqFcnIMIQR``
(reference acq_fcn_viqr.py, acq_fcn_imiqr.py), evaluate through ``vbmc_acq_is_set`` /
``vbmc_acq_is_eval`` (csrc/api_acq_is.hip): the importance state the reference prepares once per
active-sampling round (``optim_state["active_importance_sampling"]``) is uploaded once and stays
in HBM while the acquisition is evaluated thousands of times.  Preparing that state
(``active_importance_sampling``) is the reference's own code and out of scope here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._duck import ctx_of, upload_vp
from .gp import upload_gp

ACQ_STD, ACQ_LOG, ACQ_VANILLA, ACQ_NOISY = 0, 1, 2, 3


def sq_dist(a, b, *, ctx=None, return_argmin=False):
    """All pairwise squared distances between the rows of ``a`` (n, D) and ``b`` (m, D),
    computed like the reference's ``_sq_dist`` (abstract_acq_fcn.py:195-222): common mean
    removed, ``|a|^2 + |b|^2 - 2 a.b``, clamped at zero.  With ``return_argmin`` also
    ``np.argmin(c, axis=1)`` (computed on the device)."""
    ctx = _lib.default_context() if ctx is None else ctx
    a = _lib.f64(np.atleast_2d(a))
    b = _lib.f64(np.atleast_2d(b))
    if a.shape[1] != b.shape[1]:
        raise ValueError("sq_dist: a and b must have the same number of columns")
    n, m, D = a.shape[0], b.shape[0], a.shape[1]
    c = np.empty((n, m))
    idx = np.empty(n, dtype=np.int64) if return_argmin else None
    ip = idx.ctypes.data_as(C.POINTER(C.c_int64)) if return_argmin else None
    ctx.check(ctx._lib.vbmc_sq_dist(ctx._h, n, m, D, _lib.ptr(a), _lib.ptr(b), _lib.ptr(c), ip))
    return (c, idx) if return_argmin else c


def nearest_neighbour(a, b, *, ctx=None):
    """``np.argmin(_sq_dist(a, b), axis=1)`` without materialising the matrix on the host."""
    ctx = _lib.default_context() if ctx is None else ctx
    a = _lib.f64(np.atleast_2d(a))
    b = _lib.f64(np.atleast_2d(b))
    idx = np.empty(a.shape[0], dtype=np.int64)
    ctx.check(ctx._lib.vbmc_sq_dist(ctx._h, a.shape[0], b.shape[0], a.shape[1], _lib.ptr(a), _lib.ptr(b),
                                    None, idx.ctypes.data_as(C.POINTER(C.c_int64))))
    return idx


class AbstractAcqFcn:
    """Base of the closed-form acquisition functions (reference abstract_acq_fcn.py)."""

    _kind = None

    def __init__(self):
        self.acq_info = {"compute_var_log_joint": False, "log_flag": False}

    def get_info(self):
        return self.acq_info

    # reference staticmethods, kept under their names
    @staticmethod
    def _sq_dist(a, b):
        return sq_dist(a, b)

    @staticmethod
    def _real2int(X, parameter_transformer, integer_vars):
        if integer_vars is not None and np.any(integer_vars):
            X_temp = parameter_transformer.inverse(X)
            X_temp[:, integer_vars] = np.around(X_temp[:, integer_vars])
            X_temp = parameter_transformer(X_temp)
            X[:, integer_vars] = X_temp[:, integer_vars]
        return X

    def _estimate_observation_noise(self, Xs, gp, optim_state):
        """Noise at the nearest training input (abstract_acq_fcn.py:224-256)."""
        pos = nearest_neighbour(Xs / optim_state.get("gp_length_scale"),
                                gp.temporary_data.get("X_rescaled"), ctx=getattr(gp, "_ctx", None))
        return gp.temporary_data.get("sn2_new")[pos]

    def __call__(self, Xs, gp, vp, function_logger, optim_state):
        if self._kind is None:
            raise NotImplementedError("AbstractAcqFcn is abstract")
        Xs = np.asarray(Xs, dtype=np.float64)
        if Xs.ndim == 1:
            Xs = Xs[None, :]
        Xs = self._real2int(Xs, vp.parameter_transformer, optim_state.get("integer_vars"))
        ctx = ctx_of(vp)
        upload_vp(vp, ctx)
        upload_gp(gp, ctx)
        xs = _lib.f64(Xs)
        M = xs.shape[0]
        sn2 = None
        if self._kind == ACQ_NOISY:
            sn2 = _lib.f64(np.ravel(self._estimate_observation_noise(Xs, gp, optim_state)))
        tol_var = 0.0
        if optim_state.get("variance_regularized_acq_fcn"):
            tol_var = float(optim_state.get("tol_gp_var"))
        acq = np.empty(M)
        y_max = float(function_logger.y_max) if self._kind in (ACQ_STD, ACQ_LOG, ACQ_NOISY) else 0.0
        ctx.check(ctx._lib.vbmc_acq_eval(ctx._h, M, _lib.ptr(xs), self._kind, y_max, tol_var,
                                         _lib.ptr(sn2), _lib.ptr(acq), None, None))
        # hard bounds in original space (abstract_acq_fcn.py:133-139)
        X_orig = vp.parameter_transformer.inverse(Xs)
        out = ((X_orig < optim_state.get("lb_eps_orig")) | (X_orig > optim_state.get("ub_eps_orig"))).any(axis=1)
        if out.any():
            acq[out] = np.inf
        return acq


class AcqFcn(AbstractAcqFcn):
    """Prospective uncertainty search (reference acq_fcn.py)."""

    _kind = ACQ_STD


class AcqFcnLog(AbstractAcqFcn):
    """Prospective uncertainty search, log-valued (reference acq_fcn_log.py)."""

    _kind = ACQ_LOG

    def __init__(self):
        super().__init__()
        self.acq_info["log_flag"] = True


class AcqFcnVanilla(AbstractAcqFcn):
    """Vanilla uncertainty sampling (reference acq_fcn_vanilla.py)."""

    _kind = ACQ_VANILLA


class AcqFcnNoisy(AbstractAcqFcn):
    """Prospective uncertainty search for noisy targets (reference acq_fcn_noisy.py)."""

    _kind = ACQ_NOISY


class _QuantileAcq(AbstractAcqFcn):
    """Shared parts of AcqFcnVIQR / AcqFcnIMIQR (reference acq_fcn_viqr.py, acq_fcn_imiqr.py)."""

    _kind = "is"

    def __init__(self, quantile=0.75):
        from scipy.stats import norm

        self.acq_info = {"log_flag": True, "importance_sampling": True, "importance_sampling_vp": False,
                         "quantile": quantile, "compute_var_log_joint": False}
        self.u = norm.ppf(quantile)

    # importance-sampling log densities (acq_fcn_viqr.py:159-247, acq_fcn_imiqr.py:173-260): NumPy
    # on top of the device gp.predict
    def is_log_added(self, **kwargs):
        f_s = np.sqrt(kwargs["f_s2"])
        return self.u * f_s + np.log1p(-np.exp(-2 * self.u * f_s))

    @staticmethod
    def _c_tmp(gp, ais):
        """C_tmp[s] = (L'L)^-1 K(X, Xa) / sn2_eff, or L K(X, Xa) for a non-Cholesky sample -- what
        active_importance_sampling.py:279-306 stores for VIQR; IMIQR stores K_Xa_X and repeats the
        solves on every call (acq_fcn_imiqr.py:127-141): formed once here instead."""
        if ais.get("C_tmp") is not None:
            return np.ascontiguousarray(ais["C_tmp"], dtype=np.float64)
        from scipy.linalg import solve_triangular

        K_Xa_X = ais["K_Xa_X"]
        out = np.empty((K_Xa_X.shape[0], K_Xa_X.shape[2], K_Xa_X.shape[1]))
        for s, p in enumerate(gp.posteriors):
            if p.L_chol:
                sn2_eff = 1 / np.ravel(p.sW)[0] ** 2
                out[s] = solve_triangular(p.L, solve_triangular(p.L, K_Xa_X[s].T, trans=True, check_finite=False),
                                          check_finite=False) / sn2_eff
            else:
                out[s] = p.L @ K_Xa_X[s].T
        return out

    def _upload_state(self, gp, ais, ctx):
        # every array the device state is built from, and the content-aware key of the GP that
        # upload_gp (called just before) left in the context: a posterior record replaced in place
        # (active_importance_sampling.py:207-209) or a re-bound ais["C_tmp"] uploads the state again
        parts = tuple(ais.get(k) for k in ("X", "f_s2", "ln_weights", "C_tmp", "K_Xa_X"))
        key = (id(ais),) + tuple(id(a) for a in parts) + tuple(ctx.__dict__.get("_gp_key") or ())
        if ctx.__dict__.get("_acq_is_key") == key:
            return
        Xa = _lib.f64(ais["X"])
        per_sample = Xa.ndim == 3
        Na = Xa.shape[-2]
        ctmp = _lib.f64(self._c_tmp(gp, ais))
        fs2a = _lib.f64(ais["f_s2"])
        lnw = None if self.acq_info.get("variational_importance_sampling") else _lib.f64(ais["ln_weights"])
        ctx.check(ctx._lib.vbmc_acq_is_set(ctx._h, Na, _lib.ptr(Xa), int(per_sample), _lib.ptr(ctmp),
                                           _lib.ptr(fs2a), _lib.ptr(lnw)))
        # (the held references keep every keyed object alive, so an id cannot be recycled)
        ctx.__dict__["_acq_is_key"], ctx.__dict__["_acq_is_ref"] = key, (ais, gp.posteriors, parts)

    def __call__(self, Xs, gp, vp, function_logger, optim_state):
        Xs = np.asarray(Xs, dtype=np.float64)
        if Xs.ndim == 1:
            Xs = Xs[None, :]
        Xs = self._real2int(Xs, vp.parameter_transformer, optim_state.get("integer_vars"))
        ctx = ctx_of(vp)
        upload_gp(gp, ctx)
        self._upload_state(gp, optim_state["active_importance_sampling"], ctx)
        xs = _lib.f64(Xs)
        M = xs.shape[0]
        sn2 = _lib.f64(np.ravel(self._estimate_observation_noise(Xs, gp, optim_state)))
        acq = np.empty(M)
        reg = bool(optim_state.get("variance_regularized_acq_fcn"))
        var_tot = np.empty(M) if reg else None
        ctx.check(ctx._lib.vbmc_acq_is_eval(ctx._h, M, _lib.ptr(xs), _lib.ptr(sn2), float(self.u), _lib.ptr(acq),
                                            _lib.ptr(var_tot)))
        # regularisation, clamp and hard bounds of AbstractAcqFcn.__call__ (abstract_acq_fcn.py:110-139)
        if reg:
            tol_var = optim_state.get("tol_gp_var")
            low = var_tot < tol_var
            acq[low] += tol_var / var_tot[low] - 1  # log_flag is set for both classes
        acq = np.maximum(acq, -np.finfo(np.float64).max)
        X_orig = vp.parameter_transformer.inverse(Xs)
        out = ((X_orig < optim_state.get("lb_eps_orig")) | (X_orig > optim_state.get("ub_eps_orig"))).any(axis=1)
        if out.any():
            acq[out] = np.inf
        return acq


class AcqFcnVIQR(_QuantileAcq):
    """Variational interquantile range (reference acq_fcn_viqr.py): simple Monte Carlo over the VP."""

    def __init__(self, quantile=0.75):
        super().__init__(quantile)
        self.acq_info["variational_importance_sampling"] = True

    def is_log_base(self, x, **kwargs):
        return np.zeros(kwargs["f_s2"].shape)

    def is_log_full(self, x, **kwargs):
        f_s2 = kwargs.pop("f_s2", None)
        if f_s2 is None:
            gp = kwargs.get("gp")
            if gp is None:
                raise ValueError("Must provide gp as keyword argument if f_s2 is not provided.")
            __, f_s2 = gp.predict(np.atleast_2d(x), add_noise=True)
        return self.is_log_added(f_s2=f_s2, **kwargs)


class AcqFcnIMIQR(_QuantileAcq):
    """Integrated median interquantile range (reference acq_fcn_imiqr.py)."""

    def __init__(self, quantile=0.75):
        super().__init__(quantile)
        self.acq_info["variational_importance_sampling"] = False

    def is_log_base(self, x, **kwargs):
        return kwargs["f_mu"]

    def is_log_full(self, x, **kwargs):
        f_mu, f_s2 = kwargs.pop("f_mu", None), kwargs.pop("f_s2", None)
        if f_mu is None or f_s2 is None:
            gp = kwargs.get("gp")
            if gp is None:
                raise ValueError("Must provide gp as keyword argument if f_mu / f_s2 are not provided.")
            f_mu, f_s2 = gp.predict(np.atleast_2d(x), add_noise=True)
        return self.is_log_base(x, f_mu=f_mu) + self.is_log_added(f_s2=f_s2, **kwargs)


def string_to_acq(string):
    """Reference utilities.py:6: evaluate a constructor string such as ``"AcqFcnLog()"``."""
    names = {c.__name__: c for c in (AcqFcn, AcqFcnLog, AcqFcnVanilla, AcqFcnNoisy, AcqFcnVIQR, AcqFcnIMIQR)}
    import ast

    # the reference evals the string; here only `Name(...)` with literal arguments is accepted
    try:
        call = ast.parse(string.strip(), mode="eval").body
        ok = isinstance(call, ast.Call) and isinstance(call.func, ast.Name) and call.func.id in names
        args = [ast.literal_eval(a) for a in call.args] if ok else []
        kwargs = {k.arg: ast.literal_eval(k.value) for k in call.keywords} if ok else {}
    except (SyntaxError, ValueError):
        ok = False
    if not ok:
        raise ValueError(f"unknown acquisition function {string!r}")
    return names[call.func.id](*args, **kwargs)


