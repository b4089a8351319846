"""Acquisition functions -- mirror of ``pyvbmc/acquisition_functions`` for the
closed-form members (reference ``AcqFcn``, ``AcqFcnLog``, ``AcqFcnVanilla``,
``AcqFcnNoisy``) and of ``AbstractAcqFcn._sq_dist`` (row a13).

Same call signature as the reference, ``acq(Xs, gp, vp, function_logger, optim_state)``
(abstract_acq_fcn.py:36-147).  The GP prediction for every hyper-parameter sample, the
variational density and the acquisition formula run in one library call
(``vbmc_acq_eval``, csrc/api_acq.hip): the points go to the GPU once and only the M
acquisition values come back.  What needs the caller's ``parameter_transformer`` -- rounding
of integer variables and the hard-bound mask -- stays here on the host, as NumPy.

The information-theoretic members (``AcqFcnVIQR``, ``AcqFcnIMIQR``) are not part of this
path; asking ``string_to_acq`` for them raises ``NotImplementedError``.
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
        if np.any(integer_vars):
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
        out = np.logical_or(np.any(X_orig < optim_state.get("lb_eps_orig"), axis=1),
                            np.any(X_orig > optim_state.get("ub_eps_orig"), axis=1))
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


def string_to_acq(string):
    """Reference utilities.py:6: evaluate a constructor string such as ``"AcqFcnLog()"``."""
    names = {c.__name__: c for c in (AcqFcn, AcqFcnLog, AcqFcnVanilla, AcqFcnNoisy)}
    head = string.strip().split("(")[0]
    if head in ("AcqFcnVIQR", "AcqFcnIMIQR"):
        raise NotImplementedError(f"{head} is not on the accelerated path")
    if head not in names:
        raise ValueError(f"unknown acquisition function {string!r}")
    return names[head]()


