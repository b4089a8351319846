"""CPU restatement of the reference's closed-form acquisition functions.
TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline).

Follows pyvbmc/acquisition_functions/abstract_acq_fcn.py:68-147 (__call__), :195-222
(_sq_dist), :224-256 (_estimate_observation_noise) and the four
_compute_acquisition_function bodies (acq_fcn.py:38-45, acq_fcn_log.py:43-52,
acq_fcn_vanilla.py:38-42, acq_fcn_noisy.py:33-41).  Pinned by tests/golden/acq.npz,
produced by oracle/make_golden.py running the reference's own classes (with the gpyreg
stand-in supplying gp.predict, itself pinned by the MATLAB known answers).
The parameter transformer is the identity here (transformed == original space).
"""
import sys

import numpy as np

from . import gp_ref, mixture_ref

STD, LOG, VANILLA, NOISY = 0, 1, 2, 3


def sq_dist(a, b):
    """:209-222"""
    n, m = a.shape[0], b.shape[0]
    mu = (m / (n + m)) * np.mean(b, axis=0) + (n / (n + m)) * np.mean(a, axis=0)
    a = a - mu
    b = b - mu
    c = np.sum(a * a, axis=1, keepdims=True) + (np.sum(b * b, axis=1, keepdims=True).T - (2 * a @ b.T))
    return np.maximum(c, 0)


def estimate_observation_noise(Xs, X_rescaled, sn2_new, gp_length_scale):
    """:244-254"""
    pos = np.argmin(sq_dist(Xs / gp_length_scale, X_rescaled), axis=1)
    return sn2_new[pos]


def acq_call(kind, Xs, gp, mix, y_max, optim_state, X_rescaled=None, sn2_new=None):
    if Xs.ndim == 1:
        Xs = Xs[None, :]
    f_mu, f_s2 = gp_ref.predict(gp, Xs, separate_samples=True)        # :80
    Ns = f_mu.shape[1]
    f_bar = np.sum(f_mu, axis=1, keepdims=True) / Ns                   # :84
    var_bar = np.sum(f_s2, axis=1, keepdims=True) / Ns                 # :85-87
    var_f = np.sum((f_mu - f_bar) ** 2, axis=1, keepdims=True) / (Ns - 1) if Ns > 1 else 0  # :90-95
    f_bar = np.ravel(f_bar)
    var_tot = np.ravel(var_f + var_bar)                                # :98
    realmin = sys.float_info.min
    log_flag = kind == LOG
    if kind == LOG:
        log_p = np.ravel(np.maximum(mixture_ref.pdf(mix, Xs, log_flag=True), np.log(realmin)))
        acq = -(np.log(var_tot) + f_bar - y_max + log_p)
    else:
        p = np.ravel(np.maximum(mixture_ref.pdf(mix, Xs), realmin))
        if kind == STD:
            acq = -var_tot * np.exp(f_bar - y_max) * p
        elif kind == VANILLA:
            acq = -var_tot * p**2
        else:
            sn2 = estimate_observation_noise(Xs, X_rescaled, sn2_new, optim_state["gp_length_scale"])
            acq = -var_tot * (1 - sn2 / (var_tot + sn2)) * np.exp(f_bar - y_max) * p
    if optim_state.get("variance_regularized_acq_fcn"):                # :112-128
        tol_var = optim_state.get("tol_gp_var")
        low = var_tot < tol_var
        if np.any(low):
            if log_flag:
                acq[low] += tol_var / var_tot[low] - 1
            else:
                acq[low] *= np.exp(-(tol_var / var_tot[low] - 1))
    acq = np.maximum(acq, -sys.float_info.max)                         # :130-131
    out = np.logical_or(np.any(Xs < optim_state["lb_eps_orig"], axis=1),
                        np.any(Xs > optim_state["ub_eps_orig"], axis=1))  # :134-138
    acq[out] = np.inf
    return acq


def quantile_acq(gp, Xs, sn2, ais, u, use_weights):
    """AcqFcnVIQR / AcqFcnIMIQR._compute_acquisition_function restated (acq_fcn_viqr.py:80-158,
    acq_fcn_imiqr.py:77-171) on an oracle GP (gp_ref.GPData): posterior cross-covariance between
    the points and the importance points from dense solves, then the log-sum-exp of
    ln_w + u s_pred + log1p(-exp(-2 u s_pred)) over the importance points and over the GP samples.
    ``ais``: dict with X (Na, D), f_s2 (Na, S), ln_weights (S, Na)."""
    import numpy as np

    from . import gp_ref

    D, X = gp.D, gp.X
    Xa = ais["X"]
    _, f_s2 = gp_ref.predict(gp, Xs, separate_samples=True)
    y_s2 = f_s2 + np.reshape(sn2, (-1, 1))
    S = len(gp.posteriors)
    acq = np.zeros((Xs.shape[0], S))
    for s, post in enumerate(gp.posteriors):
        hyp = post.hyp[: D + 1]
        K_Xs_X = gp_ref.se_ard(hyp, Xs, X)
        K_Xs_Xa = gp_ref.se_ard(hyp, Xs, Xa)
        K_X_Xa = gp_ref.se_ard(hyp, X, Xa)
        if post.L_chol:
            import scipy.linalg as sla

            sn2_eff = 1 / post.sW[0] ** 2
            c_tmp = sla.solve_triangular(post.L, sla.solve_triangular(post.L, K_X_Xa, trans=1, check_finite=False),
                                         check_finite=False) / sn2_eff
            C = K_Xs_Xa - K_Xs_X @ c_tmp
        else:
            C = K_Xs_Xa + K_Xs_X @ (post.L @ K_X_Xa)
        tau2 = C**2 / y_s2[:, s].reshape(-1, 1)
        s_pred = np.sqrt(np.maximum(ais["f_s2"][:, s].T - tau2, 0.0))
        with np.errstate(all="ignore"):
            zz = u * s_pred + np.log1p(-np.exp(-2 * u * s_pred))
            if use_weights:
                zz = zz + ais["ln_weights"][s, :]
            ln_max = np.amax(zz, axis=1)
            ln_max[ln_max == -np.inf] = 0.0
            acq[:, s] = ln_max + np.log(np.sum(np.exp(zz - ln_max.reshape(-1, 1)), axis=1))
    if S > 1:
        with np.errstate(all="ignore"):
            M = np.amax(acq, axis=1)
            M[M == -np.inf] = 0.0
            return M + np.log(np.sum(np.exp(acq - M.reshape(-1, 1)), axis=1) / S)
    return acq.ravel()
