"""TEST INFRASTRUCTURE (not part of the product: SURVEY.md section 2 marks active importance
sampling out of scope).  Its only job is to drive the device ``gp.predict`` / ``vp.pdf`` through the
MATLAB known-answer fixtures the reference holds for them (``fess.mat``,
``activesample_proposalpdf.mat``; reference tests test_active_importance_sampling.py:113-250).

Importance-sampling helpers that sit directly on top of the accelerated primitives --
restatement of the small functions of ``pyvbmc/vbmc/active_importance_sampling.py`` (reference
``active_sample_proposal_pdf`` :317-390, ``fess`` :426-478, ``renormalize_weights`` :481-483,
``get_mcmc_opts`` :393-423) and of the importance-sampling log densities of the
information-theoretic acquisition functions (``acq_fcn_viqr.py:159-247``,
``acq_fcn_imiqr.py:173-260``).

All device work goes through ``gp.predict`` and ``vp.pdf`` (vbmc_gp_predict, vbmc_mixture_pdf);
what is left here is O(N) NumPy glue, as in the reference.  The acquisition values of
VIQR / IMIQR themselves (``_compute_acquisition_function``) and the MCMC driver
``active_importance_sampling`` are not part of the accelerated path.
"""
import sys
from math import ceil

import numpy as np


# (AcqFcnVIQR / AcqFcnIMIQR themselves are product classes now: pyvbmc_amd.acquisition)


def active_sample_proposal_pdf(Xa, gp, vp_is, w_vp, rect_delta, acq_fcn):
    """Log importance weights of the proposal ``w_vp * vp_is + (1 - w_vp) * boxes around gp.X``
    against the acquisition's fixed integrand; returns ``(ln_weights (N, Ns_gp), f_s2 (N, Ns_gp))``."""
    N, D = gp.X.shape
    Na = Xa.shape[0]
    f_mu, f_s2 = gp.predict(Xa, separate_samples=True)
    temp_lpdf = np.zeros((Na, 1 + N if w_vp < 1 else 1))
    if w_vp > 0:
        temp_lpdf[:, 0] = vp_is.pdf(Xa, orig_flag=False, log_flag=True).T + np.log(w_vp)
    else:
        temp_lpdf[:, 0] = -np.inf
    ln_y = acq_fcn.is_log_base(Xa, f_mu=f_mu, f_s2=f_s2)
    if w_vp < 1:
        VV = np.prod(2 * rect_delta)
        inside = np.all(np.abs(Xa[:, None, :] - gp.X[None, :, :]) < rect_delta, axis=2)  # (Na, N)
        temp_lpdf[:, 1:] = np.where(inside, np.log((1 - w_vp) / VV / N), -np.inf)
        m_max = np.amax(temp_lpdf, axis=1)
        if np.any(m_max == -np.inf):
            raise ValueError("Invalid value.")
        l_pdf = np.log(np.sum(np.exp(temp_lpdf - m_max.reshape(-1, 1)), axis=1))
        ln_weights = ln_y - (l_pdf + m_max).reshape(-1, 1)
    else:
        ln_weights = ln_y - temp_lpdf
    return ln_weights, f_s2


def get_mcmc_opts(Ns, thin=1, burn_in=None):
    if burn_in is None:
        burn_in = ceil(thin * Ns / 2)
    return {"display": "off", "diagnostics": False}, thin, burn_in


def fess(vp, gp, X=100):
    """Fractional effective sample size of importance sampling from ``vp`` towards the GP's
    log-density surrogate.  ``gp``: a GP (its averaged predictive mean is used) or an array of
    per-sample means (N, Ns_gp); ``X``: points (N, D) or a number of VP samples to draw."""
    if np.isscalar(X):
        N = X
        X, _ = vp.sample(N, orig_flag=False)
    else:
        N = X.shape[0]
    if hasattr(gp, "predict"):
        f_bar, __ = gp.predict(X)
        f_bar = f_bar.ravel()
    else:
        f_bar = np.mean(gp, axis=1)
    if f_bar.shape[0] != X.shape[0]:
        raise ValueError("Mismatch between number of samples from VP and GP.")
    v_ln_pdf = np.maximum(vp.pdf(X, orig_flag=False, log_flag=True), np.log(sys.float_info.min)).ravel()
    ln_weights = f_bar - np.atleast_2d(v_ln_pdf)
    weight = np.exp(ln_weights - np.amax(ln_weights))
    weight = weight / np.sum(weight)
    return (1 / np.sum(weight**2)) / N


def renormalize_weights(ln_w):
    M = np.amax(ln_w)
    return ln_w - (M + np.log(np.sum(np.exp(ln_w - M))))
