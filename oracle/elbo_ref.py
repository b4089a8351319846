"""Oracle: the negative-ELBO objective glue.  TEST INFRASTRUCTURE.

Restates /root/reference/pyvbmc/vbmc/variational_optimization.py:
``_soft_bound_loss`` (:609-657), ``_vp_bound_loss`` (:503-606) and
``_neg_elcbo`` (:991-1235).
"""
import numpy as np

from . import entropy_ref, gp_ref, mixture_ref


def soft_bound_loss(x, slb, sub, tol_con=1e-3, compute_grad=False):
    """Quadratic penalty outside [slb, sub], scale (sub-slb)*tol_con (:645-657)."""
    x = np.asarray(x, dtype=np.float64)
    ell = (sub - slb) * tol_con
    y = 0.0
    dy = np.zeros(x.shape)
    lo = x < slb
    if np.any(lo):
        y += 0.5 * np.sum(((slb[lo] - x[lo]) / ell[lo]) ** 2)
        dy[lo] = (x[lo] - slb[lo]) / ell[lo] ** 2
    hi = x > sub
    if np.any(hi):
        y += 0.5 * np.sum(((x[hi] - sub[hi]) / ell[hi]) ** 2)
        dy[hi] = (x[hi] - sub[hi]) / ell[hi] ** 2
    return (y, dy) if compute_grad else y


def vp_bound_loss(mix, theta, theta_bnd, tol_con=1e-3, compute_grad=True):
    """Soft-bound loss on mu, ln(sigma*lambda), eta and its chain rule back to theta (:537-606)."""
    D, K = mix.D, mix.K
    pos = 0
    if mix.optimize_mu:
        mu = theta[: D * K]
        pos = D * K
    else:
        mu = mix.mu.ravel(order="F")
    if mix.optimize_sigma:
        ln_sigma = theta[pos : pos + K]
        pos += K
    else:
        ln_sigma = np.log(mix.sigma)
    ln_lambd = theta[pos : pos + D] if mix.optimize_lambd else np.log(mix.lambd)
    ln_scale = ln_lambd.reshape(-1, 1) + ln_sigma.reshape(1, -1)  # (D, K)
    ext = []
    if mix.optimize_mu:
        ext.append(mu.ravel())
    if mix.optimize_sigma or mix.optimize_lambd:
        ext.append(ln_scale.ravel(order="F"))
    if mix.optimize_weights:
        ext.append(theta[-K:].ravel())
    ext = np.concatenate(ext)
    lb, ub = theta_bnd["lb"].ravel(), theta_bnd["ub"].ravel()
    if not compute_grad:
        return soft_bound_loss(ext, lb, ub, tol_con)
    L, dL = soft_bound_loss(ext, lb, ub, tol_con, compute_grad=True)
    out = []
    pos = 0
    if mix.optimize_mu:
        out.append(dL[: D * K])
        pos = D * K
    if mix.optimize_sigma or mix.optimize_lambd:
        # NB the reference reshapes C-order here (:585-587) although ln_scale was
        # flattened F-order (:561); restated as-is.
        dls = dL[pos : pos + D * K].reshape((D, K))
        if mix.optimize_sigma:
            out.append(dls.sum(axis=0))
        if mix.optimize_lambd:
            out.append(dls.sum(axis=1))
    if mix.optimize_weights:
        out.append(dL[-K:])
    return L, np.concatenate(out)


def neg_elcbo(
    theta,
    gp,
    mix,
    beta=0.0,
    Ns=0,
    compute_grad=True,
    compute_var=None,
    theta_bnd=None,
    separate_K=False,
    eps_half=None,
):
    """F = -G - H (+ beta sqrt(varF)) (+ soft-bound / weight penalties) (:1059-1235).

    Mutates ``mix`` (set_parameters + eta), like the reference (:1080-1085).
    """
    if not np.isfinite(beta):
        beta = 0
    if compute_var is None:
        compute_var = beta != 0
    if compute_grad and beta != 0 and compute_var != 2:
        raise NotImplementedError(
            "Computation of the gradient of ELBO with full variance not supported"
        )
    K = mix.K
    # NB no copy: like the reference (:1082-1085, `vp.eta = theta[-K:]` is a view
    # and `-=` shifts it in place) the caller's float64 theta gets its eta tail
    # max-shifted, and the bound loss below sees the shifted tail.
    theta = np.asarray(theta, dtype=np.float64)
    mixture_ref.set_parameters(mix, theta)
    if mix.optimize_weights:
        tail = theta[-K:]
        tail -= np.max(tail)
        mix.eta = tail.copy()
    if compute_grad:
        gf = (mix.optimize_mu, mix.optimize_sigma, mix.optimize_lambd, mix.optimize_weights)
    else:
        gf = (False,) * 4
    I_sk = J_sjk = None
    dG = None
    if separate_K:
        if compute_grad:
            raise ValueError(
                "Computing the gradient of variational parameters and "
                "requesting per-component results at the same time."
            )
        if compute_var:
            G, _, varG, _, varG_ss, I_sk, J_sjk = gp_ref.gp_log_joint(
                mix, gp, gf, True, True, compute_var, True
            )
        else:
            G, dG, _, _, _, I_sk, _ = gp_ref.gp_log_joint(mix, gp, gf, True, True, 0, True)
            varG = varG_ss = 0
    else:
        if compute_var:
            G, dG, varG, _, varG_ss = gp_ref.gp_log_joint(mix, gp, gf, True, True, compute_var)
        else:
            G, dG, _, _, _ = gp_ref.gp_log_joint(mix, gp, gf, True, True, 0)
            varG = varG_ss = 0
    if Ns > 0:
        H, dH = entropy_ref.entmc(mix, Ns, gf, True, eps_half=eps_half)
    else:
        H, dH = entropy_ref.entlb(mix, gf, True)
    F = -G - H
    if compute_grad:
        dF = -dG - dH
    else:
        dF = None
        dH = None
    varH = 0
    varF = varG + varH if compute_var else 0
    if beta != 0:
        F = F + beta * np.sqrt(varF)
    if theta_bnd is not None:
        if compute_grad:
            L, dL = vp_bound_loss(mix, theta, theta_bnd, tol_con=theta_bnd["tol_con"])
            dF = dF + dL
        else:
            L = vp_bound_loss(
                mix, theta, theta_bnd, tol_con=theta_bnd["tol_con"], compute_grad=False
            )
        F = F + L
        if mix.optimize_weights:
            thresh = theta_bnd["weight_threshold"]
            small = mix.w < thresh
            F = F + np.sum(mix.w * small + thresh * (~small)) * theta_bnd["weight_penalty"]
            if compute_grad:
                wg = theta_bnd["weight_penalty"] * small.astype(np.float64)
                dL = np.zeros(dF.shape)
                dL[-K:] = entropy_ref.softmax_jacobian(mix.eta) @ wg
                dF = dF + dL
    if separate_K:
        return F, dF, G, H, varF, dH, varG_ss, varG, varH, I_sk, J_sjk
    return F, dF, G, H, varF
