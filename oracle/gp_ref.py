"""Oracle: the GP side of the ELBO path.  TEST INFRASTRUCTURE.

Two things live here.

1. The **gpyreg boundary**, restated.  gpyreg (`gpyreg >= 0.1.0`,
   /root/reference/pyproject.toml:13; CI installs unpinned HEAD,
   .github/workflows/tests.yml:24-28) is a third-party package that is NOT in
   /root/reference, so its arithmetic is restated from what PyVBMC's call sites
   require and pinned through the reference's own
   MATLAB-derived tests at that boundary
   (pyvbmc/testing/vbmc/test_variational_optimization.py:120-211,
   pyvbmc/testing/vbmc/test_active_importance_sampling.py:113-250).
   Parity UNPINNED for: heteroskedastic/user noise, the L_chol=False branch,
   predict(add_noise=True) -- those are defined from first principles below.

2. ``gp_log_joint`` -- restatement of
   /root/reference/pyvbmc/vbmc/variational_optimization.py:1238-1606.
"""
from dataclasses import dataclass

import numpy as np
import scipy.linalg as sla

MEAN_ZERO, MEAN_CONST, MEAN_NEGQUAD = 0, 1, 2


@dataclass
class Posterior:
    hyp: np.ndarray
    alpha: np.ndarray  # (N, 1)
    sW: np.ndarray  # (N,)
    L: np.ndarray  # (N, N)
    sn2_mult: float
    L_chol: bool


@dataclass
class GPData:
    D: int
    X: np.ndarray  # (N, D)
    y: np.ndarray  # (N, 1)
    s2: object  # None or (N, 1)
    mean_kind: int
    posteriors: list
    noise_user: bool = False  # add user-provided s2 to the noise variance

    @property
    def cov_n(self):
        return self.D + 1

    @property
    def noise_n(self):
        return 1


def mean_n(kind, D):
    return {MEAN_ZERO: 0, MEAN_CONST: 1, MEAN_NEGQUAD: 1 + 2 * D}[kind]


def se_ard(hyp_cov, A, B):
    """k(a,b) = sf^2 exp(-1/2 sum_d ((a_d-b_d)/ell_d)^2), sf^2 = exp(2 hyp[D]).

    Convention restated in-tree at acquisition_functions/acq_fcn_viqr.py:108-116.
    """
    D = A.shape[1]
    ell = np.exp(hyp_cov[:D])
    sf2 = np.exp(2 * hyp_cov[D])
    a = A / ell
    b = B / ell
    # direct differences (no |a|^2+|b|^2-2ab expansion): keeps the 1e-10
    # predictive-variance comparisons free of cancellation error
    d2 = np.zeros((A.shape[0], B.shape[0]))
    for d in range(D):
        d2 += (a[:, d][:, None] - b[:, d][None, :]) ** 2
    return sf2 * np.exp(-0.5 * d2)


def mean_fn(kind, hyp_mean, X):
    """ZeroMean / ConstantMean / NegativeQuadratic:  m0 - 1/2 sum((x-xm)/omega)^2
    (layout per variational_optimization.py:1383-1392)."""
    n, D = X.shape
    if kind == MEAN_ZERO:
        return np.zeros(n)
    if kind == MEAN_CONST:
        return np.full(n, hyp_mean[0])
    m0 = hyp_mean[0]
    xm = hyp_mean[1 : 1 + D]
    om = np.exp(hyp_mean[1 + D : 1 + 2 * D])
    return m0 - 0.5 * np.sum(((X - xm) / om) ** 2, axis=1)


def noise_var(gp_noise_hyp, n, s2=None, noise_user=False):
    sn2 = np.full(n, np.exp(2 * gp_noise_hyp[0]))
    if noise_user and s2 is not None:
        sn2 = sn2 + np.asarray(s2).ravel()
    return sn2


def make_posterior(hyp, X, y, mean_kind, s2=None, noise_user=False):
    """alpha, L, sW for one hyper-parameter vector (SURVEY Appendix A 'Posterior')."""
    N, D = X.shape
    hyp = np.asarray(hyp, dtype=np.float64)
    cov_n, noise_n = D + 1, 1
    Kxx = se_ard(hyp[:cov_n], X, X)
    m = mean_fn(mean_kind, hyp[cov_n + noise_n :], X)
    sn2 = noise_var(hyp[cov_n : cov_n + noise_n], N, s2, noise_user)
    sn2_div = np.min(sn2)
    sn2_mult = 1.0
    sn2_mat = np.diag(sn2 / sn2_div)
    r = y.ravel() - m
    if sn2_div * sn2_mult >= 1e-6:
        sl = sn2_div * sn2_mult
        L = sla.cholesky(Kxx / sl + sn2_mat, lower=False)
        alpha = sla.solve_triangular(
            L, sla.solve_triangular(L, r, trans=1, check_finite=False), trans=0, check_finite=False
        ) / sl
        L_chol = True
    else:
        sl = 1.0
        L_chol = False
        L = -np.linalg.inv(Kxx + sn2_mult * sn2_div * sn2_mat)
        alpha = -L @ r
    sW = np.ones(N) / np.sqrt(sn2_div * sn2_mult)
    return Posterior(hyp.copy(), alpha.reshape(-1, 1), sW, L, sn2_mult, L_chol)


def make_gp(X, y, hyp, mean_kind=MEAN_NEGQUAD, s2=None, noise_user=False):
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
    hyp = np.atleast_2d(np.asarray(hyp, dtype=np.float64))
    posts = [make_posterior(h, X, y, mean_kind, s2, noise_user) for h in hyp]
    return GPData(X.shape[1], X, y, s2, mean_kind, posts, noise_user)


def predict(gp, x_star, s2_star=0.0, add_noise=False, separate_samples=False):
    """Latent (or noisy) predictive mean / variance (SURVEY Appendix A 'Predict').

    Per sample: fmu = m(x*) + K*^T alpha;
    fs2 = max(0, sf^2 - ||L^-T (sW o K*)||^2)   (L_chol)
        = max(0, sf^2 + diag(K*^T L K*))        (otherwise).
    Averaging law: mean over s, fs2 = mean_s fs2 + var_s(fmu, ddof=1) -- the law
    the reference re-derives at acquisition_functions/abstract_acq_fcn.py:82-97.
    """
    x_star = np.atleast_2d(np.asarray(x_star, dtype=np.float64))
    M = x_star.shape[0]
    S = len(gp.posteriors)
    D = gp.D
    fmu = np.zeros((M, S))
    fs2 = np.zeros((M, S))
    for s, post in enumerate(gp.posteriors):
        hyp = post.hyp
        Ks = se_ard(hyp[: D + 1], gp.X, x_star)  # (N, M)
        ms = mean_fn(gp.mean_kind, hyp[D + 2 :], x_star)
        sf2 = np.exp(2 * hyp[D])
        fmu[:, s] = ms + Ks.T @ post.alpha.ravel()
        if post.L_chol:
            V = sla.solve_triangular(
                post.L, post.sW[:, None] * Ks, trans=1, check_finite=False
            )
            v = sf2 - np.sum(V * V, axis=0)
        else:
            v = sf2 + np.sum(Ks * (post.L @ Ks), axis=0)
        fs2[:, s] = np.maximum(v, 0.0)
        if add_noise:
            sn2 = noise_var(hyp[D + 1 : D + 2], M, None, False) + s2_star * gp.noise_user
            fs2[:, s] += sn2 * post.sn2_mult
    if separate_samples:
        return fmu, fs2
    fbar = fmu.mean(axis=1, keepdims=True)
    v = fs2.mean(axis=1, keepdims=True)
    if S > 1:
        v = v + np.sum((fmu - fbar) ** 2, axis=1, keepdims=True) / (S - 1)
    return fbar, v


def gp_log_joint(
    mix,
    gp,
    grad_flags,
    avg_flag=True,
    jacobian_flag=True,
    compute_var=False,
    separate_K=False,
):
    """Expected log joint of the mixture under the GP surrogate (Bayesian quadrature).

    Restates variational_optimization.py:1297-1606: per GP sample s and
    component k, z_k[n] = exp(lnnf_k - 1/2 sum_d ((mu_dk - X_nd)/tau_dk)^2),
    I_k = z_k . alpha + m0 + nu_k (:1400-1425); gradients (:1430-1465);
    variance via J_jk (:1473-1514); averaging over s (:1578-1596).
    Returns (G, dG, varG, dvarG, var_ss[, I_sk, J_sjk]).
    """
    if np.isscalar(grad_flags):
        grad_flags = (bool(grad_flags),) * 4
    if compute_var and any(grad_flags) and compute_var != 2:
        raise NotImplementedError(
            "Computation of gradient of log joint variance is currently "
            "available only for diagonal approximation of the variance."
        )
    if compute_var == 2:
        raise NotImplementedError(
            "Diagonal approximation of GP log-joint variance not implemented."
        )
    D, K = mix.D, mix.K
    X = gp.X
    N = X.shape[0]
    S = len(gp.posteriors)
    mu, sigma, lambd, w = mix.mu, mix.sigma, mix.lambd.reshape(-1, 1), mix.w
    quad = gp.mean_kind == MEAN_NEGQUAD
    G = np.zeros(S)
    g_mu = np.zeros((D, K, S))
    g_sigma = np.zeros((K, S))
    g_lambd = np.zeros((D, S))
    g_w = np.zeros((K, S))
    varG = np.zeros(S)
    I_sk = np.zeros((S, K))
    J_sjk = np.zeros((S, K, K))
    tiny = np.spacing(1)
    for s, post in enumerate(gp.posteriors):
        hyp = post.hyp
        ell = np.exp(hyp[:D]).reshape(-1, 1)
        ln_sf2 = 2 * hyp[D]
        sum_lnell = np.sum(hyp[:D])
        m0 = 0.0 if gp.mean_kind == MEAN_ZERO else hyp[D + 2]
        if quad:
            xm = hyp[D + 3 : 2 * D + 3].reshape(-1, 1)
            omega = np.exp(hyp[2 * D + 3 :]).reshape(-1, 1)
        alpha = post.alpha.ravel()
        sn2_eff = 1.0 / post.sW[0] ** 2
        Z = np.zeros((K, N))
        for k in range(K):
            tau = np.sqrt(sigma[k] ** 2 * lambd**2 + ell**2)  # (D,1)
            lnnf = ln_sf2 + sum_lnell - np.sum(np.log(tau))
            delta = (mu[:, k : k + 1] - X.T) / tau  # (D, N)
            z = np.exp(lnnf - 0.5 * np.sum(delta**2, axis=0))
            Z[k] = z
            I_k = z @ alpha + m0
            if quad:
                I_k += -0.5 * np.sum(
                    (mu[:, k : k + 1] ** 2 + sigma[k] ** 2 * lambd**2 - 2 * mu[:, k : k + 1] * xm + xm**2)
                    / omega**2
                )
            G[s] += w[k] * I_k
            I_sk[s, k] = I_k
            if grad_flags[0]:
                g = w[k] * ((-(delta / tau) * z) @ alpha)
                if quad:
                    g = g - w[k] / omega.ravel() ** 2 * (mu[:, k] - xm.ravel())
                g_mu[:, k, s] = g
            if grad_flags[1]:
                dz = np.sum((lambd / tau) ** 2 * (delta**2 - 1), axis=0) * sigma[k] * z
                g = w[k] * (dz @ alpha)
                if quad:
                    g -= w[k] * sigma[k] * np.sum(lambd**2 / omega**2)
                g_sigma[k, s] = g
            if grad_flags[2]:
                dz = (sigma[k] / tau) ** 2 * (delta**2 - 1) * (lambd * z)
                g = w[k] * (dz @ alpha)
                if quad:
                    g = g - w[k] * sigma[k] ** 2 / omega.ravel() ** 2 * lambd.ravel()
                g_lambd[:, s] += g
            if grad_flags[3]:
                g_w[k, s] = I_k
            if compute_var:
                for j in range(k + 1):
                    tjk = np.sqrt((sigma[j] ** 2 + sigma[k] ** 2) * lambd**2 + ell**2)
                    lnnf_jk = ln_sf2 + sum_lnell - np.sum(np.log(tjk))
                    djk = (mu[:, j : j + 1] - mu[:, k : k + 1]) / tjk
                    J = np.exp(lnnf_jk - 0.5 * np.sum(djk**2))
                    if post.L_chol:
                        t = sla.solve_triangular(post.L, Z[j], trans=1, check_finite=False)
                        t = sla.solve_triangular(post.L, t, trans=0, check_finite=False)
                        J -= Z[k] @ t / sn2_eff
                    else:
                        J += Z[k] @ (post.L @ Z[j])
                    if j == k:
                        varG[s] += w[k] ** 2 * max(tiny, J)
                        J_sjk[s, k, k] = J
                    else:
                        varG[s] += 2 * w[j] * w[k] * J
                        J_sjk[s, j, k] = J
                        J_sjk[s, k, j] = J
    varG = np.maximum(varG, tiny) if compute_var else None
    dG = None
    if any(grad_flags):
        parts = []
        if grad_flags[0]:
            parts.append(g_mu.reshape((D * K, S), order="F"))
        # sigma/lambda/w blocks exist only under jacobian_flag (:1528-1546)
        if jacobian_flag and grad_flags[1]:
            parts.append(g_sigma * sigma.reshape(-1, 1))
        if jacobian_flag and grad_flags[2]:
            parts.append(g_lambd * lambd)
        if jacobian_flag and grad_flags[3]:
            ee = np.exp(mix.eta)
            es = ee.sum()
            Jw = -np.outer(ee, ee) / es**2 + np.diag(ee) / es
            parts.append(Jw @ g_w)
        dG = np.concatenate(parts, axis=0)
    var_ss = 0
    if S > 1 and avg_flag:
        Gbar = G.sum() / S
        if compute_var:
            vss = np.sum((G - Gbar) ** 2) / (S - 1)
            var_ss = vss + np.std(varG, ddof=1)
            varG = np.sum(varG) / S + vss
        G = Gbar
        if dG is not None:
            dG = dG.sum(axis=1) / S
    if S == 1:
        G = G[0]
        if dG is not None:
            dG = dG[:, 0]
    if separate_K:
        return G, dG, varG, None, var_ss, I_sk, (J_sjk if compute_var else None)
    return G, dG, varG, None, var_ss
