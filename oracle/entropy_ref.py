"""Oracle: entropy estimators of the mixture.  TEST INFRASTRUCTURE.

Restates /root/reference/pyvbmc/entropy/entmc_vbmc.py:6-134 (Monte-Carlo entropy,
reparameterisation gradients) and entlb_vbmc.py:6-180 (Jensen lower bound).

Structure of ``entmc``: the raw, pre-Jacobian accumulators are produced by
``entmc_partial`` for an arbitrary subset of the antithetic half-draws and are
additive over disjoint subsets; ``entmc_finalize`` applies the Jacobians and
packs ``dH``.  That split is what the multi-GPU path shards; with
the full draw set it is arithmetically the reference loop (same loops over j
and k, same [Ns, D, K] temporaries).
"""
import numpy as np


def npair_per_component(Ns):
    """Reference rounds Ns up to even (entmc_vbmc.py:61); returns Ns/2."""
    return int(np.ceil(Ns / 2))


def draw_eps_half(K, D, Ns):
    """The reference's draw order (entmc_vbmc.py:64-68): for j ascending, one
    fresh ``np.random.randn(Ns//2, D)`` from the global legacy stream."""
    h = npair_per_component(Ns)
    eps = np.empty((K, h, D))
    for j in range(K):
        eps[j] = np.random.randn(h, D)
    return eps


def entmc_partial(mix, eps_half, ns_total, grad_flags):
    """Raw accumulators over the given antithetic half-draws.

    eps_half : (K, h, D) -- for each component j, h rows of standard normals;
               each row is used twice, as +eps and -eps (entmc_vbmc.py:67-68).
    ns_total : the normaliser Ns (even) of the full job.
    Returns dict H, mu (D,K), sigma (K,), lambd (D,), w (K,) -- un-Jacobianed sums.
    """
    D, K = mix.D, mix.K
    mu, sigma, lambd, w = mix.mu, mix.sigma, mix.lambd, mix.w
    any_grad = any(grad_flags)
    sl = sigma[None, :] * lambd[:, None]  # (D, K)  :53
    nconst = 1.0 / (2 * np.pi) ** (D / 2) / np.prod(lambd)  # :54-56
    out = {
        "H": 0.0,
        "mu": np.zeros((D, K)),
        "sigma": np.zeros(K),
        "lambd": np.zeros(D),
        "w": np.zeros(K),
    }
    Ns = float(ns_total)
    for j in range(K):
        e = np.concatenate([eps_half[j], -eps_half[j]], axis=0)  # (2h, D)
        if e.shape[0] == 0:
            continue
        xs = e * lambd * sigma[j] + mu[:, j]  # :70
        ys = np.zeros(e.shape[0])
        for k in range(K):  # :74-78
            d2 = np.sum(((xs - mu[:, k]) / (sigma[k] * lambd)) ** 2, axis=1)
            ys += w[k] * nconst / sigma[k] ** D * np.exp(-0.5 * d2)
        out["H"] += -w[j] * np.sum(np.log(ys)) / Ns  # :80
        if not any_grad:
            continue
        # :85-90 densities of every component at every sample, [n, K]
        r = np.sum(((xs[:, :, None] - mu[None]) / sl[None]) ** 2, axis=1)
        r = nconst / sigma**D * np.exp(-0.5 * r)
        q = (w * r).sum(1)
        # :93-95
        ls = (xs[:, :, None] - mu[None]) / sl[None] ** 2
        ls = (ls * w * r[:, None, :]).sum(2)  # [n, D]
        if grad_flags[0]:
            out["mu"][:, j] = w[j] * (ls / q[:, None]).sum(0) / Ns  # :98
        if grad_flags[1]:
            isum = (ls * e * lambd).sum(1)  # :102
            out["sigma"][j] = (w[j] * isum / q).sum() / Ns
        if grad_flags[2]:
            out["lambd"] += (w[j] * sigma[j] * e * ls / q[:, None]).sum(0) / Ns  # :106
        if grad_flags[3]:
            out["w"][j] -= np.log(q).sum() / Ns  # :111
            out["w"] -= (w[j] * r / q[:, None]).sum(0) / Ns  # :112
    return out


def pack_partial(p):
    """[H | mu 'F' | sigma | lambd | w] -- the vector the all-reduce sums."""
    return np.concatenate(
        [[p["H"]], p["mu"].ravel(order="F"), p["sigma"], p["lambd"], p["w"]]
    )


def unpack_partial(v, D, K):
    v = np.asarray(v)
    o = 1
    mu = v[o : o + D * K].reshape((D, K), order="F")
    o += D * K
    sg = v[o : o + K]
    o += K
    lm = v[o : o + D]
    o += D
    return {"H": float(v[0]), "mu": mu, "sigma": sg, "lambd": lm, "w": v[o : o + K]}


def softmax_jacobian(eta):
    """J_w of entmc_vbmc.py:123-129 / entlb_vbmc.py:170-176."""
    ee = np.exp(eta)
    s = ee.sum()
    return -np.outer(ee, ee) / s**2 + np.diag(ee) / s


def entmc_finalize(mix, p, grad_flags, jacobian_flag):
    """Jacobians and packing (entmc_vbmc.py:114-132); disabled blocks are omitted."""
    blocks = []
    if grad_flags[0]:
        blocks.append(p["mu"].ravel(order="F"))
    if grad_flags[1]:
        blocks.append(p["sigma"] * mix.sigma if jacobian_flag else p["sigma"])
    if grad_flags[2]:
        blocks.append(p["lambd"] * mix.lambd if jacobian_flag else p["lambd"])
    if grad_flags[3]:
        blocks.append(softmax_jacobian(mix.eta) @ p["w"] if jacobian_flag else p["w"])
    dH = np.concatenate(blocks) if blocks else np.zeros(0)
    return p["H"], dH


def entmc(mix, Ns, grad_flags=(True,) * 4, jacobian_flag=True, eps_half=None):
    """Monte-Carlo entropy; draws from np.random exactly like the reference if
    ``eps_half`` is not given."""
    h = npair_per_component(Ns)
    if eps_half is None:
        eps_half = draw_eps_half(mix.K, mix.D, Ns)
    assert eps_half.shape == (mix.K, h, mix.D)
    p = entmc_partial(mix, eps_half, 2 * h, grad_flags)
    return entmc_finalize(mix, p, grad_flags, jacobian_flag)


def entlb(mix, grad_flags=(True,) * 4, jacobian_flag=True):
    """Entropy lower bound (entlb_vbmc.py:44-180)."""
    D, K = mix.D, mix.K
    mu_t = mix.mu.T  # (K, D)
    sigma, lambd, w = mix.sigma, mix.lambd, mix.w
    g_mu = np.zeros((D, K))
    g_sigma = np.zeros(K)
    g_lambd = np.zeros(D)
    g_w = np.zeros(K)
    if K == 1:
        # exact entropy of one Gaussian (:60-78)
        H = 0.5 * D * (1 + np.log(2 * np.pi)) + D * np.log(sigma).sum() + np.log(lambd).sum()
        g_mu = np.zeros(D)
        g_sigma = D / sigma
        g_lambd = 1 / lambd
        g_w = np.zeros(1)
    else:
        s2 = sigma[:, None] ** 2 + sigma[None, :] ** 2  # :84
        ss = np.sqrt(s2)
        nconst = 1 / (2 * np.pi) ** (D / 2) / np.prod(lambd)
        diff = mu_t[:, None, :] - mu_t[None, :, :]  # [K, K, D]
        d2 = np.sum((diff / (ss[..., None] * lambd)) ** 2, axis=2)
        gamma = nconst / ss**D * np.exp(-0.5 * d2)  # :94
        gsum = (w * gamma).sum(1)  # :95
        H = -(w * np.log(gsum)).sum()
        if any(grad_flags):
            gfrac = gamma / gsum  # [i,j] = gamma_ij / gsum_j  (:100-102)
            wgfrac = w * gfrac
            dmu = diff / (s2[..., None] * lambd**2)
            dsig = -D / s2 + np.sum((diff / lambd) ** 2, 2) / s2**2
            for j in range(K):
                if grad_flags[0]:
                    m1 = (wgfrac[j, :][:, None] * dmu[:, j, :]).sum(0)
                    m2 = ((dmu[:, j, :] * gamma[:, [j]] * w[:, None]) / gsum[j]).sum(0)
                    g_mu[:, j] = -w[j] * (m1 + m2)
                if grad_flags[1]:
                    s1 = (wgfrac[j, :] * dsig[:, j]).sum()
                    s2j = ((dsig[:, j] * gamma[:, j] * w) / gsum[j]).sum()
                    g_sigma[j] = -w[j] * sigma[j] * (s1 + s2j)
            if grad_flags[2]:
                dmu2 = diff**2 / s2[..., None] / lambd**2
                inner = np.sum(w[:, None, None] * gamma[:, :, None] * (dmu2 - 1), 0)
                g_lambd = -np.sum(w[:, None] * inner / gsum[:, None], 0) / lambd
            if grad_flags[3]:
                g_w = -np.log(gsum) - wgfrac.sum(1)
    blocks = []
    if grad_flags[0]:
        blocks.append(g_mu.ravel(order="F") if K > 1 else g_mu)
    if grad_flags[1]:
        blocks.append(g_sigma * sigma if jacobian_flag else g_sigma)
    if grad_flags[2]:
        blocks.append(g_lambd * lambd if jacobian_flag else g_lambd)
    if grad_flags[3]:
        blocks.append(softmax_jacobian(mix.eta) @ g_w if jacobian_flag else g_w)
    dH = np.concatenate(blocks) if blocks else np.zeros(0)
    return H, dH
