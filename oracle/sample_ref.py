"""Oracle for SURVEY 8f row 4: sampling and the Monte-Carlo KL divergence.
TEST INFRASTRUCTURE ONLY.

``sample`` restates VariationalPosterior.sample for the transformed space and Gaussian
components (reference variational_posterior/variational_posterior.py:296-327) with the draws
taken from the device generator's streams (oracle/philox_ref.py; csrc/sample.hip) instead of
NumPy's global stream, and -- like the device -- without the final shuffle of the balanced
labels.  ``kl_div_mc`` restates the gauss_flag=False branch of kl_div (:1107-1126) on those
samples.  The reference's own NumPy-stream behaviour is pinned separately through
tests/golden/vpmc.npz (oracle/make_golden.py vpmc).
"""
import sys

import numpy as np

from . import mixture_ref, philox_ref


def component_labels(w, N, seed, balance_flag):
    w = np.ravel(w)
    K = w.size
    idx = np.arange(N, dtype=np.uint64)
    if K == 1:
        return np.zeros(N, dtype=np.int64)
    lab = np.empty(N, dtype=np.int64)
    n_exact = 0
    p = w
    if balance_flag:
        repeats = np.floor(w * N).astype("int")                       # :298
        lab_exact = np.repeat(np.arange(K), repeats)                   # :299
        n_exact = min(lab_exact.size, N)
        lab[:n_exact] = lab_exact[:n_exact]
        w_extra = w * N - repeats                                      # :302
        repeats_extra = np.ceil(np.sum(w_extra))                       # :303
        w_extra = w_extra + w * (repeats_extra - np.sum(w_extra))      # :304
        tot = np.sum(w_extra)
        p = w_extra / tot if tot > 0 else w                            # :305
    if n_exact < N:
        cdf = np.cumsum(p)
        cdf[-1] = 2.0
        u = philox_ref.uniform(idx[n_exact:], seed, 3)
        lab[n_exact:] = np.searchsorted(cdf, u, side="right")          # np.random.choice's inverse CDF
    return lab


def gamma_variates(idx, shape, seed):
    """Gamma(shape, 1) variates of the device generator (csrc/sample.hip philox_gamma): Marsaglia &
    Tsang with round r taking its normal from Philox(n, 2r, 4) and its uniforms from Philox(n, 2r+1, 4)."""
    idx = np.asarray(idx, dtype=np.uint64)
    small = shape < 1.0
    sh = shape + 1.0 if small else shape
    d = sh - 1.0 / 3.0
    c = 1.0 / np.sqrt(9.0 * d)
    g = np.full(idx.size, d)
    up = np.ones(idx.size)
    todo = np.ones(idx.size, dtype=bool)
    lo, hi = philox_ref._split(idx)
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF

    def two_uniforms(r):
        x0, x1, x2, x3 = philox_ref.philox4x32_10(lo[todo], hi[todo], np.full(todo.sum(), r, dtype=np.uint32),
                                                  np.full(todo.sum(), 4, dtype=np.uint32), k0, k1)
        a = ((x0.astype(np.uint64) << np.uint64(32)) | x1.astype(np.uint64)) >> np.uint64(11)
        b = ((x2.astype(np.uint64) << np.uint64(32)) | x3.astype(np.uint64)) >> np.uint64(11)
        return a, b

    for r in range(64):
        if not todo.any():
            break
        a, b = two_uniforms(2 * r)
        u1 = (a + np.uint64(1)).astype(np.float64) * 2.0**-53
        u2 = b.astype(np.float64) * 2.0**-53
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
        t = 1.0 + c * z
        a, b = two_uniforms(2 * r + 1)
        U = (a + np.uint64(1)).astype(np.float64) * 2.0**-53
        Up = (b + np.uint64(1)).astype(np.float64) * 2.0**-53
        with np.errstate(all="ignore"):
            v = t**3
            ok = (t > 0) & (np.log(U) < 0.5 * z * z + d - d * v + d * np.log(np.where(t > 0, v, 1.0)))
        where = np.flatnonzero(todo)[ok]
        g[where] = d * v[ok]
        up[where] = Up[ok]
        todo[where] = False
    if small:
        g = g * np.exp(np.log(up) / shape)
    return g


def sample(mix, N, seed, balance_flag=False, df=np.inf):
    lab = component_labels(mix.w, N, seed, balance_flag)
    idx = np.arange(N, dtype=np.uint64)
    z = philox_ref.normals(idx, mix.D, seed, 2)
    lam = mix.lambd.reshape(1, -1)
    if not (np.isfinite(df) and df != 0):
        return mix.mu.T[lab] + lam * z * mix.sigma.ravel()[lab][:, None], lab  # :321-327
    G = gamma_variates(idx, df / 2, seed) * (df / 2)                   # np.random.gamma(df/2, df/2) (:330, :346)
    t = (df / 2 / np.sqrt(G))[:, None]
    if mix.K > 1:
        x = mix.mu.T[lab] + lam * z * t * mix.sigma.ravel()[lab][:, None]  # :332-336
    else:
        x = mix.mu.T[lab] + lam * t * z * mix.sigma.ravel()[lab][:, None]  # :349-353
    return x, lab


def kl_div_mc(mix1, mix2, N, seed):
    minp = sys.float_info.min
    xx1, _ = sample(mix1, N, seed, True)
    q1, q2 = mixture_ref.pdf(mix1, xx1).ravel(), mixture_ref.pdf(mix2, xx1).ravel()
    q1[q1 == 0] = 1.0                                                  # :1113 (as Python parses it)
    q2[q2 == 0] = minp
    kl1 = -np.mean(np.log(q2) - np.log(q1))
    xx2, _ = sample(mix2, N, seed + 1, True)
    q1, q2 = mixture_ref.pdf(mix1, xx2).ravel(), mixture_ref.pdf(mix2, xx2).ravel()
    q1[q1 == 0] = minp
    q2[q2 == 0] = 1.0
    kl2 = -np.mean(np.log(q1) - np.log(q2))
    return np.maximum(0, np.array([kl1, kl2]))
