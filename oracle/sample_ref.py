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


def sample(mix, N, seed, balance_flag=False):
    lab = component_labels(mix.w, N, seed, balance_flag)
    z = philox_ref.normals(np.arange(N, dtype=np.uint64), mix.D, seed, 2)
    lam = mix.lambd.reshape(1, -1)
    x = mix.mu.T[lab] + lam * z * mix.sigma.ravel()[lab][:, None]     # :321-327
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
