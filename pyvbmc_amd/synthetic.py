"""Seeded synthetic inputs for the BASELINE.json configurations.

Pure input generation (NumPy): mixture parameters, GP training set and GP
hyper-parameters.  No ELBO arithmetic happens here.  Shared by bench.py, the
tests and oracle/make_golden.py so that every party evaluates the same inputs.
"""
from dataclasses import dataclass

import numpy as np

# cfg index -> (D, K, N, Ns_total), BASELINE.json `configs` in order
CONFIGS = {
    1: (2, 2, 50, 1_000),
    2: (6, 20, 200, 100_000),
    3: (10, 50, 400, 1_000_000),
    4: (10, 50, 400, 8_000_000),
    5: (20, 100, 800, 4_000_000),
}


@dataclass
class Workload:
    cfg: int
    D: int
    K: int
    N: int
    Ns_total: int
    mu: np.ndarray  # (D, K)
    sigma: np.ndarray  # (K,)
    lambd: np.ndarray  # (D,)
    w: np.ndarray  # (K,)
    eta: np.ndarray  # (K,)
    X: np.ndarray  # (N, D)
    y: np.ndarray  # (N, 1)
    s2: object  # None or (N, 1) user-provided noise (cfg 5)
    hyp: np.ndarray  # (S, 3D+3): [log ell (D), log sf, log sn, m0, xm (D), log omega (D)]

    @property
    def NsK(self):
        """Per-component sample count, as the reference derives it
        (variational_optimization.py:728 then entmc_vbmc.py:61)."""
        return ns_per_component(self.Ns_total, self.K)

    @property
    def theta(self):
        """[mu 'F' | log sigma | log lambd | eta] (variational_posterior.py:653-676)."""
        return np.concatenate(
            [self.mu.ravel(order="F"), np.log(self.sigma), np.log(self.lambd), self.eta]
        )


def ns_per_component(ns_total, K):
    nsk = int(np.ceil(ns_total / K))
    return 2 * int(np.ceil(nsk / 2))


def rosenbrock_log_joint(x):
    """Rosenbrock likelihood + N(0, 3^2) prior, the target of
    examples/scripts/pyvbmc_example_1_full_code.py:9-30 (config 1's GP data)."""
    x = np.atleast_2d(x)
    a, b = x[:, :-1], x[:, 1:]
    ll = -np.sum((a**2 - b) ** 2 + (a - 1) ** 2 / 100, axis=1)
    lp = np.sum(-0.5 * (x / 3.0) ** 2 - np.log(3.0) - 0.5 * np.log(2 * np.pi), axis=1)
    return ll + lp


def make_workload(cfg, S=1, D=None, K=None, N=None, Ns_total=None):
    """Inputs for BASELINE config ``cfg``; the keyword overrides shrink a config
    (same distributions, same seed) for oracle-sized parity cases."""
    d0, k0, n0, ns0 = CONFIGS[cfg]
    D = d0 if D is None else D
    K = k0 if K is None else K
    N = n0 if N is None else N
    Ns_total = ns0 if Ns_total is None else Ns_total
    rng = np.random.default_rng(20250215 + cfg)
    mu = rng.standard_normal((K, D)).T.copy()
    sigma = 0.5 * np.exp(0.3 * rng.standard_normal(K))
    lambd = np.exp(0.2 * rng.standard_normal(D))
    nl = np.sqrt(np.sum(lambd**2) / D)
    lambd = lambd / nl
    sigma = sigma * nl
    w = rng.dirichlet(np.ones(K))
    eta = np.log(w) - np.max(np.log(w))
    X = rng.standard_normal((N, D))
    s2 = None
    if cfg == 1:
        y = rosenbrock_log_joint(X).reshape(-1, 1)
    else:
        y = (-0.5 * np.sum(X**2, axis=1) + 0.01 * rng.standard_normal(N)).reshape(-1, 1)
    if cfg == 5:
        s2 = rng.uniform(0.01, 1.0, size=(N, 1))
        y = y + np.sqrt(s2) * rng.standard_normal((N, 1))
    base = np.concatenate(
        [np.zeros(D), [np.log(3.0)], [np.log(1e-2)], [0.0], np.zeros(D), np.zeros(D)]
    )
    hyp = base[None, :] + 0.05 * rng.standard_normal((S, base.size))
    return Workload(cfg, D, K, N, Ns_total, mu, sigma, lambd, w, eta, X, y, s2, hyp)


def draw_eps_half(K, D, NsK, seed):
    """Antithetic half-draws in the reference's order (entmc_vbmc.py:64-68):
    legacy ``np.random.seed(seed)``; for j ascending, ``randn(NsK//2, D)``.
    The caller's global RNG state is saved and restored."""
    state = np.random.get_state()
    try:
        np.random.seed(seed)
        h = NsK // 2
        eps = np.empty((K, h, D))
        for j in range(K):
            eps[j] = np.random.randn(h, D)
    finally:
        np.random.set_state(state)
    return eps


def default_theta_bnd(wl, tol_con_loss=0.01, tol_weight=1e-2, weight_penalty=0.1, tol_length=1e-6):
    """Soft bounds as ``VariationalPosterior.get_bounds`` builds them from the GP
    inputs (variational_posterior.py:140-239) with the option values the
    reference's tests use (test_variational_optimization.py:178-183)."""
    lo, hi = wl.X.min(axis=0), wl.X.max(axis=0)
    ln_range = np.log(hi - lo)
    lb = np.concatenate(
        [
            np.tile(lo, wl.K),
            np.tile(ln_range + np.log(tol_length), wl.K),
            np.full(wl.K, np.log(0.5 * tol_weight)),
        ]
    )
    ub = np.concatenate([np.tile(hi, wl.K), np.tile(ln_range, wl.K), np.zeros(wl.K)])
    return {
        "lb": lb,
        "ub": ub,
        "tol_con": tol_con_loss,
        "weight_threshold": max(1 / (4 * wl.K), tol_weight),
        "weight_penalty": weight_penalty,
    }
