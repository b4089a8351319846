"""Oracle: Gaussian-mixture (variational posterior) arithmetic.  TEST INFRASTRUCTURE.

Restates, on plain arrays, what the reference's ``VariationalPosterior`` does
on the ELBO path.  All citations are to
/root/reference/pyvbmc/variational_posterior/variational_posterior.py.
"""
from dataclasses import dataclass, field

import numpy as np
from scipy.special import gammaln


@dataclass
class Mixture:
    """Plain container with the reference attribute layout (:106-138)."""

    D: int
    K: int
    mu: np.ndarray  # (D, K)
    sigma: np.ndarray  # (K,)
    lambd: np.ndarray  # (D,)
    w: np.ndarray  # (K,)
    eta: np.ndarray  # (K,)
    optimize_mu: bool = True
    optimize_sigma: bool = True
    optimize_lambd: bool = True
    optimize_weights: bool = True
    extra: dict = field(default_factory=dict)

    @staticmethod
    def make(mu, sigma, lambd, w, eta=None):
        mu = np.array(mu, dtype=np.float64)
        D, K = mu.shape
        w = np.array(w, dtype=np.float64).ravel()
        if eta is None:
            eta = np.log(w) - np.max(np.log(w))
        return Mixture(
            D,
            K,
            mu,
            np.array(sigma, dtype=np.float64).ravel().copy(),
            np.array(lambd, dtype=np.float64).ravel().copy(),
            w.copy(),
            np.array(eta, dtype=np.float64).ravel().copy(),
        )

    def copy(self):
        return Mixture(
            self.D,
            self.K,
            self.mu.copy(),
            self.sigma.copy(),
            self.lambd.copy(),
            self.w.copy(),
            self.eta.copy(),
            self.optimize_mu,
            self.optimize_sigma,
            self.optimize_lambd,
            self.optimize_weights,
        )


def _renormalise(mix):
    # :642-645 and :749-752 -- lambda is forced to unit RMS, sigma absorbs it.
    nl = np.sqrt(np.sum(mix.lambd**2) / mix.D)
    mix.lambd = mix.lambd / nl
    mix.sigma = mix.sigma * nl
    if mix.optimize_weights:
        mix.w = mix.w / np.sum(mix.w)


def get_parameters(mix, raw_flag=True):
    """theta = [mu (K blocks of D) | sigma | lambd | w], log of the tail if raw (:623-678).

    Mutates ``mix`` (renormalisation), like the reference.
    """
    _renormalise(mix)
    head = mix.mu.ravel(order="F") if mix.optimize_mu else np.zeros(0)
    tail = []
    if mix.optimize_sigma:
        tail.append(mix.sigma)
    if mix.optimize_lambd:
        tail.append(mix.lambd)
    if mix.optimize_weights:
        tail.append(mix.w)
    tail = np.concatenate(tail) if tail else np.zeros(0)
    return np.concatenate([head, np.log(tail) if raw_flag else tail])


def set_parameters(mix, theta, raw_flag=True):
    """Inverse of get_parameters (:680-759); softmax of eta with max shift (:741-747)."""
    theta = np.array(theta, dtype=np.float64)
    D, K = mix.D, mix.K
    if not raw_flag:
        n_con = (
            K * mix.optimize_weights + D * mix.optimize_lambd + K * mix.optimize_sigma
        )
        # The reference slices theta[-check_idx:] with check_idx <= 0 (:701-710),
        # i.e. theta[n_con:]; reproduce that exact slice.
        if np.any(theta[n_con:] < 0.0):
            raise ValueError(
                "sigma, lambda and weights must be positive when raw_flag = False"
            )
    pos = 0
    if mix.optimize_mu:
        mix.mu = theta[: D * K].reshape((D, K), order="F").copy()
        pos = D * K
    if mix.optimize_sigma:
        s = theta[pos : pos + K]
        mix.sigma = np.exp(s) if raw_flag else s.copy()
        pos += K
    if mix.optimize_lambd:
        l = theta[pos : pos + D]
        mix.lambd = np.exp(l) if raw_flag else l.copy()
    if mix.optimize_weights:
        e = theta[-K:]
        mix.w = np.exp(e - np.max(e)) if raw_flag else e.copy()
    _renormalise(mix)


def pdf(mix, x, log_flag=False, grad_flag=False, df=np.inf):
    """Mixture density in the transformed space (orig_flag=False branch of :365-564).

    Linear-domain accumulation over components, then log with 0 -> -inf (:531-541);
    the log-gradient is dy/y taken before the log (:532-533).
    """
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    n, D = x.shape
    lam = mix.lambd.reshape(1, -1)
    y = np.zeros((n, 1))
    dy = np.zeros((n, D)) if grad_flag else None
    gaussian = (not np.isfinite(df)) or df == 0
    if gaussian:
        nf = 1.0 / (2 * np.pi) ** (D / 2) / np.prod(lam)  # :450
    elif df > 0:
        nf = (
            np.exp(gammaln((df + D) / 2) - gammaln(df / 2))
            / (df * np.pi) ** (D / 2)
            / np.prod(lam)
        )  # :478-482
    else:
        a = abs(df)
        nf = (np.exp(gammaln((a + 1) / 2) - gammaln(a / 2)) / np.sqrt(a * np.pi)) ** D
        nf = nf / np.prod(lam)  # :508-511
    for k in range(mix.K):
        z = (x - mix.mu[:, k]) / (mix.sigma[k] * lam)
        if gaussian:
            nn = nf * mix.w[k] / mix.sigma[k] ** D * np.exp(-0.5 * np.sum(z**2, 1))
        elif df > 0:
            nn = (
                nf
                * mix.w[k]
                / mix.sigma[k] ** D
                * (1 + np.sum(z**2, 1) / df) ** (-(df + D) / 2)
            )
        else:
            a = abs(df)
            nn = (
                nf
                * mix.w[k]
                / mix.sigma[k] ** D
                * np.prod((1 + z**2 / a) ** (-(a + 1) / 2), axis=1)
            )
        y[:, 0] += nn
        if grad_flag:
            if not gaussian:
                raise NotImplementedError(
                    "Gradient of heavy-tailed pdf not supported yet."
                )
            dy -= nn[:, None] * (x - mix.mu[:, k]) / (lam**2 * mix.sigma[k] ** 2)
    if log_flag:
        if grad_flag:
            with np.errstate(divide="ignore", invalid="ignore"):
                dy = dy / y
        out = np.full_like(y, -np.inf)
        nz = y != 0
        out[nz] = np.log(y[nz])
        y = out
    return (y, dy) if grad_flag else y


def moments(mix, cov_flag=False):
    """Analytic mean/covariance in the transformed space (:793-804)."""
    mubar = mix.mu @ mix.w
    if not cov_flag:
        return mubar.reshape(1, -1)
    cov = np.sum(mix.w * mix.sigma**2) * np.diag(mix.lambd**2)
    for k in range(mix.K):
        d = (mix.mu[:, k] - mubar)[:, None]
        cov = cov + mix.w[k] * (d @ d.T)
    return mubar.reshape(1, -1), cov
