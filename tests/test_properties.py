"""Property tests (hypothesis) of the host-side logic: no GPU needed."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import elbo_ref, mixture_ref
from pyvbmc_amd import VariationalPosterior, comm
from pyvbmc_amd.variational_optimization import _soft_bound_loss, _vp_bound_loss
from pyvbmc_amd.variational_posterior import kl_div_mvn

dims = st.integers(min_value=1, max_value=6)
comps = st.integers(min_value=1, max_value=7)
flag4 = st.tuples(st.booleans(), st.booleans(), st.booleans(), st.booleans()).filter(any)


def random_vp(D, K, flags, seed):
    rng = np.random.default_rng(seed)
    vp = VariationalPosterior(D, K)
    vp.mu = rng.standard_normal((D, K))
    vp.sigma = np.exp(0.3 * rng.standard_normal((1, K)))
    vp.lambd = np.exp(0.3 * rng.standard_normal((D, 1)))
    w = rng.dirichlet(np.ones(K))
    vp.w, vp.eta = w.reshape(1, -1), np.log(w).reshape(1, -1)
    vp.optimize_mu, vp.optimize_sigma, vp.optimize_lambd, vp.optimize_weights = flags
    mix = mixture_ref.Mixture.make(vp.mu, vp.sigma.ravel(), vp.lambd.ravel(), vp.w.ravel(), vp.eta.ravel())
    mix.optimize_mu, mix.optimize_sigma, mix.optimize_lambd, mix.optimize_weights = flags
    return vp, mix, rng


@settings(max_examples=60, deadline=None)
@given(dims, comps, flag4, st.integers(0, 10**6))
def test_parameter_round_trip_and_bound_loss_match_oracle(D, K, flags, seed):
    vp, mix, rng = random_vp(D, K, flags, seed)
    theta = vp.get_parameters()
    assert np.allclose(theta, mixture_ref.get_parameters(mix), rtol=0, atol=1e-13)
    n_expected = D * K * flags[0] + K * flags[1] + D * flags[2] + K * flags[3]
    assert theta.size == n_expected
    theta2 = theta + 0.2 * rng.standard_normal(theta.size)
    vp.set_parameters(theta2)
    mixture_ref.set_parameters(mix, theta2)
    assert np.allclose(vp.mu, mix.mu) and np.allclose(vp.sigma.ravel(), mix.sigma.ravel())
    assert np.allclose(vp.lambd.ravel(), mix.lambd.ravel()) and np.allclose(vp.w.ravel(), mix.w.ravel())
    assert np.isclose(np.sum(vp.lambd**2) / D, 1.0) and np.isclose(vp.w.sum(), 1.0)
    # soft bounds in the layout get_bounds emits for these flags
    n_b = D * K * flags[0] + D * K * (flags[1] or flags[2]) + K * flags[3]
    lb, ub = -1.0 + 0.5 * rng.standard_normal(n_b), 1.0 + 0.5 * rng.standard_normal(n_b)
    lo, hi = np.minimum(lb, ub) - 0.1, np.maximum(lb, ub) + 0.1
    bnd = dict(lb=lo, ub=hi, tol_con=0.01)
    th = vp.get_parameters()
    L, dL = _vp_bound_loss(vp, th, bnd, tol_con=0.01)
    Lo, dLo = elbo_ref.vp_bound_loss(mix, th, bnd, tol_con=0.01)
    assert np.isclose(L, Lo, rtol=1e-12, atol=1e-12) and dL.shape == dLo.shape == th.shape
    assert np.allclose(dL, dLo, rtol=1e-12, atol=1e-10)


@settings(max_examples=100, deadline=None)
@given(st.lists(st.floats(-50, 50), min_size=1, max_size=12), st.floats(1e-3, 0.5))
def test_soft_bound_loss_properties(xs, tol):
    x = np.array(xs)
    lb, ub = np.full(x.size, -3.0), np.full(x.size, 5.0)
    L, dL = _soft_bound_loss(x, lb, ub, tol_con=tol, compute_grad=True)
    inside = (x >= lb) & (x <= ub)
    assert L >= 0 and np.all(dL[inside] == 0)
    assert (L == 0) == bool(np.all(inside))
    assert np.all(dL[x > ub] > 0) and np.all(dL[x < lb] < 0)  # pushes back towards the box
    ell = (ub - lb) * tol
    assert np.isclose(L, 0.5 * np.sum((np.clip(lb - x, 0, None) / ell) ** 2 + (np.clip(x - ub, 0, None) / ell) ** 2))


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 10**7), st.integers(1, 64))
def test_shard_rows_partition(n_half, world):
    edges = [comm.shard_rows(n_half, r, world) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == n_half
    assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    sizes = [e - b for b, e in edges]
    assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 0


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 5), st.integers(0, 10**6))
def test_kl_div_mvn_properties(D, seed):
    rng = np.random.default_rng(seed)
    A, B = rng.standard_normal((D, D)), rng.standard_normal((D, D))
    S1, S2 = A @ A.T + np.eye(D), B @ B.T + np.eye(D)
    m1, m2 = rng.standard_normal(D), rng.standard_normal(D)
    kl = kl_div_mvn(m1, S1, m2, S2)
    assert kl.shape == (2,) and np.all(kl >= -1e-10)
    assert np.allclose(kl_div_mvn(m1, S1, m1, S1), 0.0, atol=1e-9)
    assert np.allclose(kl, kl_div_mvn(m2, S2, m1, S1)[::-1])  # forward of one pair = reverse of the other
