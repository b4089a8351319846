"""``pyvbmc_amd.patch(vo)`` on the GPU: the sieve's candidates in ONE device call and ``optimize_vp``'s
stochastic objective in the device-resident loop, delivered through module rebinding alone.

The reference cannot be imported on the GPU box, so ``vo`` here is a stand-in module with the
reference's CALL SHAPES (not its code): a ``_sieve`` that builds candidate posteriors and evaluates
``_neg_elcbo(theta, gp, vp0, 0, 0, 0, compute_var, theta_bnd)`` one by one through the module global
(variational_optimization.py:775-787), and an optimiser entry that hands ``minimize_adam`` a closure
named ``vb_train_mc_fun`` over ``gp, vp0, elcbo_beta, ns_ent_K, compute_var, theta_bnd`` (:238-249).
The same patch is applied to the REAL module in the build container by
tools/check_integration_patch.py (identical candidates / order / routing against the reference
itself)."""
import copy
import types

import numpy as np
import pytest
from helpers import oracle_gp, oracle_mix

from oracle import elbo_ref
from pyvbmc_amd import synthetic

pytestmark = pytest.mark.gpu

FAKE_VO = '''
import copy
import numpy as np

entmc_vbmc = entlb_vbmc = _gp_log_joint = _neg_elcbo = None   # the leaf names the reference binds at import
minimize_adam = None


def _sieve(options, optim_state, vp, gp, init_N=None, best_N=1, K=None):
    theta_bnd = options["theta_bnd"]
    rng = np.random.default_rng(options["seed"])
    vp0_vec = np.empty(init_N, dtype=object)
    vp0_type = np.zeros(init_N, dtype=int)
    for i in range(init_N):
        v = copy.deepcopy(vp)
        v.mu = v.mu + 0.2 * rng.standard_normal(v.mu.shape)
        v.sigma = v.sigma * np.exp(0.1 * rng.standard_normal(v.sigma.shape))
        vp0_vec[i], vp0_type[i] = v, 1 + i % 3
    nelcbo_fill = np.zeros(init_N)
    compute_var = False
    for i, vp0 in enumerate(vp0_vec):
        theta = vp0.get_parameters()
        nelbo_tmp, _, _, _, varF_tmp = _neg_elcbo(theta, gp, vp0, 0, options["ns_fast"], 0, compute_var, theta_bnd)
        nelcbo_fill[i] = nelbo_tmp
    order = np.argsort(nelcbo_fill)
    return vp0_vec[order], vp0_type[order], 0, compute_var, options["ns_ent_K"], options["ns_fast"]


def optimize_stochastic(options, gp, vp0, theta0):
    elcbo_beta, compute_var = 0, False
    ns_ent_K, theta_bnd = options["ns_ent_K"], options["theta_bnd"]

    def vb_train_mc_fun(theta_):
        res = _neg_elcbo(theta_, gp, vp0, elcbo_beta, ns_ent_K, compute_grad=True, compute_var=compute_var,
                         theta_bnd=theta_bnd)
        return res[0], res[1]

    return minimize_adam(vb_train_mc_fun, theta0, tol_fun=options["tol_fun"], max_iter=options["max_iter"],
                         master_min=0.001, master_max=0.05, master_decay=200)
'''


@pytest.fixture()
def vo():
    m = types.ModuleType("fake_vo")
    exec(compile(FAKE_VO, "fake_vo", "exec"), m.__dict__)
    from pyvbmc_amd import minimize_adam as am

    m.minimize_adam = am.minimize_adam  # (the reference module imports its own host loop under this name)
    return m


@pytest.fixture(scope="module")
def problem():
    from pyvbmc_amd import VariationalPosterior, _lib
    from pyvbmc_amd import gp as gpm

    ctx = _lib.Context(0)
    _lib.set_default_context(ctx)
    wl = synthetic.make_workload(2, Ns_total=20 * 200)
    vp = VariationalPosterior(wl.D, wl.K)
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
    vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
    gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True))
    gp.update(X_new=wl.X, y_new=wl.y, hyp=wl.hyp)
    yield wl, vp, gp, synthetic.default_theta_bnd(wl), ctx
    _lib.set_default_context(None)
    ctx.close()


def test_patch_batches_the_sieve(vo, problem):
    import pyvbmc_amd
    from pyvbmc_amd import variational_optimization as avo

    wl, vp, gp, bnd, ctx = problem
    options = dict(theta_bnd=bnd, seed=4, ns_fast=0, ns_ent_K=wl.NsK)
    calls = []
    real_batch = avo._neg_elcbo_batch

    def counting_batch(thetas, *a, **k):
        calls.append(len(thetas))
        return real_batch(thetas, *a, **k)

    # (a) the four leaves only: the stand-in's own loop, one device call per candidate
    pyvbmc_amd.patch(vo, sieve=False, adam=False)
    assert vo._neg_elcbo is avo._neg_elcbo and vo._sieve.__name__ == "_sieve" and not hasattr(vo._sieve, "__wrapped__")
    per_call = vo._sieve(options, {}, copy.deepcopy(vp), gp, init_N=40)
    # (b) the whole patch: the same function, its loop deferred into one batched call
    pyvbmc_amd.patch(vo, _batch_eval=counting_batch)
    batched = vo._sieve(options, {}, copy.deepcopy(vp), gp, init_N=40)
    assert calls == [40]
    assert vo._neg_elcbo is avo._neg_elcbo  # the recorder was taken out again
    assert np.array_equal(per_call[1], batched[1]) and per_call[2:] == batched[2:]
    for a, b in zip(per_call[0], batched[0]):
        for attr in ("mu", "sigma", "lambd", "w", "eta"):
            # (to rounding: the per-call path takes the renormalised mixture back from the library's C code,
            # the deferred path applies vp0.set_parameters -- the same arithmetic in two languages)
            assert np.allclose(getattr(a, attr), getattr(b, attr), rtol=1e-14, atol=1e-300), attr
    # and the order is the oracle's
    ogp = oracle_gp(dict(X=wl.X, y=wl.y, hyp=wl.hyp, s2=np.zeros(0)))
    Fo = []
    for v in batched[0]:
        mix = oracle_mix(dict(mu=v.mu, sigma=v.sigma.ravel(), lambd=v.lambd.ravel(), w=v.w.ravel(), eta=v.eta.ravel()))
        Fo.append(elbo_ref.neg_elcbo(v.get_parameters(), ogp, mix, 0.0, 0, False, False, bnd, False)[0])
    assert np.all(np.diff(Fo) >= -1e-9 * np.abs(Fo[:-1]))
    # a Monte-Carlo sieve (ns_ent_fast > 0) is not batchable: every candidate takes the per-call path
    calls.clear()
    np.random.seed(0)
    out = vo._sieve(dict(options, ns_fast=64), {}, copy.deepcopy(vp), gp, init_N=6)
    assert calls == [] and len(out[0]) == 6
    pyvbmc_amd.unpatch(vo)
    assert vo._neg_elcbo is None and not hasattr(vo._sieve, "__wrapped__")


def test_patch_routes_the_stochastic_optimiser_to_the_device_loop(vo, problem):
    import pyvbmc_amd
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl, vp, gp, bnd, ctx = problem
    options = dict(theta_bnd=bnd, ns_ent_K=wl.NsK, tol_fun=0.01, max_iter=60)
    pyvbmc_amd.patch(vo)
    theta0 = vp.get_parameters()
    np.random.seed(21)
    ctx.__dict__.pop("_philox_seq", None)  # a fresh per-context seed sequence: the loop's seed is one np.random draw
    vp_a = copy.deepcopy(vp)
    got = vo.optimize_stochastic(options, gp, vp_a, theta0.copy())
    np.random.seed(21)
    ctx.__dict__.pop("_philox_seq", None)
    vp_b = copy.deepcopy(vp)
    want = minimize_adam_elbo(theta0.copy(), gp, vp_b, wl.NsK, bnd, 0.0, tol_fun=0.01, max_iter=60, master_min=0.001,
                              master_max=0.05, master_decay=200)
    # same seed, same loop: the routed call IS the device-resident loop
    assert got[4] == want[4] and np.array_equal(got[3], want[3]) and np.array_equal(got[2], want[2])
    assert np.array_equal(vp_a.mu, vp_b.mu)
    assert got[3][-1] < got[3][0]  # it optimised
    # an objective that is NOT optimize_vp's closure takes the host loop with the reference's semantics
    x, y, xt, yt, it = vo.minimize_adam(lambda t: (float(np.sum((t - 1.0) ** 2)), 2 * (t - 1.0)), np.zeros(4),
                                        max_iter=300, use_early_stopping=False)
    assert it == 300 and np.max(np.abs(x - 1.0)) < 0.05
    pyvbmc_amd.unpatch(vo)
