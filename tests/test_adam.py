"""the stochastic optimiser's loop (reference vbmc/minimize_adam.py).

CPU part: the oracle's restatement and the host mirror against trajectories produced by
the reference's own minimize_adam (tests/golden/adam.npz, oracle/make_golden.py adam).
GPU part: the host loop around the device objective against the reference's trajectory
(NumPy draw stream), and the device-resident loop against the oracle driven by the same
Philox draws.
"""
import pathlib

import numpy as np
import pytest
from helpers import oracle_gp, oracle_mix, rel_err

from oracle import adam_ref, elbo_ref, philox_ref
from pyvbmc_amd import synthetic
from pyvbmc_amd.minimize_adam import minimize_adam

ELBO_CASES = {"c1": (1, {}), "c2s": (2, dict(Ns_total=20 * 100))}
QUAD = {
    "box": lambda g: dict(lb=g["quad_lb"], ub=g["quad_ub"], max_iter=400),
    "free": lambda g: dict(max_iter=90, master_max=0.05, use_early_stopping=False),
    "short": lambda g: dict(max_iter=25, tol_fun=0.5),
}


def quad_objective(g):
    a, c = g["quad_a"], g["quad_c"]

    def f(x):
        wob = 0.01 * np.sin(37.0 * np.sum(x))
        return 0.5 * np.sum(a * (x - c) ** 2) + wob, a * (x - c) + 0.37 * np.cos(37.0 * np.sum(x))

    return f


def workload_dict(name):
    cfg, shrink = ELBO_CASES[name]
    wl = synthetic.make_workload(cfg, S=1, **shrink)
    g = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X,
             y=wl.y, hyp=wl.hyp[:1], s2=np.zeros(0) if wl.s2 is None else wl.s2)
    return wl, g


def check_traj(got, g, pre, tol):
    x, y, xt, yt, it = got
    assert it == int(g[f"{pre}_iters"])
    assert xt.shape == g[f"{pre}_x_tab"].shape and yt.shape == g[f"{pre}_y_tab"].shape
    assert rel_err(xt, g[f"{pre}_x_tab"]) < tol, rel_err(xt, g[f"{pre}_x_tab"])
    assert rel_err(yt, g[f"{pre}_y_tab"]) < tol, rel_err(yt, g[f"{pre}_y_tab"])
    assert rel_err(x, g[f"{pre}_x"]) < tol and abs(y - g[f"{pre}_y"]) <= tol * max(1.0, abs(g[f"{pre}_y"]))


# ---------------------------------------------------------------- CPU
@pytest.mark.parametrize("impl", [adam_ref.minimize_adam, minimize_adam], ids=["oracle", "host-mirror"])
@pytest.mark.parametrize("tag", list(QUAD))
def test_adam_quadratic_vs_reference(golden, impl, tag):
    g = golden("adam")
    x0 = g["quad_x0"].copy()
    got = impl(quad_objective(g), x0, **QUAD[tag](g))
    check_traj(got, g, f"quad_{tag}", 1e-13)
    # the reference applies its first update to the caller's array in place
    assert np.array_equal(x0, g[f"quad_{tag}_x_tab"][:, 0]) or tag == "box"


@pytest.mark.parametrize("name", list(ELBO_CASES))
def test_oracle_adam_elbo_vs_reference(golden, name):
    """oracle Adam around the oracle objective, NumPy draw stream == the reference's run."""
    g = golden("adam")
    wl, wd = workload_dict(name)
    mix, gp = oracle_mix(wd), oracle_gp(wd)
    bnd = synthetic.default_theta_bnd(wl)
    NsK = int(g[f"elbo_{name}_NsK"])
    assert NsK == wl.NsK

    def f(t):
        r = elbo_ref.neg_elcbo(t, gp, mix, 0.0, NsK, True, False, bnd)
        return r[0], r[1]

    np.random.seed(int(g[f"elbo_{name}_seed"]))
    n_it = int(g[f"elbo_{name}_iters"])
    got = adam_ref.minimize_adam(f, g[f"elbo_{name}_theta0"].copy(), tol_fun=0.05, max_iter=n_it,
                                 master_min=0.001, master_max=0.1, master_decay=200)
    check_traj(got, g, f"elbo_{name}", 1e-8)


def test_window_stop_matches_polyfit_rule():
    from pyvbmc_amd.minimize_adam import _window_stop

    rng = np.random.default_rng(0)
    y = 3.0 + 1e-4 * rng.standard_normal(20)
    xa, xb = rng.standard_normal(5), rng.standard_normal(5)
    assert _window_stop(y, xa, xa + 1e-5, 0.001)  # flat values, iterates at rest
    assert not _window_stop(y - 0.5 * np.arange(20), xa, xa + 1e-5, 0.001)  # still descending fast
    assert not _window_stop(y, xa, xb, 0.001)  # iterates still moving


# ---------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def ctx():
    from pyvbmc_amd import _lib

    c = _lib.Context(0)
    _lib.set_default_context(c)
    yield c
    _lib.set_default_context(None)
    c.close()


def device_objects(wd, ctx):
    from test_gpu_parity import make_gp, make_vp

    return make_vp(wd, ctx), make_gp(wd, ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ELBO_CASES))
def test_host_loop_device_objective_vs_reference(ctx, golden, name):
    """minimize_adam (host mirror) around the fused device objective fed by the NumPy stream
    reproduces the reference's own optimisation run."""
    from pyvbmc_amd.variational_optimization import _neg_elcbo

    g = golden("adam")
    wl, wd = workload_dict(name)
    vp, gp = device_objects(wd, ctx)
    bnd = synthetic.default_theta_bnd(wl)
    NsK = int(g[f"elbo_{name}_NsK"])

    def f(t):
        r = _neg_elcbo(t, gp, vp, 0.0, NsK, True, False, bnd, rng="numpy")
        return r[0], r[1]

    np.random.seed(int(g[f"elbo_{name}_seed"]))
    got = minimize_adam(f, g[f"elbo_{name}_theta0"].copy(), tol_fun=0.05,
                        max_iter=int(g[f"elbo_{name}_iters"]), master_min=0.001, master_max=0.1,
                        master_decay=200)
    check_traj(got, g, f"elbo_{name}", 1e-7)


def oracle_philox_run(wl, wd, theta0, bnd, seed, max_iter, mask_flags=None, **kw):
    mix, gp = oracle_mix(wd), oracle_gp(wd)
    if mask_flags is not None:
        mix.optimize_mu, mix.optimize_sigma, mix.optimize_lambd, mix.optimize_weights = mask_flags
    it = [0]

    def f(t):
        eps = philox_ref.eps_half(wl.K, wl.NsK // 2, wl.D, seed + it[0])
        it[0] += 1
        r = elbo_ref.neg_elcbo(t, gp, mix, 0.0, wl.NsK, True, False, bnd, eps_half=eps)
        return r[0], r[1]

    return adam_ref.minimize_adam(f, theta0.copy(), max_iter=max_iter, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ELBO_CASES))
def test_device_loop_vs_oracle(ctx, golden, name):
    """The device-resident loop (csrc/adam.hip) against oracle Adam around the oracle
    objective, both drawing Philox(seed + i) at iteration i."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    g = golden("adam")
    wl, wd = workload_dict(name)
    vp, gp = device_objects(wd, ctx)
    bnd = synthetic.default_theta_bnd(wl)
    theta0 = g[f"elbo_{name}_theta0"].copy()
    kw = dict(tol_fun=0.05, master_min=0.001, master_max=0.1, master_decay=200)
    n_it = 60 if name == "c1" else 45
    ref = oracle_philox_run(wl, wd, theta0, bnd, 1234, n_it, **kw)
    keep = theta0.copy()
    got = minimize_adam_elbo(theta0, gp, vp, wl.NsK, bnd, max_iter=n_it, seed=1234, rng="philox", **kw)
    assert np.array_equal(theta0, keep)  # documented: the start point is not modified
    assert got[4] == ref[4]
    assert rel_err(got[3], ref[3]) < 1e-7, rel_err(got[3], ref[3])
    assert rel_err(got[2], ref[2]) < 1e-7, rel_err(got[2], ref[2])
    assert rel_err(got[0], ref[0]) < 1e-7 and abs(got[1] - ref[1]) < 1e-7 * max(1.0, abs(ref[1]))
    # vp holds the last iterate's parameters
    from oracle import mixture_ref

    mix = oracle_mix(wd)
    mixture_ref.set_parameters(mix, got[2][:, -1].copy())
    assert rel_err(vp.mu, mix.mu) < 1e-12 and rel_err(vp.sigma.ravel(), mix.sigma.ravel()) < 1e-12
    assert rel_err(vp.w.ravel(), mix.w.ravel()) < 1e-12 and rel_err(vp.lambd.ravel(), mix.lambd.ravel()) < 1e-12


def masked_bounds(wl, flags):
    """default_theta_bnd cut down to the blocks get_bounds emits for these optimise flags
    (variational_posterior.py:205-239: mu | ln scale if sigma or lambda | eta)."""
    full = synthetic.default_theta_bnd(wl)
    DK, K = wl.D * wl.K, wl.K
    keep = np.concatenate([np.full(DK, flags[0]), np.full(DK, flags[1] or flags[2]), np.full(K, flags[3])])
    out = dict(full)
    out["lb"], out["ub"] = full["lb"][keep], full["ub"][keep]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [(True, True, True, False), (True, True, False, True), (False, True, True, True)],
                         ids=["warmup-no-weights", "no-lambda", "no-mu"])
def test_device_loop_partial_masks(ctx, flags):
    """Blocks that are not optimised (warm-up runs without the weights) keep their values."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl, wd = workload_dict("c2s")
    vp, gp = device_objects(wd, ctx)
    vp.optimize_mu, vp.optimize_sigma, vp.optimize_lambd, vp.optimize_weights = flags
    mix = oracle_mix(wd)
    mix.optimize_mu, mix.optimize_sigma, mix.optimize_lambd, mix.optimize_weights = flags
    from oracle import mixture_ref

    theta0 = mixture_ref.get_parameters(mix)
    theta0[0] += 5.0  # leave a soft bound so the penalty and its odd sigma/lambda reshape act
    bnd = masked_bounds(wl, flags)
    kw = dict(tol_fun=0.05, master_min=0.001, master_max=0.1, master_decay=200)
    ref = oracle_philox_run(wl, wd, theta0, bnd, 99, 40, mask_flags=flags, **kw)
    got = minimize_adam_elbo(theta0, gp, vp, wl.NsK, bnd, max_iter=40, seed=99, rng="philox", **kw)
    assert got[4] == ref[4]
    assert rel_err(got[2], ref[2]) < 1e-7, rel_err(got[2], ref[2])
    assert rel_err(got[3], ref[3]) < 1e-7


@pytest.mark.gpu
def test_device_loop_box_and_no_early_stop(ctx):
    """Box constraints clamp every iterate; without early stopping the loop is one enqueue."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl, wd = workload_dict("c1")
    vp, gp = device_objects(wd, ctx)
    bnd = synthetic.default_theta_bnd(wl)
    theta0 = wl.theta.copy()
    lb, ub = theta0 - 0.05, theta0 + 0.02
    kw = dict(master_min=0.001, master_max=0.1, master_decay=200, use_early_stopping=False)
    ref = oracle_philox_run(wl, wd, theta0, bnd, 7, 30, lb=lb, ub=ub, **kw)
    got = minimize_adam_elbo(theta0, gp, vp, wl.NsK, bnd, lb=lb, ub=ub, max_iter=30, seed=7, rng="philox", **kw)
    assert got[4] == 30
    assert np.all(got[2] >= lb[:, None]) and np.all(got[2] <= ub[:, None])
    assert rel_err(got[2], ref[2]) < 1e-7 and rel_err(got[3], ref[3]) < 1e-7


@pytest.mark.gpu
def test_device_loop_errors(ctx):
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl, wd = workload_dict("c1")
    vp, gp = device_objects(wd, ctx)
    with pytest.raises(ValueError):
        minimize_adam_elbo(wl.theta.copy(), gp, vp, 0)  # Adam is only used with the MC entropy
    with pytest.raises(ValueError):
        minimize_adam_elbo(wl.theta[:-1].copy(), gp, vp, wl.NsK, max_iter=20)  # wrong length
    with pytest.raises(NotImplementedError):
        minimize_adam_elbo(wl.theta.copy(), gp, vp, wl.NsK, beta=1.0)
    bad = wl.theta.copy()
    bad[0] = np.nan
    with pytest.raises(Exception):
        minimize_adam_elbo(bad, gp, vp, wl.NsK, max_iter=20)
    # the context is usable afterwards
    out = minimize_adam_elbo(wl.theta.copy(), gp, vp, wl.NsK, max_iter=20, seed=3)
    assert out[4] == 20 and np.all(np.isfinite(out[3]))


@pytest.mark.gpu
def test_device_loop_global_memory_kernels(ctx):
    """BASELINE config 5's shape (D = 20, K = 100): the working sets of the pre and step kernels no longer fit
    the LDS plan (150 KB) and both work from global memory -- the loop against oracle Adam on the same Philox draws."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl = synthetic.make_workload(5, S=1, N=60, Ns_total=100 * 24)
    assert (wl.D, wl.K) == (20, 100)
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y,
              hyp=wl.hyp, s2=wl.s2)
    vp, gp = device_objects(wd, ctx)
    bnd = synthetic.default_theta_bnd(wl)
    theta0 = wl.theta.copy()
    theta0[2] += 3.0
    kw = dict(tol_fun=0.05, master_min=0.001, master_max=0.1, master_decay=200)
    ref = oracle_philox_run(wl, wd, theta0, bnd, 13, 12, **kw)
    got = minimize_adam_elbo(theta0, gp, vp, wl.NsK, bnd, max_iter=12, seed=13, rng="philox", **kw)
    assert got[4] == ref[4]
    assert rel_err(got[2], ref[2]) < 1e-7 and rel_err(got[3], ref[3]) < 1e-7, (rel_err(got[2], ref[2]), rel_err(got[3], ref[3]))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [dict(D=20, K=12, N=60, S=3, NsK=40), dict(D=3, K=70, N=90, S=2, NsK=28),
                                   dict(D=17, K=5, N=40, S=1, NsK=130), dict(D=3, K=130, N=40, S=1, NsK=6)],
                         ids=["D20-S3", "K70-S2", "D17", "K130-standalone-pre"])
def test_device_loop_other_shapes(ctx, shape):
    """Shapes away from the BASELINE configurations: more than 16 dimensions (the 16-lane
    dimension groups of the pre workgroup and of the GP sums take a second round), several GP
    hyper-parameter samples, a component count that is not a multiple of four, the
    optimiser's default sample count (ns_ent = 100 K^(2/3), advanced_vbmc_options.ini:43), and K > 128: the generic
    entropy kernel has no extra row, so the pre workgroup runs as a launch of its own (adam.hip launch_pre)."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl = synthetic.make_workload(5, S=shape["S"], D=shape["D"], K=shape["K"], N=shape["N"],
                                 Ns_total=shape["NsK"] * shape["K"])
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y,
              hyp=wl.hyp, s2=wl.s2)  # config 5: user-provided noise
    vp, gp = device_objects(wd, ctx)
    bnd = synthetic.default_theta_bnd(wl)
    theta0 = wl.theta.copy()
    theta0[1] += 4.0  # one coordinate outside its soft bound
    kw = dict(tol_fun=0.05, master_min=0.001, master_max=0.1, master_decay=200)
    ref = oracle_philox_run(wl, wd, theta0, bnd, 31, 30, **kw)
    got = minimize_adam_elbo(theta0, gp, vp, wl.NsK, bnd, max_iter=30, seed=31, rng="philox", **kw)
    assert got[4] == ref[4]
    assert rel_err(got[2], ref[2]) < 1e-7, rel_err(got[2], ref[2])
    assert rel_err(got[3], ref[3]) < 1e-7, rel_err(got[3], ref[3])


@pytest.mark.gpu
def test_device_loop_random_shapes(ctx):
    """A seeded sweep over (D, K, N, S, NsK, iterations): the device loop against the oracle loop
    on the same Philox draws, whatever the shape does to the kernels' tiling (padded D, K not a
    multiple of 4, one antithetic pair per component, batches that end mid-window)."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    rng = np.random.default_rng(2024)
    for t in range(12):
        D, K = int(rng.integers(1, 25)), int(rng.integers(1, 60))
        N, S = int(rng.integers(5, 130)), int(rng.integers(1, 4))
        nsk, cfg = 2 * int(rng.integers(1, 90)), int(rng.choice([2, 3, 5]))
        wl = synthetic.make_workload(cfg, S=S, D=D, K=K, N=N, Ns_total=nsk * K)
        wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y,
                  hyp=wl.hyp, s2=np.zeros(0) if wl.s2 is None else wl.s2)
        vp, gp = device_objects(wd, ctx)
        bnd = synthetic.default_theta_bnd(wl)
        theta0 = wl.theta.copy()
        theta0[0] += 3.0
        kw = dict(tol_fun=0.05, master_min=0.001, master_max=0.1, master_decay=200)
        n_it = int(rng.choice([7, 20, 23, 41]))
        ref = oracle_philox_run(wl, wd, theta0, bnd, 5 + t, n_it, **kw)
        got = minimize_adam_elbo(theta0, gp, vp, wl.NsK, bnd, max_iter=n_it, seed=5 + t, rng="philox", **kw)
        shape = dict(D=D, K=K, N=N, S=S, NsK=wl.NsK, cfg=cfg, it=n_it)
        assert got[4] == ref[4], shape
        assert rel_err(got[2], ref[2]) < 1e-8 and rel_err(got[3], ref[3]) < 1e-8, shape



TAIL_SHAPES = [dict(cfg=2, D=6, K=20, N=60, S=2, NsK=9000), dict(cfg=3, D=10, K=13, N=50, S=1, NsK=15000),
               dict(cfg=5, D=4, K=32, N=40, S=3, NsK=6000)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", TAIL_SHAPES, ids=lambda s: "D%d-K%d-S%d-NsK%d" % (s["D"], s["K"], s["S"], s["NsK"]))
def test_two_launch_iteration_vs_four_launch_loop_and_oracle(ctx, shape):
    """The optimiser loop's two-launch iteration (csrc/adam.hip adam_tail_kernel: GP sums and pre workgroup inside the entropy
    launch, then reduction -> hand-off -> Adam step, pack and table rows in one tail launch): the same run as four
    launches per iteration (iterates, F, G, H) and as oracle Adam on the same Philox draws, across two batches of the host's
    stopping rule, with the run continued through a second call (the table and the draws the last tail launch left are
    the next call's), a box, and a switch of forms in mid-run."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl = synthetic.make_workload(shape["cfg"], S=shape["S"], D=shape["D"], K=shape["K"], N=shape["N"],
                                 Ns_total=shape["NsK"] * shape["K"])
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y,
              hyp=wl.hyp, s2=np.zeros(0) if wl.s2 is None else wl.s2)
    bnd = synthetic.default_theta_bnd(wl)
    theta0 = wl.theta.copy()
    theta0[1] += 3.0
    kw = dict(tol_fun=0.05, master_min=0.001, master_max=0.1, master_decay=200, use_early_stopping=False)
    n_it = 27  # one full batch of the host's rule and a partial one
    runs = {}
    try:
        for form in (2, 0):
            ctx.set_option("adam_tail", form)
            vp, gp = device_objects(wd, ctx)
            runs[form] = minimize_adam_elbo(theta0.copy(), gp, vp, wl.NsK, bnd, max_iter=n_it, seed=77, rng="philox", **kw)
            plan = ctx.last_entmc_plan()
            assert plan["adam_tail"] == (form == 2) and (plan["span"] or form == 0), plan
    finally:
        ctx.set_option("adam_tail", 1)
    a, b = runs[2], runs[0]
    assert a[4] == b[4] == n_it
    assert rel_err(a[2], b[2]) < 1e-10 and rel_err(a[3], b[3]) < 1e-10, (rel_err(a[2], b[2]), rel_err(a[3], b[3]))
    ref = oracle_philox_run(wl, wd, theta0, bnd, 77, 8, **kw)  # (the oracle takes seconds per iteration at these sizes)
    assert rel_err(a[2][:, :8], ref[2][:, :8]) < 1e-7 and rel_err(a[3][:8], ref[3][:8]) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("first", [0, 2], ids=["four-then-two", "two-then-four"])
def test_forms_switched_between_batches(ctx, first):
    """a two-launch iteration that follows a four-launch iteration in the same run (and the other way round).
    The four-launch iteration's finish and step launches make the share [0, 0.67) of the next iteration's draws; the
    two-launch iteration's prep launch has to make the rest (it used to assume its own kind had made them all, and
    read stale rows), and its table must be rebuilt after a batch in the other form.  Forms alternate batch by batch
    (batches of 20, the host's stopping rule with a tolerance it never meets) and the run must equal the all-four-launch
    run."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    shape = TAIL_SHAPES[0]
    wl = synthetic.make_workload(shape["cfg"], S=shape["S"], D=shape["D"], K=shape["K"], N=shape["N"],
                                 Ns_total=shape["NsK"] * shape["K"])
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y,
              hyp=wl.hyp, s2=np.zeros(0) if wl.s2 is None else wl.s2)
    bnd = synthetic.default_theta_bnd(wl)
    theta0 = wl.theta.copy()
    theta0[1] += 3.0
    kw = dict(tol_fun=0.0, master_min=0.001, master_max=0.1, master_decay=200, use_early_stopping=True, device_stop=False)
    n_it = 70  # three full batches and a partial one
    forms = []

    def switch(done):
        forms.append((done, ctx.last_entmc_plan()["adam_tail"]))  # the form the batch just done ran in
        ctx.set_option("adam_tail", 2 - first if (done // 20) % 2 == 1 else first)

    try:
        ctx.set_option("adam_tail", 0)
        vp, gp = device_objects(wd, ctx)
        want = minimize_adam_elbo(theta0.copy(), gp, vp, wl.NsK, bnd, max_iter=n_it, seed=91, rng="philox", **kw)
        ctx.set_option("adam_tail", first)
        vp, gp = device_objects(wd, ctx)
        got = minimize_adam_elbo(theta0.copy(), gp, vp, wl.NsK, bnd, max_iter=n_it, seed=91, rng="philox",
                                 _between_batches=switch, **kw)
    finally:
        ctx.set_option("adam_tail", 1)
    # the batches really ran in alternating forms
    assert [f for _, f in forms] == [first == 2, first != 2, first == 2, first != 2], forms
    assert got[4] == want[4] == n_it
    assert rel_err(got[2], want[2]) < 1e-10 and rel_err(got[3], want[3]) < 1e-10, (rel_err(got[2], want[2]), rel_err(got[3], want[3]))


FUSED_SHAPES = [dict(cfg=3, D=10, K=50, N=400, S=1, NsK=28), dict(cfg=5, D=16, K=12, N=60, S=3, NsK=40),
                dict(cfg=2, D=4, K=20, N=200, S=2, NsK=22), dict(cfg=3, D=7, K=64, N=90, S=1, NsK=2),
                dict(cfg=5, D=11, K=1, N=50, S=1, NsK=128), dict(cfg=3, D=6, K=20, N=60, S=8, NsK=28),
                # the widest builds at full register pressure (<12>, <16>: 0 B of scratch since round 4), theta near
                # its 1 024-entry limit, and the largest training set whose LDS plan fits at D = 10
                dict(cfg=3, D=12, K=40, N=300, S=2, NsK=28), dict(cfg=3, D=16, K=30, N=100, S=1, NsK=6),
                dict(cfg=3, D=10, K=50, N=800, S=1, NsK=28),
                # round 5: the builds for D = 17 .. 24 (theta beyond 1 024 entries: three per thread; the per-dimension sums in
                # two rounds of 16-lane groups), X^T resident (N = 200) and read from memory (N = 400 at D = 20; N = 2000 at
                # D = 10: beyond the old plan's N ~ 900)
                dict(cfg=5, D=20, K=50, N=400, S=1, NsK=22), dict(cfg=5, D=20, K=24, N=200, S=2, NsK=28),
                dict(cfg=3, D=24, K=40, N=150, S=1, NsK=10), dict(cfg=3, D=18, K=64, N=100, S=1, NsK=28),
                dict(cfg=3, D=10, K=50, N=2000, S=1, NsK=28),
                # round 5: more than 64 rows per component, split over R workgroups per component whose partial records the
                # gather adds up -- two slices (65 rows), four (160 rows), three at K = 64 with a ragged last one (151 rows)
                dict(cfg=3, D=10, K=50, N=400, S=1, NsK=130), dict(cfg=3, D=10, K=50, N=400, S=1, NsK=320),
                dict(cfg=2, D=6, K=64, N=100, S=2, NsK=302)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", FUSED_SHAPES, ids=lambda s: "D%d-K%d-S%d-NsK%d" % (s["D"], s["K"], s["S"], s["NsK"]))
def test_fused_loop_vs_four_launch_loop_and_oracle(ctx, shape):
    """The loop at the reference's own sample counts (ns_ent = 100 K^(2/3) in total, advanced_vbmc_options.ini:43:
    NsK = 28 at K = 50) as ONE launch per batch (csrc/adam_fused.hip) against the four-launch iteration and
    against oracle Adam on the same Philox draws: the reference's default shape, three GP hyper-parameter
    samples with the quadratic mean, more GP blocks than GP workgroups (S K > 128), K = 64 lanes full, K = 1,
    one antithetic pair per component, 64 rows per component (the fused loop's limit)."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl = synthetic.make_workload(shape["cfg"], S=shape["S"], D=shape["D"], K=shape["K"], N=shape["N"],
                                 Ns_total=shape["NsK"] * shape["K"])
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y,
              hyp=wl.hyp, s2=np.zeros(0) if wl.s2 is None else wl.s2)
    bnd = synthetic.default_theta_bnd(wl)
    theta0 = wl.theta.copy()
    theta0[0] += 4.0  # one coordinate outside its soft bound
    kw = dict(tol_fun=1e-9, master_min=0.001, master_max=0.1, master_decay=200)
    n_it = shape.get("n_it", 47)  # two full batches and a short one
    runs = {}
    for fused in (1, 0):
        ctx.set_option("adam_fused", fused)
        vp, gp = device_objects(wd, ctx)
        runs[fused] = minimize_adam_elbo(theta0, gp, vp, wl.NsK, bnd, max_iter=n_it, seed=77, rng="philox",
                                         return_parts=True, **kw)
        assert (ctx.last_entmc_plan()["kernel"] == "adam_fused") == bool(fused), ctx.last_entmc_plan()
        runs[fused, "vp"] = vp
    ctx.set_option("adam_fused", 1)
    a, b = runs[1], runs[0]
    assert a[4] == b[4] == n_it
    # the two device loops agree far inside what either owes the oracle (summation orders differ)
    assert rel_err(a[2], b[2]) < 1e-10 and rel_err(a[3], b[3]) < 1e-10, (rel_err(a[2], b[2]), rel_err(a[3], b[3]))
    assert rel_err(a[5], b[5]) < 1e-10 and rel_err(a[6], b[6]) < 1e-9  # G and H of every iteration
    for name in ("mu", "sigma", "lambd", "w"):
        assert rel_err(np.ravel(getattr(runs[1, "vp"], name)), np.ravel(getattr(runs[0, "vp"], name))) < 1e-10, name
    ref = oracle_philox_run(wl, wd, theta0, bnd, 77, n_it, **kw)
    assert a[4] == ref[4]
    assert rel_err(a[2], ref[2]) < 1e-7 and rel_err(a[3], ref[3]) < 1e-7, (rel_err(a[2], ref[2]), rel_err(a[3], ref[3]))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [(True, True, True, False), (True, True, False, True), (False, True, True, True)],
                         ids=["warmup-no-weights", "no-lambda", "no-mu"])
def test_fused_loop_partial_masks_box_and_resident_draws(ctx, flags):
    """The fused loop with blocks that are not optimised, with box constraints, and with ONE resident set of
    NumPy-stream draws reused by every iteration (rng="numpy"), each against the four-launch iteration."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl = synthetic.make_workload(3, S=2, D=6, K=18, N=70, Ns_total=18 * 30)
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y,
              hyp=wl.hyp, s2=np.zeros(0))
    mix = oracle_mix(wd)
    mix.optimize_mu, mix.optimize_sigma, mix.optimize_lambd, mix.optimize_weights = flags
    from oracle import mixture_ref

    theta0 = mixture_ref.get_parameters(mix)
    theta0[0] += 5.0
    bnd = masked_bounds(wl, flags)
    lb, ub = theta0 - 0.3, theta0 + 0.2
    for extra in (dict(seed=3, rng="philox"), dict(rng="numpy"), dict(seed=3, rng="philox", lb=lb, ub=ub)):
        out = {}
        for fused in (1, 0):
            ctx.set_option("adam_fused", fused)
            vp, gp = device_objects(wd, ctx)
            vp.optimize_mu, vp.optimize_sigma, vp.optimize_lambd, vp.optimize_weights = flags
            if extra.get("rng") == "numpy":
                np.random.seed(11)
            out[fused] = minimize_adam_elbo(theta0, gp, vp, wl.NsK, bnd, max_iter=40, use_early_stopping=False, **extra)
            assert (ctx.last_entmc_plan()["kernel"] == "adam_fused") == bool(fused)
        ctx.set_option("adam_fused", 1)
        assert rel_err(out[1][2], out[0][2]) < 1e-10 and rel_err(out[1][3], out[0][3]) < 1e-10, (flags, extra.keys())
        if "lb" in extra:
            assert np.all(out[1][2] >= lb[:, None]) and np.all(out[1][2] <= ub[:, None])


@pytest.mark.gpu
def test_fused_loop_applies_only_to_its_shapes(ctx):
    """K > 64, D > 24, more than 160 rows per component, a row slice (virtual rank) or an LDS plan that does not
    fit even with X^T left in memory keep the four-launch iteration; a non-finite iterate is reported as before."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    for kwargs, fused in ((dict(D=10, K=50, N=400, NsK=28), True), (dict(D=10, K=50, N=800, NsK=28), True),
                          (dict(D=10, K=65, N=100, NsK=28), False),
                          (dict(D=17, K=10, N=100, NsK=28), True), (dict(D=25, K=10, N=100, NsK=28), False),
                          (dict(D=10, K=20, N=100, NsK=130), True), (dict(D=10, K=20, N=100, NsK=322), False),
                          (dict(D=16, K=40, N=1200, NsK=28), True), (dict(D=24, K=64, N=9000, NsK=28), False)):
        wl = synthetic.make_workload(3, S=1, D=kwargs["D"], K=kwargs["K"], N=kwargs["N"], Ns_total=kwargs["NsK"] * kwargs["K"])
        wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y,
                  hyp=wl.hyp, s2=np.zeros(0))
        vp, gp = device_objects(wd, ctx)
        out = minimize_adam_elbo(wl.theta.copy(), gp, vp, wl.NsK, synthetic.default_theta_bnd(wl), max_iter=20, seed=1)
        assert np.all(np.isfinite(out[3]))
        assert (ctx.last_entmc_plan()["kernel"] == "adam_fused") == fused, (kwargs, ctx.last_entmc_plan())
    # a row slice: one virtual rank's share
    wl = synthetic.make_workload(3, S=1, D=10, K=50, N=400, Ns_total=50 * 28)
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y, hyp=wl.hyp,
              s2=np.zeros(0))
    vp, gp = device_objects(wd, ctx)
    minimize_adam_elbo(wl.theta.copy(), gp, vp, wl.NsK, synthetic.default_theta_bnd(wl), max_iter=20, seed=1, rows=(0, 7))
    assert ctx.last_entmc_plan()["kernel"] != "adam_fused"


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [FUSED_SHAPES[0], FUSED_SHAPES[1], FUSED_SHAPES[5]],
                         ids=lambda s: "D%d-K%d-S%d" % (s["D"], s["K"], s["S"]))
def test_fused_loop_release_acquire_exchange(ctx, shape):
    """Option adam_fused = 3: the flags of the loop's exchange as agent-scope release stores and acquire fences
    (csrc/adam_fused.hip, exchange) instead of write-through stores + relaxed flags.  The fallback form computes the
    same numbers from the same records: iterates, objective values and the stopping decision, bit for bit."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl = synthetic.make_workload(shape["cfg"], S=shape["S"], D=shape["D"], K=shape["K"], N=shape["N"],
                                 Ns_total=shape["NsK"] * shape["K"])
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y,
              hyp=wl.hyp, s2=np.zeros(0) if wl.s2 is None else wl.s2)
    bnd = synthetic.default_theta_bnd(wl)
    out = {}
    try:
        for mode in (1, 3):
            ctx.set_option("adam_fused", mode)
            for dev in (False, True):
                vp, gp = device_objects(wd, ctx)
                out[mode, dev] = minimize_adam_elbo(wl.theta.copy(), gp, vp, wl.NsK, bnd, max_iter=90, seed=9, rng="philox",
                                                    tol_fun=0.05, device_stop=dev)
                assert ctx.last_entmc_plan()["kernel"] == "adam_fused"
    finally:
        ctx.set_option("adam_fused", 1)
    for dev in (False, True):
        a, b = out[1, dev], out[3, dev]
        assert a[4] == b[4] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[0], b[0])


@pytest.mark.gpu
def test_fused_loop_bounded_wait(ctx):
    """Every spin of the fused loop is bounded: made to wait for a workgroup that does not exist (test hook), the
    launch gives up after its 20 ms limit -- leaving the state of the start of the batch untouched -- and the call
    runs that batch, and the rest of the optimisation, as four launches per iteration: same iterates."""
    import time

    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl = synthetic.make_workload(3, S=1, D=10, K=50, N=400, Ns_total=50 * 28)
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y, hyp=wl.hyp,
              s2=np.zeros(0))
    bnd = synthetic.default_theta_bnd(wl)
    kw = dict(max_iter=60, seed=1, tol_fun=1e-9)
    out = {}
    for mode in (0, 2, 1):
        ctx.set_option("adam_fused", mode)
        vp, gp = device_objects(wd, ctx)
        t0 = time.perf_counter()
        out[mode] = minimize_adam_elbo(wl.theta.copy(), gp, vp, wl.NsK, bnd, **kw)
        assert time.perf_counter() - t0 < 5.0
        assert (ctx.last_entmc_plan()["kernel"] == "adam_fused") == (mode == 1), (mode, ctx.last_entmc_plan())
    ctx.set_option("adam_fused", 1)
    assert out[2][4] == out[0][4] == out[1][4] == 60
    assert np.array_equal(out[2][2], out[0][2]) and np.array_equal(out[2][3], out[0][3])  # the same four-launch arithmetic
    assert rel_err(out[1][2], out[0][2]) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("tol_fun,step", [(0.05, 0.1), (50.0, 1e-4), (0.5, 1e-4), (0.004, 1e-4)])
def test_fused_loop_applies_the_stopping_rule_itself(ctx, tol_fun, step):
    """vbmc_adam_run_auto: the workgroups of the one-launch form apply minimize_adam's stopping rule (minimize_adam.py:107-140)
    every 20 iterations themselves.  Same number of iterations and the same iterates, bit for bit, as the same kernel run
    in batches of 20 with the rule on the host (np.polyfit), and the iteration count of oracle Adam."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl = synthetic.make_workload(3, S=1, D=10, K=50, N=400, Ns_total=50 * 28)
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y, hyp=wl.hyp,
              s2=np.zeros(0))
    bnd = synthetic.default_theta_bnd(wl)
    sched = dict(master_min=step / 100, master_max=step, master_decay=200)  # small steps: iterates nearly at rest
    kw = dict(tol_fun=tol_fun, max_iter=170, seed=21, rng="philox", **sched)
    out = {}
    for dev in (True, False):
        vp, gp = device_objects(wd, ctx)
        out[dev] = minimize_adam_elbo(wl.theta.copy(), gp, vp, wl.NsK, bnd, device_stop=dev, **kw)
        assert ctx.last_entmc_plan()["kernel"] == "adam_fused"
        out[dev, "vp"] = vp
    a, b = out[True], out[False]
    assert a[4] == b[4], (a[4], b[4])
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[0], b[0]) and a[1] == b[1]
    assert np.array_equal(out[True, "vp"].mu, out[False, "vp"].mu) and np.array_equal(out[True, "vp"].w, out[False, "vp"].w)
    ref = oracle_philox_run(wl, wd, wl.theta.copy(), bnd, 21, 170, tol_fun=tol_fun, **sched)
    assert a[4] == ref[4], (a[4], ref[4])
    assert a[4] % 20 == 0 or a[4] == 170
    print(f"tol_fun={tol_fun} step={step}: stopped after {a[4]} iterations")


@pytest.mark.gpu
def test_fused_loop_soak_is_deterministic(ctx):
    """VBMC_FUSED_SOAK_S seconds (default 6) of fused-loop optimisations at three shapes while a second context on
    another thread keeps the same GPU busy with full-size ELBO evaluations (whose 500-workgroup entropy launches
    delay, interleave with and take CUs from the loop's resident workgroups).  The loop's workgroups exchange their
    results through write-through stores, flags and sc1 loads, without fences: a record read before it has landed
    would change an iterate by an ulp or more -- so every repeat of a run must reproduce the first one bit for bit,
    whatever the interference (about 2 000 exchanges per run, each of ~100 records read by ~100 workgroups)."""
    import os
    import threading
    import time

    from pyvbmc_amd import _lib
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo
    from pyvbmc_amd.variational_optimization import _neg_elcbo

    shapes = [dict(D=10, K=50, N=400, S=1, NsK=28), dict(D=6, K=20, N=150, S=8, NsK=38), dict(D=16, K=7, N=60, S=3, NsK=2)]
    runs = []
    for sh in shapes:
        wl = synthetic.make_workload(3, S=sh["S"], D=sh["D"], K=sh["K"], N=sh["N"], Ns_total=sh["NsK"] * sh["K"])
        wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y, hyp=wl.hyp,
                  s2=np.zeros(0))
        runs.append((wl, wd, synthetic.default_theta_bnd(wl)))
    stop = threading.Event()
    other_n = [0]
    err = []

    def disturb():
        try:
            c2 = _lib.Context(0)
            wl = synthetic.make_workload(3, S=1)
            wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y, hyp=wl.hyp,
                      s2=np.zeros(0))
            vp2, gp2 = device_objects(wd, c2)
            th = wl.theta.copy()
            first = None
            while not stop.is_set():
                out = _neg_elcbo(th, gp2, vp2, 0.0, wl.NsK, True, False, None, rng="philox", seed=3, ctx=c2)
                if first is None:
                    first = (out[0], out[1].copy())
                elif not (first[0] == out[0] and np.array_equal(first[1], out[1])):
                    err.append("the disturbing context's own evaluation changed")
                    break
                other_n[0] += 1
            c2.close()
        except Exception as e:  # noqa: BLE001
            err.append(repr(e))

    th2 = threading.Thread(target=disturb)
    th2.start()
    first, n_runs, n_iter, n_fallback = {}, 0, 0, 0
    soak_s = float(os.environ.get("VBMC_FUSED_SOAK_S", "6"))
    t0 = time.time()
    try:
        while time.time() - t0 < soak_s and not err:
            for i, (wl, wd, bnd) in enumerate(runs):
                vp, gp = device_objects(wd, ctx)
                out = minimize_adam_elbo(wl.theta.copy(), gp, vp, wl.NsK, bnd, max_iter=200, tol_fun=1e-12, seed=5 + i,
                                         rng="philox", ctx=ctx)
                # (a launch whose workgroups are not all dispatched within its bound -- the other context's launches can
                # keep the CUs' LDS taken -- gives up and the run continues as four launches per iteration: the same
                # numbers to ~1e-10, in another summation order.  Such a run is compared with its own kind.)
                kern = ctx.last_entmc_plan()["kernel"]
                n_fallback += kern != "adam_fused"
                key = (out[2].tobytes(), out[3].tobytes(), np.asarray(vp.mu).tobytes(), np.asarray(vp.w).tobytes())
                if (i, kern) not in first:
                    first[i, kern] = key
                else:
                    assert key == first[i, kern], (i, kern, n_runs)
                n_runs += 1
                n_iter += out[4]
    finally:
        stop.set()
        th2.join(timeout=60)
    assert not err, err
    print(f"{n_runs} runs ({n_fallback} of them gave the one-launch form up), {n_iter} iterations, {other_n[0]} full-size "
          f"evaluations on the other context meanwhile")
    assert n_runs >= 6 and other_n[0] > 100 and n_fallback <= n_runs // 4
