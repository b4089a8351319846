"""GPU side of tests/golden/variants.npz and is_known.npz (reference-generated, see
tests/test_variants_oracle.py): every kernel / host finaliser that carries a GP-mean-kind branch
-- ``_gp_log_joint`` (csrc/api_gp.hip glj_finalize), ``gp.predict`` (gp.hip predict finish), the
acquisition tail (api_acq.hip), the sieve batch (api_batch.hip) and both forms of the optimiser
loop (adam_dev.h, adam_fused.hip) -- with ZeroMean / ConstantMean / NegativeQuadratic; avg_flag and
jacobian_flag off; the partial optimise masks of the warm-up (weights off) and of
``variable_means = False`` through the fused objective; and the orig-space density with a bounded
transformer.  Run with ``-m gpu``.
"""
from types import SimpleNamespace

import numpy as np
import pytest
from helpers import oracle_mix, rel_err
from test_variants_oracle import KINDS, MASKS, full_bnd, kind_gp, kind_hyp, mask_bnd, masked_mix, variant_transformer

from oracle import acq_ref, adam_ref, elbo_ref, entropy_ref, gp_ref, mixture_ref, philox_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pyvbmc_amd import _lib

    c = _lib.Context(0)
    _lib.set_default_context(c)
    yield c
    _lib.set_default_context(None)
    c.close()


def dev_vp(g, ctx, flags=None):
    from test_gpu_parity import make_vp

    vp = make_vp(g, ctx)
    if flags is not None:
        vp.optimize_mu, vp.optimize_sigma, vp.optimize_lambd, vp.optimize_weights = map(bool, flags)
    return vp


def dev_gp(g, ctx, kind, rows=slice(None)):
    from pyvbmc_amd import gp as gpm

    mean = {"zero": gpm.ZeroMean, "const": gpm.ConstantMean, "negquad": gpm.NegativeQuadratic}[kind]()
    gp = gpm.GP(int(g["D"]), gpm.SquaredExponential(), mean, gpm.GaussianNoise(constant_add=True))
    gp.ctx = ctx
    gp.update(X_new=g["X"], y_new=g["y"], hyp=kind_hyp(g, kind, rows))
    return gp


@pytest.mark.parametrize("kind", list(KINDS))
def test_gp_log_joint_mean_kinds_vs_reference(ctx, golden, kind):
    from pyvbmc_amd.variational_optimization import _gp_log_joint

    g = golden("variants")
    sf2 = float(np.exp(2 * g["hyp"][0, int(g["D"])]))
    for tag, rows in (("S1", slice(0, 1)), ("SM", slice(None))):
        gp = dev_gp(g, ctx, kind, rows)
        for avg in (True, False):
            for jac in (True, False):
                G, dG, _, _, _ = _gp_log_joint(dev_vp(g, ctx), gp, True, avg, jac, False, False)
                k = f"glj_{kind}_{tag}_a{int(avg)}_j{int(jac)}"
                assert np.shape(G) == g[k + "_G"].shape and dG.shape == g[k + "_dG"].shape, k
                assert rel_err(G, g[k + "_G"]) < 1e-10 and rel_err(dG, g[k + "_dG"]) < 1e-9, k
            G, _, varG, _, var_ss, I_sk, J_sjk = _gp_log_joint(dev_vp(g, ctx), gp, False, avg, True, True, True)
            k = f"glj_{kind}_{tag}_a{int(avg)}_var"
            assert np.shape(G) == g[k + "_G"].shape and np.shape(varG) == g[k + "_varG"].shape, k
            assert rel_err(G, g[k + "_G"]) < 1e-10 and rel_err(I_sk, g[k + "_I_sk"]) < 1e-10, k
            # variances are differences of O(sf^2) quantities: asserted on that scale
            assert np.max(np.abs(J_sjk - g[k + "_J_sjk"])) < 1e-10 * sf2, k
            assert np.max(np.abs(np.asarray(varG) - g[k + "_varG"])) < 1e-10 * sf2, k
            assert abs(var_ss - g[k + "_var_ss"]) < 1e-10 * sf2, k


@pytest.mark.parametrize("kind", list(KINDS))
def test_neg_elcbo_mean_kinds_vs_reference(ctx, golden, kind):
    """The fused objective (one vbmc_neg_elcbo call) with every mean kind, S = 1 and S = 3,
    Monte-Carlo (the reference's NumPy stream) and lower-bound entropy."""
    from pyvbmc_amd.variational_optimization import _neg_elcbo

    g = golden("variants")
    NsK, seed, bnd = int(g["NsK"]), int(g["seed"]), full_bnd(g)
    for gtag, rows in (("S1", slice(0, 1)), ("SM", slice(None))):
        gp = dev_gp(g, ctx, kind, rows)
        for ns_tag, Ns in (("mc", NsK), ("lb", 0)):
            vp = dev_vp(g, ctx)
            np.random.seed(seed)
            F, dF, G, H, _ = _neg_elcbo(g["theta"].copy(), gp, vp, 0.0, Ns, True, False, bnd, rng="numpy")
            k = f"elbo_{kind}_{gtag}_{ns_tag}"
            assert abs(F - g[k + "_F"]) <= 1e-10 * abs(F) and rel_err(dF, g[k + "_dF"]) < 1e-9, k
            assert abs(G - g[k + "_G"]) <= 1e-10 * abs(G) and abs(H - g[k + "_H"]) <= 1e-10 * abs(H), k


@pytest.mark.parametrize("kind", list(KINDS))
def test_predict_and_acquisition_mean_kinds(ctx, golden, kind):
    """predict's finish kernel (mean function added on the device), the small-batch acquisition
    tail (M = 40: one launch, wave per point) and the large-batch path (M = 600)."""
    from pyvbmc_amd import acquisition

    g = golden("variants")
    D = int(g["D"])
    sf2 = float(np.exp(2 * g["hyp"][:, D]).max())
    gp = dev_gp(g, ctx, kind)
    vp = dev_vp(g, ctx)
    fmu, fs2 = gp.predict(g["Xs"], separate_samples=True)
    assert np.max(np.abs(fmu - g[f"pred_{kind}_fmu"])) < 1e-10 * max(1.0, sf2)
    assert np.max(np.abs(fs2 - g[f"pred_{kind}_fs2"])) < 1e-10 * max(1.0, sf2)
    fb, v = gp.predict(g["Xs"])
    ogp = kind_gp(g, kind)
    fbo, vo = gp_ref.predict(ogp, g["Xs"])
    assert np.max(np.abs(fb - fbo)) < 1e-10 * max(1.0, sf2) and np.max(np.abs(v - vo)) < 1e-10 * max(1.0, sf2)
    length = np.exp(g["hyp"][0, :D])
    gp.temporary_data["X_rescaled"] = g["X"] / length
    flog = SimpleNamespace(y_max=float(np.max(g["y"])))
    st = dict(integer_vars=None, lb_eps_orig=g["X"].min(0) - 2.0, ub_eps_orig=g["X"].max(0) + 2.0,
              gp_length_scale=length, variance_regularized_acq_fcn=False)
    for name in ("AcqFcn", "AcqFcnLog"):
        val = getattr(acquisition, name)()(g["Xs"].copy(), gp, vp, flog, dict(st))
        ref = g[f"acq_{kind}_{name}"]
        assert np.array_equal(np.isinf(val), np.isinf(ref))
        fin = ~np.isinf(ref)
        assert np.max(np.abs(val[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin]))) < 1e-8, (kind, name)
    # a batch beyond the small-batch path: predict finish + combine kernels
    rng = np.random.default_rng(17)
    Xb = g["mu"].T[rng.integers(0, int(g["K"]), size=600)] + rng.standard_normal((600, D))
    val = acquisition.AcqFcnLog()(Xb.copy(), gp, vp, flog, dict(st))
    ref = acq_ref.acq_call(acq_ref.LOG, Xb, ogp, oracle_mix(g), flog.y_max, st)
    fin = ~np.isinf(ref)
    assert np.array_equal(np.isinf(val), ~fin)
    assert np.max(np.abs(val[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin]))) < 1e-8


@pytest.mark.parametrize("kind", list(KINDS))
def test_sieve_batch_mean_kinds(ctx, golden, kind):
    """vbmc_neg_elcbo_batch (api_batch.hip: the G / F launch carries the mean-kind branch): row 0 is the
    fixture's theta (the reference's value), the others against the oracle candidate by candidate."""
    from pyvbmc_amd.variational_optimization import _neg_elcbo_batch

    g = golden("variants")
    bnd = full_bnd(g)
    rng = np.random.default_rng(3)
    B = 37
    thetas = g["theta"][None, :] + 0.2 * rng.standard_normal((B, g["theta"].size))
    thetas[0] = g["theta"]
    thetas[5, 0] = bnd["ub"][0] + 0.4
    for gtag, rows in (("S1", slice(0, 1)), ("SM", slice(None))):
        gp, ogp = dev_gp(g, ctx, kind, rows), kind_gp(g, kind, rows)
        F, G, H = _neg_elcbo_batch(thetas, gp, dev_vp(g, ctx), bnd, return_parts=True)
        k = f"elbo_{kind}_{gtag}_lb"
        assert abs(F[0] - g[k + "_F"]) <= 1e-10 * abs(F[0]) and abs(G[0] - g[k + "_G"]) <= 1e-10 * abs(G[0])
        assert abs(H[0] - g[k + "_H"]) <= 1e-10 * abs(H[0])
        for b in range(B):
            Fo, _, Go, Ho, _ = elbo_ref.neg_elcbo(thetas[b].copy(), ogp, oracle_mix(g), 0.0, 0, False, False, bnd)
            assert abs(F[b] - Fo) <= 1e-10 * abs(Fo) and abs(G[b] - Go) <= 1e-10 * abs(Go), (kind, gtag, b)


@pytest.mark.parametrize("kind", list(KINDS))
def test_optimiser_loops_mean_kinds(ctx, golden, kind):
    """(1) the host Adam loop around the fused device objective on the NumPy stream against the REFERENCE's
    trajectory for this mean kind; (2) the device-resident loop in both forms -- four launches per iteration
    (adam_dev.h) and one launch per batch (adam_fused.hip) -- against oracle Adam on the same Philox draws,
    S = 1 and S = 3."""
    from pyvbmc_amd.minimize_adam import minimize_adam, minimize_adam_elbo
    from pyvbmc_amd.variational_optimization import _neg_elcbo

    g = golden("variants")
    bnd = full_bnd(g)
    K, D = int(g["K"]), int(g["D"])
    vp, gp = dev_vp(g, ctx), dev_gp(g, ctx, kind, slice(0, 1))

    def f(t):
        r = _neg_elcbo(t, gp, vp, 0.0, 40, True, False, bnd, rng="numpy")
        return r[0], r[1]

    np.random.seed(70)
    x, y, xt, yt, it = minimize_adam(f, g[f"adam_{kind}_theta0"].copy(), tol_fun=0.05, max_iter=30, master_min=0.001,
                                     master_max=0.1, master_decay=200)
    assert it == int(g[f"adam_{kind}_iters"])
    assert rel_err(xt, g[f"adam_{kind}_x_tab"]) < 1e-7 and rel_err(yt, g[f"adam_{kind}_y_tab"]) < 1e-7
    kw = dict(tol_fun=1e-9, master_min=0.001, master_max=0.1, master_decay=200)
    theta0 = g[f"adam_{kind}_theta0"]
    NsK, n_it, seed = 28, 45, 4711
    for rows in (slice(0, 1), slice(None)):
        ogp = kind_gp(g, kind, rows)
        mix = oracle_mix(g)
        cnt = [0]

        def fo(t):
            eps = philox_ref.eps_half(K, NsK // 2, D, seed + cnt[0])
            cnt[0] += 1
            r = elbo_ref.neg_elcbo(t, ogp, mix, 0.0, NsK, True, False, bnd, eps_half=eps)
            return r[0], r[1]

        ref = adam_ref.minimize_adam(fo, theta0.copy(), max_iter=n_it, **kw)
        for fused in (1, 0):
            ctx.set_option("adam_fused", fused)
            try:
                got = minimize_adam_elbo(theta0, dev_gp(g, ctx, kind, rows), dev_vp(g, ctx), NsK, bnd, max_iter=n_it,
                                         seed=seed, rng="philox", **kw)
                plan = ctx.last_entmc_plan()
            finally:
                ctx.set_option("adam_fused", 1)
            assert (plan["kernel"] == "adam_fused") == bool(fused), plan
            assert got[4] == ref[4]
            assert rel_err(got[2], ref[2]) < 1e-7 and rel_err(got[3], ref[3]) < 1e-7, (kind, fused, rel_err(got[3], ref[3]))


@pytest.mark.parametrize("mname", list(MASKS))
def test_partial_masks_fused_objective_vs_reference(ctx, golden, mname):
    """What every warm-up iteration evaluates (optimize_weights off, variational_optimization.py:142-143) and
    the ``variable_means = False`` masks: the fused objective with a reduced theta and the reduced theta_bnd
    of the reference's get_bounds; values, gradient blocks, the caller's theta and the vp side effects."""
    from pyvbmc_amd import entlb_vbmc, entmc_vbmc
    from pyvbmc_amd.variational_optimization import _neg_elcbo, _neg_elcbo_batch

    g = golden("variants")
    flags = MASKS[mname]
    NsK, seed = int(g["NsK"]), int(g["seed"])
    bnd = mask_bnd(g, mname)
    for gtag, gp, ogp in (("nq1", dev_gp(g, ctx, "negquad", slice(0, 1)), kind_gp(g, "negquad", slice(0, 1))),
                          ("constM", dev_gp(g, ctx, "const"), kind_gp(g, "const"))):
        for ns_tag, Ns in (("mc", NsK), ("lb", 0)):
            vp = dev_vp(g, ctx, flags)
            th = g[f"{mname}_theta"].copy()
            np.random.seed(seed)
            F, dF, G, H, _ = _neg_elcbo(th, gp, vp, 0.0, Ns, True, False, bnd, rng="numpy")
            k = f"{mname}_{gtag}_{ns_tag}"
            assert dF.shape == g[k + "_dF"].shape
            assert abs(F - g[k + "_F"]) <= 1e-10 * abs(F) and rel_err(dF, g[k + "_dF"]) < 1e-9, k
            assert np.allclose(th, g[k + "_theta_after"], rtol=0, atol=1e-15), k
            assert rel_err(vp.mu, g[k + "_mu"]) < 1e-14 and rel_err(vp.sigma.ravel(), g[k + "_sigma"]) < 1e-14, k
            assert rel_err(vp.lambd.ravel(), g[k + "_lambd"]) < 1e-14 and rel_err(vp.w.ravel(), g[k + "_w"]) < 1e-14, k
            assert rel_err(vp.eta.ravel(), g[k + "_eta"]) < 1e-14, k
            assert vp.mu.shape == g[k + "_mu"].shape and vp.sigma.shape == (1, int(g["K"]))
        # value only (what _sieve asks for), single call and as a batch of perturbed candidates
        vp = dev_vp(g, ctx, flags)
        F = _neg_elcbo(g[f"{mname}_theta"].copy(), gp, vp, 0.0, 0, False, False, bnd)[0]
        assert abs(F - g[f"{mname}_{gtag}_lb_F_nograd"]) <= 1e-10 * abs(F)
        rng = np.random.default_rng(8)
        thetas = g[f"{mname}_theta"][None, :] + 0.1 * rng.standard_normal((9, g[f"{mname}_theta"].size))
        thetas[0] = g[f"{mname}_theta"]
        Fb = _neg_elcbo_batch(thetas, gp, dev_vp(g, ctx, flags), bnd)
        assert abs(Fb[0] - g[f"{mname}_{gtag}_lb_F_nograd"]) <= 1e-10 * abs(Fb[0])
        for b in range(1, 9):
            Fo = elbo_ref.neg_elcbo(thetas[b].copy(), ogp, masked_mix(g, flags), 0.0, 0, False, False, bnd)[0]
            assert abs(Fb[b] - Fo) <= 1e-10 * abs(Fo), (mname, gtag, b)
    gf = tuple(map(bool, flags))
    vp = dev_vp(g, ctx)
    np.random.seed(seed)
    H, dH = entmc_vbmc(vp, NsK, gf, True)
    assert abs(H - g[f"{mname}_entmc_H"]) <= 1e-10 * abs(H) and rel_err(dH, g[f"{mname}_entmc_dH"]) < 1e-9
    H, dH = entlb_vbmc(dev_vp(g, ctx), gf, True)
    assert abs(H - g[f"{mname}_entlb_H"]) <= 1e-10 * abs(H) and rel_err(dH, g[f"{mname}_entlb_dH"]) < 1e-9


@pytest.mark.parametrize("mname", list(MASKS))
def test_partial_masks_optimiser_loops(ctx, golden, mname):
    """Both forms of the device-resident loop with the reference's masks and reduced bounds, a GP with
    ConstantMean and three hyper-parameter samples, against oracle Adam on the same Philox draws."""
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    g = golden("variants")
    flags, bnd = MASKS[mname], mask_bnd(g, mname)
    K, D, NsK, n_it, seed = int(g["K"]), int(g["D"]), 28, 41, 99
    ogp, mix = kind_gp(g, "const"), masked_mix(g, flags)
    theta0 = g[f"{mname}_theta"]
    cnt = [0]

    def fo(t):
        eps = philox_ref.eps_half(K, NsK // 2, D, seed + cnt[0])
        cnt[0] += 1
        r = elbo_ref.neg_elcbo(t, ogp, mix, 0.0, NsK, True, False, bnd, eps_half=eps)
        return r[0], r[1]

    kw = dict(tol_fun=1e-9, master_min=0.001, master_max=0.1, master_decay=200)
    ref = adam_ref.minimize_adam(fo, theta0.copy(), max_iter=n_it, **kw)
    for fused in (1, 0):
        ctx.set_option("adam_fused", fused)
        try:
            got = minimize_adam_elbo(theta0, dev_gp(g, ctx, "const"), dev_vp(g, ctx, flags), NsK, bnd, max_iter=n_it,
                                     seed=seed, rng="philox", **kw)
            plan = ctx.last_entmc_plan()
        finally:
            ctx.set_option("adam_fused", 1)
        assert (plan["kernel"] == "adam_fused") == bool(fused), plan
        assert got[4] == ref[4]
        assert rel_err(got[2], ref[2]) < 1e-7 and rel_err(got[3], ref[3]) < 1e-7, (mname, fused)


def test_pdf_orig_space_bounded_transformer(ctx, golden):
    """``pdf`` / ``log_pdf`` with ``orig_flag=True`` and a bounded transformer: strict-inequality mask (rows ON
    the bounds are outside), Jacobian divided out / subtracted, gradient rows, t-tails, 1-D input
    (variational_posterior.py:429-439, 543-559; the reference's test_pdf_outside_bounds)."""
    from pyvbmc_amd import VariationalPosterior

    g = golden("variants")
    pt = variant_transformer(g)
    vp = dev_vp(g, ctx)
    vp.parameter_transformer = pt
    x, m = g["pdfo_x"], g["pdfo_mask"]
    y = vp.pdf(x, orig_flag=True)
    assert y.shape == g["pdfo_y"].shape and np.all(y[~m] == 0)
    assert rel_err(y, g["pdfo_y"]) < 1e-10
    for ly in (vp.pdf(x, orig_flag=True, log_flag=True), vp.log_pdf(x, orig_flag=True)):
        assert np.all(np.isneginf(ly[~m])) and np.allclose(ly[m], g["pdfo_logy"][m], rtol=0, atol=1e-10)
    yy, dy = vp.pdf(x, orig_flag=True, grad_flag=True)
    assert rel_err(yy, g["pdfo_y_g"]) < 1e-10
    assert np.allclose(dy, g["pdfo_dy"], rtol=1e-9, atol=1e-300 + 1e-10 * np.abs(g["pdfo_dy"]).max())
    for df in (7.0, -3.0):
        assert rel_err(vp.pdf(x, orig_flag=True, df=df), g[f"pdfo_y_df{df}"]) < 1e-10
        l = vp.pdf(x, orig_flag=True, log_flag=True, df=df)
        assert np.all(np.isneginf(l[~m])) and np.allclose(l[m], g[f"pdfo_logy_df{df}"][m], rtol=0, atol=1e-10)
    one = vp.pdf(x[3], orig_flag=True)
    assert one.shape == g["pdfo_1d"].shape and rel_err(one, g["pdfo_1d"]) < 1e-10
    with pytest.raises(NotImplementedError):
        vp.pdf(x, orig_flag=True, log_flag=True, grad_flag=True)
    # samples drawn in the transformed space come back through the transformer's inverse inside the bounds
    np.random.seed(12)
    xs, _ = vp.sample(300, orig_flag=True, balance_flag=True)
    assert np.allclose(xs, g["pdfo_sample_x"], rtol=1e-12, atol=1e-12)
    # the reference's own edge test: D = 2, bounds -3 / 3
    from oracle import transform_ref

    pt2 = transform_ref.BoundedLogit(2, -3.0 * np.ones((1, 2)), 3.0 * np.ones((1, 2)))
    vp2 = VariationalPosterior(2, 2, np.array([[2.0, 2.0], [-2.0, -2.0]]), pt2)
    vp2.ctx = ctx
    vp2.mu = g["pdfo2_mu"].copy()
    vp2.sigma = np.ones((1, 2))
    y2 = vp2.pdf(g["pdfo2_x"], orig_flag=True)
    assert np.array_equal(y2[:4], np.zeros((4, 1))) and np.all(y2[4:] > 0) and rel_err(y2, g["pdfo2_y"]) < 1e-10
    l2 = vp2.log_pdf(g["pdfo2_x"], orig_flag=True)
    assert np.all(np.isneginf(l2[:4])) and np.allclose(l2[4:], g["pdfo2_logy"][4:], rtol=0, atol=1e-10)


def test_importance_sampling_inputs_on_device(ctx, golden):
    """The device's ``gp.predict`` / ``vp.pdf`` against the values the reference's ``fess`` and
    ``active_sample_proposal_pdf`` consumed on their MATLAB known-answer inputs
    (testing/vbmc/test_active_importance_sampling.py:113-250): with these inputs reproduced, the two
    helpers -- the reference's own code, unchanged -- return MATLAB's answers."""
    from pyvbmc_amd import VariationalPosterior
    from pyvbmc_amd import gp as gpm

    g = golden("is_known")
    gp = gpm.GP(3, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True))
    gp.ctx = ctx
    gp.update(X_new=g["X"], y_new=g["y"], hyp=g["hyp"])
    sf2 = float(np.exp(2 * g["hyp"][:, 3]).max())
    fbar, fs2 = gp.predict(g["Xa"])
    assert fbar.shape == g["fess_gp_fbar"].shape and fs2.shape == g["fess_gp_fs2"].shape
    assert np.max(np.abs(fbar - g["fess_gp_fbar"])) < 1e-10 * sf2 and np.max(np.abs(fs2 - g["fess_gp_fs2"])) < 1e-10 * sf2
    vp = VariationalPosterior(3, 2)
    vp.ctx = ctx
    vp.mu, vp.w, vp.lambd = g["fess_mu"].copy(), g["fess_w"].reshape(1, -1).copy(), np.ones((3, 1))
    vp.sigma = g["fess_sigma"].reshape(1, -1).copy()
    assert np.allclose(vp.pdf(g["Xa"], orig_flag=False, log_flag=True), g["fess_gp_logpdf"], rtol=1e-10)
    assert np.allclose(vp.pdf(g["X"], orig_flag=False, log_flag=True), g["fess_means_logpdf"], rtol=1e-10)
    fmu, fs2 = gp.predict(g["Xa"], separate_samples=True)
    vp.mu, vp.sigma = g["aspp_mu"].copy(), np.ones((1, 2))
    for name in ("viqr", "imiqr"):
        assert np.max(np.abs(fmu - g[f"aspp_{name}_fmu"])) < 1e-10 * sf2
        assert np.max(np.abs(fs2 - g[f"aspp_{name}_pred_fs2"])) < 1e-10 * sf2
        assert np.allclose(vp.pdf(g["Xa"], orig_flag=False, log_flag=True).ravel(), g[f"aspp_{name}_logpdf"].ravel(), rtol=1e-10)
