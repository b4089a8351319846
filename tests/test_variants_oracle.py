"""Pin the oracle on the branches the reference takes outside "all blocks optimised,
NegativeQuadratic mean": GP mean kinds, avg_flag / jacobian_flag off, the partial optimise
masks (warm-up: weights off), the orig-space density with a bounded transformer, and the
primitives the reference's importance-sampling helpers consume.

tests/golden/variants.npz and is_known.npz were produced by the REFERENCE
(oracle/make_golden.py ``variants`` / ``is_known``).  CPU only; the device side of the same
fixtures is tests/test_variants_gpu.py.
"""
import numpy as np
import pytest
from helpers import oracle_mix, rel_err

from oracle import acq_ref, adam_ref, elbo_ref, entropy_ref, gp_ref, mixture_ref, transform_ref
from pyvbmc_amd import synthetic

KINDS = {"zero": gp_ref.MEAN_ZERO, "const": gp_ref.MEAN_CONST, "negquad": gp_ref.MEAN_NEGQUAD}
MASKS = {"mask1110": (1, 1, 1, 0), "mask0111": (0, 1, 1, 1), "mask0110": (0, 1, 1, 0)}


def kind_hyp(g, kind, rows=slice(None)):
    D = int(g["D"])
    n_mean = {"zero": 0, "const": 1, "negquad": 1 + 2 * D}[kind]
    return np.ascontiguousarray(g["hyp"][rows, : D + 2 + n_mean])


def kind_gp(g, kind, rows=slice(None)):
    return gp_ref.make_gp(g["X"], g["y"], kind_hyp(g, kind, rows), KINDS[kind])


def full_bnd(g):
    return {"lb": g["bnd_full_lb"], "ub": g["bnd_full_ub"], "tol_con": float(g["bnd_tol_con"]),
            "weight_threshold": float(g["bnd_weight_threshold"]), "weight_penalty": float(g["bnd_weight_penalty"])}


def mask_bnd(g, mname):
    b = {"lb": g[f"{mname}_lb"], "ub": g[f"{mname}_ub"], "tol_con": float(g["bnd_tol_con"])}
    if int(g[f"{mname}_has_weight_keys"]):
        b["weight_threshold"], b["weight_penalty"] = float(g["bnd_weight_threshold"]), float(g["bnd_weight_penalty"])
    return b


def masked_mix(g, flags):
    m = oracle_mix(g)
    m.optimize_mu, m.optimize_sigma, m.optimize_lambd, m.optimize_weights = map(bool, flags)
    return m


@pytest.mark.parametrize("kind", list(KINDS))
def test_gp_log_joint_mean_kinds(golden, kind):
    g = golden("variants")
    for tag, rows in (("S1", slice(0, 1)), ("SM", slice(None))):
        gp = kind_gp(g, kind, rows)
        for avg in (True, False):
            for jac in (True, False):
                G, dG, _, _, _ = gp_ref.gp_log_joint(oracle_mix(g), gp, True, avg, jac, False, False)
                k = f"glj_{kind}_{tag}_a{int(avg)}_j{int(jac)}"
                assert np.shape(G) == g[k + "_G"].shape and dG.shape == g[k + "_dG"].shape, k
                assert rel_err(G, g[k + "_G"]) < 1e-12 and rel_err(dG, g[k + "_dG"]) < 1e-11, k
            G, _, varG, _, var_ss, I_sk, J_sjk = gp_ref.gp_log_joint(oracle_mix(g), gp, False, avg, True, True, True)
            k = f"glj_{kind}_{tag}_a{int(avg)}_var"
            assert np.shape(G) == g[k + "_G"].shape and np.shape(varG) == g[k + "_varG"].shape, k
            assert rel_err(G, g[k + "_G"]) < 1e-12 and rel_err(varG, g[k + "_varG"]) < 1e-9, k
            assert abs(var_ss - g[k + "_var_ss"]) <= 1e-9 * max(abs(var_ss), 1e-300), k
            assert rel_err(I_sk, g[k + "_I_sk"]) < 1e-12 and rel_err(J_sjk, g[k + "_J_sjk"]) < 1e-9, k


@pytest.mark.parametrize("kind", list(KINDS))
def test_neg_elcbo_mean_kinds(golden, kind):
    g = golden("variants")
    K, D, NsK, seed = int(g["K"]), int(g["D"]), int(g["NsK"]), int(g["seed"])
    eps = synthetic.draw_eps_half(K, D, NsK, seed)
    bnd = full_bnd(g)
    for gtag, rows in (("S1", slice(0, 1)), ("SM", slice(None))):
        gp = kind_gp(g, kind, rows)
        for ns_tag, Ns in (("mc", NsK), ("lb", 0)):
            F, dF, G, H, _ = elbo_ref.neg_elcbo(g["theta"].copy(), gp, oracle_mix(g), 0.0, Ns, True, False, bnd,
                                                False, eps_half=eps if Ns else None)
            k = f"elbo_{kind}_{gtag}_{ns_tag}"
            assert abs(F - g[k + "_F"]) <= 1e-11 * abs(F) and rel_err(dF, g[k + "_dF"]) < 1e-10, k
            assert abs(G - g[k + "_G"]) <= 1e-12 * abs(G) and abs(H - g[k + "_H"]) <= 1e-12 * abs(H), k


@pytest.mark.parametrize("kind", list(KINDS))
def test_acquisition_and_predict_mean_kinds(golden, kind):
    g = golden("variants")
    gp = kind_gp(g, kind)
    D = int(g["D"])
    fmu, fs2 = gp_ref.predict(gp, g["Xs"], separate_samples=True)
    assert np.array_equal(fmu, g[f"pred_{kind}_fmu"]) and np.array_equal(fs2, g[f"pred_{kind}_fs2"])
    st = dict(lb_eps_orig=g["X"].min(0) - 2.0, ub_eps_orig=g["X"].max(0) + 2.0,
              gp_length_scale=np.exp(g["hyp"][0, :D]), variance_regularized_acq_fcn=False)
    for name, code in (("AcqFcn", acq_ref.STD), ("AcqFcnLog", acq_ref.LOG)):
        v = acq_ref.acq_call(code, g["Xs"], gp, oracle_mix(g), float(np.max(g["y"])), st)
        assert rel_err(v, g[f"acq_{kind}_{name}"]) < 1e-12, (kind, name)


@pytest.mark.parametrize("kind", list(KINDS))
def test_adam_trajectory_mean_kinds(golden, kind):
    """oracle Adam around the oracle objective on the NumPy stream == the reference's run."""
    g = golden("variants")
    gp, mix, bnd = kind_gp(g, kind, slice(0, 1)), oracle_mix(g), full_bnd(g)

    def f(t):
        r = elbo_ref.neg_elcbo(t, gp, mix, 0.0, 40, True, False, bnd)
        return r[0], r[1]

    np.random.seed(70)
    x, y, xt, yt, it = adam_ref.minimize_adam(f, g[f"adam_{kind}_theta0"].copy(), tol_fun=0.05, max_iter=30,
                                              master_min=0.001, master_max=0.1, master_decay=200)
    assert it == int(g[f"adam_{kind}_iters"])
    assert rel_err(xt, g[f"adam_{kind}_x_tab"]) < 1e-8 and rel_err(yt, g[f"adam_{kind}_y_tab"]) < 1e-8


@pytest.mark.parametrize("mname", list(MASKS))
def test_partial_masks(golden, mname):
    g = golden("variants")
    flags = MASKS[mname]
    K, D, NsK, seed = int(g["K"]), int(g["D"]), int(g["NsK"]), int(g["seed"])
    eps = synthetic.draw_eps_half(K, D, NsK, seed)
    bnd = mask_bnd(g, mname)
    n_theta = D * K * flags[0] + K * flags[1] + D * flags[2] + K * flags[3]
    assert g[f"{mname}_theta"].size == n_theta
    assert bnd["lb"].size == D * K * flags[0] + D * K + K * flags[3]
    for gtag, gp in (("nq1", kind_gp(g, "negquad", slice(0, 1))), ("constM", kind_gp(g, "const"))):
        for ns_tag, Ns in (("mc", NsK), ("lb", 0)):
            mix = masked_mix(g, flags)
            th = g[f"{mname}_theta"].copy()
            F, dF, G, H, _ = elbo_ref.neg_elcbo(th, gp, mix, 0.0, Ns, True, False, bnd, False,
                                                eps_half=eps if Ns else None)
            k = f"{mname}_{gtag}_{ns_tag}"
            assert dF.shape == g[k + "_dF"].shape == (n_theta,)
            assert abs(F - g[k + "_F"]) <= 1e-11 * abs(F) and rel_err(dF, g[k + "_dF"]) < 1e-10, k
            assert np.array_equal(th, g[k + "_theta_after"]), k
            assert rel_err(mix.mu, g[k + "_mu"]) < 1e-15 and rel_err(mix.sigma, g[k + "_sigma"]) < 1e-15, k
            assert rel_err(mix.lambd, g[k + "_lambd"]) < 1e-15 and rel_err(mix.w, g[k + "_w"]) < 1e-15, k
            assert rel_err(mix.eta, g[k + "_eta"]) < 1e-15, k
        mix = masked_mix(g, flags)
        F = elbo_ref.neg_elcbo(g[f"{mname}_theta"].copy(), gp, mix, 0.0, 0, False, False, bnd, False)[0]
        assert abs(F - g[f"{mname}_{gtag}_lb_F_nograd"]) <= 1e-11 * abs(F)
    gf = tuple(map(bool, flags))
    H, dH = entropy_ref.entmc(oracle_mix(g), NsK, gf, True, eps_half=eps)
    assert abs(H - g[f"{mname}_entmc_H"]) <= 1e-12 * abs(H) and rel_err(dH, g[f"{mname}_entmc_dH"]) < 1e-11
    H, dH = entropy_ref.entlb(oracle_mix(g), gf, True)
    assert abs(H - g[f"{mname}_entlb_H"]) <= 1e-12 * abs(H) and rel_err(dH, g[f"{mname}_entlb_dH"]) < 1e-11


def variant_transformer(g):
    return transform_ref.BoundedLogit(int(g["D"]), g["pt_lb"], g["pt_ub"], g["pt_plb"], g["pt_pub"])


def test_bounded_transformer_is_the_references(golden):
    g = golden("variants")
    pt = variant_transformer(g)
    assert np.array_equal(pt.mu, g["pt_mu"]) and np.array_equal(pt.delta, g["pt_delta"])
    assert np.array_equal(pt.bounded, g["pt_type"] != 0)
    x, m = g["pdfo_x"], g["pdfo_mask"]
    assert np.array_equal(np.logical_and(np.all(x > pt.lb_orig, 1), np.all(x < pt.ub_orig, 1)), m)
    assert 0 < m.sum() < m.size
    u = pt(x[m])
    assert np.allclose(u, g["pdfo_u"], rtol=1e-14, atol=1e-14)
    assert np.allclose(pt.log_abs_det_jacobian(u), g["pdfo_ladj"], rtol=1e-13, atol=1e-13)
    assert np.allclose(pt.inverse(u), g["pdfo_inv"], rtol=1e-14, atol=1e-14)


def test_pdf_orig_space(golden):
    g = golden("variants")
    pt, mix, x, m = variant_transformer(g), oracle_mix(g), g["pdfo_x"], g["pdfo_mask"]
    y = transform_ref.pdf_orig(mix, pt, x)
    assert y.shape == g["pdfo_y"].shape and np.all(y[~m] == 0) and np.all(g["pdfo_y"][~m] == 0)
    assert rel_err(y, g["pdfo_y"]) < 1e-13
    ly = transform_ref.pdf_orig(mix, pt, x, log_flag=True)
    assert np.all(np.isneginf(ly[~m])) and np.all(np.isneginf(g["pdfo_logy"][~m]))
    assert np.allclose(ly[m], g["pdfo_logy"][m], rtol=0, atol=1e-12)
    assert np.array_equal(g["pdfo_logy"], g["pdfo_logy_method"], equal_nan=True)
    yy, dy = transform_ref.pdf_orig(mix, pt, x, grad_flag=True)
    assert rel_err(yy, g["pdfo_y_g"]) < 1e-13
    assert np.allclose(dy, g["pdfo_dy"], rtol=1e-10, atol=1e-300)  # incl. the rows outside the bounds
    for df in (7.0, -3.0):
        assert rel_err(transform_ref.pdf_orig(mix, pt, x, df=df), g[f"pdfo_y_df{df}"]) < 1e-12
        l = transform_ref.pdf_orig(mix, pt, x, log_flag=True, df=df)
        assert np.allclose(l[m], g[f"pdfo_logy_df{df}"][m], rtol=0, atol=1e-11)
    with pytest.raises(NotImplementedError):
        transform_ref.pdf_orig(mix, pt, x, log_flag=True, grad_flag=True)
    # the reference's own edge test: on and outside the bounds 0 / -inf, inside positive / finite
    pt2 = transform_ref.BoundedLogit(2, -3.0 * np.ones((1, 2)), 3.0 * np.ones((1, 2)))
    mix2 = mixture_ref.Mixture.make(g["pdfo2_mu"], np.ones(2), np.ones(2), np.full(2, 0.5))
    y2 = transform_ref.pdf_orig(mix2, pt2, g["pdfo2_x"])
    assert np.array_equal(y2[:4], np.zeros((4, 1))) and np.all(y2[4:] > 0)
    assert rel_err(y2, g["pdfo2_y"]) < 1e-13
    l2 = transform_ref.pdf_orig(mix2, pt2, g["pdfo2_x"], log_flag=True)
    assert np.all(np.isneginf(l2[:4])) and np.allclose(l2[4:], g["pdfo2_logy"][4:], rtol=0, atol=1e-13)


def is_known_gp(g):
    return gp_ref.make_gp(g["X"], g["y"], g["hyp"], gp_ref.MEAN_NEGQUAD)


def test_importance_sampling_inputs(golden):
    """The values ``gp.predict`` / ``vp.pdf`` handed to the reference's ``fess`` and
    ``active_sample_proposal_pdf`` on its MATLAB known-answer inputs, and (from those recorded values
    alone, no restatement of the two functions) the MATLAB answers themselves."""
    g, m = golden("is_known"), golden("matlab_known")
    gp = is_known_gp(g)
    fbar, fs2 = gp_ref.predict(gp, g["Xa"])
    assert np.allclose(fbar, g["fess_gp_fbar"], rtol=1e-13, atol=1e-13) and np.allclose(fs2, g["fess_gp_fs2"], rtol=1e-12, atol=1e-13)
    mix = mixture_ref.Mixture.make(g["fess_mu"], g["fess_sigma"], np.ones(3), g["fess_w"])
    assert np.allclose(mixture_ref.pdf(mix, g["Xa"], log_flag=True), g["fess_gp_logpdf"], rtol=1e-13)
    assert np.isclose(float(g["fess_means"]), m["fess_fess_means"].item())
    assert np.isclose(float(g["fess_gp"]), m["fess_fess_gp"].item())
    fmu, fs2 = gp_ref.predict(gp, g["Xa"], separate_samples=True)
    mix = mixture_ref.Mixture.make(g["aspp_mu"], np.ones(2), np.ones(3), np.array([0.7, 0.3]))
    for name in ("viqr", "imiqr"):
        assert np.allclose(fmu, g[f"aspp_{name}_fmu"], rtol=1e-13, atol=1e-13)
        assert np.allclose(fs2, g[f"aspp_{name}_pred_fs2"], rtol=1e-12, atol=1e-13)
        assert np.allclose(mixture_ref.pdf(mix, g["Xa"], log_flag=True).ravel(), g[f"aspp_{name}_logpdf"].ravel(), rtol=1e-13)
        assert np.array_equal(g[f"aspp_{name}_f_s2"], g[f"aspp_{name}_pred_fs2"])
        assert np.allclose(g[f"aspp_{name}_ln_weights"], m[f"activesample_proposalpdf_ln_weights_{name}"])
        assert np.allclose(g[f"aspp_{name}_f_s2"], m[f"activesample_proposalpdf_f_s2_{name}"])
