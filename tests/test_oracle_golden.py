"""Pin the oracle: NumPy restatement vs outputs of the actual reference.

tests/golden/*.npz were produced by oracle/make_golden.py importing
/root/reference; matlab_known.npz re-packs the MATLAB-derived known answers the
reference's own tests assert.  CPU only.
"""
import itertools

import numpy as np
import pytest
from conftest import CASES, MID_CASES
from helpers import oracle_gp, oracle_mix, rel_err

from oracle import elbo_ref, entropy_ref, gp_ref, mixture_ref
from pyvbmc_amd import synthetic

TOL = 1e-12


def fl(f):
    return "".join("1" if b else "0" for b in f)


@pytest.mark.parametrize("name", CASES)
def test_entmc_matches_reference(golden, name):
    g = golden(name)
    K, D, NsK, seed = int(g["K"]), int(g["D"]), int(g["NsK"]), int(g["seed"])
    eps = synthetic.draw_eps_half(K, D, NsK, seed)
    combos = list(itertools.product([False, True], repeat=4)) if name == "c1" else [(False,) * 4, (True,) * 4]
    for gf in combos:
        for jac in (True, False):
            H, dH = entropy_ref.entmc(oracle_mix(g), NsK, gf, jac, eps_half=eps)
            assert abs(H - g[f"entmc_H_{fl(gf)}_{int(jac)}"]) <= TOL * abs(H)
            ref = g[f"entmc_dH_{fl(gf)}_{int(jac)}"]
            assert dH.shape == ref.shape
            if ref.size:
                assert rel_err(dH, ref) < 1e-11


@pytest.mark.parametrize("name", MID_CASES)
def test_mid_cases_match_reference(golden, name):
    """The mid-size goldens (reference-generated) pin the oracle at the sizes where the
    device kernel runs several batches per workgroup."""
    g = golden(name)
    K, D, NsK, seed = int(g["K"]), int(g["D"]), int(g["NsK"]), int(g["seed"])
    eps = synthetic.draw_eps_half(K, D, NsK, seed)
    H, dH = entropy_ref.entmc(oracle_mix(g), NsK, (True,) * 4, True, eps_half=eps)
    assert abs(H - g["entmc_H_1111_1"]) <= TOL * abs(H)
    assert rel_err(dH, g["entmc_dH_1111_1"]) < 1e-11
    if name == "c5m":
        return  # the ELBO part below costs another ~20 s of oracle time at D=20, K=100
    wl = synthetic.make_workload(int(g["cfg"]), S=1, D=D, K=K, N=int(g["N"]), Ns_total=int(g["Ns_total"]))
    bnd = synthetic.default_theta_bnd(wl)
    gp = oracle_gp(g, g["hyp"][:1])
    for tag, th in (("bnd", g["theta"]), ("bndout", g["theta_out"])):
        th_in = th.copy()
        F, dF, G, H, _ = elbo_ref.neg_elcbo(th_in, gp, oracle_mix(g), 0.0, NsK, True, False, bnd, False, eps_half=eps)
        key = f"elbo_{tag}_mc"
        assert abs(F - g[key + "_F"]) <= 1e-11 * abs(F), key
        assert rel_err(dF, g[key + "_dF"]) < 1e-10, key
        assert np.array_equal(th_in, g[key + "_theta_after"]), key


@pytest.mark.parametrize("name", CASES)
def test_entmc_global_rng_draw_order(golden, name):
    """Without eps the oracle consumes np.random exactly like the reference."""
    g = golden(name)
    np.random.seed(int(g["seed"]))
    H, dH = entropy_ref.entmc(oracle_mix(g), int(g["NsK"]), (True,) * 4, True)
    assert abs(H - g["entmc_H_1111_1"]) <= TOL * abs(H)
    assert rel_err(dH, g["entmc_dH_1111_1"]) < 1e-11


@pytest.mark.parametrize("name", CASES)
def test_entlb_matches_reference(golden, name):
    g = golden(name)
    combos = list(itertools.product([False, True], repeat=4)) if name == "c1" else [(False,) * 4, (True,) * 4]
    for gf in combos:
        for jac in (True, False):
            H, dH = entropy_ref.entlb(oracle_mix(g), gf, jac)
            assert abs(H - g[f"entlb_H_{fl(gf)}_{int(jac)}"]) <= TOL * abs(H)
            ref = g[f"entlb_dH_{fl(gf)}_{int(jac)}"]
            assert dH.shape == ref.shape
            if ref.size:
                assert rel_err(dH, ref) < 1e-11


@pytest.mark.parametrize("name", CASES)
def test_gp_log_joint_matches_reference(golden, name):
    g = golden(name)
    for tag, hyp in (("S1", g["hyp"][:1]), ("SM", g["hyp"])):
        gp = oracle_gp(g, hyp)
        mix = oracle_mix(g)
        G, dG, _, _, _ = gp_ref.gp_log_joint(mix, gp, True, True, True, False, False)
        assert abs(G - g[f"glj_{tag}_G"]) <= 1e-12 * abs(G)
        assert rel_err(dG, g[f"glj_{tag}_dG"]) < 1e-11
        G, _, varG, _, var_ss, I_sk, J_sjk = gp_ref.gp_log_joint(mix, gp, False, True, True, True, True)
        assert rel_err(varG, g[f"glj_{tag}_varG"]) < 1e-9
        assert abs(var_ss - g[f"glj_{tag}_var_ss"]) <= 1e-9 * max(abs(var_ss), 1e-300)
        assert rel_err(I_sk, g[f"glj_{tag}_I_sk"]) < 1e-12
        assert rel_err(J_sjk, g[f"glj_{tag}_J_sjk"]) < 1e-9


@pytest.mark.parametrize("name", CASES)
def test_neg_elcbo_matches_reference(golden, name):
    g = golden(name)
    K, D, NsK, seed = int(g["K"]), int(g["D"]), int(g["NsK"]), int(g["seed"])
    wl = synthetic.make_workload(int(g["cfg"]), S=1, D=D, K=K, N=int(g["N"]), Ns_total=int(g["Ns_total"]))
    bnd = synthetic.default_theta_bnd(wl)
    gp = oracle_gp(g, g["hyp"][:1])
    eps = synthetic.draw_eps_half(K, D, NsK, seed)
    for tag, th, tb in (("nobnd", g["theta"], None), ("bnd", g["theta"], bnd), ("bndout", g["theta_out"], bnd)):
        for ns_tag, Ns in (("mc", NsK), ("lb", 0)):
            th_in = th.copy()
            F, dF, G, H, varF = elbo_ref.neg_elcbo(
                th_in, gp, oracle_mix(g), 0.0, Ns, True, False, tb, False, eps_half=eps if Ns else None
            )
            key = f"elbo_{tag}_{ns_tag}"
            assert abs(F - g[key + "_F"]) <= 1e-11 * abs(F), key
            assert rel_err(dF, g[key + "_dF"]) < 1e-10, key
            assert abs(G - g[key + "_G"]) <= 1e-11 * abs(G)
            assert abs(H - g[key + "_H"]) <= 1e-11 * abs(H)
            # reference quirk: the caller's theta gets its eta tail max-shifted in place
            assert np.array_equal(th_in, g[key + "_theta_after"]), key
    r = elbo_ref.neg_elcbo(g["theta"].copy(), gp, oracle_mix(g), 0.0, NsK, False, True, None, True, eps_half=eps)
    assert abs(r[0] - g["elbo_full_F"]) <= 1e-11 * abs(r[0])
    assert rel_err(r[4], g["elbo_full_varF"]) < 1e-9
    assert rel_err(r[9], g["elbo_full_I_sk"]) < 1e-12
    assert rel_err(r[10], g["elbo_full_J_sjk"]) < 1e-9
    assert r[1] is None and r[5] is None


@pytest.mark.parametrize("name", CASES)
def test_pdf_matches_reference(golden, name):
    g = golden(name)
    mix = oracle_mix(g)
    x = g["pdf_x"]
    assert rel_err(mixture_ref.pdf(mix, x), g["pdf_y"]) < 1e-12
    ly = mixture_ref.pdf(mix, x, log_flag=True)
    fin = np.isfinite(g["pdf_logy"])
    assert np.array_equal(np.isneginf(ly), np.isneginf(g["pdf_logy"]))
    assert rel_err(ly[fin], g["pdf_logy"][fin]) < 1e-12
    _, dy = mixture_ref.pdf(mix, x, grad_flag=True)
    assert rel_err(dy, g["pdf_dy"]) < 1e-12
    with np.errstate(all="ignore"):
        _, dly = mixture_ref.pdf(mix, x, log_flag=True, grad_flag=True)
    ok = np.isfinite(g["pdf_dlogy"])
    assert np.array_equal(ok, np.isfinite(dly))
    assert rel_err(dly[ok], g["pdf_dlogy"][ok]) < 1e-11
    for df in (10.0, -2.0, 3.5, -7.0):
        assert rel_err(mixture_ref.pdf(mix, x, df=df), g[f"pdf_y_df{df}"]) < 1e-12
        ly = mixture_ref.pdf(mix, x, log_flag=True, df=df)
        assert rel_err(ly, g[f"pdf_logy_df{df}"]) < 1e-12


@pytest.mark.parametrize("name", CASES)
def test_moments_and_parameter_round_trip(golden, name):
    g = golden(name)
    m, c = mixture_ref.moments(oracle_mix(g), cov_flag=True)
    assert rel_err(m, g["mom_mean"]) < 1e-13 and rel_err(c, g["mom_cov"]) < 1e-13
    mix = oracle_mix(g)
    mixture_ref.set_parameters(mix, g["rt_theta_in"])
    for k, v in (("mu", mix.mu), ("sigma", mix.sigma), ("lambd", mix.lambd), ("w", mix.w)):
        assert rel_err(v, g["rt_" + k]) < 1e-14
    assert rel_err(mixture_ref.get_parameters(mix), g["rt_theta_out"]) < 1e-14
    assert rel_err(mixture_ref.get_parameters(mix, raw_flag=False), g["rt_theta_out_noraw"]) < 1e-14


# ---- MATLAB-derived known answers the reference's own tests assert ----------


def test_matlab_entropy(golden):
    m = golden("matlab_known")
    mix = mixture_ref.Mixture.make(m["ent_mu"], m["ent_sigma"], m["ent_lambd"], m["ent_w"], m["ent_eta"])
    Hl, dHl = entropy_ref.entlb(mix, (True,) * 4, bool(m["ent_jacobian_flag"]))
    assert np.isclose(Hl, m["ent_Hl"], rtol=1e-13)
    assert np.allclose(dHl, m["ent_dHl"], rtol=1e-9, atol=1e-12)
    # MC entropy: different RNG stream than MATLAB -> the reference's own rtol 1e-2
    np.random.seed(42)
    H, dH = entropy_ref.entmc(mix, int(m["ent_Ns"]), (True,) * 4, bool(m["ent_jacobian_flag"]))
    assert np.isclose(H, m["ent_H"], rtol=1e-2)
    assert np.allclose(dH, m["ent_dH"], rtol=1e-2, atol=1e-2)


def test_matlab_gp_log_joint_and_elbo(golden):
    m = golden("matlab_known")
    D = K = 2
    mix = mixture_ref.Mixture.make(m["vbmc_mu"], 1e-3 * np.ones(K), np.ones(D), np.ones(K) / K, np.ones(K) / K)
    gp = gp_ref.make_gp(m["vbmc_X"], m["vbmc_y"], m["vbmc_hyp"])
    assert all(p.L_chol for p in gp.posteriors)
    G, dG, varG, _, var_ss, _, _ = gp_ref.gp_log_joint(mix, gp, False, True, True, True, True)
    assert np.isclose(G, m["vbmc_G"]) and dG is None
    assert np.isclose(varG, m["vbmc_varG"]) and np.isclose(var_ss, m["vbmc_var_ss"])
    G, dG, _, _, _ = gp_ref.gp_log_joint(mix, gp, True, True, True, False, False)
    assert np.allclose(dG, m["vbmc_dG_gp_log_joint"])
    theta = mixture_ref.get_parameters(mix)
    r = elbo_ref.neg_elcbo(theta, gp, mix, 0.0, 0, False, True, None, True)
    assert np.isclose(r[0], m["vbmc_F"]) and np.isclose(r[3], m["vbmc_H"], rtol=1e-14)
    F, dF, _, _, _ = elbo_ref.neg_elcbo(theta, gp, mix, 0.0, 0, True, False, None, False)
    assert np.allclose(dF, m["vbmc_dF"])


def test_matlab_moments(golden):
    m = golden("matlab_known")
    D, K = 6, 3
    w = np.arange(1, 4) / 6.0
    mix = mixture_ref.Mixture.make(
        np.linspace(-3, 3, D * K).reshape((D, K), order="F"), np.arange(2, 5), np.arange(3, 9), w
    )
    mean, cov = mixture_ref.moments(mix, cov_flag=True)
    assert np.allclose(mean, m["mom_mubar"]) and np.allclose(cov, m["mom_sigma"])


def test_matlab_gp_predict(golden):
    """gp.predict pinned through test_active_importance_sampling.py:178-250."""
    m = golden("matlab_known")
    D = 3
    X = np.arange(-7, 8).reshape((5, 3), order="F").astype(float)
    y = (-0.5 * np.sum(X**2, axis=1) - 0.5 * D * np.log(2 * np.pi)).reshape(-1, 1)
    hyp = np.array([-2.0, -3.0, -4.0, 1.0, 0.0, -(D / 2) * np.log(2 * np.pi), 0.0, 0.25, 0.5, -0.5, 0.0, 0.5])
    gp = gp_ref.make_gp(X, y, np.vstack([hyp, 2 * hyp]))
    Xa = 2 * np.arange(-4, 5).reshape((3, 3), order="F") / np.pi
    fmu, fs2 = gp_ref.predict(gp, Xa, separate_samples=True)
    assert np.allclose(fs2, m["activesample_proposalpdf_f_s2_viqr"])
    assert np.allclose(fs2, m["activesample_proposalpdf_f_s2_imiqr"])
    # fess(vp, gp, Xa): restated from active_importance_sampling.py:461-478 (test-only)
    mix = mixture_ref.Mixture.make(
        np.array([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]).T, 1e-3 * np.ones(2), np.ones(3), [0.7, 0.3]
    )
    fbar, _ = gp_ref.predict(gp, Xa)
    lp = np.maximum(mixture_ref.pdf(mix, Xa, log_flag=True), np.log(np.finfo(float).tiny)).ravel()
    lw = fbar.ravel() - lp
    wgt = np.exp(lw - lw.max())
    wgt /= wgt.sum()
    assert np.isclose(1 / np.sum(wgt**2) / 3, m["fess_fess_gp"].item())


def test_soft_bound_loss_known(golden):
    g = golden("misc")
    L, dL = elbo_ref.soft_bound_loss(g["sbl_x"], np.full(3, -10.0), np.full(3, 10.0), compute_grad=True)
    assert np.isclose(L, 156250.0) and np.isclose(L, g["sbl_L"])
    assert np.allclose(dL, [12500.0, -25000.0, 0.0]) and np.allclose(dL, g["sbl_dL"])


GPCOV_CASES = ["homo", "hetero", "tiny"]


def gpcov_gp(c, name):
    s2 = c["s2"] if name == "hetero" else None
    return gp_ref.make_gp(c["X"], c["y"], c[f"{name}_hyp"], gp_ref.MEAN_NEGQUAD, s2=s2, noise_user=s2 is not None)


@pytest.mark.parametrize("name", GPCOV_CASES)
def test_posterior_records_vs_reference_in_tree_solves(golden, name):
    """tests/golden/gpcov.npz: the reference's own solve_triangular code
    (active_importance_sampling.py:279-306) run on the posterior records -- homoskedastic,
    heteroskedastic, and L_chol=False -- gives these predictive variances / cross terms.
    gp_ref.predict and a dense first-principles solve must agree at 1e-10 (of sf^2)."""
    c = golden("gpcov")
    D, N = int(c["D"]), int(c["N"])
    gp = gpcov_gp(c, name)
    hyp = c[f"{name}_hyp"]
    assert [int(p.L_chol) for p in gp.posteriors] == list(c[f"{name}_L_chol"])
    if name == "tiny":
        assert list(c[f"{name}_L_chol"]) == [0, 1]  # both branches in one GP
    for cls in ("AcqFcnVIQR", "AcqFcnIMIQR"):
        Xa = c[f"{name}_{cls}_Xa"]
        imp = c[f"{name}_{cls}_fs2_implied"]
        fmu, fs2 = gp_ref.predict(gp, Xa, separate_samples=True)
        for s in range(hyp.shape[0]):
            sf2 = np.exp(2 * hyp[s, D])
            assert np.max(np.abs(fs2[:, s] - imp[:, s])) <= 1e-10 * sf2, (name, cls, s)
            # dense first principles: (K + diag(sn2))^-1 with no Cholesky / no record at all
            Kxx = gp_ref.se_ard(hyp[s, : D + 1], c["X"], c["X"])
            sn2 = np.exp(2 * hyp[s, D + 1]) + (c["s2"].ravel() if name == "hetero" else 0.0)
            Ks = gp_ref.se_ard(hyp[s, : D + 1], c["X"], Xa)
            sol = np.linalg.solve(Kxx + np.diag(np.full(N, 1.0) * sn2), Ks)
            dense = sf2 - np.sum(Ks * sol, axis=0)
            # (the dense solve itself loses ~cond(K) * eps: compare at 1e-8 of sf^2)
            assert np.max(np.abs(dense - imp[:, s])) <= 1e-8 * sf2, (name, cls, s)
            cross = Ks[:, :8].T @ sol[:, :8]
            sign = 1.0 if c[f"{name}_L_chol"][s] else -1.0  # C_tmp = L K with L = -(K+S)^-1 when !L_chol
            assert np.max(np.abs(sign * cross - c[f"{name}_{cls}_cross_implied"][s])) <= 1e-8 * sf2
        # the f_s2 the reference's importance-sampling bookkeeping stored came from predict: same numbers
        assert np.max(np.abs(fs2 - c[f"{name}_{cls}_ais_f_s2"])) <= 1e-10 * np.exp(2 * np.max(hyp[:, D]))
