#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE ACTUAL REFERENCE.  TEST INFRASTRUCTURE.

Runs only in the build container (needs /root/reference).  It imports the
reference's hot-path modules -- with oracle/_stubs/ standing in for the
third-party packages that are not installed (gpyreg, corner, cma, imageio) --
evaluates them on the seeded inputs of pyvbmc_amd/synthetic.py and stores
inputs + reference outputs as small .npz fixtures.  It also re-packs the
MATLAB-derived known-answer DATA files the reference's own tests hold
(pyvbmc/testing/**.mat/.txt) into npz; no reference source text is copied.

    python oracle/make_golden.py        # rewrites tests/golden/
"""
import itertools
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT / "oracle" / "_stubs"))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(REF))

import gpyreg as gpr  # noqa: E402  (the stand-in)
import scipy.io  # noqa: E402
from pyvbmc.entropy import entlb_vbmc, entmc_vbmc  # noqa: E402
from pyvbmc.variational_posterior import VariationalPosterior  # noqa: E402
from pyvbmc.vbmc.minimize_adam import minimize_adam  # noqa: E402
from pyvbmc.vbmc.variational_optimization import (  # noqa: E402
    _gp_log_joint,
    _neg_elcbo,
    _soft_bound_loss,
    _vp_bound_loss,
)

from pyvbmc_amd import synthetic  # noqa: E402

OUT = ROOT / "tests" / "golden"


def ref_vp(wl):
    vp = VariationalPosterior(wl.D, wl.K)
    vp.mu = wl.mu.copy()
    vp.sigma = wl.sigma.reshape(1, -1).copy()
    vp.lambd = wl.lambd.reshape(-1, 1).copy()
    vp.w = wl.w.reshape(1, -1).copy()
    vp.eta = wl.eta.reshape(1, -1).copy()
    return vp


def ref_gp(wl, hyp):
    noise = gpr.noise_functions.GaussianNoise(
        constant_add=True, user_provided_add=wl.s2 is not None
    )
    gp = gpr.GP(
        D=wl.D,
        covariance=gpr.covariance_functions.SquaredExponential(),
        mean=gpr.mean_functions.NegativeQuadratic(),
        noise=noise,
    )
    gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=hyp)
    return gp


def flagname(f):
    return "".join("1" if b else "0" for b in f)


def case(name, cfg, S_multi, seed, all_flag_combos=False, **shrink):
    wl = synthetic.make_workload(cfg, S=S_multi, **shrink)
    NsK = wl.NsK
    out = dict(
        cfg=cfg, D=wl.D, K=wl.K, N=wl.N, Ns_total=wl.Ns_total, NsK=NsK, seed=seed,
        mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta,
        X=wl.X, y=wl.y, hyp=wl.hyp, theta=wl.theta,
        s2=np.zeros(0) if wl.s2 is None else wl.s2,
    )
    combos = (
        list(itertools.product([False, True], repeat=4))
        if all_flag_combos
        else [(False,) * 4, (True,) * 4]
    )
    # --- entropy ------------------------------------------------------------
    for gf in combos:
        for jac in (True, False):
            vp = ref_vp(wl)
            np.random.seed(seed)
            H, dH = entmc_vbmc(vp, NsK, gf, jac)
            out[f"entmc_H_{flagname(gf)}_{int(jac)}"] = H
            out[f"entmc_dH_{flagname(gf)}_{int(jac)}"] = dH
            H, dH = entlb_vbmc(ref_vp(wl), gf, jac)
            out[f"entlb_H_{flagname(gf)}_{int(jac)}"] = H
            out[f"entlb_dH_{flagname(gf)}_{int(jac)}"] = dH
    # --- GP expected log joint ------------------------------------------------
    for tag, hyp in (("S1", wl.hyp[:1]), ("SM", wl.hyp)):
        gp = ref_gp(wl, hyp)
        vp = ref_vp(wl)
        G, dG, _, _, _ = _gp_log_joint(vp, gp, True, True, True, False, False)
        out[f"glj_{tag}_G"], out[f"glj_{tag}_dG"] = G, dG
        G, _, varG, _, var_ss, I_sk, J_sjk = _gp_log_joint(vp, gp, False, True, True, True, True)
        out[f"glj_{tag}_var_G"] = G
        out[f"glj_{tag}_varG"] = np.asarray(varG)
        out[f"glj_{tag}_var_ss"] = var_ss
        out[f"glj_{tag}_I_sk"] = I_sk
        out[f"glj_{tag}_J_sjk"] = J_sjk
    # --- negative ELBO ----------------------------------------------------------
    gp = ref_gp(wl, wl.hyp[:1])
    bnd = synthetic.default_theta_bnd(wl)
    # push a few parameters outside the soft bounds so the penalty is exercised
    theta_out = wl.theta.copy()
    theta_out[0] = bnd["ub"][0] + 0.3
    theta_out[wl.D * wl.K] = bnd["ub"][wl.D * wl.K] + 0.2 - np.log(wl.lambd[0])
    theta_out[-1] += 0.7
    out["theta_out"] = theta_out
    for tag, th, tb in (("nobnd", wl.theta, None), ("bnd", wl.theta, bnd), ("bndout", theta_out, bnd)):
        for ns_tag, Ns in (("mc", NsK), ("lb", 0)):
            vp = ref_vp(wl)
            np.random.seed(seed)
            th_in = th.copy()
            F, dF, G, H, varF = _neg_elcbo(th_in, gp, vp, 0.0, Ns, True, False, tb, 0.0, False)
            out[f"elbo_{tag}_{ns_tag}_F"] = F
            out[f"elbo_{tag}_{ns_tag}_dF"] = dF
            out[f"elbo_{tag}_{ns_tag}_G"] = G
            out[f"elbo_{tag}_{ns_tag}_H"] = H
            out[f"elbo_{tag}_{ns_tag}_theta_after"] = th_in
    vp = ref_vp(wl)
    np.random.seed(seed)
    r = _neg_elcbo(wl.theta.copy(), gp, vp, 0.0, NsK, False, True, None, 0.0, True)
    out["elbo_full_F"], out["elbo_full_G"], out["elbo_full_H"] = r[0], r[2], r[3]
    out["elbo_full_varF"] = np.asarray(r[4])
    out["elbo_full_I_sk"], out["elbo_full_J_sjk"] = r[9], r[10]
    # --- mixture pdf --------------------------------------------------------------
    rng = np.random.default_rng(77 + cfg)
    comp = rng.integers(0, wl.K, size=48)
    xq = wl.mu.T[comp] + wl.lambd * wl.sigma[comp, None] * rng.standard_normal((48, wl.D))
    xq = np.vstack([xq, 3.0 * rng.standard_normal((12, wl.D)), 60.0 * np.ones((4, wl.D))])
    out["pdf_x"] = xq
    vp = ref_vp(wl)
    out["pdf_y"] = vp.pdf(xq, orig_flag=False)
    out["pdf_logy"] = vp.pdf(xq, orig_flag=False, log_flag=True)
    yy, dy = vp.pdf(xq, orig_flag=False, grad_flag=True)
    out["pdf_dy"] = dy
    with np.errstate(all="ignore"):
        yy, dy = vp.pdf(xq, orig_flag=False, log_flag=True, grad_flag=True)
    out["pdf_dlogy"] = dy
    for df in (10.0, -2.0, 3.5, -7.0):
        out[f"pdf_y_df{df}"] = vp.pdf(xq, orig_flag=False, df=df)
        out[f"pdf_logy_df{df}"] = vp.pdf(xq, orig_flag=False, log_flag=True, df=df)
    out["pdf_1d"] = vp.pdf(xq[0], orig_flag=False)  # 1-D input -> raveled output
    # --- moments / parameter round trip ---------------------------------------------
    m, c = ref_vp(wl).moments(orig_flag=False, cov_flag=True)
    out["mom_mean"], out["mom_cov"] = m, c
    vp = ref_vp(wl)
    th_raw = wl.theta + 0.1 * np.random.default_rng(5).standard_normal(wl.theta.size)
    vp.set_parameters(th_raw)
    out["rt_theta_in"] = th_raw
    out["rt_mu"], out["rt_sigma"], out["rt_lambd"], out["rt_w"] = (
        vp.mu, vp.sigma.ravel(), vp.lambd.ravel(), vp.w.ravel())
    out["rt_theta_out"] = vp.get_parameters()
    out["rt_theta_out_noraw"] = vp.get_parameters(raw_flag=False)
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(f"wrote {name}: D={wl.D} K={wl.K} N={wl.N} NsK={NsK} keys={len(out)}")


def mid(name, cfg, seed, **shrink):
    """Mid-size cases whose only job is to drive the wave-split entropy kernel through its
    MULTI-BATCH workgroup loop (K * NsK/2 > 32768 rows => rg >= 2, csrc/entropy.hip entmc_plan)
    with reference-generated values: ``entmc_vbmc`` value and all four gradient blocks, and the
    call Adam makes (``_neg_elcbo(compute_grad=True, theta_bnd=...)``, inside and outside the
    soft bounds).  Inputs are the seeded synthetic ones; only small arrays are stored (the
    draws are re-created from ``seed`` by the tests: legacy ``np.random.seed``)."""
    import time

    wl = synthetic.make_workload(cfg, S=1, **shrink)
    NsK = wl.NsK
    out = dict(
        cfg=cfg, D=wl.D, K=wl.K, N=wl.N, Ns_total=wl.Ns_total, NsK=NsK, seed=seed,
        mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta,
        X=wl.X, y=wl.y, hyp=wl.hyp, theta=wl.theta,
        s2=np.zeros(0) if wl.s2 is None else wl.s2,
    )
    t0 = time.time()
    for gf in ((False,) * 4, (True,) * 4):
        vp = ref_vp(wl)
        np.random.seed(seed)
        H, dH = entmc_vbmc(vp, NsK, gf, True)
        out[f"entmc_H_{flagname(gf)}_1"], out[f"entmc_dH_{flagname(gf)}_1"] = H, dH
    gp = ref_gp(wl, wl.hyp[:1])
    bnd = synthetic.default_theta_bnd(wl)
    theta_out = wl.theta.copy()
    theta_out[0] = bnd["ub"][0] + 0.3
    theta_out[wl.D * wl.K] = bnd["ub"][wl.D * wl.K] + 0.2 - np.log(wl.lambd[0])
    theta_out[-1] += 0.7
    out["theta_out"] = theta_out
    for tag, th in (("bnd", wl.theta), ("bndout", theta_out)):
        vp = ref_vp(wl)
        np.random.seed(seed)
        th_in = th.copy()
        F, dF, G, H, _ = _neg_elcbo(th_in, gp, vp, 0.0, NsK, True, False, bnd, 0.0, False)
        out[f"elbo_{tag}_mc_F"], out[f"elbo_{tag}_mc_dF"] = F, dF
        out[f"elbo_{tag}_mc_G"], out[f"elbo_{tag}_mc_H"] = G, H
        out[f"elbo_{tag}_mc_theta_after"] = th_in
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(f"wrote {name}: D={wl.D} K={wl.K} N={wl.N} NsK={NsK} K*rows={wl.K * NsK // 2} "
          f"({time.time() - t0:.0f} s of reference time)")


def gpcov():
    """Pin the GP posterior records' consumption conventions with the reference's IN-TREE code.

    gpyreg is absent, so (L, sW, L_chol, alpha) of a posterior are built by oracle/gp_ref.py.  What
    PyVBMC itself does with those records is in the reference tree: step 3 of
    ``active_importance_sampling`` (vbmc/active_importance_sampling.py:262-306) forms
    ``C_tmp = (L'L)^-1 K(X,Xa) / sn2_eff`` (``L_chol``) or ``L K(X,Xa)`` (otherwise) with
    ``solve_triangular`` on ``posteriors[s].L``, and ``AcqFcnVIQR/IMIQR._compute_acquisition_function``
    (acq_fcn_viqr.py:93-141, acq_fcn_imiqr.py:100-141) turn that into posterior cross-covariances.
    Here the reference runs on stub GPs with MODERATE length scales (K* far from 0) -- homoskedastic,
    heteroskedastic (user s2) and one sample with sn2 < 1e-6 (``L_chol=False``) -- and the implied
    predictive variances / covariances are stored: sf^2 -/+ sum_n K(xa,X)_n C_tmp_n.  The tests hold
    gp_ref.predict, a dense first-principles solve and the device kernels to them at 1e-10.
    The two acquisition classes' values on the same GPs are stored too."""
    from types import SimpleNamespace

    from pyvbmc.acquisition_functions import AcqFcnIMIQR, AcqFcnVIQR
    from pyvbmc.vbmc.active_importance_sampling import active_importance_sampling

    class Opts(dict):
        def eval(self, key, env):
            return self[key]

    out = {}
    D, N, K = 3, 60, 2
    rng = np.random.default_rng(4242)
    X = rng.standard_normal((N, D))
    y = (-0.5 * np.sum(X**2, axis=1) + 0.3 * np.sin(2 * X[:, 0]) + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    s2 = rng.uniform(0.01, 1.0, size=(N, 1))

    def hyp_rows(log_sn):
        rows = []
        for i, ls in enumerate(log_sn):
            ell = np.log(np.array([0.8, 1.3, 1.0]) * (1.0 + 0.4 * i))
            rows.append(np.concatenate([ell, [np.log(2.0 + i)], [ls], [0.3], 0.1 * np.ones(D), np.log(2.0) * np.ones(D)]))
        return np.array(rows)

    cases = {
        "homo": (hyp_rows([np.log(0.05), np.log(0.08)]), None),
        "hetero": (hyp_rows([np.log(0.05), np.log(0.02)]), s2),
        "tiny": (hyp_rows([np.log(3e-4), np.log(0.05)]), None),  # sample 0: sn2 = 9e-8 < 1e-6 -> L_chol False
    }
    out.update(X=X, y=y, s2=s2, D=D, N=N)
    vp = VariationalPosterior(D, K)
    vp.mu = np.array([[0.3, -0.6], [-0.2, 0.5], [0.1, 0.4]])
    vp.sigma = np.array([[0.6, 0.9]])
    vp.lambd = np.array([[1.1], [0.8], [1.0]])
    vp.lambd = vp.lambd / np.sqrt(np.mean(vp.lambd**2))
    vp.w = np.array([[0.35, 0.65]])
    vp.eta = np.log(vp.w)
    out.update(vp_mu=vp.mu, vp_sigma=vp.sigma.ravel(), vp_lambd=vp.lambd.ravel(), vp_w=vp.w.ravel())
    Xs = rng.standard_normal((32, D)) * 1.2
    Xs[:4] = X[:4] + 0.02 * rng.standard_normal((4, D))
    out["Xs"] = Xs
    for name, (hyp, s2c) in cases.items():
        wl = SimpleNamespace(D=D, X=X, y=y, s2=s2c)
        gp = ref_gp(wl, hyp)
        S = hyp.shape[0]
        out[f"{name}_hyp"] = hyp
        out[f"{name}_L_chol"] = np.array([int(p.L_chol) for p in gp.posteriors])
        sf2 = np.exp(2 * hyp[:, D])
        length = np.exp(hyp[0, :D])
        gp.temporary_data["X_rescaled"] = X / length
        gp.temporary_data["sn2_new"] = 0.01 + rng.random(N)
        out[f"{name}_sn2_new"] = gp.temporary_data["sn2_new"]
        for acq_cls, opts in (
            (AcqFcnVIQR, Opts(active_importance_sampling_mcmc_samples=48)),
            (AcqFcnIMIQR, Opts(active_importance_sampling_vp_samples=30, active_importance_sampling_box_samples=18,
                               active_importance_sampling_mcmc_samples=0)),
        ):
            tag = f"{name}_{acq_cls.__name__}"
            acq = acq_cls()
            np.random.seed(11)
            ais = active_importance_sampling(vp, gp, acq, opts)
            Xa, KaX, Ct = ais["X"], ais["K_Xa_X"], ais["C_tmp"]
            assert Xa.ndim == 2 and KaX.shape == (S, Xa.shape[0], N) and Ct.shape == (S, N, Xa.shape[0])
            vr = np.einsum("san,sna->as", KaX, Ct)  # sum_n K(xa, X_n) C_tmp[n, a]
            sign = np.where(out[f"{name}_L_chol"] == 1, -1.0, 1.0)
            out[f"{tag}_Xa"] = Xa
            out[f"{tag}_fs2_implied"] = sf2[None, :] + sign[None, :] * vr  # (Na, S), unclamped
            # cross terms between the first 8 points: K(xa,X) C_tmp, to be subtracted from / added to K(xa,xa')
            out[f"{tag}_cross_implied"] = np.einsum("san,snb->sab", KaX[:, :8, :], Ct[:, :, :8])
            out[f"{tag}_ais_f_s2"], out[f"{tag}_ais_ln_weights"] = ais["f_s2"], ais["ln_weights"]
            st = dict(integer_vars=None, lb_eps_orig=X.min(0) - 3.0, ub_eps_orig=X.max(0) + 3.0,
                      gp_length_scale=length, variance_regularized_acq_fcn=False, active_importance_sampling=ais)
            flog = SimpleNamespace(y_max=float(np.max(y)))
            with np.errstate(all="ignore"):
                out[f"{tag}_acq"] = acq(Xs.copy(), gp, vp, flog, st)
        print(f"gpcov {name}: L_chol={out[f'{name}_L_chol']}, "
              f"min implied fs2/sf2 = {np.min(out[f'{name}_AcqFcnVIQR_fs2_implied'] / sf2):.3e}")
    np.savez_compressed(OUT / "gpcov.npz", **out)
    print("wrote gpcov:", len(out), "keys")


def matlab_known():
    """Known-answer DATA held by the reference's own tests, re-packed as npz."""
    out = {}
    m = scipy.io.loadmat(
        REF / "pyvbmc/testing/entropy/entropy-test.mat", struct_as_record=False, squeeze_me=True
    )
    vp = m["vp"]
    out.update(
        ent_D=int(m["D"]), ent_K=int(m["K"]), ent_Ns=int(m["Ns"]), ent_H=float(m["H"]),
        ent_Hl=float(m["Hl"]), ent_dH=np.asarray(m["dH"]).ravel(), ent_dHl=np.asarray(m["dHl"]).ravel(),
        ent_jacobian_flag=int(m["jacobian_flag"]),
        ent_mu=np.asarray(vp.mu, dtype=float), ent_sigma=np.asarray(vp.sigma, dtype=float).ravel(),
        ent_lambd=np.asarray(getattr(vp, "lambda"), dtype=float).ravel(),
        ent_w=np.asarray(vp.w, dtype=float).ravel(), ent_eta=np.asarray(vp.eta, dtype=float).ravel(),
    )
    m = scipy.io.loadmat(
        REF / "pyvbmc/testing/variational_posterior/test_moments_no_orig_flag_2_MATLAB.mat",
        struct_as_record=False, squeeze_me=True,
    )
    for k, v in m.items():
        if not k.startswith("__"):
            out[f"mom_{k}"] = np.asarray(v, dtype=float)
    vb = REF / "pyvbmc/testing/vbmc"
    for f in ("X", "y", "hyp", "mu", "dG_gp_log_joint", "dF"):
        out[f"vbmc_{f}"] = np.loadtxt(vb / f"{f}.txt", delimiter=",")
    # constants asserted at test_variational_optimization.py:143-147,192-198
    out.update(
        vbmc_G=-0.461812484952867, vbmc_varG=6.598768992700180e-05,
        vbmc_var_ss=1.031705745662353e-04, vbmc_F=11.746298071422430,
        vbmc_H=-11.284485586469563,
    )
    for f in ("fess", "activesample_proposalpdf"):
        m = scipy.io.loadmat(vb / "compare_MATLAB" / f"{f}.mat")
        for k, v in m.items():
            if not k.startswith("__"):
                out[f"{f}_{k}"] = np.asarray(v, dtype=float)
    vpdir = REF / "pyvbmc/testing/variational_posterior"
    for f in ("X", "mu", "bnd_lb", "bnd_ub"):
        out[f"vp_{f}"] = np.loadtxt(vpdir / f"{f}.txt", delimiter=",")
    np.savez_compressed(OUT / "matlab_known.npz", **out)
    print("wrote matlab_known:", sorted(out))


def misc():
    """Small direct calls: soft-bound loss and vp-bound loss known inputs."""
    out = {}
    x = np.zeros(3)
    x[0], x[1] = 15.0, -20.0
    L, dL = _soft_bound_loss(x, np.full(3, -10.0), np.full(3, 10.0), compute_grad=True)
    out["sbl_x"], out["sbl_L"], out["sbl_dL"] = x, L, dL
    np.savez_compressed(OUT / "misc.npz", **out)


def adam():
    """The reference's minimize_adam (vbmc/minimize_adam.py) on (1) a deterministic,
    box-constrained quadratic with a wiggle and (2) the objective PyVBMC gives it,
    _neg_elcbo with fresh np.random draws per iteration (variational_optimization.py:238-249),
    on config 1 and a shrunken config 2."""
    out = {}
    rng = np.random.default_rng(123)
    n = 7
    a, c = np.exp(rng.standard_normal(n)), rng.standard_normal(n)
    x0 = c + 2.0 * rng.standard_normal(n)
    lb, ub = c - 0.5, c + 3.0
    lb[0], ub[1] = c[0] + 0.2, c[1] - 0.3  # active constraints

    def fq(x):
        wob = 0.01 * np.sin(37.0 * np.sum(x))
        return 0.5 * np.sum(a * (x - c) ** 2) + wob, a * (x - c) + 0.37 * np.cos(37.0 * np.sum(x))

    out.update(quad_a=a, quad_c=c, quad_x0=x0.copy(), quad_lb=lb, quad_ub=ub)
    for tag, kw in (("box", dict(lb=lb, ub=ub, max_iter=400)),
                    ("free", dict(max_iter=90, master_max=0.05, use_early_stopping=False)),
                    ("short", dict(max_iter=25, tol_fun=0.5))):
        x, y, xt, yt, it = minimize_adam(fq, x0.copy(), **kw)
        out[f"quad_{tag}_x"], out[f"quad_{tag}_y"] = x, y
        out[f"quad_{tag}_x_tab"], out[f"quad_{tag}_y_tab"], out[f"quad_{tag}_iters"] = xt, yt, it
    for name, cfg, shrink, max_iter in (("c1", 1, {}, 80), ("c2s", 2, dict(Ns_total=20 * 100), 45)):
        wl = synthetic.make_workload(cfg, S=1, **shrink)
        gp = ref_gp(wl, wl.hyp[:1])
        bnd = synthetic.default_theta_bnd(wl)
        vp = ref_vp(wl)
        theta0 = wl.theta.copy()
        theta0[0] = bnd["ub"][0] + 0.1  # start outside a soft bound
        out[f"elbo_{name}_theta0"] = theta0.copy()
        out[f"elbo_{name}_NsK"] = wl.NsK

        def f(t):
            r = _neg_elcbo(t, gp, vp, 0.0, wl.NsK, True, False, bnd)
            return r[0], r[1]

        np.random.seed(40 + cfg)
        out[f"elbo_{name}_seed"] = 40 + cfg
        x, y, xt, yt, it = minimize_adam(f, theta0, tol_fun=0.05, max_iter=max_iter,
                                         master_min=0.001, master_max=0.1, master_decay=200)
        out[f"elbo_{name}_x"], out[f"elbo_{name}_y"] = x, y
        out[f"elbo_{name}_x_tab"], out[f"elbo_{name}_y_tab"], out[f"elbo_{name}_iters"] = xt, yt, it
        print(f"adam elbo {name}: {it} iterations, y {yt[0]:.4f} -> {yt[-1]:.4f}")
    np.savez_compressed(OUT / "adam.npz", **out)
    print("wrote adam:", len(out), "keys")


def acq():
    """The reference's closed-form acquisition classes (acquisition_functions/acq_fcn*.py via
    AbstractAcqFcn.__call__) on seeded points: gp.predict through the gpyreg stand-in, vp.pdf
    and the formulas are the reference's own code."""
    from types import SimpleNamespace

    from pyvbmc.acquisition_functions import AcqFcn, AcqFcnLog, AcqFcnNoisy, AcqFcnVanilla
    from pyvbmc.acquisition_functions.abstract_acq_fcn import AbstractAcqFcn

    out = {}
    rng = np.random.default_rng(2024)
    # _sq_dist on its own (the reference's test_sq_dist checks it against a direct loop)
    a, b = 3.0 + rng.standard_normal((37, 5)), 3.0 + 2.0 * rng.standard_normal((71, 5))
    out["sq_a"], out["sq_b"], out["sq_c"] = a, b, AbstractAcqFcn._sq_dist(a, b)
    for name, cfg, S, shrink in (("c1", 1, 2, {}), ("c2s", 2, 3, dict(Ns_total=20 * 100))):
        wl = synthetic.make_workload(cfg, S=S, **shrink)
        gp = ref_gp(wl, wl.hyp)
        vp = ref_vp(wl)
        M = 96
        comp = rng.integers(0, wl.K, size=M)
        Xs = wl.mu.T[comp] + 1.5 * wl.lambd * wl.sigma[comp, None] * rng.standard_normal((M, wl.D))
        Xs[:8] = wl.X[:8] + 1e-3 * rng.standard_normal((8, wl.D))  # near training inputs: tiny variance
        lo, hi = wl.X.min(axis=0) - 1.0, wl.X.max(axis=0) + 1.0
        Xs[-3:] = hi + 0.5  # beyond the hard bounds
        length = np.exp(wl.hyp[0, : wl.D])
        gp.temporary_data["X_rescaled"] = wl.X / length
        gp.temporary_data["sn2_new"] = 0.01 + rng.random(wl.N)
        flog = SimpleNamespace(y_max=float(np.max(wl.y)))
        base = dict(integer_vars=None, lb_eps_orig=lo, ub_eps_orig=hi, gp_length_scale=length)
        out[f"{name}_Xs"], out[f"{name}_lo"], out[f"{name}_hi"] = Xs, lo, hi
        out[f"{name}_sn2_new"], out[f"{name}_y_max"], out[f"{name}_S"] = gp.temporary_data["sn2_new"], flog.y_max, S
        f_mu, f_s2 = gp.predict(Xs, separate_samples=True)
        var_tot = f_s2.mean(axis=1) + f_mu.var(axis=1, ddof=1)
        tol = float(np.sort(var_tot)[12])  # a dozen points fall below the variance tolerance
        out[f"{name}_tol_gp_var"] = tol
        for cls in (AcqFcn, AcqFcnLog, AcqFcnVanilla, AcqFcnNoisy):
            for reg in (False, True):
                st = dict(base, variance_regularized_acq_fcn=reg, tol_gp_var=tol)
                with np.errstate(all="ignore"):
                    v = cls()(Xs.copy(), gp, vp, flog, st)
                out[f"{name}_{cls.__name__}_{int(reg)}"] = v
        out[f"{name}_one"] = AcqFcnLog()(Xs[20].copy(), gp, vp, flog, dict(base))  # 1-D input
    np.savez_compressed(OUT / "acq.npz", **out)
    print("wrote acq:", len(out), "keys")


def vpmc():
    """The reference's RNG-bound VariationalPosterior methods under a seeded NumPy stream:
    sample (plain / balanced), Monte-Carlo moments, kl_div (both branches) and kl_div_mvn."""
    from pyvbmc.stats import kl_div_mvn

    out = {}
    for name, cfg, shrink in (("c1", 1, {}), ("c2s", 2, dict(Ns_total=20 * 100))):
        wl = synthetic.make_workload(cfg, S=1, **shrink)
        vp = ref_vp(wl)
        vp2 = ref_vp(wl)
        rng = np.random.default_rng(60 + cfg)
        vp2.mu = vp2.mu + 0.2 * rng.standard_normal(vp2.mu.shape)
        vp2.sigma = vp2.sigma * np.exp(0.1 * rng.standard_normal(vp2.sigma.shape))
        w2 = vp2.w * np.exp(0.3 * rng.standard_normal(vp2.w.shape))
        vp2.w = w2 / w2.sum()
        out[f"{name}_mu2"], out[f"{name}_sigma2"], out[f"{name}_w2"] = vp2.mu, vp2.sigma.ravel(), vp2.w.ravel()
        for bal in (False, True):
            np.random.seed(7)
            x, i = vp.sample(500, orig_flag=False, balance_flag=bal)
            out[f"{name}_sample_x_{int(bal)}"], out[f"{name}_sample_i_{int(bal)}"] = x, i
        np.random.seed(8)
        m, c = vp.moments(20000, orig_flag=True, cov_flag=True)
        out[f"{name}_mom_mc_mean"], out[f"{name}_mom_mc_cov"] = m, c
        np.random.seed(9)
        out[f"{name}_kl_mc"] = vp.kl_div(vp2, N=20000)
        np.random.seed(10)
        out[f"{name}_kl_gauss"] = vp.kl_div(vp2, N=20000, gauss_flag=True)
        m1, c1 = vp.moments(orig_flag=False, cov_flag=True)
        m2, c2 = vp2.moments(orig_flag=False, cov_flag=True)
        out[f"{name}_kl_mvn"] = kl_div_mvn(m1, c1, m2, c2)
    np.savez_compressed(OUT / "vpmc.npz", **out)
    print("wrote vpmc:", len(out), "keys")


MEAN_CLASSES = {"zero": "ZeroMean", "const": "ConstantMean", "negquad": "NegativeQuadratic"}


def _mean_hyp(wl, kind, hyp):
    """Columns of the synthetic hyper-parameter rows a GP with this mean function carries:
    [cov (D+1) | noise (1) | mean (0 / 1 / 1+2D)] (variational_optimization.py:1383-1392)."""
    n_mean = {"zero": 0, "const": 1, "negquad": 1 + 2 * wl.D}[kind]
    return np.ascontiguousarray(hyp[:, : wl.D + 2 + n_mean])


def ref_gp_kind(wl, kind, hyp):
    gp = gpr.GP(
        D=wl.D,
        covariance=gpr.covariance_functions.SquaredExponential(),
        mean=getattr(gpr.mean_functions, MEAN_CLASSES[kind])(),
        noise=gpr.noise_functions.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None),
    )
    gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=_mean_hyp(wl, kind, hyp))
    return gp


def variants():
    """The branches the reference takes outside "all blocks optimised, NegativeQuadratic mean",
    run by the REFERENCE on a D=6, K=20, N=200 workload (config 2's shape, S = 3 hyper-samples):

    * GP mean kinds zero / const / negquad (gaussian_process_train.py:219-224; vbmc.py:1064-1078
      switches to "const" temporarily): ``_gp_log_joint`` with avg_flag x jacobian_flag, variance +
      separate_K, ``_neg_elcbo`` (lower-bound and Monte-Carlo entropy), the closed-form acquisition
      classes and a ``minimize_adam`` trajectory per kind;
    * the partial optimise masks the reference can produce -- weights off (every warm-up iteration,
      variational_optimization.py:142-143), means off (options variable_means, vbmc.py:249), both --
      through ``_neg_elcbo`` with the reduced ``theta_bnd`` of the reference's own
      ``vp.get_bounds`` (variational_posterior.py:140-239);
    * ``pdf`` / ``log_pdf`` with ``orig_flag=True`` on a bounded ``ParameterTransformer`` (finite and
      infinite bounds mixed, plausible bounds => centring), rows inside, ON and outside the bounds
      (variational_posterior.py:429-439, 543-559), with the transformer's own outputs stored so
      that oracle/transform_ref.py is pinned as well.
    """
    from types import SimpleNamespace

    from pyvbmc.acquisition_functions import AcqFcn, AcqFcnLog
    from pyvbmc.parameter_transformer import ParameterTransformer

    out = {}
    S = 3
    wl = synthetic.make_workload(2, S=S, Ns_total=20 * 200)
    D, K, NsK = wl.D, wl.K, wl.NsK
    seed = 21
    out.update(D=D, K=K, N=wl.N, NsK=NsK, seed=seed, S=S, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w,
               eta=wl.eta, X=wl.X, y=wl.y, hyp=wl.hyp, theta=wl.theta, s2=np.zeros(0))
    opts = {"tol_length": 1e-6, "tol_weight": 1e-2, "tol_con_loss": 0.01, "weight_penalty": 0.1}
    rng = np.random.default_rng(909)
    Xs = wl.mu.T[rng.integers(0, K, size=40)] + 1.2 * rng.standard_normal((40, D))
    Xs[:4] = wl.X[:4] + 1e-3 * rng.standard_normal((4, D))
    out["Xs"] = Xs

    # ---- GP mean kinds -----------------------------------------------------------------------
    for kind in MEAN_CLASSES:
        for tag, hyp in (("S1", wl.hyp[:1]), ("SM", wl.hyp)):
            gp = ref_gp_kind(wl, kind, hyp)
            for avg in (True, False):
                for jac in (True, False):
                    G, dG, _, _, _ = _gp_log_joint(ref_vp(wl), gp, True, avg, jac, False, False)
                    k = f"glj_{kind}_{tag}_a{int(avg)}_j{int(jac)}"
                    out[k + "_G"], out[k + "_dG"] = np.asarray(G), np.asarray(dG)
                G, _, varG, _, var_ss, I_sk, J_sjk = _gp_log_joint(ref_vp(wl), gp, False, avg, True, True, True)
                k = f"glj_{kind}_{tag}_a{int(avg)}_var"
                out[k + "_G"], out[k + "_varG"], out[k + "_var_ss"] = np.asarray(G), np.asarray(varG), var_ss
                out[k + "_I_sk"], out[k + "_J_sjk"] = I_sk, J_sjk
        gp = ref_gp_kind(wl, kind, wl.hyp[:1])
        gpM = ref_gp_kind(wl, kind, wl.hyp)
        vp = ref_vp(wl)
        bnd = vp.get_bounds(wl.X, opts)
        if kind == "zero":
            out["bnd_full_lb"], out["bnd_full_ub"] = bnd["lb"], bnd["ub"]
            out["bnd_tol_con"], out["bnd_weight_threshold"] = bnd["tol_con"], bnd["weight_threshold"]
            out["bnd_weight_penalty"] = bnd["weight_penalty"]
        for g_, gtag in ((gp, "S1"), (gpM, "SM")):
            for ns_tag, Ns in (("mc", NsK), ("lb", 0)):
                vp = ref_vp(wl)
                np.random.seed(seed)
                th = wl.theta.copy()
                F, dF, G, H, _ = _neg_elcbo(th, g_, vp, 0.0, Ns, True, False, bnd, 0.0, False)
                k = f"elbo_{kind}_{gtag}_{ns_tag}"
                out[k + "_F"], out[k + "_dF"], out[k + "_G"], out[k + "_H"] = F, dF, G, H
        # the acquisition classes through AbstractAcqFcn.__call__ (predict by the gpyreg stand-in)
        length = np.exp(wl.hyp[0, :D])
        gpM.temporary_data["X_rescaled"] = wl.X / length
        flog = SimpleNamespace(y_max=float(np.max(wl.y)))
        st = dict(integer_vars=None, lb_eps_orig=wl.X.min(0) - 2.0, ub_eps_orig=wl.X.max(0) + 2.0,
                  gp_length_scale=length, variance_regularized_acq_fcn=False)
        for cls in (AcqFcn, AcqFcnLog):
            with np.errstate(all="ignore"):
                out[f"acq_{kind}_{cls.__name__}"] = cls()(Xs.copy(), gpM, ref_vp(wl), flog, dict(st))
        # what the gpyreg stand-in's predict returned for those calls (restated third-party arithmetic)
        fmu, fs2 = gpM.predict(Xs, separate_samples=True)
        out[f"pred_{kind}_fmu"], out[f"pred_{kind}_fs2"] = fmu, fs2
        # a short optimiser trajectory per kind: the reference's minimize_adam around its _neg_elcbo
        vp = ref_vp(wl)
        theta0 = wl.theta.copy()
        theta0[0] = bnd["ub"][0] + 0.1

        def f(t, gp=gp, vp=vp, bnd=bnd):
            r = _neg_elcbo(t, gp, vp, 0.0, 40, True, False, bnd)
            return r[0], r[1]

        np.random.seed(70)
        x, y, xt, yt, it = minimize_adam(f, theta0.copy(), tol_fun=0.05, max_iter=30, master_min=0.001,
                                         master_max=0.1, master_decay=200)
        out[f"adam_{kind}_theta0"] = theta0
        out[f"adam_{kind}_x"], out[f"adam_{kind}_y"] = x, y
        out[f"adam_{kind}_x_tab"], out[f"adam_{kind}_y_tab"], out[f"adam_{kind}_iters"] = xt, yt, it
        print(f"variants {kind}: adam {it} iterations, y {yt[0]:.4f} -> {yt[-1]:.4f}")

    # ---- partial optimise masks -----------------------------------------------------------------
    gp = ref_gp_kind(wl, "negquad", wl.hyp[:1])
    gpc = ref_gp_kind(wl, "const", wl.hyp)
    for mname, flags in (("mask1110", (1, 1, 1, 0)), ("mask0111", (0, 1, 1, 1)), ("mask0110", (0, 1, 1, 0))):
        def masked_vp():
            vp = ref_vp(wl)
            vp.optimize_mu, vp.optimize_sigma, vp.optimize_lambd, vp.optimize_weights = map(bool, flags)
            return vp

        vp = masked_vp()
        bnd = vp.get_bounds(wl.X, opts)
        theta = vp.get_parameters()
        theta = theta + 0.05 * np.random.default_rng(31).standard_normal(theta.size)
        if flags[0]:
            theta[0] = bnd["ub"][0] + 0.3  # outside a soft bound
        out[f"{mname}_theta"], out[f"{mname}_lb"], out[f"{mname}_ub"] = theta, bnd["lb"], bnd["ub"]
        out[f"{mname}_has_weight_keys"] = int("weight_threshold" in bnd)
        for g_, gtag in ((gp, "nq1"), (gpc, "constM")):
            for ns_tag, Ns in (("mc", NsK), ("lb", 0)):
                vp = masked_vp()
                np.random.seed(seed)
                th = theta.copy()
                F, dF, G, H, _ = _neg_elcbo(th, g_, vp, 0.0, Ns, True, False, bnd, 0.0, False)
                k = f"{mname}_{gtag}_{ns_tag}"
                out[k + "_F"], out[k + "_dF"], out[k + "_G"], out[k + "_H"] = F, dF, G, H
                out[k + "_theta_after"] = th
                out[k + "_mu"], out[k + "_sigma"], out[k + "_lambd"] = vp.mu, vp.sigma.ravel(), vp.lambd.ravel()
                out[k + "_w"], out[k + "_eta"] = vp.w.ravel(), vp.eta.ravel()
            # value only, no bounds: what _sieve's candidates see when theta_bnd is None
            vp = masked_vp()
            th = theta.copy()
            out[f"{mname}_{gtag}_lb_F_nograd"] = _neg_elcbo(th, g_, vp, 0.0, 0, False, False, bnd, 0.0, False)[0]
        # the entropy estimators with exactly these blocks (disabled blocks omitted, entmc_vbmc.py:48-51)
        gf = tuple(map(bool, flags))
        vp = ref_vp(wl)  # (the constructor consumes np.random: build first, seed after)
        np.random.seed(seed)
        out[f"{mname}_entmc_H"], out[f"{mname}_entmc_dH"] = entmc_vbmc(vp, NsK, gf, True)
        out[f"{mname}_entlb_H"], out[f"{mname}_entlb_dH"] = entlb_vbmc(ref_vp(wl), gf, True)

    # ---- orig-space density with a bounded transformer ----------------------------------------------
    lb = np.array([[-3.0, -np.inf, 0.0, -2.0, -np.inf, -5.0]])
    ub = np.array([[3.0, np.inf, 4.0, 6.0, np.inf, 5.0]])
    plb = np.array([[-2.0, -1.5, 0.5, -1.0, -2.0, -4.0]])
    pub = np.array([[2.5, 1.5, 3.0, 4.0, 2.0, 4.5]])
    pt = ParameterTransformer(D, lb, ub, plb, pub)
    out.update(pt_lb=lb, pt_ub=ub, pt_plb=plb, pt_pub=pub, pt_mu=pt.mu, pt_delta=pt.delta, pt_type=pt.type)
    vp = ref_vp(wl)
    vp.parameter_transformer = pt
    rngp = np.random.default_rng(444)
    comp = rngp.integers(0, K, size=40)
    u = wl.mu.T[comp] + wl.lambd * wl.sigma[comp, None] * rngp.standard_normal((40, D))
    x_in = pt.inverse(u)
    x_edge = pt.inverse(wl.mu.T[:6].copy())
    x_edge[0, 0] = lb[0, 0]            # ON the lower bound
    x_edge[1, 2] = ub[0, 2]            # ON the upper bound
    x_edge[2, 3] = lb[0, 3] - 1e-3     # just outside
    x_edge[3, 5] = ub[0, 5] + 2.0      # far outside
    x_edge[4, 0] = np.nextafter(lb[0, 0], np.inf)   # the closest inside value
    x_edge[5, 2] = np.nextafter(ub[0, 2], -np.inf)
    xo = np.vstack([x_in, x_edge])
    out["pdfo_x"] = xo
    mask = np.logical_and(np.all(xo > lb, axis=1), np.all(xo < ub, axis=1))
    out["pdfo_mask"] = mask
    out["pdfo_u"] = pt(xo[mask])
    out["pdfo_ladj"] = pt.log_abs_det_jacobian(out["pdfo_u"])
    out["pdfo_inv"] = pt.inverse(out["pdfo_u"])
    with np.errstate(all="ignore"):
        out["pdfo_y"] = vp.pdf(xo, orig_flag=True)
        out["pdfo_logy"] = vp.pdf(xo, orig_flag=True, log_flag=True)
        out["pdfo_logy_method"] = vp.log_pdf(xo, orig_flag=True)
        yy, dy = vp.pdf(xo, orig_flag=True, grad_flag=True)
        out["pdfo_y_g"], out["pdfo_dy"] = yy, dy
        for df in (7.0, -3.0):
            out[f"pdfo_y_df{df}"] = vp.pdf(xo, orig_flag=True, df=df)
            out[f"pdfo_logy_df{df}"] = vp.pdf(xo, orig_flag=True, log_flag=True, df=df)
        out["pdfo_1d"] = vp.pdf(xo[3], orig_flag=True)
        # the reference's own edge test (test_variational_posterior.py:304-332): D = 2, lb = -3, ub = 3
        lb2, ub2 = -3.0 * np.ones((1, 2)), 3.0 * np.ones((1, 2))
        pt2 = ParameterTransformer(2, lb2, ub2)
        vp2 = VariationalPosterior(2, 2, np.array([[2.0, 2.0], [-2.0, -2.0]]), pt2)
        vp2.sigma = np.ones((1, 2))
        pts = np.vstack([lb2, lb2 - 1e-3, ub2, ub2 + 1e-3, lb2 + 0.5, ub2 - 0.5])
        out["pdfo2_mu"], out["pdfo2_x"] = vp2.mu, pts
        out["pdfo2_y"] = vp2.pdf(pts, orig_flag=True)
        out["pdfo2_logy"] = vp2.log_pdf(pts, orig_flag=True)
    np.random.seed(12)
    xs_o, _ = vp.sample(300, orig_flag=True, balance_flag=True)
    out["pdfo_sample_x"] = xs_o  # (inverse transform of the reference's NumPy-stream samples)
    np.savez_compressed(OUT / "variants.npz", **out)
    print("wrote variants:", len(out), "keys")


def is_known():
    """What the reference's ``fess`` and ``active_sample_proposal_pdf`` return on the inputs of its own
    MATLAB known-answer tests (testing/vbmc/test_active_importance_sampling.py:113-250), together with
    the values of ``gp.predict`` and ``vp.pdf`` those two functions consumed on the way -- recorded by
    wrapping the two methods while the reference runs.  The tests hold the device's predict / pdf to
    the recorded values; no restatement of the two functions travels."""
    import scipy.stats as sps

    from pyvbmc.acquisition_functions import AcqFcnIMIQR, AcqFcnVIQR
    from pyvbmc.vbmc.active_importance_sampling import active_sample_proposal_pdf, fess

    out = {}
    D, K = 3, 2
    X = np.arange(-7, 8).reshape((5, 3), order="F").astype(float)
    y = np.array([sps.multivariate_normal.logpdf(x, mean=np.zeros(D)) for x in X]).reshape((-1, 1))
    hyp = np.array([-2.0, -3.0, -4.0, 1.0, 0.0, -(D / 2) * np.log(2 * np.pi), 0.0, 0.25, 0.5, -0.5, 0.0, 0.5])
    hyp = np.vstack([hyp, 2 * hyp])
    wl = SimpleNamespaceWL(D=D, X=X, y=y, s2=None)
    Xa = 2 * np.arange(-4, 5).reshape((3, 3), order="F") / np.pi
    out.update(X=X, y=y, hyp=hyp, Xa=Xa)

    def recording(gp, vp, log):
        p0, d0 = gp.predict, vp.pdf

        def predict(x, *a, **k):
            r = p0(x, *a, **k)
            log.append(("predict", np.array(x), bool(k.get("separate_samples", False)), r))
            return r

        def pdf(x, *a, **k):
            r = d0(x, *a, **k)
            log.append(("pdf", np.array(x), dict(k), r))
            return r

        gp.predict, vp.pdf = predict, pdf

    # fess (test_fess)
    vp = VariationalPosterior(D=D, K=K)
    vp.mu = np.array([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]).T
    vp.w = np.array([[0.7, 0.3]])
    vp.lambd = np.ones(vp.lambd.shape)
    out.update(fess_mu=vp.mu, fess_sigma=vp.sigma.ravel(), fess_w=vp.w.ravel())
    gp = ref_gp(wl, hyp)
    gp_means = np.arange(-5, 5).reshape((5, 2), order="F") * np.pi
    log = []
    recording(gp, vp, log)
    out["fess_means"] = fess(vp, gp_means, X)
    out["fess_means_logpdf"] = log[-1][3]   # vp.pdf(X, log) consumed by the call above
    log.clear()
    out["fess_gp"] = fess(vp, gp, Xa)
    (k1, x1, sep1, r1), (k2, x2, kw2, r2) = log
    assert k1 == "predict" and k2 == "pdf" and not sep1 and np.array_equal(x1, Xa) and np.array_equal(x2, Xa)
    out["fess_gp_fbar"], out["fess_gp_fs2"], out["fess_gp_logpdf"] = r1[0], r1[1], r2
    # active_sample_proposal_pdf (test_active_sample_proposal_pdf)
    vp = VariationalPosterior(D=D, K=K)
    vp.mu = np.array([[-1.0, -2.0, -3.0], [3.0, 2.0, 1.0]]).T
    vp.w = np.array([[0.7, 0.3]])
    vp.sigma = np.ones(vp.sigma.shape)
    vp.lambd = np.ones(vp.lambd.shape)
    out["aspp_mu"] = vp.mu
    gp = ref_gp(wl, hyp)
    rect_delta = 2 * np.std(gp.X, ddof=1, axis=0)
    for name, acq in (("viqr", AcqFcnVIQR()), ("imiqr", AcqFcnIMIQR())):
        log = []
        g2, v2 = ref_gp(wl, hyp), vp
        p_keep = v2.pdf
        recording(g2, v2, log)
        lw, fs2 = active_sample_proposal_pdf(Xa, g2, v2, 0.5, rect_delta, acq)
        v2.pdf = p_keep
        out[f"aspp_{name}_ln_weights"], out[f"aspp_{name}_f_s2"] = lw, fs2
        pred = [e for e in log if e[0] == "predict"][0]
        dens = [e for e in log if e[0] == "pdf"][0]
        assert pred[2] and np.array_equal(pred[1], Xa)
        out[f"aspp_{name}_fmu"], out[f"aspp_{name}_pred_fs2"], out[f"aspp_{name}_logpdf"] = pred[3][0], pred[3][1], dens[3]
    np.savez_compressed(OUT / "is_known.npz", **out)
    print("wrote is_known:", sorted(out))


class SimpleNamespaceWL:
    def __init__(self, **kw):
        self.__dict__.update(kw)


if __name__ == "__main__":
    OUT.mkdir(parents=True, exist_ok=True)
    jobs = {
        "c1": lambda: case("c1", 1, S_multi=2, seed=1, all_flag_combos=True),
        "c2s": lambda: case("c2s", 2, S_multi=3, seed=2, Ns_total=20 * 200),
        "c3s": lambda: case("c3s", 3, S_multi=8, seed=3, Ns_total=50 * 200),
        "c5s": lambda: case("c5s", 5, S_multi=2, seed=5, Ns_total=100 * 100),
        # multi-batch workgroups of the entropy kernel (K * NsK / 2 rows: 50 000, 50 000, 100 000)
        "c2f": lambda: mid("c2f", 2, seed=12),
        "c3m": lambda: mid("c3m", 3, seed=13, Ns_total=50 * 2000),
        "c5m": lambda: mid("c5m", 5, seed=15, Ns_total=100 * 2000),
        "gpcov": gpcov,
        "matlab_known": matlab_known,
        "misc": misc,
        "adam": adam,
        "acq": acq,
        "vpmc": vpmc,
        "variants": variants,
        "is_known": is_known,
    }
    for name in sys.argv[1:] or list(jobs):  # no argument: rewrite everything
        jobs[name]()
