"""SURVEY 8f row 3 (acquisition evaluation) and row a13 (_sq_dist).

tests/golden/acq.npz holds outputs of the reference's own AcqFcn* classes
(oracle/make_golden.py acq).  CPU: the oracle restatement against them.  GPU: the device
path (one vbmc_acq_eval call per batch) against them and, at the batch size the reference
uses for its cached search (2**13 points), against the oracle.
"""
from types import SimpleNamespace

import numpy as np
import pytest
from helpers import oracle_gp, oracle_mix, rel_err

from oracle import acq_ref
from pyvbmc_amd import synthetic

CASES = {"c1": (1, 2, {}), "c2s": (2, 3, dict(Ns_total=20 * 100))}
KINDS = {"AcqFcn": acq_ref.STD, "AcqFcnLog": acq_ref.LOG, "AcqFcnVanilla": acq_ref.VANILLA,
         "AcqFcnNoisy": acq_ref.NOISY}


def setup_case(g, name):
    cfg, S, shrink = CASES[name]
    wl = synthetic.make_workload(cfg, S=S, **shrink)
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X,
              y=wl.y, hyp=wl.hyp, s2=np.zeros(0))
    length = np.exp(wl.hyp[0, : wl.D])
    state = dict(integer_vars=None, lb_eps_orig=g[f"{name}_lo"], ub_eps_orig=g[f"{name}_hi"],
                 gp_length_scale=length, tol_gp_var=float(g[f"{name}_tol_gp_var"]))
    return wl, wd, state, wl.X / length


def close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    inf = np.isinf(b)
    assert np.array_equal(np.isinf(a), inf)
    assert np.array_equal(a[inf], b[inf])
    err = np.abs(a[~inf] - b[~inf]) / np.maximum(1.0, np.abs(b[~inf]))
    assert err.max() < tol, err.max()


# ---------------------------------------------------------------- CPU
def test_oracle_sq_dist_vs_reference(golden):
    g = golden("acq")
    c = acq_ref.sq_dist(g["sq_a"], g["sq_b"])
    assert np.max(np.abs(c - g["sq_c"])) < 1e-12
    direct = ((g["sq_a"][:, None, :] - g["sq_b"][None, :, :]) ** 2).sum(-1)  # the reference's test_sq_dist
    assert np.max(np.abs(c - direct)) < 1e-10


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_acq_vs_reference(golden, name):
    g = golden("acq")
    wl, wd, state, Xr = setup_case(g, name)
    mix, gp = oracle_mix(wd), oracle_gp(wd)
    for cls, kind in KINDS.items():
        for reg in (0, 1):
            st = dict(state, variance_regularized_acq_fcn=bool(reg))
            with np.errstate(all="ignore"):
                v = acq_ref.acq_call(kind, g[f"{name}_Xs"].copy(), gp, mix, float(g[f"{name}_y_max"]), st,
                                     X_rescaled=Xr, sn2_new=g[f"{name}_sn2_new"])
            close(v, g[f"{name}_{cls}_{reg}"], 1e-11)
    one = acq_ref.acq_call(acq_ref.LOG, g[f"{name}_Xs"][20].copy(), gp, mix, float(g[f"{name}_y_max"]),
                           dict(state))
    close(one, g[f"{name}_one"], 1e-11)


def test_string_to_acq_names():
    from pyvbmc_amd.acquisition import AcqFcnLog, AcqFcnNoisy, string_to_acq

    assert isinstance(string_to_acq("AcqFcnLog()"), AcqFcnLog)
    assert isinstance(string_to_acq("AcqFcnNoisy()"), AcqFcnNoisy)
    assert string_to_acq("AcqFcnLog()").get_info()["log_flag"] is True
    assert string_to_acq("AcqFcnVIQR()").get_info()["importance_sampling"] is True
    assert string_to_acq("AcqFcnIMIQR(quantile=0.9)").get_info()["quantile"] == 0.9
    for bad in ("os.system('true')", "AcqFcnLog().__class__", "AcqFcnLog(__import__('os'))", "Nope()"):
        with pytest.raises(ValueError):
            string_to_acq(bad)


# ---------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def ctx():
    from pyvbmc_amd import _lib

    c = _lib.Context(0)
    _lib.set_default_context(c)
    yield c
    _lib.set_default_context(None)
    c.close()


def device_objects(wd, ctx, Xr, sn2_new):
    from test_gpu_parity import make_gp, make_vp

    vp, gp = make_vp(wd, ctx), make_gp(wd, ctx)
    gp.temporary_data["X_rescaled"] = Xr
    gp.temporary_data["sn2_new"] = sn2_new
    return vp, gp


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_device_acq_vs_reference(ctx, golden, name):
    from pyvbmc_amd import acquisition

    g = golden("acq")
    wl, wd, state, Xr = setup_case(g, name)
    vp, gp = device_objects(wd, ctx, Xr, g[f"{name}_sn2_new"])
    flog = SimpleNamespace(y_max=float(g[f"{name}_y_max"]))
    for cls in KINDS:
        for reg in (0, 1):
            st = dict(state, variance_regularized_acq_fcn=bool(reg))
            v = getattr(acquisition, cls)()(g[f"{name}_Xs"].copy(), gp, vp, flog, st)
            # mean / variance are good to 1e-10 (the contract); the log-valued form adds
            # log(var_tot) of variances that are tiny near training inputs
            close(v, g[f"{name}_{cls}_{reg}"], 1e-8)
    one = acquisition.AcqFcnLog()(g[f"{name}_Xs"][20].copy(), gp, vp, flog, dict(state))
    assert one.shape == (1,)
    close(one, g[f"{name}_one"], 1e-8)


@pytest.mark.gpu
def test_device_acq_search_batch_vs_oracle(ctx, golden):
    """2**13 points (the reference's cached search batch, active_sample.py) at config 3 size."""
    from pyvbmc_amd import acquisition

    wl = synthetic.make_workload(3, S=2)
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X,
              y=wl.y, hyp=wl.hyp, s2=np.zeros(0))
    rng = np.random.default_rng(3)
    M = 8192
    comp = rng.integers(0, wl.K, size=M)
    Xs = wl.mu.T[comp] + 1.5 * wl.lambd * wl.sigma[comp, None] * rng.standard_normal((M, wl.D))
    length = np.exp(wl.hyp[0, : wl.D])
    state = dict(integer_vars=None, lb_eps_orig=wl.X.min(0) - 2.0, ub_eps_orig=wl.X.max(0) + 2.0,
                 gp_length_scale=length, variance_regularized_acq_fcn=True, tol_gp_var=1e-4)
    sn2_new = 0.01 + rng.random(wl.N)
    vp, gp = device_objects(wd, ctx, wl.X / length, sn2_new)
    flog = SimpleNamespace(y_max=float(np.max(wl.y)))
    mix, ogp = oracle_mix(wd), oracle_gp(wd)
    sub = slice(0, 1024)  # the oracle's predict is O(M N^2) on the CPU
    for cls, kind in KINDS.items():
        v = getattr(acquisition, cls)()(Xs.copy(), gp, vp, flog, state)
        with np.errstate(all="ignore"):
            ref = acq_ref.acq_call(kind, Xs[sub].copy(), ogp, mix, flog.y_max, state,
                                   X_rescaled=wl.X / length, sn2_new=sn2_new)
        close(v[sub], ref, 1e-8)
        assert np.all(np.isfinite(v[~np.isinf(v)]))


@pytest.mark.gpu
def test_device_sq_dist(ctx, golden):
    from pyvbmc_amd.acquisition import nearest_neighbour, sq_dist

    g = golden("acq")
    c, idx = sq_dist(g["sq_a"], g["sq_b"], ctx=ctx, return_argmin=True)
    assert c.shape == g["sq_c"].shape
    assert np.max(np.abs(c - g["sq_c"])) < 1e-12 * max(1.0, g["sq_c"].max())
    assert np.array_equal(idx, np.argmin(g["sq_c"], axis=1))
    # ragged sizes around the 64-wide tiles, D not a multiple of 4, duplicate rows (ties)
    rng = np.random.default_rng(9)
    for n, m, D in ((1, 1, 1), (63, 65, 3), (130, 257, 10), (5, 700, 32)):
        a, b = rng.standard_normal((n, D)), rng.standard_normal((m, D))
        if m > 4:
            b[m // 2] = b[1]  # exact tie: np.argmin keeps the first
            a[0] = b[1]
        ref = acq_ref.sq_dist(a, b)
        c, idx = sq_dist(a, b, ctx=ctx, return_argmin=True)
        assert np.max(np.abs(c - ref)) < 1e-12 * max(1.0, ref.max())
        assert np.all(c >= 0.0)
        assert np.array_equal(idx, np.argmin(c, axis=1))
        assert np.array_equal(nearest_neighbour(a, b, ctx=ctx), idx)
    with pytest.raises(ValueError):
        sq_dist(np.zeros((3, 2)), np.zeros((3, 4)), ctx=ctx)
    with pytest.raises(NotImplementedError):
        sq_dist(np.zeros((3, 40)), np.zeros((3, 40)), ctx=ctx)


@pytest.mark.gpu
def test_predict_far_from_origin(ctx):
    """The centred distance expansion keeps 1e-10 when the data sit far from the origin."""
    from test_gpu_parity import make_gp

    from oracle import gp_ref

    wl = synthetic.make_workload(2, S=1)
    shift = 1.0e3
    X = wl.X + shift
    hyp = wl.hyp.copy()
    hyp[:, wl.D + 3 : 2 * wl.D + 3] += shift  # the quadratic mean's location moves along
    wd = dict(D=wl.D, K=wl.K, X=X, y=wl.y, hyp=hyp, s2=np.zeros(0))
    gp = make_gp(wd, ctx)
    ogp = gp_ref.make_gp(X, wl.y, hyp, gp_ref.MEAN_NEGQUAD)
    xs = X[:64] + 0.3 * np.random.default_rng(1).standard_normal((64, wl.D))
    fmu, fs2 = gp.predict(xs, separate_samples=True)
    omu, os2 = gp_ref.predict(ogp, xs, separate_samples=True)
    sf2 = float(np.exp(2 * hyp[0, wl.D]))
    assert np.max(np.abs(fmu - omu)) <= 1e-10 * max(1.0, np.max(np.abs(omu)))
    assert np.max(np.abs(fs2 - os2)) <= 1e-10 * max(1.0, sf2)


# ---------------------------------------------------------------- importance-sampled acquisitions (noisy targets)
GPCOV = ["homo", "hetero", "tiny"]


def gpcov_objects(c, name):
    from scipy.stats import norm

    from oracle import gp_ref

    s2 = c["s2"] if name == "hetero" else None
    hyp = c[f"{name}_hyp"]
    ogp = gp_ref.make_gp(c["X"], c["y"], hyp, gp_ref.MEAN_NEGQUAD, s2=s2, noise_user=s2 is not None)
    length = np.exp(hyp[0, : int(c["D"])])
    d = ((c["Xs"][:, None, :] / length - (c["X"] / length)[None, :, :]) ** 2).sum(-1)
    sn2 = c[f"{name}_sn2_new"][np.argmin(d, axis=1)]
    return ogp, length, sn2, norm.ppf(0.75)


@pytest.mark.parametrize("name", GPCOV)
def test_oracle_quantile_acq_vs_reference(golden, name):
    """tests/golden/gpcov.npz holds AcqFcnVIQR / AcqFcnIMIQR values produced by the reference's own
    classes (oracle/make_golden.py gpcov): the oracle restatement reproduces them."""
    c = golden("gpcov")
    ogp, length, sn2, u = gpcov_objects(c, name)
    for cls, usew in (("AcqFcnVIQR", False), ("AcqFcnIMIQR", True)):
        ais = dict(X=c[f"{name}_{cls}_Xa"], f_s2=c[f"{name}_{cls}_ais_f_s2"], ln_weights=c[f"{name}_{cls}_ais_ln_weights"])
        a = acq_ref.quantile_acq(ogp, c["Xs"], sn2, ais, u, usew)
        assert np.max(np.abs(a - c[f"{name}_{cls}_acq"])) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name", GPCOV)
def test_device_quantile_acq_vs_reference(name, golden):
    """The mirror classes' full __call__ (device predict, cross-covariance product on the matrix
    cores, log-sum-exp) against the reference's values -- homoskedastic, heteroskedastic and a GP
    with a non-Cholesky sample -- with attribute-only GP / VP objects."""
    from helpers import PlainGP, PlainVP

    from oracle import gp_ref, mixture_ref
    from pyvbmc_amd import _lib
    from pyvbmc_amd.acquisition import AcqFcnIMIQR, AcqFcnVIQR, string_to_acq

    c = golden("gpcov")
    ctx = _lib.Context(0)
    _lib.set_default_context(ctx)
    try:
        ogp, length, sn2, u = gpcov_objects(c, name)
        gp = PlainGP(ogp)
        gp.temporary_data["X_rescaled"] = c["X"] / length
        gp.temporary_data["sn2_new"] = c[f"{name}_sn2_new"]
        vp = PlainVP(mixture_ref.Mixture.make(c["vp_mu"], c["vp_sigma"], c["vp_lambd"], c["vp_w"]))
        flog = SimpleNamespace(y_max=float(np.max(c["y"])))
        for cls in (AcqFcnVIQR, AcqFcnIMIQR):
            tag = f"{name}_{cls.__name__}"
            Xa = c[f"{tag}_Xa"]
            ais = dict(X=Xa, f_s2=c[f"{tag}_ais_f_s2"], ln_weights=c[f"{tag}_ais_ln_weights"])
            # what the reference's active_importance_sampling step 3 stores: K_Xa_X (and C_tmp for VIQR)
            ais["K_Xa_X"] = np.stack([gp_ref.se_ard(p.hyp[:4], Xa, c["X"]) for p in ogp.posteriors])
            st = dict(integer_vars=None, lb_eps_orig=c["X"].min(0) - 3.0, ub_eps_orig=c["X"].max(0) + 3.0,
                      gp_length_scale=length, variance_regularized_acq_fcn=False, active_importance_sampling=ais)
            acq = cls()
            v = acq(c["Xs"].copy(), gp, vp, flog, st)
            err = float(np.max(np.abs(v - c[f"{tag}_acq"])))
            print(f"{tag}: max |acq - reference| = {err:.2e}")
            assert err < 1e-9
            assert np.array_equal(acq(c["Xs"][5].copy(), gp, vp, flog, st), v[5:6])  # 1-D input, cached state
        assert isinstance(string_to_acq("AcqFcnVIQR()"), AcqFcnVIQR)
        assert string_to_acq("AcqFcnIMIQR(quantile=0.9)").acq_info["quantile"] == 0.9
    finally:
        _lib.set_default_context(None)
        ctx.close()


@pytest.mark.gpu
def test_device_quantile_acq_vs_oracle_larger():
    """A larger case against the oracle: M = 1500 points, 420 importance points, S = 3 (one
    non-Cholesky sample), per-sample importance points (IMIQR after MCMC) and the variance
    regularisation."""
    from helpers import PlainGP, PlainVP

    from oracle import gp_ref, mixture_ref
    from pyvbmc_amd import _lib
    from pyvbmc_amd.acquisition import AcqFcnIMIQR, AcqFcnVIQR
    from scipy.stats import norm

    rng = np.random.default_rng(31)
    D, N, S, M, Na = 4, 150, 3, 1500, 420
    X = rng.standard_normal((N, D))
    y = (-0.5 * np.sum(X**2, axis=1) + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    hyp = np.array([np.concatenate([np.log(0.8 + 0.3 * rng.random(D)), [np.log(2.0)], [ls], [0.1], np.zeros(D), np.zeros(D)])
                    for ls in (np.log(0.05), np.log(3e-4), np.log(0.1))])
    ogp = gp_ref.make_gp(X, y, hyp, gp_ref.MEAN_NEGQUAD)
    assert [p.L_chol for p in ogp.posteriors] == [True, False, True]
    ctx = _lib.Context(0)
    _lib.set_default_context(ctx)
    try:
        gp = PlainGP(ogp)
        length = np.exp(hyp[0, :D])
        gp.temporary_data["X_rescaled"] = X / length
        gp.temporary_data["sn2_new"] = 0.01 + rng.random(N)
        vp = PlainVP(mixture_ref.Mixture.make(rng.standard_normal((D, 3)), [0.5, 0.7, 0.9], np.ones(D), [0.2, 0.3, 0.5]))
        Xs = 1.3 * rng.standard_normal((M, D))
        d = ((Xs[:, None, :] / length - (X / length)[None, :, :]) ** 2).sum(-1)
        sn2 = gp.temporary_data["sn2_new"][np.argmin(d, axis=1)]
        Xa = rng.standard_normal((Na, D))
        _, fs2a = gp_ref.predict(ogp, Xa, separate_samples=True)
        lnw = rng.standard_normal((S, Na))
        base = dict(integer_vars=None, lb_eps_orig=np.full(D, -50.0), ub_eps_orig=np.full(D, 50.0), gp_length_scale=length)
        flog = SimpleNamespace(y_max=0.0)
        for cls, usew in ((AcqFcnVIQR, False), (AcqFcnIMIQR, True)):
            ais = dict(X=Xa, f_s2=fs2a, ln_weights=lnw,
                       K_Xa_X=np.stack([gp_ref.se_ard(p.hyp[: D + 1], Xa, X) for p in ogp.posteriors]))
            ref = acq_ref.quantile_acq(ogp, Xs, sn2, ais, norm.ppf(0.75), usew)
            v = cls()(Xs.copy(), gp, vp, flog, dict(base, variance_regularized_acq_fcn=False, active_importance_sampling=ais))
            assert np.max(np.abs(v - ref)) < 1e-9 * max(1.0, np.max(np.abs(ref)))
            # variance regularisation on top
            _, f_s2 = gp_ref.predict(ogp, Xs, separate_samples=True)
            fmu, _ = gp_ref.predict(ogp, Xs, separate_samples=True)
            var_tot = f_s2.mean(axis=1) + fmu.var(axis=1, ddof=1)
            tol = float(np.sort(var_tot)[100])
            ref_reg = ref.copy()
            low = var_tot < tol
            ref_reg[low] += tol / var_tot[low] - 1
            v = cls()(Xs.copy(), gp, vp, flog, dict(base, variance_regularized_acq_fcn=True, tol_gp_var=tol,
                                                   active_importance_sampling=dict(ais)))
            assert np.max(np.abs(v - ref_reg) / np.maximum(1.0, np.abs(ref_reg))) < 1e-8
        # IMIQR with per-sample importance points (X of shape (S, Na, D))
        Xa3 = rng.standard_normal((S, Na, D))
        fs2a3 = np.stack([gp_ref.predict(ogp, Xa3[s], separate_samples=True)[1][:, s] for s in range(S)], axis=1)
        ais3 = dict(X=Xa3, f_s2=fs2a3, ln_weights=lnw,
                    K_Xa_X=np.stack([gp_ref.se_ard(p.hyp[: D + 1], Xa3[s], X) for s, p in enumerate(ogp.posteriors)]))
        v = AcqFcnIMIQR()(Xs[:200].copy(), gp, vp, flog, dict(base, variance_regularized_acq_fcn=False,
                                                             active_importance_sampling=ais3))
        # oracle per sample, then the reference's log-mean-exp over the samples
        per = np.stack([acq_ref.quantile_acq(gp_ref.GPData(D, X, y, None, gp_ref.MEAN_NEGQUAD, [ogp.posteriors[s]]),
                                             Xs[:200], sn2[:200], dict(X=Xa3[s], f_s2=fs2a3[:, s:s + 1], ln_weights=lnw[s:s + 1]),
                                             norm.ppf(0.75), True) for s in range(S)], axis=1)
        mx = per.max(axis=1)
        ref3 = mx + np.log(np.sum(np.exp(per - mx[:, None]), axis=1) / S)
        assert np.max(np.abs(v - ref3)) < 1e-9 * max(1.0, np.max(np.abs(ref3)))
    finally:
        _lib.set_default_context(None)
        ctx.close()


@pytest.mark.gpu
def test_small_batches_polled_completion_equals_copy_path(ctx):
    """Batches of up to 256 points (a CMA-ES population, a single point) go up by CPU stores into host-writable device
    memory, predict's finish, the density and the formula are ONE launch, and the results come back through pinned memory
    and a polled completion word (csrc/api_acq.hip).  The results are those of the five-launch copy-and-synchronise path
    (the density's K terms are a running sum here and a tree there: 1e-13) and reproduce bit for bit from call to call --
    at every batch size around the wave and workgroup boundaries, for every acquisition kind, and when such calls
    alternate with other entry points."""
    from pyvbmc_amd import acquisition

    wl = synthetic.make_workload(3, S=3, N=100)
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X, y=wl.y, hyp=wl.hyp,
              s2=np.zeros(0))
    from test_gpu_parity import make_gp, make_vp

    vp, gp = make_vp(wd, ctx), make_gp(wd, ctx)
    rng = np.random.default_rng(0)
    state = dict(integer_vars=None, lb_eps_orig=wl.X.min(0) - 20.0, ub_eps_orig=wl.X.max(0) + 20.0,
                 gp_length_scale=np.exp(wl.hyp[0, : wl.D]), variance_regularized_acq_fcn=True, tol_gp_var=1e-4)
    flog = SimpleNamespace(y_max=float(np.max(wl.y)))
    fns = [acquisition.AcqFcn(), acquisition.AcqFcnLog(), acquisition.AcqFcnVanilla()]
    for M in (1, 2, 15, 16, 63, 64, 65, 127, 129, 200, 255, 256, 257):
        comp = rng.integers(0, wl.K, size=M)
        Xs = wl.mu.T[comp] + 1.5 * wl.lambd * wl.sigma[comp, None] * rng.standard_normal((M, wl.D))
        for fn in fns:
            ctx.set_option("acq_poll", 1)
            a = fn(Xs, gp, vp, flog, state)
            if M % 3 == 0:
                vp.pdf(Xs[:2])  # another entry point in between
            a2 = fn(Xs, gp, vp, flog, state)
            ctx.set_option("acq_poll", 0)
            b = fn(Xs, gp, vp, flog, state)
            assert np.array_equal(a, a2), (M, type(fn).__name__)
            close(a, b, 1e-13)
    ctx.set_option("acq_poll", 1)
