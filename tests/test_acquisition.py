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
    with pytest.raises(NotImplementedError):
        string_to_acq("AcqFcnVIQR()")
    with pytest.raises(ValueError):
        string_to_acq("os.system('true')")


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
