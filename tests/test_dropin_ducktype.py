"""The drop-in boundary with the REFERENCE's own kind of objects.

The reference hands its own ``VariationalPosterior`` and a ``gpyreg.GP`` to ``_neg_elcbo``
(/root/reference/pyvbmc/vbmc/vbmc.py:1172-1180, variational_optimization.py:1080-1085), so the
mirrors may only touch the public attributes those classes have.  ``PlainVP`` / ``PlainGP``
(tests/helpers.py) are attribute-only stand-ins with ``__slots__``: any access to a private member
raises.  Every mirror entry point is driven with them against the reference goldens, and the GP is
overwritten IN PLACE between calls (active_importance_sampling.py:207-209 does that) to check the
device never keeps a stale GP.
"""
import numpy as np
import pytest
from helpers import PlainGP, PlainVP, oracle_gp, oracle_mix, rel_err

from oracle import acq_ref, elbo_ref, gp_ref
from pyvbmc_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pyvbmc_amd import _lib

    c = _lib.Context(0)
    _lib.set_default_context(c)  # reference objects carry no context: the default one is used
    yield c
    _lib.set_default_context(None)
    c.close()


def fl(f):
    return "".join("1" if b else "0" for b in f)


@pytest.mark.parametrize("name", ["c1", "c2s", "c3s"])
def test_every_entry_point_accepts_reference_style_objects(ctx, golden, name):
    from pyvbmc_amd import entlb_vbmc, entmc_vbmc
    from pyvbmc_amd.variational_optimization import _gp_log_joint, _neg_elcbo, _neg_elcbo_batch

    g = golden(name)
    K, D, NsK, seed = int(g["K"]), int(g["D"]), int(g["NsK"]), int(g["seed"])
    # entropy
    for gf in ((False,) * 4, (True,) * 4):
        np.random.seed(seed)
        H, dH = entmc_vbmc(PlainVP(g), NsK, gf, True)
        assert abs(H - g[f"entmc_H_{fl(gf)}_1"]) <= 1e-10 * abs(H)
        if gf[0]:
            assert rel_err(dH, g[f"entmc_dH_{fl(gf)}_1"]) < 1e-9
        H, dH = entlb_vbmc(PlainVP(g), gf, True)
        assert abs(H - g[f"entlb_H_{fl(gf)}_1"]) <= 1e-10 * abs(H)
    # GP expected log joint, S = 1 and S > 1
    for tag, hyp in (("S1", g["hyp"][:1]), ("SM", g["hyp"])):
        gp = PlainGP(oracle_gp(g, hyp))
        G, dG, _, _, _ = _gp_log_joint(PlainVP(g), gp, True, True, True, False, False)
        assert abs(G - g[f"glj_{tag}_G"]) <= 1e-10 * abs(G) and rel_err(dG, g[f"glj_{tag}_dG"]) < 1e-9
    # the objective, with the reference's side effects on vp and theta
    wl = synthetic.make_workload(int(g["cfg"]), S=1, D=D, K=K, N=int(g["N"]), Ns_total=int(g["Ns_total"]))
    bnd = synthetic.default_theta_bnd(wl)
    gp = PlainGP(oracle_gp(g, g["hyp"][:1]))
    for tag, th, tb in (("nobnd", g["theta"], None), ("bnd", g["theta"], bnd), ("bndout", g["theta_out"], bnd)):
        for ns_tag, Ns in (("mc", NsK), ("lb", 0)):
            vp, th_in = PlainVP(g), th.copy()
            np.random.seed(seed)
            F, dF, G, H, varF = _neg_elcbo(th_in, gp, vp, 0.0, Ns, True, False, tb, 0.0, False)
            key = f"elbo_{tag}_{ns_tag}"
            assert abs(F - g[key + "_F"]) <= 1e-9 * abs(g[key + "_F"]), key
            assert rel_err(dF, g[key + "_dF"]) < 1e-8, key
            assert np.allclose(th_in, g[key + "_theta_after"], rtol=0, atol=1e-15)
            ref = PlainVP(g)
            ref.set_parameters(th)
            assert rel_err(vp.mu, ref.mu) < 1e-15 and vp.mu.shape == (D, K)
            assert rel_err(vp.sigma, ref.sigma) < 1e-14 and vp.sigma.shape == (1, K)
            assert rel_err(vp.lambd, ref.lambd) < 1e-14 and vp.lambd.shape == (D, 1)
            assert rel_err(vp.w, ref.w) < 1e-14 and vp.w.shape == (1, K) and vp.eta.shape == (1, K)
    # variance / per-component form (composed path: goes through vp.set_parameters)
    vp = PlainVP(g)
    np.random.seed(seed)
    r = _neg_elcbo(g["theta"].copy(), gp, vp, 0.0, NsK, False, True, None, 0.0, True)
    assert len(r) == 11 and abs(r[0] - g["elbo_full_F"]) <= 1e-9 * abs(r[0])
    # the sieve batch
    thetas = g["theta"][None, :] + 0.2 * np.random.default_rng(3).standard_normal((5, g["theta"].size))
    F = _neg_elcbo_batch(thetas, gp, PlainVP(g), bnd)
    for b in range(5):
        Fo = elbo_ref.neg_elcbo(thetas[b].copy(), oracle_gp(g, g["hyp"][:1]), oracle_mix(g), 0.0, 0, False, False,
                                bnd, False)[0]
        assert abs(F[b] - Fo) <= 1e-10 * abs(Fo)


def test_device_adam_loop_and_acquisition_accept_reference_style_objects(ctx, golden):
    from types import SimpleNamespace

    from pyvbmc_amd.acquisition import AcqFcnLog
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    g = golden("c2s")
    K, D, NsK = int(g["K"]), int(g["D"]), int(g["NsK"])
    wl = synthetic.make_workload(2, S=1, D=D, K=K, N=int(g["N"]), Ns_total=int(g["Ns_total"]))
    bnd = synthetic.default_theta_bnd(wl)
    outs = []
    for cls in ("plain", "mirror"):
        gp = PlainGP(oracle_gp(g, g["hyp"][:1]))  # the same posterior records for both
        if cls == "plain":
            vp = PlainVP(g)
        else:
            from test_gpu_multibatch import make_vp

            vp = make_vp(g, ctx)
        x, y, xt, yt, it = minimize_adam_elbo(g["theta"].copy(), gp, vp, NsK, bnd, max_iter=40, seed=11, rng="philox")
        outs.append((x, y, xt, yt, vp.mu.copy()))
        assert vp.mu.shape == (D, K) and vp.sigma.shape == (1, K)
    assert np.array_equal(outs[0][3], outs[1][3]) and np.array_equal(outs[0][2], outs[1][2])
    assert np.array_equal(outs[0][4], outs[1][4])
    # acquisition: reference goldens (acq.npz) through attribute-only objects
    a = golden("acq")
    S = int(a["c2s_S"])
    wl = synthetic.make_workload(2, S=S, Ns_total=20 * 100)
    ogp = gp_ref.make_gp(wl.X, wl.y, wl.hyp, gp_ref.MEAN_NEGQUAD)
    gp = PlainGP(ogp)
    vp = PlainVP(oracle_mix(dict(mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta)))
    length = np.exp(wl.hyp[0, : wl.D])
    st = dict(integer_vars=None, lb_eps_orig=a["c2s_lo"], ub_eps_orig=a["c2s_hi"], gp_length_scale=length,
              variance_regularized_acq_fcn=True, tol_gp_var=float(a["c2s_tol_gp_var"]))
    flog = SimpleNamespace(y_max=float(a["c2s_y_max"]))
    v = AcqFcnLog()(a["c2s_Xs"].copy(), gp, vp, flog, st)
    ref = a["c2s_AcqFcnLog_1"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(v), fin) and rel_err(v[fin], ref[fin]) < 1e-8


def test_gp_overwritten_in_place_is_seen(ctx, golden):
    """``gp.posteriors[s] = ...`` and ``gp.X[...] = ...`` between two calls (same array
    objects, same ids): the device must evaluate the NEW GP."""
    from pyvbmc_amd.variational_optimization import _gp_log_joint, _neg_elcbo

    g = golden("c2s")
    hyp = g["hyp"]
    ogp_a = oracle_gp(g, hyp[:2])
    gp = PlainGP(ogp_a)
    vp = PlainVP(g)
    Ga = _gp_log_joint(vp, gp, False, True, True, False, False)[0]
    Ga_ref = gp_ref.gp_log_joint(oracle_mix(g), ogp_a, False, True, True, False, False)[0]
    assert abs(Ga - Ga_ref) <= 1e-10 * abs(Ga_ref)
    # (i) replace one posterior record in place (same posteriors array)
    ogp_b = oracle_gp(g, np.vstack([hyp[2:3], hyp[1:2]]))
    pb = ogp_b.posteriors[0]
    from types import SimpleNamespace

    gp.posteriors[0] = SimpleNamespace(hyp=pb.hyp.copy(), alpha=pb.alpha.copy(), sW=pb.sW.copy(), L=pb.L.copy(),
                                       sn2_mult=pb.sn2_mult, L_chol=pb.L_chol)
    Gb = _gp_log_joint(vp, gp, False, True, True, False, False)[0]
    Gb_ref = gp_ref.gp_log_joint(oracle_mix(g), ogp_b, False, True, True, False, False)[0]
    assert abs(Gb - Gb_ref) <= 1e-10 * abs(Gb_ref) and abs(Gb - Ga) > 1e-6 * abs(Ga)
    # (ii) edit the arrays of an existing record in place (same record object)
    pc = oracle_gp(g, np.vstack([hyp[0:1], hyp[1:2]])).posteriors[0]
    rec = gp.posteriors[0]
    rec.hyp[:], rec.alpha[:], rec.L[:], rec.sW[:] = pc.hyp, pc.alpha, pc.L, pc.sW
    Gc = _gp_log_joint(vp, gp, False, True, True, False, False)[0]
    assert abs(Gc - Ga_ref) <= 1e-10 * abs(Ga_ref)
    # (iii) overwrite the training inputs in place (same X object) and rebuild the records
    X2 = g["X"] + 0.05 * np.random.default_rng(1).standard_normal(g["X"].shape)
    g2 = dict(g, X=X2)
    ogp_d = oracle_gp(g2, hyp[:2])
    gp.X[:] = X2
    for s in range(2):
        pd_ = ogp_d.posteriors[s]
        gp.posteriors[s] = SimpleNamespace(hyp=pd_.hyp.copy(), alpha=pd_.alpha.copy(), sW=pd_.sW.copy(),
                                           L=pd_.L.copy(), sn2_mult=pd_.sn2_mult, L_chol=pd_.L_chol)
    th = g["theta"].copy()
    F = _neg_elcbo(th, gp, PlainVP(g), 0.0, 0, True, False, None)[0]
    Fo = elbo_ref.neg_elcbo(g["theta"].copy(), ogp_d, oracle_mix(g), 0.0, 0, True, False, None, False)[0]
    assert abs(F - Fo) <= 1e-10 * abs(Fo)
    # predict through the mirror GP class after an in-place posterior swap
    from test_gpu_multibatch import make_gp

    mgp = make_gp(g, ctx, hyp[:2])
    xs = np.random.default_rng(5).standard_normal((40, int(g["D"])))
    mgp.predict(xs)
    mgp.posteriors[0] = mgp._posterior(hyp[2])
    fmu, fs2 = mgp.predict(xs, separate_samples=True)
    omu, os2 = gp_ref.predict(oracle_gp(g, np.vstack([hyp[2:3], hyp[1:2]])), xs, separate_samples=True)
    assert np.max(np.abs(fmu - omu)) <= 1e-10 * max(1.0, np.max(np.abs(omu)))
    assert np.max(np.abs(fs2 - os2)) <= 1e-10 * float(np.exp(2 * hyp[0, int(g["D"])]))


def test_gp_interior_edit_in_place_is_seen(ctx, golden):
    """An in-place edit strictly INSIDE ``alpha`` (or ``hyp``) -- same objects, same ids, same first
    and last elements -- must reach the device: the GP key carries a checksum of every element of
    every posterior's alpha and hyp (round 2's key only saw the ends)."""
    from pyvbmc_amd.gp import _gp_fingerprint
    from pyvbmc_amd.variational_optimization import _gp_log_joint

    g = golden("c2s")
    ogp = oracle_gp(g, g["hyp"][:2])
    gp = PlainGP(ogp)
    vp = PlainVP(g)
    G0 = _gp_log_joint(vp, gp, False, True, True, False, False)[0]
    N = gp.X.shape[0]
    k0 = _gp_fingerprint(gp, ctx)[0]
    assert _gp_fingerprint(gp, ctx)[0] == k0  # stable while nothing changes
    gp.posteriors[1].alpha[N // 2, 0] *= 1.5  # interior element of the second record
    assert _gp_fingerprint(gp, ctx)[0] != k0
    G1 = _gp_log_joint(vp, gp, False, True, True, False, False)[0]
    ogp.posteriors[1].alpha[N // 2, 0] *= 1.5
    G1_ref = gp_ref.gp_log_joint(oracle_mix(g), ogp, False, True, True, False, False)[0]
    assert abs(G1 - G1_ref) <= 1e-10 * abs(G1_ref) and G1 != G0
    D = int(g["D"])
    gp.posteriors[0].hyp[D + 2] += 0.3  # interior hyper-parameter (the mean's m0): not an end element
    ogp.posteriors[0].hyp[D + 2] += 0.3
    G2 = _gp_log_joint(vp, gp, False, True, True, False, False)[0]
    G2_ref = gp_ref.gp_log_joint(oracle_mix(g), ogp, False, True, True, False, False)[0]
    assert abs(G2 - G2_ref) <= 1e-10 * abs(G2_ref) and G2 != G1
    # non-contiguous / non-float64 arrays take the slow hash and are seen too
    rec = gp.posteriors[0]
    big = np.zeros((N, 2))
    big[:, 0] = rec.alpha[:, 0]
    rec.alpha = big[:, :1]  # strided view
    assert not rec.alpha.flags["C_CONTIGUOUS"]
    _gp_log_joint(vp, gp, False, True, True, False, False)
    k3 = _gp_fingerprint(gp, ctx)[0]
    rec.alpha[N // 3, 0] += 1.0
    assert _gp_fingerprint(gp, ctx)[0] != k3
    # cost of the key (host): stated in DESIGN 4.3b; generous bound here, the figure is printed
    import time

    t0 = time.perf_counter()
    for _ in range(2000):
        _gp_fingerprint(PLAIN8, ctx)
    dt = (time.perf_counter() - t0) / 2000
    print(f"_gp_fingerprint at S=8, N=800: {1e6 * dt:.2f} us")
    assert dt < 50e-6


def test_gp_edited_in_place_between_fused_evaluations(ctx, golden):
    """The optimiser's inner call does not checksum the GP arrays up front: the library does it while
    the device works and answers W_GP_CHANGED, and the mirror uploads and evaluates again.  An
    interior in-place edit between two Monte-Carlo evaluations (consecutive seeds: the second one is
    armed, i.e. already queued with the OLD GP on the device) must give the NEW GP's value."""
    from pyvbmc_amd import synthetic as syn
    from pyvbmc_amd.variational_optimization import _neg_elcbo

    g = golden("c2s")
    ogp = oracle_gp(g, g["hyp"][:1])
    gp = PlainGP(ogp)
    K, D = int(g["K"]), int(g["D"])
    wl = syn.make_workload(int(g["cfg"]), S=1, D=D, K=K, N=int(g["N"]), Ns_total=int(g["Ns_total"]))
    bnd = syn.default_theta_bnd(wl)
    NsK = 2 * 64 * 4
    from oracle import philox_ref

    def oracle(seed):
        eps = philox_ref.eps_half(K, NsK // 2, D, seed)
        return elbo_ref.neg_elcbo(g["theta"].copy(), ogp, oracle_mix(g), 0.0, NsK, True, False, bnd, False, eps_half=eps)

    vp = PlainVP(g)
    for i in range(3):
        F, dF, G, H, _ = _neg_elcbo(g["theta"].copy(), gp, vp, 0.0, NsK, True, False, bnd, rng="philox", seed=300 + i)
    Fo = oracle(302)
    assert abs(F - Fo[0]) <= 1e-10 * abs(Fo[0]) and abs(G - Fo[2]) <= 1e-10 * abs(Fo[2])
    N = gp.X.shape[0]
    for rec in (gp.posteriors[0], ogp.posteriors[0]):
        rec.alpha[N // 2, 0] *= 1.25  # interior, in place: ids and end elements unchanged
    F2, dF2, G2, H2, _ = _neg_elcbo(g["theta"].copy(), gp, vp, 0.0, NsK, True, False, bnd, rng="philox", seed=303)
    Fo2 = oracle(303)
    assert abs(G2 - Fo2[2]) <= 1e-10 * abs(Fo2[2]) and abs(G2 - G) > 1e-8 * abs(G)
    assert abs(F2 - Fo2[0]) <= 1e-10 * abs(Fo2[0]) and rel_err(dF2, Fo2[1]) < 1e-9
    # and the evaluations after it run on the new GP without another upload
    F3, _, G3, _, _ = _neg_elcbo(g["theta"].copy(), gp, vp, 0.0, NsK, True, False, bnd, rng="philox", seed=304)
    Fo3 = oracle(304)
    assert abs(F3 - Fo3[0]) <= 1e-10 * abs(Fo3[0]) and abs(G3 - G2) <= 1e-12 * abs(G2)


def _plain8():
    from types import SimpleNamespace

    rng = np.random.default_rng(0)
    gp = SimpleNamespace(X=rng.standard_normal((800, 20)), posteriors=np.empty(8, dtype=object))
    for s in range(8):
        gp.posteriors[s] = SimpleNamespace(hyp=rng.standard_normal(63), alpha=rng.standard_normal((800, 1)),
                                           L=rng.standard_normal((8, 8)), sW=np.ones(800), L_chol=True)
    return gp


PLAIN8 = _plain8()


def test_theta_bnd_edited_in_place_is_seen(ctx, golden):
    """Reassigning / editing ``theta_bnd`` entries between calls (advisor finding): the cached
    argument block must not keep the old arrays or scalars."""
    from pyvbmc_amd.variational_optimization import _neg_elcbo

    g = golden("c1")
    D, K = int(g["D"]), int(g["K"])
    wl = synthetic.make_workload(1, S=1)
    bnd = synthetic.default_theta_bnd(wl)
    gp = PlainGP(oracle_gp(g, g["hyp"][:1]))
    ogp = oracle_gp(g, g["hyp"][:1])

    def both(b):
        F = _neg_elcbo(g["theta_out"].copy(), gp, PlainVP(g), 0.0, 0, True, False, b)[0]
        Fo = elbo_ref.neg_elcbo(g["theta_out"].copy(), ogp, oracle_mix(g), 0.0, 0, True, False, b, False)[0]
        assert abs(F - Fo) <= 1e-10 * abs(Fo)
        return F

    F0 = both(bnd)
    bnd["ub"] = bnd["ub"] - 0.5  # reassigned entry
    F1 = both(bnd)
    bnd["lb"][0] += 0.25  # edited in place
    bnd["tol_con"] = 0.02
    bnd["weight_penalty"] = 0.3
    F2 = both(bnd)
    assert len({F0, F1, F2}) == 3
    assert both(None) != F2


@pytest.mark.gpu
def test_vp_attributes_edited_in_place_rebound_or_odd_typed_are_seen(ctx, golden):
    """upload_vp keeps the ctypes pointers of the attribute ARRAYS of the last vp (pyvbmc_amd/_duck.py) and lets the library
    compare their contents with what the device holds: an interior in-place edit, a rebound attribute, another vp object,
    a float32 / non-contiguous attribute and a vp of another shape must each reach the device."""
    g = golden("c2s")
    D = int(g["D"])
    xs = np.random.default_rng(2).standard_normal((23, D)) * 0.5 + g["mu"].mean(axis=1)
    from oracle import mixture_ref

    def check(vp, mix):
        y = vp.pdf(xs, orig_flag=False)
        yo = mixture_ref.pdf(mix, xs)
        assert rel_err(np.ravel(y), np.ravel(yo)) < 1e-10

    from test_gpu_parity import make_vp

    vp = make_vp(g, ctx)
    mix = oracle_mix(g)
    check(vp, mix)
    check(vp, mix)  # same arrays, same contents: the cached pointers
    vp.mu[1, 2] += 0.37  # interior, in place
    mix.mu[1, 2] += 0.37
    check(vp, mix)
    vp.sigma = vp.sigma * 1.3  # rebound
    mix.sigma = mix.sigma * 1.3
    check(vp, mix)
    vp.w = (vp.w[::-1] if vp.w.ndim == 1 else vp.w[:, ::-1])  # a non-contiguous view
    mix.w = np.ascontiguousarray(mix.w[:, ::-1] if mix.w.ndim == 2 else mix.w[::-1])
    check(vp, mix)
    vp.lambd = vp.lambd.astype(np.float32).astype(np.float64) * 1.0 + 0.0
    mix.lambd = vp.lambd.copy()
    check(vp, mix)
    vp2 = make_vp(g, ctx)  # another object holding other arrays with the golden values
    check(vp2, oracle_mix(g))
    check(vp, mix)
    g1 = golden("c1")  # another (D, K) through the same context
    xs1 = np.random.default_rng(3).standard_normal((9, int(g1["D"])))
    vp1 = make_vp(g1, ctx)
    assert rel_err(np.ravel(vp1.pdf(xs1, orig_flag=False)), np.ravel(mixture_ref.pdf(oracle_mix(g1), xs1))) < 1e-10
    check(vp, mix)
