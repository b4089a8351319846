"""The repeat path of ``_neg_elcbo`` (pyvbmc_amd/variational_optimization.py ``_FastElbo``).

From its second call with the same objects on, the optimiser's inner call
(/root/reference/pyvbmc/vbmc/variational_optimization.py:238-249) goes straight to the C entry.  Everything that
can change behind an unchanged object identity must still be seen: every scenario below is run once with the repeat
path and once with ``_FAST_PATH = False`` (every call through the general path) and the two must agree bit for bit,
values and side effects; the general path itself is pinned to the reference goldens elsewhere (test_gpu_parity.py).
"""
import numpy as np
import pytest
from helpers import PlainGP, PlainVP, oracle_gp

from pyvbmc_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pyvbmc_amd import _lib

    c = _lib.Context(0)
    _lib.set_default_context(c)
    yield c
    _lib.set_default_context(None)
    c.close()


def scenario(g, fast):
    """A sequence of fused evaluations with something changed between every two of them."""
    from pyvbmc_amd import variational_optimization as vo

    vo._FAST_PATH = fast
    vo._fast_last[0] = None
    hits = [0]
    orig = vo._FastElbo.call

    def counting(self, theta, seed):
        r = orig(self, theta, seed)
        hits[0] += r is not None
        return r

    vo._FastElbo.call = counting
    try:
        wl = synthetic.make_workload(1, S=1)
        bnd = synthetic.default_theta_bnd(wl)
        gp = PlainGP(oracle_gp(g, g["hyp"][:1]))
        vp = PlainVP(g)
        th0 = g["theta_out"].copy()
        out = []

        def ev(theta, Ns=40, b=bnd, v=None, gpo=None, seed=None, cg=True):
            v = vp if v is None else v
            th = theta.copy()
            r = vo._neg_elcbo(th, gp if gpo is None else gpo, v, 0.0, Ns, cg, False, b, rng="philox", seed=seed)
            out.append((r[0], None if r[1] is None else r[1].copy(), r[2], r[3], th.copy(), v.mu.copy(), v.sigma.copy(),
                        v.lambd.copy(), v.w.copy(), v.eta.copy()))

        for i in range(4):  # plain repeats: calls 2.. take the repeat path
            ev(th0 + 1e-3 * i, seed=11 + i)
        bnd["lb"][0] += 0.25  # bounds edited in place: same array object, the pointer sees it
        ev(th0, seed=3)
        bnd["ub"] = bnd["ub"] - 0.5  # entry rebound
        ev(th0, seed=3)
        bnd["tol_con"] = 0.02  # scalars changed
        bnd["weight_penalty"] = 0.3
        ev(th0, seed=3)
        ev(th0, seed=3, b=None)  # no bounds
        ev(th0, seed=3, b=None)
        ev(th0, seed=3)  # and back
        vp.optimize_lambd = False  # an optimise flag flipped: a shorter theta and another mask
        K, D = vp.K, vp.D
        th_short = np.concatenate([th0[: D * K + K], th0[D * K + K + D:]])
        ev(th_short, seed=4)
        ev(th_short, seed=4)
        vp.optimize_lambd = True
        ev(th0, seed=4)
        ev(th0, Ns=80, seed=4)  # another sample count
        ev(th0, Ns=80, seed=5)
        ev(th0, Ns=0)  # the lower-bound entropy
        ev(th0, Ns=0)
        ev(th0, seed=6, cg=False)  # value only
        ev(th0, seed=6, cg=False)
        vp2 = PlainVP(g)  # another vp object
        ev(th0, seed=7, v=vp2)
        ev(th0, seed=7, v=vp)
        gp.posteriors[0].alpha[3, 0] *= 1.0 + 1e-3  # GP edited in place (interior element): the library's checksum
        ev(th0, seed=8)
        gp.posteriors[0].alpha = gp.posteriors[0].alpha * (1.0 - 1e-3)  # GP record's array rebound
        ev(th0, seed=8)
        gp2 = PlainGP(oracle_gp(g, g["hyp"][:1]))  # another GP object
        ev(th0, seed=8, gpo=gp2)
        ev(th0, seed=8, gpo=gp2)
        ev(th0)  # seeds from the context's own sequence
        ev(th0)
        return out, hits[0]
    finally:
        vo._FastElbo.call = orig
        vo._FAST_PATH = True
        vo._fast_last[0] = None


def test_repeat_path_is_the_general_path_bit_for_bit(ctx, golden):
    from pyvbmc_amd.entropy import philox_seed

    g = golden("c1")
    # the context's seed sequence advances with every unseeded call: start both runs from the same point
    np.random.seed(5)
    ctx.__dict__.pop("_philox_seq", None)
    slow, hits_slow = scenario(g, False)
    np.random.seed(5)
    ctx.__dict__.pop("_philox_seq", None)
    fast, hits_fast = scenario(g, True)
    assert hits_slow == 0
    assert hits_fast >= 10, hits_fast  # the repeats really took the repeat path
    assert len(slow) == len(fast)
    for i, (a, b) in enumerate(zip(slow, fast)):
        for x, y in zip(a, b):
            if x is None:
                assert y is None
            else:
                assert np.array_equal(np.asarray(x), np.asarray(y)), f"evaluation {i} differs"
    # and the sequence is not constant (the changes were seen at all)
    assert len({r[0] for r in fast}) >= 10
    philox_seed(ctx)


def test_side_effects_with_and_without_the_release_callback(ctx, golden):
    """``vp.set_parameters(theta)``'s side effects are applied by a callback the library makes once its launches are out
    (vbmc_set_release_callback) or, with VBMC_RELEASE_CB=0, after the C call: same values either way, also through the
    W_GP_CHANGED re-evaluation; and a vp that refuses an assignment raises in both (ctypes swallows exceptions raised
    inside a callback: the mirror then applies -- and raises -- after the call)."""
    from pyvbmc_amd import variational_optimization as vo

    g = golden("c1")
    wl = synthetic.make_workload(1, S=1)
    bnd = synthetic.default_theta_bnd(wl)
    th0 = g["theta_out"].copy()

    def run(cb):
        vo._RELEASE_CB = cb
        vo._fast_last[0] = None
        for k in ("_fused_last", "_fused_cache"):
            ctx.__dict__.pop(k, None)
        gp = PlainGP(oracle_gp(g, g["hyp"][:1]))
        vp = PlainVP(g)
        out = []
        for i in range(3):
            th = th0 + 1e-2 * i
            r = vo._neg_elcbo(th, gp, vp, 0.0, 40, True, False, bnd, rng="philox", seed=5 + i)
            out.append((r[0], r[1].copy(), th.copy(), vp.mu.copy(), vp.sigma.copy(), vp.lambd.copy(), vp.w.copy(), vp.eta.copy()))
        gp.posteriors[0].alpha[2, 0] *= 1.0 + 1e-3  # in place: the library answers W_GP_CHANGED, the mirror re-evaluates
        th = th0.copy()
        r = vo._neg_elcbo(th, gp, vp, 0.0, 40, True, False, bnd, rng="philox", seed=9)
        out.append((r[0], r[1].copy(), th.copy(), vp.mu.copy(), vp.sigma.copy(), vp.lambd.copy(), vp.w.copy(), vp.eta.copy()))

        class Refusing(PlainVP):
            __slots__ = ()

            def __setattr__(self, name, value):
                if name == "sigma" and getattr(self, "armed", False):
                    raise RuntimeError("no")
                object.__setattr__(self, name, value)

            armed = False

        bad = Refusing(g)
        Refusing.armed = True
        try:
            with pytest.raises(RuntimeError):
                vo._neg_elcbo(th0.copy(), gp, bad, 0.0, 40, True, False, bnd, rng="philox", seed=10)
        finally:
            Refusing.armed = False
        return out

    try:
        on, off = run(True), run(False)
    finally:
        vo._RELEASE_CB = True
        vo._fast_last[0] = None
        for k in ("_fused_last", "_fused_cache"):
            ctx.__dict__.pop(k, None)
    for i, (a, b) in enumerate(zip(on, off)):
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x), np.asarray(y)), i


def test_a_raise_on_the_general_path_leaves_no_repeat_record(ctx, golden):
    """the general path re-binds the shared argument block's soft bounds (``_fused_call``) before it can
    still raise (``rows=`` with the wrong draw source here).  The repeat record of the call BEFORE must not survive
    that: it would pass every identity check of the next call with the original arguments and evaluate with the
    other call's bounds (with ``theta_bnd=None`` in between: no bound penalty at all).  Also: ``Context.close``,
    ``unpatch`` and ``invalidate_gp`` drop the record (it holds vp, gp and the context)."""
    from pyvbmc_amd import variational_optimization as vo
    from pyvbmc_amd.gp import invalidate_gp

    g = golden("c1")
    wl = synthetic.make_workload(1, S=1)
    bnd = synthetic.default_theta_bnd(wl)
    bnd["lb"] = bnd["lb"] + 0.5  # (tight enough that the penalty is not zero)
    gp = PlainGP(oracle_gp(g, g["hyp"][:1]))
    vp = PlainVP(g)
    th0 = g["theta_out"].copy()
    vo._fast_last[0] = None

    def ev(b):
        return vo._neg_elcbo(th0.copy(), gp, vp, 0.0, 40, True, False, b, rng="philox", seed=21)

    want = ev(bnd)
    assert vo._fast_last[0] is not None
    free = ev(None)
    assert want[0] != free[0]  # the bounds matter at this theta
    again = ev(bnd)
    assert again[0] == want[0]
    rec = vo._fast_last[0]
    assert rec is not None and rec.bnd is bnd
    # a call with other bounds that raises AFTER the argument block was re-bound
    with pytest.raises(ValueError):
        vo._neg_elcbo(th0.copy(), gp, vp, 0.0, 40, True, False, None, rng="numpy", rows=(0, 10))
    assert vo._fast_last[0] is None
    got = ev(bnd)  # (with the stale record: the fast path, n_bnd = 0, no penalty)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])
    # the record does not outlive what it points at
    assert vo._fast_last[0] is not None
    invalidate_gp(ctx)
    assert vo._fast_last[0] is None
    ev(bnd)
    vo.clear_fast_path()
    assert vo._fast_last[0] is None
