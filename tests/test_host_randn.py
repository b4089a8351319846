"""The library's host-side restatement of NumPy's legacy normal stream (csrc/host_randn.hip) against
np.random.randn itself: same values bit for bit and the same generator state left behind, on every
path -- a cached second value going in or coming out, odd and even counts, requests that end inside
the current MT19937 block, at its boundary or many blocks later, one thread and several.  The
reference draws the entropy's eps from that stream (entropy/entmc_vbmc.py:67); rng="numpy" (the
drop-in default) ships exactly these values to the device."""
import numpy as np
import pytest

from pyvbmc_amd import entropy as ent


def _same_as_numpy(n, threads, seed, pre):
    np.random.seed(seed)
    if pre:
        np.random.randn(pre)  # odd `pre` leaves a cached value behind
    s0 = np.random.get_state()
    want = np.random.randn(n)
    want_after = np.random.randn(7)
    np.random.set_state(s0)
    got = ent.host_randn(n, threads)
    got_after = np.random.randn(7)
    assert got.shape == (n,)
    assert np.array_equal(want, got)
    assert np.array_equal(want_after, got_after)


@pytest.mark.parametrize("threads", [1, 3, 0])
@pytest.mark.parametrize("pre", [0, 3, 10])
def test_host_randn_matches_numpy_small(threads, pre):
    # one MT19937 block serves 156 attempts (624 words / 4): counts around whole blocks
    for n in (0, 1, 2, 3, 7, 100, 243, 244, 245, 246, 311, 312, 313, 1000, 4097):
        _same_as_numpy(n, threads, seed=n + 17, pre=pre)


@pytest.mark.parametrize("threads", [1, 5, 0])
def test_host_randn_matches_numpy_large(threads):
    # above the multi-thread threshold (16 384 accepted attempts), odd and even, cached value in
    for n, pre in ((99_999, 1), (100_000, 0), (1_234_567, 3)):
        _same_as_numpy(n, threads, seed=5, pre=pre)


def test_draw_eps_half_is_the_reference_draw_order():
    K, D, Ns = 7, 5, 4000  # K * Ns/2 * D = 70 000 values: the threaded path
    np.random.seed(11)
    want = np.stack([np.random.randn(Ns // 2, D) for _ in range(K)])
    after = np.random.rand(3)
    np.random.seed(11)
    got = ent.draw_eps_half(K, D, Ns)
    assert np.array_equal(want, got)
    assert np.array_equal(after, np.random.rand(3))
    # below the threshold the plain NumPy loop runs: same contract
    np.random.seed(12)
    want = np.stack([np.random.randn(10, 3) for _ in range(2)])
    np.random.seed(12)
    assert np.array_equal(want, ent.draw_eps_half(2, 3, 20))


def test_host_randn_declines_other_generators(monkeypatch):
    # a global state that is not MT19937 cannot be restated: the caller falls back to NumPy's own draw
    monkeypatch.setattr(np.random, "get_state", lambda legacy=True: ("PCG64", None, 0, 0, 0.0))
    assert ent.host_randn(10) is None
    monkeypatch.setattr(np.random, "get_state", lambda legacy=True: {"bit_generator": "PCG64", "state": {}})
    assert ent.host_randn(10) is None


def test_vp_sample_numpy_path_keeps_the_reference_stream():
    """``vp.sample`` on the NumPy stream (variational_posterior.py:296-327) draws its N x D normals
    through the restated generator once the request is large: same samples, same state after."""
    from pyvbmc_amd import VariationalPosterior

    np.random.seed(3)
    vp = VariationalPosterior(4, 3)
    vp.mu = np.random.randn(4, 3)
    vp.sigma = np.exp(0.3 * np.random.randn(1, 3))
    vp.lambd = np.exp(0.2 * np.random.randn(4, 1))
    vp.w = np.array([[0.2, 0.5, 0.3]])
    N = 30_000  # 120 000 normals: above the threshold
    np.random.seed(8)
    i_want = np.random.choice(range(3), size=N, p=vp.w.ravel())
    x_want = vp.mu.T[i_want] + vp.lambd.reshape(1, -1) * np.random.randn(N, 4) * vp.sigma[:, i_want].T
    after = np.random.rand(2)
    np.random.seed(8)
    x, i = vp.sample(N, orig_flag=False, rng="numpy")
    assert np.array_equal(i, i_want)
    assert np.array_equal(x, x_want)
    assert np.array_equal(after, np.random.rand(2))


@pytest.fixture(scope="module")
def ctx():
    from pyvbmc_amd import _lib

    c = _lib.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("device_stream", [0, 1], ids=["host-cores", "device"])
@pytest.mark.parametrize("K,h,D", [(3, 40, 4), (7, 3000, 5)])
def test_set_eps_numpy_uploads_the_reference_draws(ctx, K, h, D, device_stream):
    """vbmc_set_eps_numpy = np.random.randn + vbmc_set_eps: the device ends up with the same resident
    draws (checked through the raw entropy accumulator of a row shard and of the whole job) and
    NumPy's state where randn would leave it -- the whole job's values are consumed on every rank."""
    import ctypes as C

    from pyvbmc_amd import _lib

    rng = np.random.default_rng(K)
    mu, sigma = rng.standard_normal((D, K)), np.exp(0.3 * rng.standard_normal(K))
    lambd, w = np.exp(0.2 * rng.standard_normal(D)), rng.dirichlet(np.ones(K))
    ctx.set_mixture(mu, sigma, lambd, w, np.log(w))
    n = 1 + D * K + 2 * K + D

    def raw(r0, rows):
        H, out = C.c_double(), np.empty(n)
        ctx.check(ctx._lib.vbmc_entmc(ctx._h, 2 * h, _lib.EPS_RESIDENT, 0, r0, rows, 15, 1, C.byref(H), None, _lib.ptr(out)))
        return out

    # device_stream = 1: jobs of >= 65 536 values whose rows this context holds in full are generated ON the device
    # (csrc/device_randn.hip: same words, same accepted attempts, same state; ~0.1 % of the values one to three units in the
    # last place away, tests/test_device_randn.py) -- the accumulator then agrees to rounding instead of bit for bit
    ctx.set_option("randn_device", device_stream)
    on_device = device_stream and K * h * D >= 65536
    for r0, rows in ((0, h), (h // 3, h - h // 3)):
        np.random.seed(21)
        np.random.randn(1)  # a cached value going in
        want_eps = np.stack([np.random.randn(h, D) for _ in range(K)])
        want_after = np.random.rand(3)
        ctx.set_eps(want_eps, r0, rows)
        want = raw(r0, rows)
        np.random.seed(21)
        np.random.randn(1)
        assert ctx.set_eps_numpy(K, h, D, r0, rows)
        assert np.array_equal(want_after, np.random.rand(3))
        got = raw(r0, rows)
        if on_device and r0 == 0:
            assert np.max(np.abs(got - want)) <= 1e-13 * np.max(np.abs(want))
        else:
            assert np.array_equal(got, want)
    ctx.set_option("randn_device", 1)
