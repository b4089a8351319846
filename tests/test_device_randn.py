"""The reference's draw stream generated on the device (csrc/device_randn.hip, csrc/mt_jump.h).

``rng="numpy"`` (the drop-in's default) consumes NumPy's legacy stream exactly as the reference does
(/root/reference/pyvbmc/entropy/entmc_vbmc.py:64-68).  On one GPU the library now generates that stream where it is used:
MT19937 with a GF(2) jump-ahead per workgroup.  CPU part: the jump arithmetic (characteristic polynomial by
Berlekamp-Massey, polynomial powers, the sliding-window jump) against NumPy's own generator.  GPU part: the cases of
tests/test_host_randn.py -- cached value in / out, odd and even counts, positions inside and at the end of a block --
with words, accept / reject decisions and the generator state bit for bit, and the values bit for bit in > 99 % of the
draws, within three units in the last place in the rest (neither glibc's ``log`` nor the device's is correctly rounded
everywhere).
"""
import ctypes as C

import numpy as np
import pytest

from pyvbmc_amd import _lib

U32P = C.POINTER(C.c_uint32)


def numpy_key_after(key0, n_words):
    """key and pos NumPy holds after drawing n_words 32-bit words from (key0, pos=624)."""
    rs = np.random.RandomState()
    rs.set_state(("MT19937", key0, 624, 0, 0.0))
    if n_words:
        rs.randint(0, 2**32, size=n_words, dtype=np.uint32)
    st = rs.get_state()
    return np.array(st[1], dtype=np.uint32), int(st[2])


@pytest.mark.parametrize("blocks", [1, 2, 33, 80, 160, 80 * 7])
def test_jump_equals_the_recurrence(blocks):
    lib = _lib.load()
    key0 = np.random.RandomState(123 + blocks).get_state()[1].astype(np.uint32)
    # NumPy regenerates on the first draw: after 624 * b words it holds block b (key0 is block 0) with pos = 624
    want, pos = numpy_key_after(key0, 624 * blocks)
    assert pos == 624
    got = np.zeros(624, dtype=np.uint32)
    assert lib.vbmc_mt_jump_host(key0.ctypes.data_as(U32P), 624 * blocks, got.ctypes.data_as(U32P)) == 0
    assert np.array_equal(got, want)


def test_polynomial_chain_is_consistent_with_single_jumps():
    """G_m = t^(m J - 1): applying G_3 of the chain (what the device uses) equals the single jump by 3 J words."""
    lib = _lib.load()
    J, count = 80 * 624, 5
    polys = np.zeros((count, 624), dtype=np.uint32)
    assert lib.vbmc_mt_jump_polys(J, count, polys.ctypes.data_as(U32P)) == 0
    key0 = np.random.RandomState(7).get_state()[1].astype(np.uint32)
    # the window x_0 .. by NumPy itself
    rs = np.random.RandomState()
    rs.set_state(("MT19937", key0, 624, 0, 0.0))
    words = [key0]
    for _ in range(33):
        rs.randint(0, 2**32, size=624, dtype=np.uint32)
        words.append(np.array(rs.get_state()[1], dtype=np.uint32))
    win = np.concatenate(words)
    for m in (1, 3, 5):
        bits = np.unpackbits(polys[m - 1].view(np.uint8), bitorder="little")[:19937]
        idx = np.nonzero(bits)[0]
        assert idx.size > 1000  # (t^(J-1) is still sparse -- phi has few terms --, later ones fill up towards half the coefficients)
        acc = np.zeros(624, dtype=np.uint32)
        for i in idx:
            acc ^= win[1 + i: 1 + i + 624]
        want, _ = numpy_key_after(key0, m * J)
        assert np.array_equal(acc, want), m


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def device_randn(ctx, n):
    st = np.random.get_state(legacy=True)
    key = np.array(st[1], dtype=np.uint32)
    pos, hg, g = C.c_int(int(st[2])), C.c_int(int(st[3])), C.c_double(float(st[4]))
    out = np.empty(n)
    ctx.check(ctx._lib.vbmc_mt19937_randn_dev(ctx._h, key.ctypes.data_as(U32P), C.byref(pos), C.byref(hg), C.byref(g),
                                              _lib.ptr(out), n))
    np.random.set_state(("MT19937", key, pos.value, hg.value, g.value))
    return out


def ulp_diff(a, b):
    ia, ib = a.view(np.int64), b.view(np.int64)
    return np.abs(ia - ib)


def same_as_numpy(ctx, n, seed, pre):
    np.random.seed(seed)
    if pre:
        np.random.randn(pre)  # odd `pre` leaves a cached value behind
    s0 = np.random.get_state()
    want = np.random.randn(n)
    s_want = np.random.get_state()
    after = np.random.randn(5)
    np.random.set_state(s0)
    got = device_randn(ctx, n)
    s_got = np.random.get_state()
    # the state NumPy is left in: key, position, cached value -- bit for bit
    assert np.array_equal(s_got[1], s_want[1]) and s_got[2:] == s_want[2:], (n, seed, pre)
    assert np.array_equal(np.random.randn(5), after)
    # the values: the same attempts were accepted (any other choice would shift everything); a logarithm that rounds the
    # other way (neither glibc's nor the device's is correctly rounded everywhere) moves f by a unit in the last place
    # and the product f x by up to three (measured: 99.9 % bit-identical)
    d = ulp_diff(got, want)
    assert d.max() <= 4, (n, seed, pre, int(d.max()))
    return float(np.mean(d == 0))


@pytest.mark.gpu
@pytest.mark.parametrize("pre", [0, 3, 10])
def test_device_randn_matches_numpy_small(ctx, pre):
    for n in (1, 2, 3, 7, 243, 244, 245, 311, 312, 313, 1000, 4097):
        same_as_numpy(ctx, n, seed=n + 17, pre=pre)


@pytest.mark.gpu
def test_device_randn_matches_numpy_large(ctx):
    exact = []
    # (6 000 001 values need 308 streams: more than one workgroup per CU, polynomials beyond the first 255)
    for n, pre in ((99_999, 1), (100_000, 0), (1_234_567, 3), (5_000_000, 0), (5_000_001, 155), (6_000_001, 2)):
        exact.append(same_as_numpy(ctx, n, seed=5, pre=pre))
    assert min(exact) > 0.995, exact  # (bit-identical but for the draws whose logarithm rounds the other way: ~0.1 %)


@pytest.mark.gpu
def test_consecutive_requests_take_their_window_from_the_pass_before(ctx):
    """Round 6: a request that continues the stream where the last one left it finds the 33 blocks behind its incoming
    state in that pass's word sequence and skips the one sequential kernel (vbmc_randn_dev_info says so).  Values and
    states as np.random.randn's through a chain of requests of different sizes, odd ones included (cached second value),
    with the window computed afresh after a foreign draw and after a re-seed -- and the same chain with the reuse switched
    off (option randn_device = 3) gives the same values bit for bit."""
    def reused():
        v = C.c_int(-1)
        ctx.check(ctx._lib.vbmc_randn_dev_info(ctx._h, C.byref(v)))
        return v.value

    sizes = [300_000, 300_000, 70_001, 70_001, 1_000_000, 65_536]
    chains = {}
    for mode in (1, 3):
        ctx.set_option("randn_device", mode)
        np.random.seed(2024)
        want_state = []
        got, flags = [], []
        for i, n in enumerate(sizes):
            if i == 4:
                np.random.randint(0, 2**32, size=5, dtype=np.uint32)  # somebody else draws: another position, same or next block
            got.append(device_randn(ctx, n))
            flags.append(reused())
            want_state.append(np.random.get_state())
        chains[mode] = (got, flags, want_state)
    ctx.set_option("randn_device", 1)
    np.random.seed(2024)
    for i, n in enumerate(sizes):
        if i == 4:
            np.random.randint(0, 2**32, size=5, dtype=np.uint32)
        want = np.random.randn(n)
        s_want = np.random.get_state()
        for mode in (1, 3):
            s_got = chains[mode][2][i]
            assert np.array_equal(s_got[1], s_want[1]) and s_got[2:] == s_want[2:], (mode, i)
            assert ulp_diff(chains[mode][0][i], want).max() <= 4, (mode, i)
        assert np.array_equal(chains[1][0][i], chains[3][0][i]), i  # with and without the reuse: the same bits
    assert chains[3][1] == [0] * len(sizes)
    # request 0 follows a re-seed (nothing to reuse); 1 .. 3 continue; 4 follows five foreign words -- still inside the block
    # the last pass ended in or the next one, both in its sequence: the key decides, not the position; 5 continues
    assert chains[1][1][0] == 0 and chains[1][1][1:4] == [1, 1, 1] and chains[1][1][5] == 1, chains[1][1]


@pytest.mark.gpu
def test_device_randn_at_block_boundaries(ctx):
    """Positions 620 .. 624 of the current block going in (attempts that straddle two blocks, a block that is used up)."""
    for pre_words in (620, 621, 622, 623, 624, 625, 1247, 1248):
        np.random.seed(77)
        np.random.randint(0, 2**32, size=pre_words, dtype=np.uint32)
        s0 = np.random.get_state()
        want = np.random.randn(70_001)
        s_want = np.random.get_state()
        np.random.set_state(s0)
        got = device_randn(ctx, 70_001)
        s_got = np.random.get_state()
        assert np.array_equal(s_got[1], s_want[1]) and s_got[2:] == s_want[2:], pre_words
        assert ulp_diff(got, want).max() <= 4


@pytest.mark.gpu
def test_default_draw_source_uses_the_device_generator(ctx):
    """``_neg_elcbo(rng="numpy")`` through vbmc_set_eps_numpy: the device stream and the host-core stream give the same
    objective (1e-13: a last-place difference in a tenth of 5e5 draws) and leave NumPy in the same state."""
    from pyvbmc_amd import VariationalPosterior, synthetic
    from pyvbmc_amd import gp as gpm
    from pyvbmc_amd.variational_optimization import _neg_elcbo

    wl = synthetic.make_workload(2)  # D = 6, K = 20, Ns = 1e5: 3e5 draws per evaluation
    vp = VariationalPosterior(wl.D, wl.K)
    vp.ctx = ctx
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
    vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
    gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True))
    gp.ctx = ctx
    gp.update(X_new=wl.X, y_new=wl.y, hyp=wl.hyp)
    bnd = synthetic.default_theta_bnd(wl)
    res = {}
    for dev in (1, 0):
        ctx.set_option("randn_device", dev)
        np.random.seed(31)
        out = [_neg_elcbo(wl.theta.copy(), gp, vp, 0.0, wl.NsK, True, False, bnd, rng="numpy") for _ in range(3)]
        res[dev] = (out, np.random.get_state())
    ctx.set_option("randn_device", 1)
    for a, b in zip(res[1][0], res[0][0]):
        assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0])
        assert np.max(np.abs(a[1] - b[1])) <= 1e-11 * np.max(np.abs(b[1]))
    sa, sb = res[1][1], res[0][1]
    assert np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]


@pytest.mark.gpu
def test_margin_fall_back_leaves_the_state_alone(ctx):
    """the device pass must not touch the caller's state unless it succeeds.  ``randn_device`` = 2 makes it
    report its word margin exceeded after the kernels ran (about once in 1e9 calls otherwise): the direct entry point
    answers VBMC_W_NOT_FUSED with key, position and the cached value exactly as they came in, and vbmc_set_eps_numpy
    then draws on the host cores -- NumPy ends in the state np.random.randn would have left, cached value included."""
    np.random.seed(19)
    np.random.randn(3)  # leaves a cached second value
    st = np.random.get_state(legacy=True)
    assert st[3] == 1
    key = np.array(st[1], dtype=np.uint32)
    key0 = key.copy()
    pos, hg, g = C.c_int(int(st[2])), C.c_int(int(st[3])), C.c_double(float(st[4]))
    out = np.empty(100_000)
    ctx.set_option("randn_device", 2)
    try:
        rc = ctx._lib.vbmc_mt19937_randn_dev(ctx._h, key.ctypes.data_as(U32P), C.byref(pos), C.byref(hg), C.byref(g),
                                             _lib.ptr(out), out.size)
        assert rc == _lib.W_NOT_FUSED
        assert np.array_equal(key, key0) and (pos.value, hg.value, g.value) == (int(st[2]), 1, float(st[4]))
        # through the mirror: 7 * 3123 * 3 = 65 583 values (odd: a cached value goes in AND one comes out)
        from pyvbmc_amd.entropy import upload_reference_eps

        np.random.set_state(st)
        upload_reference_eps(ctx, 7, 3, 6246)
        got = np.random.get_state()
        np.random.set_state(st)
        for _ in range(7):
            np.random.randn(3123, 3)
        want = np.random.get_state()
        assert np.array_equal(got[1], want[1]) and got[2:] == want[2:]
    finally:
        ctx.set_option("randn_device", 1)


@pytest.mark.gpu
def test_state_hand_off_in_place(ctx):
    """Round 5: from the second ``rng="numpy"`` evaluation on, NumPy's key and position are read and written back through
    the bit generator's ctypes view instead of get_state / set_state (``Context.set_eps_numpy(view=...)``) -- only while the
    state is exactly what the previous call left and that call left no cached second value.  The sequence below walks
    through every case: repeats (in place), an odd number of values (the cached value goes in through set_state, the next
    call takes get_state again), the stream consumed by someone else in between (fingerprint changed), a value drawn from
    the cache only (invisible to the fingerprint -- which is why a call that left a cached value never takes the view), a
    re-seed.  After every step NumPy's full state equals what np.random.randn itself would have left."""
    from pyvbmc_amd.entropy import _np_fingerprint, upload_reference_eps

    def ours(K, D, ns):
        upload_reference_eps(ctx, K, D, ns)

    def numpys(K, D, ns):
        for _ in range(K):
            np.random.randn(ns // 2, D)

    steps = [("draw", 20, 6, 5000), ("draw", 20, 6, 5000), ("draw", 20, 6, 5000),  # 300 000 values: even
             ("draw", 7, 3, 6246),                                                  # 7 * 3123 * 3 = 65 583: odd
             ("draw", 20, 6, 5000), ("draw", 20, 6, 5000),
             ("foreign", lambda: np.random.randint(0, 10, size=3)), ("draw", 20, 6, 5000), ("draw", 20, 6, 5000),
             ("draw", 7, 3, 6246), ("foreign", lambda: np.random.randn()),          # consumes the cached value only
             ("draw", 20, 6, 5000), ("foreign", lambda: np.random.seed(5)), ("draw", 20, 6, 5000), ("draw", 20, 6, 5000)]
    states = {}
    for who, fn in (("numpy", numpys), ("ours", ours)):
        np.random.seed(123)
        seq = []
        for st in steps:
            if st[0] == "foreign":
                st[1]()
            else:
                fn(*st[1:])
            s = np.random.get_state()
            seq.append((s[1].copy(), s[2], s[3], s[4]))
        states[who] = seq
    for i, (a, b) in enumerate(zip(states["numpy"], states["ours"])):
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:], (i, steps[i][0], a[1:], b[1:])
    assert _np_fingerprint() is not None  # (the view is bound in this process: the in-place branch did run)
    assert ctx.__dict__.get("_np_left") == _np_fingerprint()
