"""Edge shapes of the hot path on the GPU, each against the oracle on the same inputs:
smallest and largest supported (D, K), component counts around the 4-wave split and the
128-component switch to the generic kernel, sample counts of one antithetic pair, GP sizes
around the 64-wide matrix tiles, the L_chol=False posterior branch, degenerate mixtures
(densities that underflow to exactly zero) and the loud failures (D > 32, non-finite input).
"""
import numpy as np
import pytest
from helpers import oracle_gp, oracle_mix, rel_err

from oracle import elbo_ref, entropy_ref, gp_ref, mixture_ref, philox_ref
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


def case(D, K, N, NsK, cfg=2, S=1):
    wl = synthetic.make_workload(cfg, S=S, D=D, K=K, N=N, Ns_total=NsK * K)
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X,
              y=wl.y, hyp=wl.hyp, s2=np.zeros(0))
    return wl, wd


def objects(wd, ctx):
    from test_gpu_parity import make_gp, make_vp

    return make_vp(wd, ctx), make_gp(wd, ctx)


@pytest.mark.parametrize("D,K", [(1, 1), (1, 5), (2, 3), (3, 4), (5, 7), (7, 33), (11, 9), (13, 2), (17, 6),
                                 (21, 5), (25, 3), (32, 4), (4, 128), (3, 129), (2, 140)])
def test_entropy_shapes(ctx, D, K):
    """entmc (Philox draws, value + all gradients) and entlb for awkward (D, K): padded D,
    K not a multiple of the 4-wave split, K = 128 / 129 either side of the kernel switch."""
    from pyvbmc_amd import entlb_vbmc, entmc_vbmc

    NsK = 64 if K < 100 else 16
    wl, wd = case(D, K, 20, NsK)
    vp, _ = objects(wd, ctx)
    mix = oracle_mix(wd)
    H, dH = entmc_vbmc(vp, NsK, (True,) * 4, True, rng="philox", seed=17)
    eps = philox_ref.eps_half(K, NsK // 2, D, 17)
    Ho, dHo = entropy_ref.entmc(oracle_mix(wd), NsK, (True,) * 4, True, eps_half=eps)
    assert abs(H - Ho) <= 1e-10 * max(1.0, abs(Ho)), (D, K, H, Ho)
    assert dH.shape == dHo.shape and rel_err(dH, dHo) < 1e-8, (D, K, rel_err(dH, dHo))
    Hl, dHl = entlb_vbmc(vp, (True,) * 4, True)
    Hlo, dHlo = entropy_ref.entlb(mix, (True,) * 4, True)
    assert abs(Hl - Hlo) <= 1e-10 * max(1.0, abs(Hlo))
    assert rel_err(dHl, dHlo) < 1e-9


@pytest.mark.parametrize("NsK", [2, 3, 7, 126, 130])
def test_entropy_sample_counts(ctx, NsK):
    """One antithetic pair, odd requests (the reference rounds Ns up to even,
    entmc_vbmc.py:61), counts around the 64-row batch."""
    from pyvbmc_amd import entmc_vbmc

    wl, wd = case(3, 5, 20, 8)
    vp, _ = objects(wd, ctx)
    even = 2 * int(np.ceil(NsK / 2))
    H, dH = entmc_vbmc(vp, NsK, (True,) * 4, True, rng="philox", seed=5)
    eps = philox_ref.eps_half(5, even // 2, 3, 5)
    Ho, dHo = entropy_ref.entmc(oracle_mix(wd), even, (True,) * 4, True, eps_half=eps)
    assert abs(H - Ho) <= 1e-10 * max(1.0, abs(Ho))
    assert rel_err(dH, dHo) < 1e-8


@pytest.mark.parametrize("NsK,kernel", [(28, "small"), (32, "small"), (34, "ws"), (100, "ws")])
def test_entropy_small_sample_counts_from_uploaded_draws(ctx, NsK, kernel):
    """The optimiser's default sample counts (ns_ent = 100 K^(2/3) in total: 28 per component at
    K = 50) with draws from memory: up to 16 antithetic rows per component take the row-split small
    kernel (csrc/entropy_small.hip), more the wave-split one -- both against the oracle on the same
    draws, and the choice asserted."""
    from pyvbmc_amd import entmc_vbmc

    wl, wd = case(6, 9, 20, NsK)
    vp, _ = objects(wd, ctx)
    eps = np.random.default_rng(NsK).standard_normal((9, NsK // 2, 6))
    H, dH = entmc_vbmc(vp, NsK, (True,) * 4, True, eps_half=eps)
    assert ctx.last_entmc_plan()["kernel"] == kernel
    Ho, dHo = entropy_ref.entmc(oracle_mix(wd), NsK, (True,) * 4, True, eps_half=eps)
    assert abs(H - Ho) <= 1e-10 * max(1.0, abs(Ho))
    assert rel_err(dH, dHo) < 1e-8


def test_entropy_far_apart_components(ctx):
    """Components hundreds of widths apart: cross densities underflow to exactly 0, sigma
    ratios of 1e3; the value stays finite and equals the oracle's."""
    from pyvbmc_amd import entmc_vbmc

    wl, wd = case(4, 6, 20, 200)
    wd["mu"] = wd["mu"] * 200.0
    wd["sigma"] = wd["sigma"] * np.array([1e-2, 1.0, 10.0, 1e-1, 1.0, 3.0])
    vp, _ = objects(wd, ctx)
    H, dH = entmc_vbmc(vp, 200, (True,) * 4, True, rng="philox", seed=2)
    eps = philox_ref.eps_half(6, 100, 4, 2)
    Ho, dHo = entropy_ref.entmc(oracle_mix(wd), 200, (True,) * 4, True, eps_half=eps)
    assert np.isfinite(H) and abs(H - Ho) <= 1e-9 * max(1.0, abs(Ho))
    assert np.all(np.isfinite(dH)) and rel_err(dH, dHo) < 1e-7


@pytest.mark.parametrize("N", [17, 130, 200, 448, 800, 1100])
def test_blocked_triangular_inverse(ctx, N):
    """vbmc_set_gp forms L^-1 with the blocked kernels (64 x 64 diagonal blocks inverted in LDS,
    16-column strips by back substitution on the FP64 matrix cores; N = 1100 takes the
    one-thread-per-column fallback): the predictive variance -- which is sf^2 - |L^-T k*|^2 -- against
    the oracle's triangular solves, for N on and off the block boundaries, S = 2."""
    wl, wd = case(3, 5, N, 40, S=2)
    _, gp = objects(wd, ctx)
    ogp = oracle_gp(wd)
    rng = np.random.default_rng(N)
    xs = np.vstack([rng.standard_normal((60, 3)), wl.X[: min(N, 10)] + 1e-3 * rng.standard_normal((min(N, 10), 3))])
    fmu, fs2 = gp.predict(xs, separate_samples=True)
    omu, os2 = gp_ref.predict(ogp, xs, separate_samples=True)
    sf2 = float(np.exp(2 * wl.hyp[0, 3]))
    err = float(np.max(np.abs(fs2 - os2)))
    print(f"N={N}: max |fs2 - oracle| = {err:.2e} (sf2 = {sf2:.3g})")
    assert np.max(np.abs(fmu - omu)) <= 1e-10 * max(1.0, np.max(np.abs(omu)))
    assert err <= 1e-10 * max(1.0, sf2)


@pytest.mark.parametrize("N,M,S", [(17, 33, 1), (64, 64, 2), (130, 333, 3), (449, 1000, 2), (512, 129, 1), (700, 2100, 1)])
def test_predict_lds_direct_product(ctx, N, M, S):
    """Batches of more than 32 points on Cholesky samples take predict_var_dma_kernel (padded
    operands, LDS-direct loads, folded column tiles); `predict_dma = 0` forces the plain 64 x 64
    kernel.  Same partial sums in a different order: the two must agree to rounding, and both with
    the oracle, for N / M on and off the 64-wide tiles and the 32-deep panels."""
    wl, wd = case(3, 5, N, 40, S=S)
    _, gp = objects(wd, ctx)
    ogp = oracle_gp(wd)
    rng = np.random.default_rng(N + M)
    xs = rng.standard_normal((M, 3))
    xs[: min(N, 8)] = wl.X[: min(N, 8)] + 1e-3 * rng.standard_normal((min(N, 8), 3))
    sf2 = float(np.exp(2 * wl.hyp[:, 3]).max())
    try:
        fmu, fs2 = gp.predict(xs, separate_samples=True)
        ctx.set_option("predict_dma", 0)
        pmu, ps2 = gp.predict(xs, separate_samples=True)
    finally:
        ctx.set_option("predict_dma", 1)
    omu, os2 = gp_ref.predict(ogp, xs, separate_samples=True)
    assert np.array_equal(fmu, pmu)  # the mean does not depend on the variance kernel
    assert np.max(np.abs(fs2 - ps2)) <= 1e-12 * max(1.0, sf2)
    assert np.max(np.abs(fs2 - os2)) <= 1e-10 * max(1.0, sf2)
    assert np.max(np.abs(fmu - omu)) <= 1e-10 * max(1.0, np.max(np.abs(omu)))


@pytest.mark.parametrize("N,M,S", [(17, 33, 1), (64, 64, 2), (130, 333, 3), (449, 1000, 2), (512, 129, 1), (700, 2100, 1),
                                   (400, 8192, 1)])
def test_predict_finish_in_the_product_epilogue(ctx, N, M, S):
    """`predict_fused`: the product kernel's last-arriving workgroup of a 64-point row tile adds the tile's partial
    sums up in slot order and writes fmu / fs2 (two launches) -- bit-identical to predict_finish_kernel (three), for
    row tiles whose column tiles are folded into one workgroup, shared between two, and the even-count middle tile;
    repeated calls reuse the tickets."""
    wl, wd = case(3, 5, N, 40, S=S)
    _, gp = objects(wd, ctx)
    xs = np.random.default_rng(N + M).standard_normal((M, 3))
    out = {}
    try:
        for mode in (0, 2, 1, 2):
            ctx.set_option("predict_fused", mode)
            out.setdefault(mode, []).append(gp.predict(xs, separate_samples=True))
    finally:
        ctx.set_option("predict_fused", 1)
    mu0, s0 = out[0][0]
    for mode in (1, 2):
        for mu, s2 in out[mode]:
            assert np.array_equal(mu, mu0) and np.array_equal(s2, s0)


@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 129])
def test_gp_sizes_around_the_tiles(ctx, N):
    """_gp_log_joint (+variance), predict and the fused objective for N around the 64-wide
    MFMA tiles, S = 2 hyper-samples."""
    from pyvbmc_amd.variational_optimization import _gp_log_joint, _neg_elcbo

    wl, wd = case(3, 5, N, 40, S=2)
    vp, gp = objects(wd, ctx)
    mix, ogp = oracle_mix(wd), oracle_gp(wd)
    G, dG, _, _, _ = _gp_log_joint(vp, gp, True, True, True, False, False)
    Go, dGo, _, _, _ = gp_ref.gp_log_joint(mix, ogp, True, True, True, False, False)
    assert abs(G - Go) <= 1e-10 * max(1.0, abs(Go)) and rel_err(dG, dGo) < 1e-9
    r = _gp_log_joint(vp, gp, False, True, True, True, True)
    ro = gp_ref.gp_log_joint(mix, ogp, False, True, True, True, True)
    scale = max(1.0, float(np.max(np.abs(ro[6]))))
    assert np.max(np.abs(r[6] - ro[6])) <= 1e-9 * scale  # J_sjk
    assert abs(np.ravel(r[2])[0] - np.ravel(ro[2])[0]) <= 1e-9 * max(1.0, abs(np.ravel(ro[2])[0]))
    xs = np.random.default_rng(N).standard_normal((70, 3))
    fmu, fs2 = gp.predict(xs, separate_samples=True)
    omu, os2 = gp_ref.predict(ogp, xs, separate_samples=True)
    sf2 = float(np.exp(2 * wl.hyp[0, 3]))
    assert np.max(np.abs(fmu - omu)) <= 1e-10 * max(1.0, np.max(np.abs(omu)))
    assert np.max(np.abs(fs2 - os2)) <= 1e-10 * max(1.0, sf2)
    eps = synthetic.draw_eps_half(5, 3, 40, 3)
    F = _neg_elcbo(wl.theta.copy(), gp, vp, 0.0, 40, True, False, None, eps_half=eps)
    Fo = elbo_ref.neg_elcbo(wl.theta.copy(), ogp, oracle_mix(wd), 0.0, 40, True, False, None, eps_half=eps)
    assert abs(F[0] - Fo[0]) <= 1e-10 * max(1.0, abs(Fo[0])) and rel_err(F[1], Fo[1]) < 1e-8


def test_predict_batch_shapes(ctx):
    wl, wd = case(3, 5, 50, 40, S=3)
    _, gp = objects(wd, ctx)
    ogp = oracle_gp(wd)
    fmu, fs2 = gp.predict(np.zeros((0, 3)))
    assert fmu.shape == (0, 1) and fs2.shape == (0, 1)
    x1 = np.array([[0.1, -0.2, 0.3]])
    for sep in (False, True):
        fmu, fs2 = gp.predict(x1, separate_samples=sep)
        omu, os2 = gp_ref.predict(ogp, x1, separate_samples=sep)
        assert fmu.shape == omu.shape and np.allclose(fmu, omu, rtol=0, atol=1e-10 * max(1.0, np.abs(omu).max()))
        assert np.allclose(fs2, os2, rtol=0, atol=1e-9)
    fmu, fs2 = gp.predict(x1, add_noise=True)
    omu, os2 = gp_ref.predict(ogp, x1, add_noise=True)
    assert np.allclose(fs2, os2, rtol=0, atol=1e-9)


def test_low_noise_posterior_branch(ctx):
    """sn2 < 1e-6 switches the posterior to L = -(K + sn2 I)^-1 (L_chol = False): the variance
    then uses z^T L z / K*^T L K* instead of triangular products."""
    from pyvbmc_amd.variational_optimization import _gp_log_joint

    wl, wd = case(3, 4, 40, 40)
    hyp = wd["hyp"].copy()
    hyp[:, 3 + 1] = np.log(3e-4)  # log sn -> sn2 = 9e-8
    wd["hyp"] = hyp
    vp, gp = objects(wd, ctx)
    assert not gp.posteriors[0].L_chol
    mix, ogp = oracle_mix(wd), oracle_gp(wd)
    r = _gp_log_joint(vp, gp, False, True, True, True, True)
    ro = gp_ref.gp_log_joint(mix, ogp, False, True, True, True, True)
    assert abs(r[0] - ro[0]) <= 1e-9 * max(1.0, abs(ro[0]))
    assert np.max(np.abs(r[6] - ro[6])) <= 1e-7 * max(1.0, float(np.max(np.abs(ro[6]))))
    xs = np.random.default_rng(0).standard_normal((33, 3))
    fmu, fs2 = gp.predict(xs, separate_samples=True)
    omu, os2 = gp_ref.predict(ogp, xs, separate_samples=True)
    sf2 = float(np.exp(2 * hyp[0, 3]))
    assert np.max(np.abs(fmu - omu)) <= 1e-8 * max(1.0, np.max(np.abs(omu)))
    assert np.max(np.abs(fs2 - os2)) <= 1e-8 * max(1.0, sf2)


def test_pdf_underflow_and_extremes(ctx):
    wl, wd = case(4, 6, 20, 40)
    vp, _ = objects(wd, ctx)
    mix = oracle_mix(wd)
    x = np.vstack([wd["mu"].T, 1e3 * np.ones((2, 4)), -1e6 * np.ones((1, 4)), np.zeros((1, 4))])
    with np.errstate(all="ignore"):
        y = vp.pdf(x, orig_flag=False)
        yo = mixture_ref.pdf(mix, x)
        ly = vp.pdf(x, orig_flag=False, log_flag=True)
        lyo = mixture_ref.pdf(mix, x, log_flag=True)
    assert np.array_equal(y == 0, yo == 0) and rel_err(y, yo) < 1e-12
    assert np.array_equal(np.isneginf(ly), np.isneginf(lyo))
    fin = np.isfinite(lyo)
    assert np.max(np.abs(ly[fin] - lyo[fin])) < 1e-11 * max(1.0, np.max(np.abs(lyo[fin])))


def test_loud_failures(ctx):
    from pyvbmc_amd import VariationalPosterior, entmc_vbmc
    from pyvbmc_amd._lib import VbmcHipError
    from pyvbmc_amd.variational_optimization import _neg_elcbo

    big = VariationalPosterior(33, 2)
    big.ctx = ctx
    with pytest.raises(NotImplementedError):
        entmc_vbmc(big, 10, rng="philox", seed=1)
    with pytest.raises(NotImplementedError):
        big.pdf(np.zeros((2, 33)), orig_flag=False)
    wl, wd = case(3, 4, 20, 20)
    vp, gp = objects(wd, ctx)
    bad = wl.theta.copy()
    bad[1] = np.inf
    with pytest.raises((VbmcHipError, ValueError)):
        _neg_elcbo(bad, gp, vp, 0.0, 20, True, False, None, rng="philox", seed=1)
    with pytest.raises(ValueError):
        _neg_elcbo(wl.theta[:-2].copy(), gp, vp, 0.0, 20, True, False, None, rng="philox", seed=1)
    vp.sigma = np.zeros_like(vp.sigma)
    with pytest.raises((VbmcHipError, ValueError)):
        entmc_vbmc(vp, 10, rng="philox", seed=1)
    # the context survives every one of them
    vp2, _ = objects(wd, ctx)
    H, _ = entmc_vbmc(vp2, 20, rng="philox", seed=1)
    assert np.isfinite(H)


@pytest.mark.parametrize("low_noise", [False, True], ids=["chol", "no-chol"])
def test_small_batches_take_the_small_kernels(ctx, low_noise):
    """A handful of points (a CMA-ES population, one candidate) goes through kernels of its own --
    predict_var_small_kernel (M <= 32) and mixture_pdf_wave_kernel (n <= 2^16) -- which must agree
    with the oracle exactly as the large-batch kernels do, on both posterior branches, for several
    GP samples, and for N not a multiple of the 64-column tile."""
    wl, wd = case(5, 9, 150, 40, S=3)
    if low_noise:
        hyp = wd["hyp"].copy()
        hyp[:, 5 + 1] = np.log(3e-4)
        wd["hyp"] = hyp
    vp, gp = objects(wd, ctx)
    assert gp.posteriors[0].L_chol != low_noise
    mix, ogp = oracle_mix(wd), oracle_gp(wd)
    sf2 = float(np.exp(2 * wd["hyp"][0, 5]))
    rng = np.random.default_rng(3)
    for M in (1, 3, 4, 5, 16, 17, 32, 33, 70):
        xs = rng.standard_normal((M, 5))
        for sep in (True, False):
            fmu, fs2 = gp.predict(xs, separate_samples=sep)
            omu, os2 = gp_ref.predict(ogp, xs, separate_samples=sep)
            assert fmu.shape == omu.shape and fs2.shape == os2.shape
            assert np.max(np.abs(fmu - omu)) <= 1e-8 * max(1.0, np.max(np.abs(omu))), M
            assert np.max(np.abs(fs2 - os2)) <= 1e-8 * max(1.0, sf2), M
        for log_flag in (False, True):
            y, dy = vp.pdf(xs, orig_flag=False, log_flag=log_flag, grad_flag=True)
            yo, dyo = mixture_ref.pdf(mix, xs, log_flag=log_flag, grad_flag=True)
            assert rel_err(y, yo) < 1e-12 and rel_err(dy, dyo) < 1e-11, (M, log_flag)
    # the same points through both pdf kernels: a large batch containing the small one
    big = rng.standard_normal((70000, 5))
    yb = vp.pdf(big, orig_flag=False, log_flag=True)
    ys = vp.pdf(big[:1000], orig_flag=False, log_flag=True)
    assert rel_err(ys, yb[:1000]) < 1e-13


def test_empty_row_slice_contributes_zero(ctx):
    """A rank whose slice of the antithetic rows is empty (n_half < world, or rows=(b, 0)): every entropy workgroup's
    partial row must still be WRITTEN -- as zeros -- because the finish kernel sums all of them (the
    wave-split kernel's chunk mode skipped an empty chunk and left stale scratch rows behind)."""
    from pyvbmc_amd import entmc_vbmc

    wl, wd = case(6, 20, 30, 4000)
    vp, _ = objects(wd, ctx)
    # a full evaluation first: the scratch rows of the next launch then hold this one's values
    _, _, full = entmc_vbmc(vp, wl.NsK, (True,) * 4, True, rng="philox", seed=3, return_raw=True)
    assert np.abs(full).max() > 0
    for begin in (0, 7, wl.NsK // 2):
        _, _, raw = entmc_vbmc(vp, wl.NsK, (True,) * 4, True, rng="philox", seed=3, return_raw=True, rows=(begin, 0))
        assert np.array_equal(raw, np.zeros_like(raw)), (begin, np.abs(raw).max())
    # and the two halves still add up to the whole
    h = wl.NsK // 2
    _, _, a = entmc_vbmc(vp, wl.NsK, (True,) * 4, True, rng="philox", seed=3, return_raw=True, rows=(0, h // 3))
    _, _, b = entmc_vbmc(vp, wl.NsK, (True,) * 4, True, rng="philox", seed=3, return_raw=True, rows=(h // 3, h - h // 3))
    assert rel_err(a + b, full) < 1e-13


@pytest.mark.parametrize("D,N", [(1, 3), (2, 255), (4, 1025), (5, 256), (8, 1100), (9, 257), (10, 513), (12, 700),
                                 (13, 300), (16, 515), (17, 511), (20, 800), (20, 1030), (21, 257), (24, 520),
                                 (25, 130), (32, 300)])
def test_gp_block_forms(ctx, D, N):
    """The GP expected-log-joint block (csrc/glj_block.h) in every build of its one-pass form (D <= 24: thread = point,
    P points per thread and round, the 1 + 2D sums in registers) and in the two-pass form (D > 24), with point counts
    around one and several rounds of 256 P points: value + gradient, value only, and the variance path (which takes the
    z_n from the same block) against the oracle; then the same sums through the fused objective's placements (prep
    launch, riders of the entropy launch)."""
    from pyvbmc_amd.variational_optimization import _gp_log_joint, _neg_elcbo

    K = 3
    wl, wd = case(D, K, N, 64, cfg=3, S=2)
    vp, gp = objects(wd, ctx)
    mix, ogp = oracle_mix(wd), oracle_gp(wd)
    G, dG, _, _, _ = _gp_log_joint(vp, gp, True, True, True, False, False)
    Go, dGo, _, _, _ = gp_ref.gp_log_joint(mix, ogp, True, True, True, False, False)
    assert abs(G - Go) <= 1e-10 * max(1.0, abs(Go)) and rel_err(dG, dGo) < 1e-9
    G0 = _gp_log_joint(vp, gp, False, True, True, False, False)[0]
    assert abs(G0 - Go) <= 1e-10 * max(1.0, abs(Go))
    r = _gp_log_joint(vp, gp, False, True, True, True, True)
    ro = gp_ref.gp_log_joint(mix, ogp, False, True, True, True, True)
    scale = max(1.0, float(np.max(np.abs(ro[6]))))
    assert np.max(np.abs(r[6] - ro[6])) <= 1e-9 * scale  # J_sjk
    eps = synthetic.draw_eps_half(K, D, 64, 3)
    bnd = synthetic.default_theta_bnd(wl)
    F = _neg_elcbo(wl.theta.copy(), gp, vp, 0.0, 64, True, False, bnd, eps_half=eps)
    Fo = elbo_ref.neg_elcbo(wl.theta.copy(), ogp, oracle_mix(wd), 0.0, 64, True, False, bnd, eps_half=eps)
    assert abs(F[0] - Fo[0]) <= 1e-10 * max(1.0, abs(Fo[0])) and rel_err(F[1], Fo[1]) < 1e-8
    assert abs(F[2] - Fo[2]) <= 1e-10 * max(1.0, abs(Fo[2]))
    # Philox draws: the polled step (GP sums as riders of the entropy launch or in the prep launch, armed or not)
    for seed in (5, 6, 7):
        Fp = _neg_elcbo(wl.theta.copy(), gp, vp, 0.0, 64, True, False, bnd, rng="philox", seed=seed)
        assert abs(Fp[2] - Fo[2]) <= 1e-10 * max(1.0, abs(Fo[2]))  # G does not depend on the draws
