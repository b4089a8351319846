"""The device draw generator (csrc/philox.h: Philox4x32-10, one block = four normals through two
Box-Muller transforms on 32-bit words) read back from the device: against its restatement
(oracle/philox_ref.py: integer side bit for bit, libm on the float side) and as a distribution --
what the reference's ``np.random.randn`` draws are (entropy/entmc_vbmc.py:64-68)."""
import numpy as np
import pytest

from oracle import philox_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pyvbmc_amd import _lib

    c = _lib.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("K,n_half,D,r0,n", [(3, 1000, 10, 0, 1000), (5, 777, 5, 100, 333), (2, 4096, 1, 4000, 96),
                                             (4, 300, 3, 0, 300), (50, 10000, 10, 8750, 1250), (7, 513, 20, 1, 511),
                                             (3, 64, 32, 0, 64), (100, 2500, 20, 625, 625), (1, 9, 2, 0, 9), (6, 50, 7, 25, 25)])
def test_device_draws_vs_restatement(ctx, K, n_half, D, r0, n):
    seed = 0x1234_5678_9ABC + 17 * D
    got = ctx.philox_normals(K, n_half, D, seed, r0, n)
    want = philox_ref.eps_half(K, n_half, D, seed, r0, n)
    err = float(np.max(np.abs(got - want)))
    print(f"K={K} n_half={n_half} D={D} rows [{r0},+{n}): max |device - restatement| = {err:.2e}")
    assert err <= 1e-12
    # a slice is the corresponding part of the whole: the values depend on (seed, global row, block) only
    if r0 > 0:
        whole = ctx.philox_normals(K, n_half, D, seed)
        assert np.array_equal(whole[:, r0:r0 + n, :], got)


def test_device_draws_are_standard_normal(ctx):
    """5e6 normals of BASELINE config 3's shape: moments, Kolmogorov-Smirnov, tail counts, independence
    of the four normals of a block and of neighbouring rows."""
    from scipy import stats

    K, h, D = 50, 10000, 10
    z = ctx.philox_normals(K, h, D, 20260928)
    x = z.ravel()
    n = x.size
    m1, m2 = x.mean(), x.var()
    m3, m4 = np.mean(x**3), np.mean(x**4)
    print(f"n={n}: mean {m1:.2e} var {m2:.5f} skew {m3:.2e} kurt {m4:.4f} max |z| {np.max(np.abs(x)):.3f}")
    assert abs(m1) < 5 / np.sqrt(n) and abs(m2 - 1) < 5 * np.sqrt(2 / n)
    assert abs(m3) < 5 * np.sqrt(15 / n) and abs(m4 - 3) < 5 * np.sqrt(96 / n)
    assert np.max(np.abs(x)) <= np.sqrt(-2 * np.log(0.5 * 2.0**-32)) + 1e-9  # 6.76: the 32-bit radius word's bound
    ks = stats.kstest(x[::5], "norm")
    print(f"KS on {x[::5].size} draws: D = {ks.statistic:.2e}, p = {ks.pvalue:.3f}")
    assert ks.pvalue > 1e-3
    for thr in (2.0, 3.0, 4.0, 4.5):
        p = 2 * stats.norm.sf(thr)
        cnt = int(np.sum(np.abs(x) > thr))
        print(f"|z| > {thr}: {cnt} (expected {n * p:.1f} +- {np.sqrt(n * p):.1f})")
        assert abs(cnt - n * p) < 5 * np.sqrt(n * p) + 1
    # every dimension on its own, and the pairs that share a Box-Muller transform / a Philox block
    for d in range(D):
        col = z[:, :, d].ravel()
        assert abs(col.mean()) < 5 / np.sqrt(col.size) and abs(col.var() - 1) < 5 * np.sqrt(2 / col.size)
    c = np.corrcoef(z.reshape(-1, D), rowvar=False)
    off = np.max(np.abs(c - np.eye(D)))
    print(f"max |corr| between dimensions: {off:.2e}")
    assert off < 5 / np.sqrt(K * h)
    c2 = np.corrcoef((z.reshape(-1, D) ** 2), rowvar=False)
    assert np.max(np.abs(c2 - np.eye(D))) < 5 / np.sqrt(K * h)  # radii of a pair are shared, their squares' cross-terms not
    # neighbouring rows (consecutive counters) and neighbouring seeds
    a, b = z[:, :-1, :].ravel(), z[:, 1:, :].ravel()
    assert abs(np.corrcoef(a, b)[0, 1]) < 5 / np.sqrt(a.size)
    z2 = ctx.philox_normals(K, h, D, 20260929)
    assert abs(np.corrcoef(x, z2.ravel())[0, 1]) < 5 / np.sqrt(n)
