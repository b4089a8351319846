"""Oracle: the device RNG, restated.  TEST INFRASTRUCTURE.

NumPy restatement of pyvbmc_amd/csrc/philox.h (Philox4x32-10, Salmon et al.
SC'11, + float64 Box-Muller on 32-bit words: one block = four normals of a row) so
that the Monte-Carlo entropy of the HIP kernel in VBMC_EPS_PHILOX mode can be checked
against the oracle on the SAME draws.  There is no reference counterpart (the
reference draws from NumPy's MT19937 stream).  Integer side is bit-exact; the float
side (log, sqrt, cos/sin by libm here, fitted polynomials on the device) agrees to
<= 2e-13 absolute.  ``normals`` / ``uniform`` restate the sampler's streams
(csrc/sample.hip: 53-bit uniforms, one block per pair).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over equal-shaped uint32 arrays c0..c3; scalar keys."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32).copy() for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & MASK).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def box_muller32(xr, xa):
    """Two standard normals from two 32-bit words, as csrc/philox.h philox_bm32 defines them:
    u = (xr + 1/2) 2^-32 in (0, 1), angle = 2 pi xa 2^-32 (libm here; the device's fitted
    polynomials agree to <= 2e-13 absolute)."""
    u = (xr.astype(np.float64) + 0.5) * 2.0**-32
    ang = 2.0 * np.pi * (xa.astype(np.float64) * 2.0**-32)
    rad = np.sqrt(-2.0 * np.log(u))
    return rad * np.cos(ang), rad * np.sin(ang)


def eps_half(K, n_half, D, seed, row_begin=0, row_count=None):
    """[K][row_count][D] standard normals exactly as the kernels generate them (csrc/philox.h):
    counter = (row_lo, row_hi, blk, 0), key = (seed_lo, seed_hi), row = j * n_half + i (global
    antithetic-pair row index); block blk gives dimensions 4 blk .. 4 blk + 3: (x0, x1) the first
    pair, (x2, x3) the second."""
    if row_count is None:
        row_count = n_half - row_begin
    seed = int(seed)
    out = np.empty((K, row_count, D))
    j = np.arange(K, dtype=np.uint64)[:, None]
    i = np.arange(row_begin, row_begin + row_count, dtype=np.uint64)[None, :]
    row = j * np.uint64(n_half) + i
    lo = (row & MASK).astype(np.uint32)
    hi = (row >> np.uint64(32)).astype(np.uint32)
    for b in range((D + 3) // 4):
        x = philox4x32_10(lo, hi, np.full_like(lo, b), np.zeros_like(lo), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        for h in range(2):
            d0 = 4 * b + 2 * h
            if d0 >= D:
                break
            z0, z1 = box_muller32(x[2 * h], x[2 * h + 1])
            out[:, :, d0] = z0
            if d0 + 1 < D:
                out[:, :, d0 + 1] = z1
    return out


def _split(idx):
    idx = np.asarray(idx, dtype=np.uint64)
    return (idx & MASK).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32)


def normals(idx, D, seed, c3):
    """[len(idx)][D] standard normals for counter = (idx_lo, idx_hi, pair, c3): the
    generator of csrc/philox.h with the stream word c3 free (csrc/sample.hip uses c3 = 2)."""
    seed = int(seed)
    lo, hi = _split(idx)
    out = np.empty((lo.size, D))
    for p in range((D + 1) // 2):
        x0, x1, x2, x3 = philox4x32_10(lo, hi, np.full_like(lo, p), np.full_like(lo, c3),
                                       seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        a = ((x0.astype(np.uint64) << np.uint64(32)) | x1.astype(np.uint64)) >> np.uint64(11)
        b = ((x2.astype(np.uint64) << np.uint64(32)) | x3.astype(np.uint64)) >> np.uint64(11)
        u1 = (a + np.uint64(1)).astype(np.float64) * 2.0**-53
        u2 = b.astype(np.float64) * 2.0**-53
        rad = np.sqrt(-2.0 * np.log(u1))
        out[:, 2 * p] = rad * np.cos(2.0 * np.pi * u2)
        if 2 * p + 1 < D:
            out[:, 2 * p + 1] = rad * np.sin(2.0 * np.pi * u2)
    return out


def uniform(idx, seed, c3):
    """[len(idx)] uniforms in [0,1) for counter = (idx_lo, idx_hi, 0, c3)."""
    seed = int(seed)
    lo, hi = _split(idx)
    x0, x1, _, _ = philox4x32_10(lo, hi, np.zeros_like(lo), np.full_like(lo, c3),
                                 seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a = ((x0.astype(np.uint64) << np.uint64(32)) | x1.astype(np.uint64)) >> np.uint64(11)
    return a.astype(np.float64) * 2.0**-53
