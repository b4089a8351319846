// Counter-based standard-normal generator of the entropy kernels' VBMC_EPS_PHILOX mode.
// Philox4x32-10 (Salmon et al., SC'11); ONE 128-bit block gives FOUR normals of a row through two
// Box-Muller transforms on 32-bit words.  oracle/philox_ref.py restates it (integer side bit for bit,
// float side with libm: the routines below agree with libm to <= 2e-13 absolute, far inside the
// 1e-10 the parity tests ask of H), so the oracle is evaluated on the same draws.
//
//   counter = (row_lo, row_hi, blk, 0)   key = (seed_lo, seed_hi)
//   row   = global antithetic-pair row index  j * n_half + i
//   blk   = d / 4: the block's words (x0, x1) give the normals of dimensions 4 blk, 4 blk + 1,
//           (x2, x3) those of 4 blk + 2, 4 blk + 3
//   pair (xr, xa):  u  = (xr + 1/2) 2^-32  in (0, 1)           (never 0 or 1: no guards)
//                   z0 = sqrt(-2 ln u) cos(2 pi xa 2^-32),  z1 = sqrt(-2 ln u) sin(2 pi xa 2^-32)
//
// Round 3.  Rounds 1-2 spent one Philox block and 53-bit uniforms on every pair,
// <= 1 ulp log / sqrt / sincospi and 64-bit index arithmetic: ~240 vector instructions per pair
// (PMC), 21 us for the 5e6 normals of BASELINE config 3 -- the largest piece of an evaluation after
// the entropy kernel.  A random INPUT needs neither: 32-bit radius / angle words leave the normal
// law intact to 2^-33 in distribution (max |z| 6.7), so one block serves two pairs; log, sin and cos
// are degree-5 fits (tools/fit_polys.py --gen: <= 1e-14 / 4e-15 / 6e-14), sqrt is v_rsq_f64 + one
// coupled Newton step, the quadrant comes from the angle word's top two bits and the sign flips are
// integer xors; thread = block with 32-bit magic-number index arithmetic.  Measured back to back
// (tools/ubench_gen2.hip): 21.2 -> 13.7 us; with seven Philox rounds 13.0 (the Crush-resistant
// minimum; the three rounds kept are 0.25 us each and buy the generator's published safety margin);
// conversions by integer tricks instead of v_cvt / v_frexp: slower (15.2); write-through (sc1) or
// nt stores: slower (16.7); stores through an LDS image of the wave's span: slower (16.5).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"
#include "fastmath.h"

struct Philox4 {
  uint32_t x[4];
};

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                 uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0;
    k1 += W1;
  }
  Philox4 o;
  o.x[0] = c0; o.x[1] = c1; o.x[2] = c2; o.x[3] = c3;
  return o;
}

// Box-Muller on two 32-bit words (see the header of this file).
__device__ __forceinline__ void philox_bm32(uint32_t xr, uint32_t xa, double& z0, double& z1) {
  // -2 ln u, u = f 2^-32, f = xr + 1/2 = m 2^e with m in [sqrt(1/2), sqrt(2)): ln m = 2 atanh((m-1)/(m+1))
  const double f = (double)xr + 0.5;  // exact
  double m = __builtin_amdgcn_frexp_mant(f);
  int e = __builtin_amdgcn_frexp_exp(f);
  const bool lo = m < 0x1.6a09e667f3bcdp-1;
  m = __builtin_amdgcn_ldexp(m, lo ? 1 : 0);
  e = lo ? e - 1 : e;
  const double den = m + 1.0, num = m - 1.0;
  double r = __builtin_amdgcn_rcp(den);  // ~24 bits
  r = fma(fma(-den, r, 1.0), r, r);      // ~48 bits: one Newton step is enough here
  const double s = num * r, u = s * s;
  double p = 0x1.9192e478c4308p-4;       // atanh(s)/s in u = s^2, degree 5: ln m to 9.4e-15
  p = fma(p, u, 0x1.c620ee6e2b4a3p-4);
  p = fma(p, u, 0x1.2494381ee5869p-3);
  p = fma(p, u, 0x1.9999962c06032p-3);
  p = fma(p, u, 0x1.555555567148cp-2);
  p = fma(p, u, 0x1.fffffffffff12p-1);
  const double lm2 = s * p;              // ln(m) / 2
  const double x2 = fma(-4.0, lm2, (double)(32 - e) * 0x1.62e42fefa39efp+0);  // in [2.3e-10, 45.8]
  // sqrt(x2): v_rsq_f64 and one coupled Newton step
  const double y = __builtin_amdgcn_rsq(x2);
  double g = x2 * y;
  g = fma(g, fma(-0.5 * y, g, 0.5), g);
  // angle = pi (q/2 + 1/4 + t), q = the word's top two bits, t = v 2^-33 in [-1/4, 1/4); the
  // polynomials in t^2 are evaluated in v^2 with the powers of 2^-33 folded into the coefficients
  const double v = (double)(int)((xa << 2) ^ 0x80000000u);
  const double w = v * v;
  double ps = -0x1.dd54805f3f706p-8 * 0x1p-363;  // sin(pi t)/t, degree 5: 3.5e-15
  ps = fma(ps, w, 0x1.5071ce4b47930p-4 * 0x1p-297);
  ps = fma(ps, w, -0x1.32d2c644adc0bp-1 * 0x1p-231);
  ps = fma(ps, w, 0x1.466bc67123fa1p+1 * 0x1p-165);
  ps = fma(ps, w, -0x1.4abbce6257a2ap+2 * 0x1p-99);
  ps = fma(ps, w, 0x1.921fb54442cfap+1 * 0x1p-33);
  const double S = ps * v;
  double pc = -0x1.a0ee132c60c1fp-6 * 0x1p-330;  // cos(pi t), degree 5: 5.6e-14
  pc = fma(pc, w, 0x1.e1e7ccccb387ap-3 * 0x1p-264);
  pc = fma(pc, w, -0x1.55d3ba300cd50p+0 * 0x1p-198);
  pc = fma(pc, w, 0x1.03c1f074ded21p+2 * 0x1p-132);
  pc = fma(pc, w, -0x1.3bd3cc9bd2c35p+2 * 0x1p-66);
  const double C = fma(pc, w, 0x1.ffffffffffe0bp-1);
  // cos(a + pi t) = ca C - sa S, sin(a + pi t) = sa C + ca S with a = (2q + 1) pi / 4:
  // (ca, sa) = sqrt(1/2) x (+,+), (-,+), (-,-), (+,-) for q = 0..3 -- sign flips by integer xor
  const uint32_t ms = xa & 0x80000000u, mc = ((xa << 1) ^ xa) & 0x80000000u;
  auto flip = [](double d, uint32_t mask) { return __hiloint2double(__double2hiint(d) ^ (int)mask, __double2loint(d)); };
  const double gh = g * 0x1.6a09e667f3bcdp-1;
  z0 = gh * (flip(C, mc) - flip(S, ms));
  z1 = gh * (flip(C, ms) + flip(S, mc));
}

// the four normals of block `blk` of Philox row `row`
__device__ __forceinline__ void philox_normal_quad(uint64_t row, uint32_t blk, uint64_t seed, double (&z)[4]) {
  const Philox4 r = philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), blk, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  philox_bm32(r.x[0], r.x[1], z[0], z[1]);
  philox_bm32(r.x[2], r.x[3], z[2], z[3]);
}

// Item t of a slice = (component j, row i of the slice, block b): t = (j * rows + i) * nb + b with
// nb = ceil(D / 4), so r = t / nb = j * rows + i is the row of the slice's eps block, the block's
// normals go to eps[r * D + 4 b ...], and its Philox row is j * n_half + row_begin + i.  Items are
// consecutive in memory.  When items, rows and offsets fit 32 bits (checked per call: every case
// the library runs) the two divisions are a multiply-high by floor(2^32 / d) and one correction.
__device__ __forceinline__ uint32_t gen_div32(uint32_t t, uint32_t d, uint32_t magic) {
  uint32_t q = __umulhi(t, magic);  // t/d - 1 <= q <= t/d
  if (t - q * d >= d) ++q;
  return q;
}
__device__ inline bool gen_fits32(const GenSlice& g) {
  const uint64_t items = (uint64_t)(g.item_begin + g.item_count), elems = (uint64_t)g.K * (uint64_t)g.rows * (uint64_t)g.D;
  return ((items | elems | (uint64_t)g.rows | (uint64_t)g.n_half | (uint64_t)(g.row_begin + g.rows)) >> 32) == 0;
}
__device__ __forceinline__ void gen_item(const GenSlice& g, int64_t t, uint64_t seed, bool fits32) {
  uint64_t grow;
  int64_t off;
  uint32_t b;
  if (fits32) {
    const uint32_t tt = (uint32_t)t, r = gen_div32(tt, (uint32_t)g.nb, g.nb_magic);
    b = tt - r * (uint32_t)g.nb;
    const uint32_t j = gen_div32(r, (uint32_t)g.rows, g.rows_magic), i = r - j * (uint32_t)g.rows;
    grow = (uint64_t)j * (uint64_t)(uint32_t)g.n_half + ((uint64_t)g.row_begin + i);
    off = (int64_t)(r * (uint32_t)g.D + 4 * b);
  } else {
    const int64_t r = t / g.nb;
    b = (uint32_t)(t - r * g.nb);
    const int64_t j = r / g.rows, i = r - j * g.rows;
    grow = (uint64_t)j * (uint64_t)g.n_half + (uint64_t)(g.row_begin + i);
    off = r * g.D + 4 * (int64_t)b;
  }
  double z[4];
  philox_normal_quad(grow, b, seed, z);
  double* dst = g.eps + off;
  const int left = g.D - 4 * (int)b;  // normals of this block that exist (1..4 and more)
  if ((g.D & 1) == 0) {  // rows start 16-byte aligned: one or two 16-byte stores
    *reinterpret_cast<double2*>(dst) = make_double2(z[0], z[1]);
    if (left > 2) *reinterpret_cast<double2*>(dst + 2) = make_double2(z[2], z[3]);
  } else {
    dst[0] = z[0];
    if (left > 1) dst[1] = z[1];
    if (left > 2) dst[2] = z[2];
    if (left > 3) dst[3] = z[3];
  }
}

// workgroup `block` (0-based among the slice's n_blocks) of a 256-thread launch: one item per thread
__device__ inline void gen_slice_block(const GenSlice& g, int block, int tid) {
  const uint64_t seed = g.seed + (g.seed_add ? (uint64_t)g.seed_add[0] : 0);
  const int64_t local = (int64_t)block * 256 + tid;
  if (local < g.item_count) gen_item(g, g.item_begin + local, seed, gen_fits32(g));
}
