// Counter-based standard-normal generator used by the entropy kernels in
// VBMC_EPS_PHILOX mode.  Philox4x32-10 (Salmon et al., SC'11) + Box-Muller in
// float64.  oracle/philox_ref.py restates this bit-for-bit on the integer side so
// the parity tests can evaluate the oracle on identical draws.
//
//   counter = (row_lo, row_hi, pair, 0)   key = (seed_lo, seed_hi)
//   row   = global antithetic-pair row index  j * n_half + i
//   pair  = d / 2  (one Philox call yields the normals of dimensions 2p, 2p+1)
//   u1 = (((x0<<32 | x1) >> 11) + 1) * 2^-53   in (0,1]
//   u2 =  ((x2<<32 | x3) >> 11)      * 2^-53   in [0,1)
//   z0 = sqrt(-2 ln u1) cos(2 pi u2),  z1 = sqrt(-2 ln u1) sin(2 pi u2)
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

__device__ inline void philox_normal_pair(uint64_t row, uint32_t pair, uint64_t seed, double& z0, double& z1);

// Item t of a slice = (component j, row i of the slice, pair p): t = (j * rows + i) * np + p.
// 64-bit division is ~50 instructions on this ISA and the generator needs two per item; items and
// rows fit 32 bits in every case the library runs (checked per call), and a thread that generates
// several items a fixed stride apart walks the position instead of dividing again.
struct GenPos {
  int64_t j, i;
  int p;
};
__device__ inline GenPos gen_pos(const GenSlice& g, int64_t t, int np) {
  GenPos q;
  if (((uint64_t)t | (uint64_t)g.rows) >> 32) {
    const int64_t r = t / np;
    q.p = (int)(t - r * np);
    q.j = r / g.rows;
    q.i = r - q.j * g.rows;
  } else {
    const uint32_t tt = (uint32_t)t, r = tt / (uint32_t)np, rows = (uint32_t)g.rows;
    q.p = (int)(tt - r * (uint32_t)np);
    const uint32_t jj = r / rows;
    q.j = jj;
    q.i = r - jj * rows;
  }
  return q;
}
__device__ inline void gen_emit(const GenSlice& g, const GenPos& q, uint64_t seed, double& z0, double& z1, double*& dst,
                                bool& two) {
  const uint64_t grow = (uint64_t)q.j * (uint64_t)g.n_half + (uint64_t)(g.row_begin + q.i);
  philox_normal_pair(grow, (uint32_t)q.p, seed, z0, z1);
  dst = g.eps + (q.j * g.rows + q.i) * g.D + 2 * q.p;
  two = 2 * q.p + 1 < g.D;
}

// workgroup `block` (0-based among the slice's n_blocks) of a 256-thread launch
__device__ inline void gen_slice_block(const GenSlice& g, int block, int tid) {
  const int np = (g.D + 1) / 2;
  const uint64_t seed = g.seed + (g.seed_add ? (uint64_t)g.seed_add[0] : 0);
  if (g.per_thread <= 1) {
    const int64_t local = (int64_t)block * 256 + tid;
    if (local >= g.item_count) return;
    double z0, z1;
    double* dst;
    bool two;
    gen_emit(g, gen_pos(g, g.item_begin + local, np), seed, z0, z1, dst, two);
    dst[0] = z0;
    if (two) dst[1] = z1;
    return;
  }
  // Deferred-store form (the speculative generation inside the finish launch, api_elbo.hip): a
  // thread computes its 8 pairs (items block * 2048 + i * 256 + tid: every store instruction of a
  // wave still writes contiguous memory) and only then stores them.  The generation is arithmetic
  // for its first two thirds, so the 40 MB of stores reach the memory system in the last third.
  constexpr int PT = 8;
  double z[PT][2];
  double* dst[PT];
  bool two[PT];
  const int64_t base = (int64_t)block * (256 * PT) + tid;
  GenPos q = gen_pos(g, g.item_begin + base, np);
  const int step_r = 256 / np, step_p = 256 - step_r * np;  // 256 items further
  const bool walk = g.rows > step_r + 1;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int64_t local = base + (int64_t)i * 256;
    dst[i] = nullptr;
    two[i] = false;
    if (local < g.item_count) gen_emit(g, q, seed, z[i][0], z[i][1], dst[i], two[i]);
    if (walk) {
      q.p += step_p;
      q.i += step_r;
      if (q.p >= np) {
        q.p -= np;
        ++q.i;
      }
      if (q.i >= g.rows) {  // (rows > step_r + 1: at most one component boundary per step)
        q.i -= g.rows;
        ++q.j;
      }
    } else {
      q = gen_pos(g, g.item_begin + local + 256, np);
    }
  }
#pragma unroll
  for (int i = 0; i < PT; ++i)
    if (dst[i]) {
      dst[i][0] = z[i][0];
      if (two[i]) dst[i][1] = z[i][1];
    }
}

__device__ inline void philox_normal_pair(uint64_t row, uint32_t pair, uint64_t seed,
                                          double& z0, double& z1) {
  Philox4 r = philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), pair, 0u, (uint32_t)seed,
                            (uint32_t)(seed >> 32));
  uint64_t a = (((uint64_t)r.x[0] << 32) | r.x[1]) >> 11;
  uint64_t b = (((uint64_t)r.x[2] << 32) | r.x[3]) >> 11;
  double u1 = (double)(a + 1) * 0x1.0p-53;
  double u2 = (double)b * 0x1.0p-53;
  double rad = sqrt(-2.0 * fm::log_fast(u1));
  double s, c;
  fm::sincospi_fast(2.0 * u2, s, c);
  z0 = rad * c;
  z1 = rad * s;
}
