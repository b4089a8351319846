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

// Item t of a slice = (component j, row i of the slice, pair p): t = (j * rows + i) * np + p, so
// r = t / np = j * rows + i is the row of the slice's eps block, the pair's normals go to
// eps[r * D + 2p (+1)], and its Philox row is j * n_half + row_begin + i.
// 64-bit division and multiplication are tens of instructions each on this ISA -- 32-bit integer
// multiplies issue at a quarter of the rate, and this index arithmetic cost as much as the ten
// Philox rounds -- so when items, rows and offsets fit 32 bits (checked per call, every case the
// library runs) the position is derived with 32-bit operations, and a thread that generates several
// items a fixed stride apart walks it with additions.
struct GenPos {
  uint64_t grow;  // Philox row
  int64_t off;    // element offset of the pair's first normal in eps
  int64_t i;      // row within the component's slice
  int p;
};
__device__ inline bool gen_fits32(const GenSlice& g, int np) {
  const uint64_t items = (uint64_t)(g.item_begin + g.item_count), elems = (uint64_t)g.K * (uint64_t)g.rows * (uint64_t)g.D;
  return ((items | elems | (uint64_t)g.rows | (uint64_t)g.n_half) >> 32) == 0;
}
__device__ inline GenPos gen_pos(const GenSlice& g, int64_t t, int np, bool fits32) {
  GenPos q;
  if (!fits32) {
    const int64_t r = t / np;
    q.p = (int)(t - r * np);
    const int64_t j = r / g.rows;
    q.i = r - j * g.rows;
    q.grow = (uint64_t)j * (uint64_t)g.n_half + (uint64_t)(g.row_begin + q.i);
    q.off = r * g.D + 2 * q.p;
  } else {
    const uint32_t tt = (uint32_t)t, r = tt / (uint32_t)np, rows = (uint32_t)g.rows;
    const uint32_t p = tt - r * (uint32_t)np;
    const uint32_t j = r / rows, i = r - j * rows;
    q.p = (int)p;
    q.i = i;
    q.grow = (uint64_t)j * (uint64_t)(uint32_t)g.n_half + ((uint64_t)g.row_begin + i);  // one 32 x 32 -> 64 multiply-add
    q.off = (int64_t)(r * (uint32_t)g.D + 2 * p);
  }
  return q;
}
__device__ inline void gen_emit(const GenSlice& g, const GenPos& q, uint64_t seed, double& z0, double& z1, double*& dst,
                                bool& two) {
  philox_normal_pair(q.grow, (uint32_t)q.p, seed, z0, z1);
  dst = g.eps + q.off;
  two = 2 * q.p + 1 < g.D;
}

// workgroup `block` (0-based among the slice's n_blocks) of a 256-thread launch
__device__ inline void gen_slice_block(const GenSlice& g, int block, int tid) {
  const int np = (g.D + 1) / 2;
  const uint64_t seed = g.seed + (g.seed_add ? (uint64_t)g.seed_add[0] : 0);
  const bool fits32 = gen_fits32(g, np);
  if (g.per_thread <= 1) {
    const int64_t local = (int64_t)block * 256 + tid;
    if (local >= g.item_count) return;
    double z0, z1;
    double* dst;
    bool two;
    gen_emit(g, gen_pos(g, g.item_begin + local, np, fits32), seed, z0, z1, dst, two);
    dst[0] = z0;
    if (two) dst[1] = z1;
    return;
  }
  // Deferred-store form (the speculative generation inside the finish launch, api_elbo.hip): a
  // thread computes its 8 pairs (items block * 2048 + i * 256 + tid: every store instruction of a
  // wave still writes contiguous memory) and only then stores them.  The generation is arithmetic
  // for its first two thirds, so the 40 MB of stores reach the memory system in the last third.
  constexpr int PT = 8;
  const int64_t base = (int64_t)block * (256 * PT) + tid;
  const int step_r = 256 / np, step_p = 256 - step_r * np;  // 256 items further
  if (g.rows <= step_r + 1) {  // slices too small to walk (more than one component boundary per step): item by item
    for (int i = 0; i < PT; ++i) {
      const int64_t local = base + (int64_t)i * 256;
      if (local >= g.item_count) break;
      double z0, z1;
      double* dst;
      bool two;
      gen_emit(g, gen_pos(g, g.item_begin + local, np, fits32), seed, z0, z1, dst, two);
      dst[0] = z0;
      if (two) dst[1] = z1;
    }
    return;
  }
  double z[PT][2];
  double* dst[PT];
  bool two[PT];
  GenPos q = gen_pos(g, g.item_begin + base, np, fits32);
  const int64_t step_off = (int64_t)step_r * g.D + 2 * step_p, wrap_off = (int64_t)g.D - 2 * np;
  const uint64_t next_comp = (uint64_t)(g.n_half - g.rows);
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int64_t local = base + (int64_t)i * 256;
    dst[i] = nullptr;
    two[i] = false;
    if (local < g.item_count) gen_emit(g, q, seed, z[i][0], z[i][1], dst[i], two[i]);
    q.p += step_p;
    q.i += step_r;
    q.grow += (uint64_t)step_r;
    q.off += step_off;
    if (q.p >= np) {
      q.p -= np;
      ++q.i;
      ++q.grow;
      q.off += wrap_off;
    }
    if (q.i >= g.rows) {  // (rows > step_r + 1: at most one component boundary per step)
      q.i -= g.rows;
      q.grow += next_comp;
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
