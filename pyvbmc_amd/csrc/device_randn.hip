// The reference's draw stream, generated on the device: the next n values of np.random.randn (reference
// entropy/entmc_vbmc.py:64-68 -- NumPy's legacy global RandomState: MT19937 words -> 53-bit uniforms -> Marsaglia's
// polar method with its cached second value) written straight into HBM, and the state NumPy would be left in handed
// back.  The drop-in's default (rng="numpy") then needs neither the host cores (csrc/host_randn.hip: 3 ms for config
// 3's 5e6 normals on 64 cores, 48 ms with np.random.randn itself) nor 40 MB over PCIe per evaluation.
//
// MT19937 is sequential only in its recurrence; S streams of J = 80 blocks (49 920 words) run side by side once every
// stream has its first block, and that block is a jump over GF(2) (mt_jump.h): each of its 624 words is the XOR of the
// ~10 000 words x_(1+i+j), g_i = 1, of ONE window of 20 560 words behind the current state, the same for all streams.
//   mt_window_kernel   one workgroup: the window, by the plain recurrence (33 blocks)
//   mt_stream_kernel   workgroup m: jump to block m * 80 (window and the polynomial's exponent list in LDS), then its blocks
//                      by the recurrence, 40 at a time in LDS, and per half stream the number of accepted polar attempts
//                      among the attempts that START in its words (attempt t reads words [pos + 4 t, pos + 4 t + 4) of the
//                      sequence whether it is accepted or not: csrc/host_randn.hip).  Round 6: it leaves its blocks' words
//                      in memory (one flat sequence, 51 MB at config 3), so that nothing sequential is left for the values:
//   mt_scan_kernel     exclusive prefix sum of the 2 S counts
//   mt_values_kernel   one workgroup per half stream (2 S of them; round 5's second pass ran the 80 blocks of every stream
//                      AGAIN on S workgroups: 73 us): the half stream's words through LDS, ranks of its accepted attempts by a block scan, the
//                      pair (f x2, f x1), f = sqrt(-2 log(r2) / r2), written where it belongs; the workgroup that holds
//                      the request's last pair hands back the block and position NumPy's state ends at and that
//                      attempt's (x1, x2, r2) -- the cached second value is then formed on the HOST with libm, so the
//                      state handed back to NumPy is bit-identical to the one np.random.randn leaves.
// Integer stream, accept / reject decisions, counts and final state: bit-exact.  Values: NumPy's expression, IEEE
// multiply / divide / sqrt, and a logarithm rounded from a double-double value (mt_log_dd) -- neither it nor glibc's log
// is correctly rounded everywhere, so about one value in a thousand differs from np.random.randn's by one to three units
// in the last place; tests/test_device_randn.py says so.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"
#include "mt_jump.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int BLK_PER_STREAM = 80, BLK_PER_PASS = 40;
constexpr int J_WORDS = BLK_PER_STREAM * MT_N;                    // 49 920
constexpr int WIN_WORDS = 33 * MT_N;                               // x_0 .. x_20591 (x_1 .. x_20560 are used)
constexpr int PASS_WORDS = (BLK_PER_PASS + 1) * MT_N;              // 25 584 words = 102 KB: the blocks of a pass
constexpr int TAP_OFF = 33 * MT_N + 640;                           // (words) the jump's exponent list behind the window and its zero block
constexpr int LDS_WORDS = TAP_OFF + 20032 / 2;                     // 31 248 words = 125 KB
constexpr int NT = 1024;
constexpr int TAP_U = 16;                  // LDS reads in flight per thread in the jump
constexpr int TAP_STRIDE = 20032;          // entries per polynomial: 19 937 + four lists' padding, a multiple of TAP_U
constexpr int TAP_PAD = WIN_WORDS;         // exponents TAP_PAD + 3 - c, c = 0 .. 3: class-c exponents whose window words are zeros
static_assert(LDS_WORDS >= PASS_WORDS && TAP_OFF >= WIN_WORDS + 4 + MT_N + 8 && (TAP_OFF % 4) == 0, "LDS plan");
static_assert(TAP_STRIDE % TAP_U == 0 && TAP_PAD + 3 < 65536 && TAP_U == 16 && (TAP_PAD % 4) == 0, "tap list layout");
static_assert(19937 + 4 * (TAP_U - 1) <= TAP_STRIDE, "the four padded lists fit a polynomial's slot");
static_assert(1024 + 4 * 6 * (MT_N / 4 + 1) * 4 + 4 <= WIN_WORDS, "the jump's partial sums fit over the window");

__device__ __forceinline__ uint32_t mt_twist(uint32_t hi, uint32_t lo) {
  const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
  return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}
// The recurrence on a buffer of consecutive words in LDS: x_n = x_(n-227) ^ twist(x_(n-624), x_(n-623)) for n in
// [624, n_end) given x_0 .. x_623 -- 227 words per step (the nearest operand is 227 words back), one barrier per step:
// 2.75 barriers per block where round 5's block-by-block form (three phases and the last word) had four.  All NT threads call it.
__device__ __forceinline__ void mt_extend_lds(uint32_t* x, int n_end) {
  constexpr int Q = MT_N - MT_M;
  const int i = threadIdx.x;
  for (int n0 = MT_N; n0 < n_end; n0 += Q) {
    const int n = n0 + i;
    if (i < Q && n < n_end) x[n] = x[n - Q] ^ mt_twist(x[n - MT_N], x[n - MT_N + 1]);
    __syncthreads();
  }
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
__device__ __forceinline__ double mt_double(uint32_t w0, uint32_t w1) {
  const int32_t a = (int32_t)(mt_temper(w0) >> 5), b = (int32_t)(mt_temper(w1) >> 6);
  return (a * 67108864.0 + b) / 9007199254740992.0;
}
// legacy_gauss's loop body on four consecutive words (host twin: host_randn.hip attempt_at); contraction is off
__device__ __forceinline__ bool mt_attempt(const uint32_t* u, double& x1, double& x2, double& r2) {
  x1 = 2.0 * mt_double(u[0], u[1]) - 1.0;
  x2 = 2.0 * mt_double(u[2], u[3]) - 1.0;
  r2 = x1 * x1 + x2 * x2;
  return !(r2 >= 1.0 || r2 == 0.0);
}

// ln(x) for normal x > 0, rounded from a double-double value: within ~0.001 ulp of the exact value before the final
// rounding, i.e. the correctly rounded result for all but ~0.2 % of the arguments (those whose logarithm lies that close to
// a rounding boundary).  glibc's log (NumPy's) is that good too and misses the correctly rounded result in ~0.06 % -- so
// the two differ, by one unit in the last place, in about one draw of 500, which f = sqrt(-2 l / r2) and the product f x
// turn into one to three units (measured: 99.9 % of 1.2e6 values bit-identical, 1 061 / 200 / 1 off by 1 / 2 / 3 ulp).
// The device library's own log (<= 1 ulp) left a tenth of the values off.
//   x = 2^e m, m in [sqrt(1/2), sqrt(2));  ln m = 2 atanh(s), s = (m - 1) / (m + 1) as head + tail
__device__ __forceinline__ double mt_log_dd(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  if (m < 0x1.6a09e667f3bcdp-1) {
    m *= 2.0;
    e -= 1;
  }
  const double num = m - 1.0;          // exact
  const double dh = m + 1.0;           // head of m + 1 ...
  const double dl = m - (dh - 1.0);    // ... and its tail (exact: fast two-sum, |m| <= 2)
  const double sh = num / dh;
  const double res = fma(-sh, dh, num);               // num - sh dh, exact
  const double sl = (res - sh * dl) / dh;              // tail of the quotient
  const double u = sh * sh;
  double p = 1.0 / 27.0;
  p = fma(p, u, 1.0 / 25.0);
  p = fma(p, u, 1.0 / 23.0);
  p = fma(p, u, 1.0 / 21.0);
  p = fma(p, u, 1.0 / 19.0);
  p = fma(p, u, 1.0 / 17.0);
  p = fma(p, u, 1.0 / 15.0);
  p = fma(p, u, 1.0 / 13.0);
  p = fma(p, u, 1.0 / 11.0);
  p = fma(p, u, 1.0 / 9.0);
  p = fma(p, u, 1.0 / 7.0);
  p = fma(p, u, 1.0 / 5.0);
  p = fma(p, u, 1.0 / 3.0);
  const double tail = fma(sh * u, p, sl);  // s^3/3 + s^5/5 + ... + the quotient's tail
  const double A = 2.0 * sh, B = 2.0 * tail;
  const double ed = (double)e;
  const double L1 = ed * 0x1.62e42fefa38p-1;            // e * ln2_hi: exact (ln2_hi has 11 trailing zero bits)
  const double L2 = ed * 0x1.ef35793c7673p-45;           // e * ln2_lo
  // (L1 + A) as head + tail (two-sum), then everything small
  const double S = L1 + A;
  const double bb = S - L1;
  const double Sl = (L1 - (S - bb)) + (A - bb);
  return S + (Sl + (B + L2));
}

struct RandnArgs {
  uint32_t* key;         // [624] the current block (device copy: written by mt_window_kernel from its argument)
  uint32_t* win;         // [WIN_WORDS]
  const uint16_t* taps;  // [S - 1][TAP_STRIDE]: the set exponents e of G_m, m = 1 .., as four lists by c = (1 + e) mod 4, each padded with TAP_PAD + 3 - c
  const int* n_taps;     // [S - 1][4]: entries of the four lists (multiples of TAP_U)
  unsigned long long* counts;  // [2 S + 1]: accepted attempts per half stream (m, pass), then (after the scan) exclusive prefix sums; [2 S] = total
  uint32_t* masks;       // [2 S][NT]: which of a thread's attempts of a half stream were accepted (bit c - c0), from the count pass
  uint32_t* words;       // [S J_WORDS + 624]: every stream's blocks as one sequence (word w of the stream sequence that starts at the key's word 0)
  int S;
  int pos0;               // NumPy's position inside the current block (0 .. 624)
  long long attempts;     // attempts the streams cover: t < attempts
  long long pairs;        // accepted attempts to consume
  long long rest;         // values to write (2 pairs or 2 pairs - 1)
  double* out;            // first value's address
  // hand-back (written by the workgroup that holds the last pair)
  // (round 6: these point into pinned HOST memory -- the kernels store there directly and the host reads it after its one
  // wait for the stream; round 5 copied a key up, cleared two words and copied four pieces back: six copy launches, ~25 us)
  uint32_t* end_key;      // [624]
  long long* end_info;    // [0] word index behind the last attempt (pos0 + 4 (t_end + 1)), [1] 1 = written
  double* end_vals;       // x1, x2, r2 of the last attempt
  unsigned long long* end_total;  // accepted attempts in all streams (the scan's last entry)
};

struct KeyArg {
  uint32_t k[MT_N];  // NumPy's current block, as a kernel argument (2 496 B)
};
__global__ __launch_bounds__(NT) void mt_window_kernel(RandnArgs a, KeyArg key) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
  for (int i = threadIdx.x; i < MT_N; i += NT) {
    sm[i] = key.k[i];
    a.key[i] = key.k[i];  // stream 0 starts from it (mt_stream_kernel, the next launch)
  }
  __syncthreads();
  mt_extend_lds(sm, WIN_WORDS);
  for (int i = threadIdx.x; i < WIN_WORDS; i += NT) a.win[i] = sm[i];
}

// The stream's first block into sm[0 .. 624): the key itself (m = 0) or the jump.
__device__ __forceinline__ void stream_start(const RandnArgs& a, int m, uint32_t* sm) {
  const int tid = threadIdx.x;
  if (m == 0) {
    for (int i = tid; i < MT_N; i += NT) sm[i] = a.key[i];
    __syncthreads();
    return;
  }
  for (int i = tid; i < WIN_WORDS; i += NT) sm[i] = a.win[i];
  for (int i = WIN_WORDS + tid; i < TAP_OFF; i += NT) sm[i] = 0;  // what the padding taps read
  __syncthreads();
  // x_(m J + j) = XOR_{i : g_i} x_(1 + i + j), j < 624.  The polynomial comes as the lists of its set exponents (16-bit,
  // staged in LDS; per step a thread reads 16 of them, broadcast, and has 16 window reads in flight).
  // Round 6: wide window reads (first eight bytes: 141 -> 108 us per jump; then sixteen).  Rounds 5's loop read one word per exponent and output word (two words j, j + 312
  // per thread): 141 us per jump where the LDS-bandwidth floor of its 25 MB is 81 -- 97 500 ds_read_b32 per jump and SIMD
  // group at ~3.5 cycles each: bound by the LDS instruction rate, not by its bytes.  A thread now owns two ADJACENT output
  // words and reads them with one ds_read_b64, which wants an even word address: for an odd exponent e the pair
  // (x_(1+e+j), x_(2+e+j)), j = 2 j2, is aligned and serves outputs (j, j + 1); for an even exponent the aligned pair is
  // (x_(e+j), x_(1+e+j)) and serves outputs (j - 1, j) -- so the exponents come as two lists (odd: "A", even: "B", each
  // padded to a multiple of 16 with an exponent of its own parity whose words are zeros), the B pairs are kept apart and
  // shifted by one when the partial sums are put together, and a 313th thread of each group covers output 623 of the B
  // pairs (its A pair, and output -1 of thread 0's B pair, fall outside and are dropped).
  // (Measured on the way, round 5: walking the bits of the polynomial's words, one dependent read at a time: 400 us per
  // jump; the list read from memory inside the loop: 230; from LDS, one word per thread on 10 waves: 150; two words per
  // thread on 15 waves: 141.)
  // (then sixteen-byte reads: four adjacent output words per thread, four lists by (1 + e) mod 4)
  int n_c[4], off_c[4];
  {
    int o = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      n_c[c] = a.n_taps[4 * (m - 1) + c];  // multiples of TAP_U
      off_c[c] = o;
      o += n_c[c];
    }
    // the lists into LDS first (40 KB; read from memory inside the loop, every step waited ~0.3 us for its 32 bytes)
    const uint32_t* tg = (const uint32_t*)(a.taps + (size_t)(m - 1) * TAP_STRIDE);
    for (int i = tid; i < o / 2; i += NT) sm[TAP_OFF + i] = tg[i];
  }
  __syncthreads();
  // Class c = (1 + e) mod 4: the aligned quad at word 1 + e - c + 4 j4 holds outputs 4 j4 - c .. 4 j4 - c + 3; j4 = 0 .. 156
  // covers outputs 0 .. 623 for every c (what falls outside is dropped when the partial sums are put together).
  constexpr int GRP = MT_N / 4 + 1, NGRP = 6;  // six groups of 157 threads, a sixth of every list's steps each
  static_assert(GRP * NGRP <= NT, "the jump's groups fit the workgroup");
  const int grp = tid / GRP, j4 = tid - grp * GRP;
  uint4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = make_uint4(0u, 0u, 0u, 0u);
  if (grp < NGRP) {
    const uint4* tl = (const uint4*)(sm + TAP_OFF);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t* base = sm + 1 - c + 4 * j4;  // + e: a multiple of four words for every exponent of the class
      const int steps = n_c[c] / TAP_U, t0 = off_c[c] / TAP_U;
      for (int t = grp; t < steps; t += NGRP) {
        const uint4 p0 = tl[2 * (t0 + t)], p1 = tl[2 * (t0 + t) + 1];  // 16 exponents, the same for every thread (broadcast reads)
        const uint32_t pw[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        uint4 v[TAP_U];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          v[2 * q] = *(const uint4*)(base + (pw[q] & 0xFFFFu));
          v[2 * q + 1] = *(const uint4*)(base + (pw[q] >> 16));
        }
#pragma unroll
        for (int q = 0; q < TAP_U; ++q) {
          acc[c].x ^= v[q].x;
          acc[c].y ^= v[q].y;
          acc[c].z ^= v[q].z;
          acc[c].w ^= v[q].w;
        }
      }
    }
  }
  __syncthreads();  // every read of the window is done: the partial sums go over it (behind the 624 words of the result)
  uint32_t* part = sm + 1024;  // [4 classes][NGRP][4 GRP]: entry k of class c = output k - c
  if (grp < NGRP) {
#pragma unroll
    for (int c = 0; c < 4; ++c) *(uint4*)(part + ((c * NGRP + grp) * GRP + j4) * 4) = acc[c];
  }
  __syncthreads();
  if (tid < MT_N) {
    uint32_t x = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int g = 0; g < NGRP; ++g) x ^= part[(c * NGRP + g) * GRP * 4 + tid + c];
    sm[tid] = x;
  }
  __syncthreads();
}

// Attempts that start in stream m: t in [t_lo, t_hi) with pos0 + 4 t in [m J, (m + 1) J), t < attempts.
__device__ __forceinline__ void stream_attempts(const RandnArgs& a, int m, long long& t_lo, long long& t_hi) {
  const long long w_lo = (long long)m * J_WORDS, w_hi = w_lo + J_WORDS;
  t_lo = w_lo <= a.pos0 ? 0 : (w_lo - a.pos0 + 3) / 4;
  t_hi = w_hi <= a.pos0 ? 0 : (w_hi - a.pos0 + 3) / 4;
  if (t_hi > a.attempts) t_hi = a.attempts;
  if (t_lo > t_hi) t_lo = t_hi;
}

constexpr int N_PASS = BLK_PER_STREAM / BLK_PER_PASS;  // half streams per stream
static_assert(N_PASS == 2, "two passes per stream");

// attempts of half stream (m, pass): first word in [w0, w0 + PB * 624) of the sequence, inside the stream's range [t_lo, t_hi)
__device__ __forceinline__ void pass_attempts(const RandnArgs& a, int m, int pass, long long t_lo, long long t_hi, long long& p_lo,
                                              long long& p_hi) {
  const long long w0 = (long long)m * J_WORDS + (long long)pass * BLK_PER_PASS * MT_N, w1 = w0 + (long long)BLK_PER_PASS * MT_N;
  p_lo = w0 <= a.pos0 ? 0 : (w0 - a.pos0 + 3) / 4;
  p_hi = w1 <= a.pos0 ? 0 : (w1 - a.pos0 + 3) / 4;
  p_lo = p_lo < t_lo ? t_lo : p_lo;
  p_hi = p_hi > t_hi ? t_hi : p_hi;
}

// Workgroup m: the jump to its first block, its 80 blocks by the recurrence (40 at a time in LDS), every block's words
// to memory, and the number of accepted attempts of each of its two halves.
// (Measured and not kept, round 6: the jump, the recurrence on four-wave workgroups and the counts as three kernels --
// 141 + 50 + 17.5 us against this kernel's 187.)
__global__ __launch_bounds__(NT) void mt_stream_kernel(RandnArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
  __shared__ unsigned long long s_red[NT / 64];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long long t_lo, t_hi;
  stream_attempts(a, m, t_lo, t_hi);
  stream_start(a, m, sm);
  for (int pass = 0; pass < N_PASS; ++pass) {
    // blocks [pass * PB, pass * PB + PB] of the stream (one more than the pass owns: an attempt may straddle into it)
    if (pass > 0) {
      for (int i = tid; i < MT_N; i += NT) sm[i] = sm[BLK_PER_PASS * MT_N + i];
      __syncthreads();
    }
    mt_extend_lds(sm, (BLK_PER_PASS + 1) * MT_N);
    // the pass's own blocks to memory (the last stream's last pass: the block behind them too -- an attempt may end there)
    const long long w0 = (long long)m * J_WORDS + (long long)pass * BLK_PER_PASS * MT_N;
    {
      const int n_w = (BLK_PER_PASS + ((m == a.S - 1 && pass == N_PASS - 1) ? 1 : 0)) * MT_N;
      uint32_t* dst = a.words + w0;
      for (int i = tid; i < n_w; i += NT) dst[i] = sm[i];
    }
    long long p_lo, p_hi;
    pass_attempts(a, m, pass, t_lo, t_hi, p_lo, p_hi);
    const int n_att = p_hi > p_lo ? (int)(p_hi - p_lo) : 0;
    // a contiguous chunk of attempts per thread
    const int per = (n_att + NT - 1) / NT;
    const int c0 = min(tid * per, n_att), c1 = min(c0 + per, n_att);
    int cnt = 0;
    uint32_t acc_mask = 0;  // bit c - c0: attempt c was accepted (per <= MAX_PER: the values pass does not try them again)
    for (int c = c0; c < c1; ++c) {
      const long long t = p_lo + c;
      const uint32_t* u = sm + (int)(a.pos0 + 4 * t - w0);
      double x1, x2, r2;
      const bool ok = mt_attempt(u, x1, x2, r2);
      cnt += ok ? 1 : 0;
      acc_mask |= (ok ? 1u : 0u) << (c - c0);
    }
    a.masks[((size_t)m * N_PASS + pass) * NT + tid] = acc_mask;
    unsigned long long v = (unsigned long long)cnt;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    if (tid == 0) {
      unsigned long long total = 0;
      for (int q = 0; q < NT / 64; ++q) total += s_red[q];
      a.counts[(size_t)m * N_PASS + pass] = total;
    }
    __syncthreads();
  }
}

// Workgroup u = (m, pass): the values of the accepted attempts of that half stream, from the words in memory.
constexpr int NTV = 1024;
constexpr int MAX_PER = (BLK_PER_PASS * (MT_N / 4) + 1 + NTV - 1) / NTV;  // attempts per thread and half stream: 7
static_assert(NTV == NT && MAX_PER <= 32, "the count pass and the values pass cut a half stream's attempts the same way; a bit per attempt");
__global__ __launch_bounds__(NTV) void mt_values_kernel(RandnArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];  // the half stream's words (+ the block behind them): PASS_WORDS
  __shared__ unsigned long long s_red[NTV / 64];
  __shared__ unsigned long long s_scan[NTV / 64];
  __shared__ long long s_end_blk;
  const int u_ = blockIdx.x, m = u_ / N_PASS, pass = u_ - m * N_PASS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long base_pair = a.counts[u_];  // accepted attempts in front of this half stream
  if ((long long)base_pair >= a.pairs) return;       // the request ends in front of it
  long long t_lo, t_hi, p_lo, p_hi;
  stream_attempts(a, m, t_lo, t_hi);
  pass_attempts(a, m, pass, t_lo, t_hi, p_lo, p_hi);
  const int n_att = p_hi > p_lo ? (int)(p_hi - p_lo) : 0;
  if (n_att == 0) return;
  if (tid == 0) s_end_blk = -1;
  // the words through LDS, coalesced (a thread's attempts are 4 * per CONSECUTIVE words: read straight from memory every
  // load instruction touched 64 different lines and the working set of a CU's waves did not fit its L1 -- 77 us, as long as
  // the second recurrence pass this kernel replaces)
  const long long w0 = (long long)m * J_WORDS + (long long)pass * BLK_PER_PASS * MT_N;
  {
    const uint4* src = (const uint4*)(a.words + w0);  // (w0 is a multiple of 624 words = 2 496 bytes: 16-byte aligned)
    uint4* dst = (uint4*)sm;
    for (int i = tid; i < PASS_WORDS / 4; i += NTV) dst[i] = src[i];
  }
  __syncthreads();
  // a contiguous chunk of attempts per thread (ranks then follow attempt order)
  const int per = (n_att + NTV - 1) / NTV;
  const int c0 = min(tid * per, n_att), c1 = min(c0 + per, n_att);
  const uint32_t* wbase = sm + (int)(a.pos0 + 4 * p_lo - w0);  // attempt c of the pass reads wbase[4 c .. 4 c + 4)
  // which of the thread's attempts were accepted: the count pass (mt_stream_kernel, the same chunking) left a bit each
  const uint32_t acc_mask = a.masks[(size_t)u_ * NTV + tid] & (c1 > c0 ? (0xFFFFFFFFu >> (32 - (c1 - c0))) : 0u);
  const int cnt = __popc(acc_mask);
  // exclusive scan of the per-thread counts over the workgroup
  unsigned long long v = (unsigned long long)cnt, incl = v;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s_red[wave] = incl;
  __syncthreads();
  if (tid == 0) {
    unsigned long long run = 0;
    for (int q = 0; q < NTV / 64; ++q) {
      s_scan[q] = run;
      run += s_red[q];
    }
  }
  __syncthreads();
  unsigned long long p = base_pair + s_scan[wave] + (incl - v);
  for (int c = c0; c < c1; ++c) {
    if (!((acc_mask >> (c - c0)) & 1u)) continue;
    const long long t = p_lo + c;
    double x1, x2, r2;
    (void)mt_attempt(wbase + 4 * (long long)c, x1, x2, r2);  // (accepted: the bit says so; its numbers)
    if ((long long)p < a.pairs) {
      const double f = sqrt(-2.0 * mt_log_dd(r2) / r2);  // (division and square root are correctly rounded on the device too)
      const long long o = 2 * (long long)p;
      a.out[o] = f * x2;
      if (o + 1 < a.rest) a.out[o + 1] = f * x1;
      if ((long long)p == a.pairs - 1) {  // the request's last pair: what NumPy's state ends at
        a.end_vals[0] = x1;
        a.end_vals[1] = x2;
        a.end_vals[2] = r2;
        a.end_info[0] = a.pos0 + 4 * (t + 1);
        s_end_blk = (a.pos0 + 4 * t + 3) / MT_N;  // the block of the sequence holding the last word read
      }
    }
    ++p;
  }
  __syncthreads();
  if (s_end_blk >= 0) {
    const uint32_t* src = a.words + s_end_blk * MT_N;  // (from memory: the block may be the one behind this half stream's)
    for (int i = tid; i < MT_N; i += NTV) a.end_key[i] = src[i];
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      a.end_info[1] = 1;
    }
  }
}

__global__ __launch_bounds__(256) void mt_scan_kernel(RandnArgs a) {
  // exclusive prefix sum of the 2 S counts by one workgroup: a contiguous chunk per thread, then a scan of the 256 chunk sums
  __shared__ unsigned long long s_w[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = N_PASS * a.S;
  const int per = (n + 255) / 256;
  const int c0 = min(tid * per, n), c1 = min(c0 + per, n);
  unsigned long long sum = 0;
  for (int m = c0; m < c1; ++m) sum += a.counts[m];
  unsigned long long incl = sum;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  unsigned long long run = incl - sum;
  for (int q = 0; q < wave; ++q) run += s_w[q];
  for (int m = c0; m < c1; ++m) {
    const unsigned long long c = a.counts[m];
    a.counts[m] = run;
    run += c;
  }
  if (tid == 255) {
    a.counts[n] = run;
    *a.end_total = run;
  }
}

struct RandnDev {
  uint32_t* d_key = nullptr;
  uint32_t* d_win = nullptr;
  uint16_t* d_taps = nullptr;
  int* d_ntaps = nullptr;
  int poly_count = 0;  // polynomials on the device (G_1 .. G_count)
  unsigned long long* d_counts = nullptr;
  // the streams' words of the last two calls, alternating: a call whose incoming block is the one the call before handed
  // back finds its window -- the 33 blocks from there on -- in that call's sequence (the streams cover ~1 % more than a
  // request consumes) and does not launch mt_window_kernel, the one sequential kernel of the pass (13.6 us)
  uint32_t* d_words[2] = {nullptr, nullptr};
  int cur = 0;               // the buffer the last successful call wrote
  bool prev_valid = false;
  long long prev_blk = 0;    // block of that sequence the handed-back key is
  int prev_S = 0;
  std::vector<uint32_t> prev_key;
  uint32_t* d_masks = nullptr;
  int cap_S = 0;
  uint32_t* h_stage = nullptr;  // pinned, written by the kernels: end_key (624) | end_info (2 x 8 B) | end_vals (3 x 8 B) | total (8 B)
  uint32_t* hd_stage = nullptr;  // its device-side address
  std::vector<uint32_t> polys;  // host copy
  double first_val = 0.0;       // NumPy's cached second value on its way to the device
};

RandnDev* randn_of(vbmc_ctx* ctx) {
  if (!ctx->randn_dev) ctx->randn_dev = new RandnDev();
  return (RandnDev*)ctx->randn_dev;
}

}  // namespace

void randn_dev_free(vbmc_ctx* ctx) {
  RandnDev* r = (RandnDev*)ctx->randn_dev;
  if (!r) return;
  void* bufs[] = {r->d_key, r->d_win, r->d_taps, r->d_ntaps, r->d_counts, r->d_words[0], r->d_words[1], r->d_masks};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (r->h_stage) (void)hipHostFree(r->h_stage);
  delete r;
  ctx->randn_dev = nullptr;
}

// n values of the stream into d_out (device memory), state advanced.  VBMC_W_NOT_FUSED: not applicable (the caller
// takes the host generator).
int randn_device(vbmc_ctx* ctx, uint32_t* key, int* pos, int* has_gauss, double* gauss, double* d_out, int64_t n) {
  if (!key || !pos || !has_gauss || !gauss || n < 0 || *pos < 0 || *pos > MT_N) return VBMC_E_ARG;
  if (n == 0) return VBMC_OK;
  RandnDev* r = randn_of(ctx);
  hipStream_t sm = ctx->stream;
  // The caller's state (key, pos, has_gauss, gauss) is written ONLY on success, at the end: a HIP error or the
  // margin fall-back below leaves NumPy's stream exactly where it was (the cached value used to be
  // consumed up front, so a later failure handed back a state the reference could never be in).
  int64_t produced = 0;
  if (*has_gauss) {
    r->first_val = *gauss;  // (a member: the copy is asynchronous)
    HIP_TRY(ctx, hipMemcpyAsync(d_out, &r->first_val, sizeof(double), hipMemcpyHostToDevice, sm));
    HIP_TRY(ctx, hipStreamSynchronize(sm));
    produced = 1;
  }
  const int64_t rest = n - produced;
  if (rest == 0) {
    *has_gauss = 0;
    *gauss = 0.0;
    return VBMC_OK;
  }
  const int64_t pairs = (rest + 1) / 2;
  const double expect = (double)pairs / 0.7853981633974483;
  const int64_t attempts_need = (int64_t)(expect + 6.0 * std::sqrt(expect) + 64.0);
  const int pos0 = *pos;
  const int64_t words = (int64_t)pos0 + 4 * attempts_need + 4;
  const int64_t S64 = (words + J_WORDS - 1) / J_WORDS;
  if (S64 > 16384) return VBMC_W_NOT_FUSED;  // (> 8e8 words: not this path's size)
  const int S = (int)S64;
  const int64_t attempts = ((int64_t)S * J_WORDS - pos0 - 3) / 4;  // every attempt whose four words the streams hold... of stream S-1's extra block too
  // ---- polynomials (once per process and size) ----
  if (r->poly_count < S - 1) {
    const int want = std::max(S - 1, 255);
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    int nthreads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u);
    if (!mtj::jump_polys((uint64_t)J_WORDS, want, nthreads, r->polys)) return vbmc_fail(ctx, VBMC_E_HIP, "randn: MT19937's characteristic polynomial was not found");
    // the polynomials as lists of their set exponents
    std::vector<uint16_t> taps((size_t)want * TAP_STRIDE, (uint16_t)TAP_PAD);
    std::vector<int> ntaps(4 * (size_t)want);
    for (int m = 0; m < want; ++m) {
      const uint32_t* g = r->polys.data() + (size_t)m * MT_N;
      uint16_t* tp = taps.data() + (size_t)m * TAP_STRIDE;
      int n = 0;
      for (int c = 0; c < 4; ++c) {  // class c: (1 + e) mod 4 == c, i.e. e = c - 1 (mod 4); each list padded with its own zero exponent
        const int n0 = n;
        for (int i = (c + 3) & 3; i < mtj::DEG; i += 4)
          if ((g[i >> 5] >> (i & 31)) & 1u) tp[n++] = (uint16_t)i;
        while ((n - n0) % TAP_U) tp[n++] = (uint16_t)(TAP_PAD + ((c + 3) & 3));
        ntaps[4 * (size_t)m + c] = n - n0;
      }
    }
    if (r->d_taps) HIP_TRY(ctx, hipFree(r->d_taps));
    if (r->d_ntaps) HIP_TRY(ctx, hipFree(r->d_ntaps));
    r->d_taps = nullptr;
    r->d_ntaps = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&r->d_taps, sizeof(uint16_t) * taps.size()));
    HIP_TRY(ctx, hipMalloc((void**)&r->d_ntaps, sizeof(int) * 4 * (size_t)want));
    HIP_TRY(ctx, hipMemcpy(r->d_taps, taps.data(), sizeof(uint16_t) * taps.size(), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(r->d_ntaps, ntaps.data(), sizeof(int) * 4 * (size_t)want, hipMemcpyHostToDevice));
    r->poly_count = want;
  }
  if (!r->d_key) {
    HIP_TRY(ctx, hipMalloc((void**)&r->d_key, sizeof(uint32_t) * MT_N));
    HIP_TRY(ctx, hipMalloc((void**)&r->d_win, sizeof(uint32_t) * WIN_WORDS));
    HIP_TRY(ctx, hipHostMalloc((void**)&r->h_stage, sizeof(uint32_t) * MT_N + 64, hipHostMallocDefault));
    HIP_TRY(ctx, hipHostGetDevicePointer((void**)&r->hd_stage, r->h_stage, 0));
    const size_t lds = sizeof(uint32_t) * LDS_WORDS;
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)mt_window_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)mt_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)mt_values_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(uint32_t) * PASS_WORDS)));
  }
  if (r->cap_S < S) {
    if (r->d_counts) HIP_TRY(ctx, hipFree(r->d_counts));
    for (int q = 0; q < 2; ++q) {
      if (r->d_words[q]) HIP_TRY(ctx, hipFree(r->d_words[q]));
      r->d_words[q] = nullptr;
    }
    if (r->d_masks) HIP_TRY(ctx, hipFree(r->d_masks));
    r->d_counts = nullptr;
    r->d_masks = nullptr;
    r->cap_S = 0;
    r->prev_valid = false;
    HIP_TRY(ctx, hipMalloc((void**)&r->d_counts, sizeof(unsigned long long) * ((size_t)N_PASS * S + 1)));
    for (int q = 0; q < 2; ++q) HIP_TRY(ctx, hipMalloc((void**)&r->d_words[q], sizeof(uint32_t) * ((size_t)S * J_WORDS + 2 * MT_N)));
    HIP_TRY(ctx, hipMalloc((void**)&r->d_masks, sizeof(uint32_t) * (size_t)N_PASS * S * NT));
    r->cap_S = S;
  }
  // hand-back block in pinned memory (nothing of an earlier call is in flight: every call ends with a wait for the stream)
  uint32_t* h_end_key = r->h_stage;
  long long* h_info = (long long*)(r->h_stage + MT_N);
  double* h_vals = (double*)(h_info + 2);
  unsigned long long* h_total = (unsigned long long*)(h_vals + 3);
  h_info[0] = h_info[1] = 0;
  *h_total = 0;
  // the window: in the last call's sequence when this call continues where that one ended (see RandnDev), else computed
  const bool reuse = r->prev_valid && ctx->opt_randn_dev != 3 /* test hook: never */ && r->prev_key.size() == (size_t)MT_N &&
                     std::memcmp(key, r->prev_key.data(), sizeof(uint32_t) * MT_N) == 0 &&
                     r->prev_blk + WIN_WORDS / MT_N <= (long long)r->prev_S * BLK_PER_STREAM + 1;
  r->prev_valid = false;  // (until this call succeeds)
  RandnArgs a;
  if (reuse) {
    a.win = r->d_words[r->cur] + r->prev_blk * MT_N;
    a.key = a.win;
  } else {
    a.key = r->d_key;
    a.win = r->d_win;
  }
  a.taps = r->d_taps;
  a.n_taps = r->d_ntaps;
  a.counts = r->d_counts;
  a.words = r->d_words[r->cur ^ 1];
  a.masks = r->d_masks;
  a.S = S;
  a.pos0 = pos0;
  a.attempts = attempts;
  a.pairs = pairs;
  a.rest = rest;
  a.out = d_out + produced;
  a.end_key = r->hd_stage;
  a.end_info = (long long*)(r->hd_stage + MT_N);
  a.end_vals = (double*)(a.end_info + 2);
  a.end_total = (unsigned long long*)(a.end_vals + 3);
  const size_t lds = sizeof(uint32_t) * LDS_WORDS;
  if (!reuse) {
    KeyArg karg;
    std::memcpy(karg.k, key, sizeof(uint32_t) * MT_N);
    hipLaunchKernelGGL(mt_window_kernel, dim3(1), dim3(NT), lds, sm, a, karg);
  }
  ctx->randn_last_reused = reuse ? 1 : 0;
  hipLaunchKernelGGL(mt_stream_kernel, dim3(S), dim3(NT), lds, sm, a);
  hipLaunchKernelGGL(mt_scan_kernel, dim3(1), dim3(256), 0, sm, a);
  hipLaunchKernelGGL(mt_values_kernel, dim3(N_PASS * S), dim3(NTV), sizeof(uint32_t) * PASS_WORDS, sm, a);
  HIP_TRY(ctx, hipGetLastError());
  // hand-back: end block, word index, the last attempt's numbers, the total
  // (the kernels stored end block, word index, the last attempt's numbers and the total into the pinned block themselves)
  HIP_TRY(ctx, stream_wait(ctx));
  // fewer accepted attempts than pairs in the words the streams cover (the 6 sigma margin: ~1e-9 per call): nothing
  // of the caller's state has been touched -- the host generator takes the request over (vbmc_set_eps_numpy)
  if ((int64_t)*h_total < pairs || h_info[1] != 1 || ctx->opt_randn_dev == 2 /* test hook: as if the margin had been exceeded */)
    return VBMC_W_NOT_FUSED;
  const int64_t w = h_info[0];            // word index behind the final attempt, >= 4
  const int64_t b = (w - 1) / MT_N;       // the block holding the last word read
  if (b > 0) std::memcpy(key, h_end_key, sizeof(uint32_t) * MT_N);
  *pos = (int)(w - b * MT_N);
  r->cur ^= 1;  // this call's sequence: where the next call may find its window
  r->prev_key.assign(key, key + MT_N);
  r->prev_blk = b;
  r->prev_S = S;
  r->prev_valid = true;
  if (rest & 1) {
    // NumPy's cached second value, with the host's libm: bit-identical to what legacy_gauss would hold
    const double r2 = h_vals[2];
    const double f = std::sqrt(-2.0 * std::log(r2) / r2);
    *has_gauss = 1;
    *gauss = f * h_vals[0];
  } else {
    *has_gauss = 0;
    *gauss = 0.0;
  }
  return VBMC_OK;
}

extern "C" int vbmc_mt19937_randn_dev(vbmc_ctx* ctx, uint32_t* key, int* pos, int* has_gauss, double* gauss, double* out,
                                       int64_t n) {
  if (!ctx || (n > 0 && !out)) return VBMC_E_ARG;
  NEED_DEVICE(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_dev(ctx, &ctx->d_scratch, &ctx->d_scratch_cap, (size_t)n + 2);
  if (rc) return rc;
  rc = randn_device(ctx, key, pos, has_gauss, gauss, ctx->d_scratch, n);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(out, ctx->d_scratch, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  return VBMC_OK;
}

// whether the last device pass of this context took its window from the pass before (tests)
extern "C" int vbmc_randn_dev_info(const vbmc_ctx* ctx, int* window_reused) {
  if (!ctx || !window_reused) return VBMC_E_ARG;
  *window_reused = ctx->randn_last_reused;
  return VBMC_OK;
}

// host twin of the jump (CPU tests): key_out = the block that starts n_words words after key_in[0]
extern "C" int vbmc_mt_jump_host(const uint32_t* key_in, uint64_t n_words, uint32_t* key_out) {
  if (!key_in || !key_out || n_words < 1) return VBMC_E_ARG;
  return mtj::jump_host(key_in, n_words, key_out) ? VBMC_OK : VBMC_E_HIP;
}
// the polynomial chain the device uses, for the same tests: out[count][624]
extern "C" int vbmc_mt_jump_polys(uint64_t stride_words, int count, uint32_t* out) {
  if (!out || count < 1) return VBMC_E_ARG;
  std::vector<uint32_t> v;
  if (!mtj::jump_polys(stride_words, count, 8, v)) return VBMC_E_HIP;
  std::memcpy(out, v.data(), sizeof(uint32_t) * v.size());
  return VBMC_OK;
}
