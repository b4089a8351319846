// Round 3: candidates for the draw generator.
// The shipped generator spends ~240 VALU instructions per Box-Muller pair (PMC): Philox4x32-10 per
// pair with 53-bit uniforms, <= 1 ulp log / sqrt / sincospi, 64-bit index arithmetic with a walked
// position.  Candidates here: one Philox block per TWO pairs (32-bit uniforms for radius and angle),
// 7 or 10 rounds, a Box-Muller built for 32-bit inputs (integer quadrant, degree-5 polynomials good
// to ~1e-14, one Newton step where one suffices), thread = Philox block, 32-bit magic-number index
// arithmetic, and three kinds of stores (plain, write-through sc1, through an LDS transpose).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off ubench_gen2.hip -o ubench_gen2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../pyvbmc_amd/csrc/philox.h"

template <int R>
__device__ __forceinline__ Philox4 philox4x32_r(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 o; o.x[0] = c0; o.x[1] = c1; o.x[2] = c2; o.x[3] = c3;
  return o;
}

// Box-Muller from two 32-bit words: radius from u = (xr + 1/2) 2^-32 in (0,1), angle 2 pi xa 2^-32.
__device__ __forceinline__ void bm32(uint32_t xr, uint32_t xa, double& z0, double& z1) {
  const double f = (double)xr + 0.5;  // exact
  double m = __builtin_amdgcn_frexp_mant(f);
  int e = __builtin_amdgcn_frexp_exp(f);
  const bool lo = m < 0x1.6a09e667f3bcdp-1;
  m = __builtin_amdgcn_ldexp(m, lo ? 1 : 0);
  e = lo ? e - 1 : e;
  const double den = m + 1.0, num = m - 1.0;
  double r = __builtin_amdgcn_rcp(den);
  r = fma(fma(-den, r, 1.0), r, r);
  const double s = num * r, u = s * s;
  double p = 0x1.9192e478c4308p-4;
  p = fma(p, u, 0x1.c620ee6e2b4a3p-4);
  p = fma(p, u, 0x1.2494381ee5869p-3);
  p = fma(p, u, 0x1.9999962c06032p-3);
  p = fma(p, u, 0x1.555555567148cp-2);
  p = fma(p, u, 0x1.fffffffffff12p-1);
  const double lm2 = s * p;  // ln(m) / 2
  const double x2 = fma(-4.0, lm2, (double)(32 - e) * 0x1.62e42fefa39efp+0);  // -2 ln u
  // sqrt(x2), x2 in [2.3e-10, 45.8]: rsq + one coupled Newton step
  const double y = __builtin_amdgcn_rsq(x2);
  double g = x2 * y;
  const double h = 0.5 * y;
  g = fma(g, fma(-h, g, 0.5), g);
  // angle = pi (q/2 + 1/4 + rr), q = top two bits, rr = v 2^-33 in [-1/4, 1/4)
  const double v = (double)(int)((xa << 2) ^ 0x80000000u);
  const double w = v * v;
  double ps = -0x1.dd54805f3f706p-8 * 0x1p-363;  // sin(pi rr)/rr, coefficients scaled for the argument v
  ps = fma(ps, w, 0x1.5071ce4b47930p-4 * 0x1p-297);
  ps = fma(ps, w, -0x1.32d2c644adc0bp-1 * 0x1p-231);
  ps = fma(ps, w, 0x1.466bc67123fa1p+1 * 0x1p-165);
  ps = fma(ps, w, -0x1.4abbce6257a2ap+2 * 0x1p-99);
  ps = fma(ps, w, 0x1.921fb54442cfap+1 * 0x1p-33);
  const double S = ps * v;
  double pc = -0x1.a0ee132c60c1fp-6 * 0x1p-330;
  pc = fma(pc, w, 0x1.e1e7ccccb387ap-3 * 0x1p-264);
  pc = fma(pc, w, -0x1.55d3ba300cd50p+0 * 0x1p-198);
  pc = fma(pc, w, 0x1.03c1f074ded21p+2 * 0x1p-132);
  pc = fma(pc, w, -0x1.3bd3cc9bd2c35p+2 * 0x1p-66);
  const double Cc = fma(pc, w, 0x1.ffffffffffe0bp-1);
  // cos(a + t) = ca C - sa S, sin(a + t) = sa C + ca S, a = (2q+1) pi/4: (ca, sa) = sqrt(1/2) (+,+),(-,+),(-,-),(+,-)
  const uint32_t ms = xa & 0x80000000u, mc = ((xa << 1) ^ xa) & 0x80000000u;
  auto flip = [](double d, uint32_t mask) { return __hiloint2double(__double2hiint(d) ^ (int)mask, __double2loint(d)); };
  const double gh = g * 0x1.6a09e667f3bcdp-1;
  z0 = gh * (flip(Cc, mc) - flip(S, ms));
  z1 = gh * (flip(Cc, ms) + flip(S, mc));
}

// the same values' formulas with the conversions done by integer operations (v_cvt / v_frexp / v_ldexp are
// quarter-rate): mantissa and exponent of 2 xr + 1 from a count of leading zeros and a 64-bit shift,
// integers -> double through the 2^52 trick
__device__ __forceinline__ void bm32b(uint32_t xr, uint32_t xa, double& z0, double& z1) {
  const int lz = __clz((int)xr);  // 32 for xr == 0
  const uint64_t X = ((uint64_t)xr << 1) | 1ull;
  const uint64_t M = X << (20 + lz);  // leading one at bit 52
  uint32_t mh = ((uint32_t)(M >> 32) & 0x000FFFFFu) | 0x3FF00000u;
  int k = 1 + lz;  // 33 - e2, e2 = 32 - lz: -2 ln u = -2 ln m + 2 ln2 (33 - e2)
  const bool big = mh >= 0x3FF6A09Eu;
  mh = big ? mh - 0x00100000u : mh;
  k = big ? k - 1 : k;
  const double m = __hiloint2double((int)mh, (int)(uint32_t)M);
  const double den = m + 1.0, num = m - 1.0;
  double r = __builtin_amdgcn_rcp(den);
  r = fma(fma(-den, r, 1.0), r, r);
  const double s = num * r, u = s * s;
  double p = 0x1.9192e478c4308p-4;
  p = fma(p, u, 0x1.c620ee6e2b4a3p-4);
  p = fma(p, u, 0x1.2494381ee5869p-3);
  p = fma(p, u, 0x1.9999962c06032p-3);
  p = fma(p, u, 0x1.555555567148cp-2);
  p = fma(p, u, 0x1.fffffffffff12p-1);
  const double lm2 = s * p;
  const double kd = __hiloint2double(0x43300000, k) - 0x1p52;
  const double x2 = fma(-4.0, lm2, kd * 0x1.62e42fefa39efp+0);
  const double y = __builtin_amdgcn_rsq(x2);
  double g = x2 * y;
  const double h = 0.5 * y;
  g = fma(g, fma(-h, g, 0.5), g);
  const double v = __hiloint2double(0x43300000, (int)(xa << 2)) - (0x1p52 + 0x1p31);
  const double w = v * v;
  double ps = -0x1.dd54805f3f706p-8 * 0x1p-363;
  ps = fma(ps, w, 0x1.5071ce4b47930p-4 * 0x1p-297);
  ps = fma(ps, w, -0x1.32d2c644adc0bp-1 * 0x1p-231);
  ps = fma(ps, w, 0x1.466bc67123fa1p+1 * 0x1p-165);
  ps = fma(ps, w, -0x1.4abbce6257a2ap+2 * 0x1p-99);
  ps = fma(ps, w, 0x1.921fb54442cfap+1 * 0x1p-33);
  const double S = ps * v;
  double pc = -0x1.a0ee132c60c1fp-6 * 0x1p-330;
  pc = fma(pc, w, 0x1.e1e7ccccb387ap-3 * 0x1p-264);
  pc = fma(pc, w, -0x1.55d3ba300cd50p+0 * 0x1p-198);
  pc = fma(pc, w, 0x1.03c1f074ded21p+2 * 0x1p-132);
  pc = fma(pc, w, -0x1.3bd3cc9bd2c35p+2 * 0x1p-66);
  const double Cc = fma(pc, w, 0x1.ffffffffffe0bp-1);
  const uint32_t ms = xa & 0x80000000u, mc = ((xa << 1) ^ xa) & 0x80000000u;
  auto flip = [](double d, uint32_t mask) { return __hiloint2double(__double2hiint(d) ^ (int)mask, __double2loint(d)); };
  const double gh = g * 0x1.6a09e667f3bcdp-1;
  z0 = gh * (flip(Cc, mc) - flip(S, ms));
  z1 = gh * (flip(Cc, ms) + flip(S, mc));
}

template <int R>
__device__ __forceinline__ void philox_normal_quad(uint64_t row, uint32_t blk, uint64_t seed, double (&z)[4]) {
  const Philox4 r = philox4x32_r<(R > 100 ? R - 100 : R)>((uint32_t)row, (uint32_t)(row >> 32), blk, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  if (R > 100) {
    bm32b(r.x[0], r.x[1], z[0], z[1]);
    bm32b(r.x[2], r.x[3], z[2], z[3]);
  } else {
    bm32(r.x[0], r.x[1], z[0], z[1]);
    bm32(r.x[2], r.x[3], z[2], z[3]);
  }
}

struct Gen2 {
  double* eps;
  int D, nb;            // nb = ceil(D / 4) Philox blocks per row
  uint32_t rows, n_half, row_begin;
  uint32_t items;       // K * rows * nb
  uint32_t nb_magic, rows_magic;  // ceil(2^32 / d): q = mulhi(t, magic) is t / d or t / d + 1 ... corrected below
  uint64_t seed;
};
__device__ __forceinline__ uint32_t div_magic(uint32_t t, uint32_t d, uint32_t magic) {
  uint32_t q = __umulhi(t, magic);  // magic = floor(2^32 / d): q in {t/d - 1, t/d} ... corrected upwards
  uint32_t rem = t - q * d;
  if (rem >= d) { ++q; rem -= d; }
  if (rem >= d) ++q;
  return q;
}
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16(double* p, double a, double b, int mode) {
  v2d v;
  v.x = a;
  v.y = b;
  if (mode == 1) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  } else if (mode == 2) {
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  } else {
    *reinterpret_cast<double2*>(p) = make_double2(a, b);
  }
}
// thread = Philox block (row, blk): up to four normals, two 16-byte stores
template <int R, int MODE, bool STORE>
__global__ __launch_bounds__(256) void k_quad(Gen2 g, double* sink) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= g.items) return;
  const uint32_t r = div_magic(t, (uint32_t)g.nb, g.nb_magic), b = t - r * (uint32_t)g.nb;
  const uint32_t j = div_magic(r, g.rows, g.rows_magic), i = r - j * g.rows;
  const uint64_t grow = (uint64_t)j * g.n_half + (g.row_begin + i);
  double z[4];
  philox_normal_quad<R>(grow, b, g.seed, z);
  if (STORE) {
    double* dst = g.eps + (size_t)r * g.D + 4 * b;
    store16(dst, z[0], z[1], MODE);
    if (4 * (int)b + 2 < g.D) store16(dst + 2, z[2], z[3], MODE);
  } else if (z[0] == 123.456 && z[3] == 1.0) {
    sink[0] = z[1] + z[2];
  }
}
// the same through an LDS image of the wave's contiguous span: every store instruction writes 512 contiguous bytes
template <int R>
__global__ __launch_bounds__(256) void k_quad_lds(Gen2 g) {
  __shared__ double img[4][64 * 4];
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t t0 = t - lane;  // the wave's first item
  auto off = [&](uint32_t tt) {  // element offset of item tt
    const uint32_t rr = div_magic(tt, (uint32_t)g.nb, g.nb_magic), bb = tt - rr * (uint32_t)g.nb;
    return (size_t)rr * g.D + 4 * bb;
  };
  const uint32_t tend = min(t0 + 64, g.items);
  if (t0 >= g.items) return;
  const size_t o0 = off(t0), o1 = tend == g.items ? (size_t)(g.items / g.nb) * g.D : off(tend);
  if (t < g.items) {
    const uint32_t r = div_magic(t, (uint32_t)g.nb, g.nb_magic), b = t - r * (uint32_t)g.nb;
    const uint32_t j = div_magic(r, g.rows, g.rows_magic), i = r - j * g.rows;
    const uint64_t grow = (uint64_t)j * g.n_half + (g.row_begin + i);
    double z[4];
    philox_normal_quad<R>(grow, b, g.seed, z);
    const int lo = (int)((size_t)r * g.D + 4 * b - o0);
    img[wave][lo] = z[0];
    img[wave][lo + 1] = z[1];
    if (4 * (int)b + 2 < g.D) { img[wave][lo + 2] = z[2]; img[wave][lo + 3] = z[3]; }
  }
  const int n = (int)(o1 - o0);
  for (int k = lane; k < n; k += 64) g.eps[o0 + k] = img[wave][k];
}
__global__ __launch_bounds__(256) void k_old(GenSlice g) { gen_slice_block(g, blockIdx.x, threadIdx.x); }
__global__ __launch_bounds__(256) void k_read(const double* eps, size_t n, double* sink) {  // a consumer sweep of the draws
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * 256) {
    const double2 v = reinterpret_cast<const double2*>(eps)[i];
    acc += v.x + v.y;
  }
  if (acc == 123.456) sink[0] = acc;
}

int main() {
  const int K = 50, D = 10;
  const uint32_t rows = 10000;
  const size_t n = (size_t)K * rows * D;
  double *eps, *sink;
  hipMalloc(&eps, sizeof(double) * n);
  hipMalloc(&sink, 64);
  Gen2 g;
  g.eps = eps; g.D = D; g.nb = (D + 3) / 4; g.rows = rows; g.n_half = rows; g.row_begin = 0; g.seed = 12345;
  g.items = K * rows * g.nb;
  g.nb_magic = (uint32_t)(0x100000000ull / g.nb);
  g.rows_magic = (uint32_t)(0x100000000ull / g.rows);
  GenSlice o;
  o.eps = eps; o.K = K; o.D = D; o.rows = rows; o.n_half = rows; o.row_begin = 0; o.seed = 12345; o.seed_add = nullptr;
  o.item_begin = 0; o.item_count = (int64_t)K * rows * 5; o.n_blocks = (int)((o.item_count + 255) / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.2f us per launch\n", name, ms * 1000 / 20);
  };
  const int nbq = (g.items + 255) / 256;
  GenSlice o8 = o;
  // (the eight-items-per-thread form was dropped from the library in round 4)
  o8.n_blocks = (int)((o.item_count + 2047) / 2048);
  time("old, 1 pair/thread", [&] { hipLaunchKernelGGL(k_old, dim3(o.n_blocks), dim3(256), 0, 0, o); });
  time("old, 8 pairs/thread deferred", [&] { hipLaunchKernelGGL(k_old, dim3(o8.n_blocks), dim3(256), 0, 0, o8); });
  time("quad r10 plain", [&] { hipLaunchKernelGGL((k_quad<10, 0, true>), dim3(nbq), dim3(256), 0, 0, g, sink); });
  time("quad r7 plain", [&] { hipLaunchKernelGGL((k_quad<7, 0, true>), dim3(nbq), dim3(256), 0, 0, g, sink); });
  time("quad r7 sc1 (write-through)", [&] { hipLaunchKernelGGL((k_quad<7, 1, true>), dim3(nbq), dim3(256), 0, 0, g, sink); });
  time("quad r7 nt", [&] { hipLaunchKernelGGL((k_quad<7, 2, true>), dim3(nbq), dim3(256), 0, 0, g, sink); });
  time("quad r7 LDS image", [&] { hipLaunchKernelGGL((k_quad_lds<7>), dim3(nbq), dim3(256), 0, 0, g); });
  time("quad r10 int-conv plain", [&] { hipLaunchKernelGGL((k_quad<110, 0, true>), dim3(nbq), dim3(256), 0, 0, g, sink); });
  time("quad r10 int-conv compute only", [&] { hipLaunchKernelGGL((k_quad<110, 0, false>), dim3(nbq), dim3(256), 0, 0, g, sink); });
  time("quad r10 compute only", [&] { hipLaunchKernelGGL((k_quad<10, 0, false>), dim3(nbq), dim3(256), 0, 0, g, sink); });
  time("quad r7 compute only", [&] { hipLaunchKernelGGL((k_quad<7, 0, false>), dim3(nbq), dim3(256), 0, 0, g, sink); });
  // generator followed by a consumer sweep (what the entropy kernel does with the draws): the pair's time
  time("quad r7 plain + read", [&] { hipLaunchKernelGGL((k_quad<7, 0, true>), dim3(nbq), dim3(256), 0, 0, g, sink);
                                     hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, eps, n, sink); });
  time("quad r7 sc1 + read", [&] { hipLaunchKernelGGL((k_quad<7, 1, true>), dim3(nbq), dim3(256), 0, 0, g, sink);
                                   hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, eps, n, sink); });
  time("quad r7 nt + read", [&] { hipLaunchKernelGGL((k_quad<7, 2, true>), dim3(nbq), dim3(256), 0, 0, g, sink);
                                  hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, eps, n, sink); });
  time("read alone", [&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, eps, n, sink); });

  // accuracy and moments of the new pair against libm on the same integers
  hipLaunchKernelGGL((k_quad<7, 0, true>), dim3(nbq), dim3(256), 0, 0, g, sink);
  hipDeviceSynchronize();
  std::vector<double> h(n);
  hipMemcpy(h.data(), eps, sizeof(double) * n, hipMemcpyDeviceToHost);
  double s1 = 0, s2 = 0, s4 = 0, mx = 0;
  for (size_t i = 0; i < n; ++i) { const double v = h[i]; s1 += v; s2 += v * v; s4 += v * v * v * v; mx = fmax(mx, fabs(v)); }
  printf("moments of %zu normals: mean %.3e  var %.6f  kurt %.5f  max |z| %.3f\n", n, s1 / n, s2 / n, (s4 / n) / ((s2 / n) * (s2 / n)), mx);
  double worst = 0;
  for (uint32_t row = 0; row < 2000; ++row)
    for (uint32_t b = 0; b < (uint32_t)g.nb; ++b) {
      Philox4 r;
      {
        uint32_t c0 = row, c1 = 0, c2 = b, c3 = 0, k0 = (uint32_t)g.seed, k1 = (uint32_t)(g.seed >> 32);
        for (int q = 0; q < 7; ++q) {
          const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
          const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
          c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        r.x[0] = c0; r.x[1] = c1; r.x[2] = c2; r.x[3] = c3;
      }
      for (int pr = 0; pr < 2 && 4 * (int)b + 2 * pr < D; ++pr) {
        const double u = ((double)r.x[2 * pr] + 0.5) * 0x1p-32, ang = 2.0 * M_PI * ((double)r.x[2 * pr + 1] * 0x1p-32);
        const double rad = sqrt(-2.0 * log(u));
        const double a0 = rad * cos(ang), a1 = rad * sin(ang);
        const double d0 = h[(size_t)row * D + 4 * b + 2 * pr], d1 = h[(size_t)row * D + 4 * b + 2 * pr + 1];
        worst = fmax(worst, fmax(fabs(d0 - a0), fabs(d1 - a1)));
      }
    }
  printf("max |device - libm| over 2000 rows: %.3e\n", worst);
  {
    std::vector<double> h2(n);
    hipLaunchKernelGGL((k_quad<107, 0, true>), dim3(nbq), dim3(256), 0, 0, g, sink);
    hipDeviceSynchronize();
    hipMemcpy(h2.data(), eps, sizeof(double) * n, hipMemcpyDeviceToHost);
    double dmax = 0;
    for (size_t i = 0; i < n; ++i) dmax = fmax(dmax, fabs(h2[i] - h[i]));
    printf("max |int-conv - cvt| over all %zu normals: %.3e\n", n, dmax);
  }
  return 0;
}
