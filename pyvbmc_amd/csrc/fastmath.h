// float64 elementary functions tailored to the entropy kernels (gfx950).
//
// The Monte-Carlo entropy spends most of its issue slots in exp(): ocml's exp() costs
// ~110 cycles per wave (tools/ubench_fp64.hip), log() ~370.  These versions drop the
// generality the kernels do not need (no NaN/Inf plumbing, known argument ranges) and
// keep full double accuracy (<= 1-2 ulp; coefficients from tools/fit_polys.py, a
// high-precision Chebyshev-node fit).
#pragma once
#include <hip/hip_runtime.h>

namespace fm {

// 2^x for finite x <= ~1000; large negative x underflows smoothly to 0: v_cvt_i32_f64
// saturates, so n = INT_MIN for x < -2^31 and v_ldexp_f64 then returns 0 (p is in
// [0.7, 1.42]); no clamp needed.
// rint + exact remainder + degree-11 polynomial on [-1/2,1/2] + v_ldexp_f64.
__device__ __forceinline__ double exp2_fast(double x) {
  const double t = __builtin_rint(x);
  const double f = x - t;  // exact, |f| <= 1/2
  const int n = (int)t;
  double p = 0x1.e9ec1fcb69a7fp-32;
  p = fma(p, f, 0x1.e6228acd1c6e5p-28);
  p = fma(p, f, 0x1.b524ebd13a55fp-24);
  p = fma(p, f, 0x1.62bfc2c86d700p-20);
  p = fma(p, f, 0x1.ffcbfc6da6ed1p-17);
  p = fma(p, f, 0x1.430913112c61bp-13);
  p = fma(p, f, 0x1.5d87fe78a3f9cp-10);
  p = fma(p, f, 0x1.3b2ab6fb9f1a5p-7);
  p = fma(p, f, 0x1.c6b08d704a0c6p-5);
  p = fma(p, f, 0x1.ebfbdff82c5aep-3);
  p = fma(p, f, 0x1.62e42fefa39efp-1);
  p = fma(p, f, 1.0);
  return __builtin_amdgcn_ldexp(p, n);
}

// The entropy kernels' own 2^f on |f| <= 1/2 (entropy_ws.hip, entropy_mfma.hip): 1 + f (c0 + f (c1 + ...)), degree 8,
// max relative error 1.07e-12 in float64 Horner evaluation (tools/fit_polys.py).  Their densities do not need
// exp2_fast's 1.6e-16: the Monte-Carlo entropy is an average of log sum_k w_k N_k over 10^5 .. 10^7 draws, compared
// with the reference at 1e-10 (tests; BASELINE's bar is 1e-6) -- a relative error e in every density moves log q by at
// most e, H by at most 1.1e-12 absolute, its gradients by 2e times their terms' cancellation; measured against the
// float64 oracle at BASELINE config 3: H 1.2e-15, gradients 1.6e-15 relative, against 7.0e-16 / 1.5e-15 with degree
// 10 (the interpolation error equi-oscillates around zero in f and averages out over the draws).  Rounds 1-3 used degree 10 (4.1e-16); each degree is one FMA of the ~70 instructions per evaluated
// (pair, component): 73.7 -> 70.2 us per launch at config 3 (profiles/r04_exp2_degree.md).
// The guarantee without any averaging (the same kernels serve calls of a few rows per component): every density within 1.07e-12
// relative, hence log q within 1.1e-12 and H within 1.1e-12 ABSOLUTE, gradients within 2.2e-12 of their terms' scale; the fused
// optimiser loop (adam_fused.hip) keeps exp2_fast, so the two forms of the loop agree to ~1e-12, not to rounding -- tests on
// small-sample shapes assert at 1e-10 or looser.
#define VBMC_ENT_EXP2_N 8
#define VBMC_ENT_EXP2_COEFFS                                                                                        \
  {0x1.62e42fef84cf0p-1, 0x1.ebfbdff823cedp-3, 0x1.c6b08dd6fd234p-5, 0x1.3b2ab7181b755p-7, 0x1.5d8745a728441p-10, \
   0x1.4308ac85aa947p-13, 0x1.00dc4a532fb8ep-16, 0x1.63d136366db24p-20}

// 1/x to ~1 ulp: v_rcp_f64 + two Newton steps.
__device__ __forceinline__ double rcp_fast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// 1/sqrt(x) for normal x > 0 to ~1 ulp: v_rsq_f64 + two Newton steps.
__device__ __forceinline__ double rsqrt_fast(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  r = fma(r, fma(-h * r, r, 0.5), r);
  r = fma(r, fma(-h * r, r, 0.5), r);
  return r;
}

// ln(x) for finite x > 0 (normal or subnormal); x == 0 -> -inf.
// x = 2^e m, m in [sqrt(1/2), sqrt(2)); ln m = 2 atanh(s), s = (m-1)/(m+1).
__device__ __forceinline__ double log_fast(double x) {
  if (x == 0.0) return -INFINITY;
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0x1.6a09e667f3bcdp-1;  // sqrt(1/2)
  m = lo ? 2.0 * m : m;
  e = lo ? e - 1 : e;
  const double den = m + 1.0, num = m - 1.0;
  const double r = rcp_fast(den);
  double s = num * r;
  s = fma(fma(-s, den, num), r, s);  // one correction step of the quotient
  const double u = s * s;
  double p = 0x1.0f9b9e3c7c9f8p-4;
  p = fma(p, u, 0x1.0f55accc791adp-4);
  p = fma(p, u, 0x1.3b2109cc7b988p-4);
  p = fma(p, u, 0x1.745cdb9f0b4c8p-4);
  p = fma(p, u, 0x1.c71c726358ec5p-4);
  p = fma(p, u, 0x1.249249241f857p-3);
  p = fma(p, u, 0x1.9999999999ee0p-3);
  p = fma(p, u, 0x1.5555555555555p-2);
  // ln m = 2 s (1 + u p)
  const double lm = fma(2.0 * s * u, p, 2.0 * s);
  const double ed = (double)e;
  return fma(ed, 0x1.62e42fefa39efp-1, fma(ed, 0x1.abc9e3b39803fp-56, lm));  // e ln2 (hi+lo)
}

// sin(pi y), cos(pi y) for y in [0, 2).
__device__ __forceinline__ void sincospi_fast(double y, double& s, double& c) {
  const double k = __builtin_rint(2.0 * y);  // 0..4
  const double r = fma(-0.5, k, y);          // [-1/4, 1/4], exact
  const double u = r * r;
  double ps = 0x1.e3f38399551bfp-12;
  ps = fma(ps, u, -0x1.e30071afc3e59p-8);
  ps = fma(ps, u, 0x1.50782fda12d96p-4);
  ps = fma(ps, u, -0x1.32d2cce2e5b19p-1);
  ps = fma(ps, u, 0x1.466bc677587f8p+1);
  ps = fma(ps, u, -0x1.4abbce625be41p+2);
  ps = fma(ps, u, 0x1.921fb54442d18p+1);
  ps *= r;
  double pc = 0x1.f3dbcea61b1a4p-10;
  pc = fma(pc, u, -0x1.a6c9c1be9eb49p-6);
  pc = fma(pc, u, 0x1.e1f4fb60281f6p-3);
  pc = fma(pc, u, -0x1.55d3c7dbfd139p+0);
  pc = fma(pc, u, 0x1.03c1f081b0780p+2);
  pc = fma(pc, u, -0x1.3bd3cc9be458bp+2);
  pc = fma(pc, u, 1.0);
  const int q = (int)k & 3;
  const double ss = (q & 1) ? pc : ps;
  const double cc = (q & 1) ? ps : pc;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

// Sum over the 64 lanes of a wave, result in every lane.  DPP row operations (no LDS traffic):
// __shfl_xor compiles to ds_bpermute_b32 pairs with a wait after each step (~100 cycles per
// step and double), this is 6 steps of two v_mov_dpp + one v_add_f64.
//   quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: every lane of a 16-lane
//   row holds its row's sum; row_bcast15 (rows 1, 3) and row_bcast31 (rows 2, 3) carry the row
//   sums upwards, lane 63 ends with the total, v_readlane broadcasts it.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
// the value of lane 63 in every lane
__device__ __forceinline__ double wave_last(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_get<0xB1, 0xf>(v);
  v += dpp_get<0x4E, 0xf>(v);
  v += dpp_get<0x141, 0xf>(v);
  v += dpp_get<0x140, 0xf>(v);
  v += dpp_get<0x142, 0xa>(v);
  v += dpp_get<0x143, 0xc>(v);
  return wave_last(v);
}
// sum over each aligned group of 16 lanes (a DPP row), result in all 16 lanes
__device__ __forceinline__ double row16_sum_dpp(double v) {
  v += dpp_get<0xB1, 0xf>(v);
  v += dpp_get<0x4E, 0xf>(v);
  v += dpp_get<0x141, 0xf>(v);
  v += dpp_get<0x140, 0xf>(v);
  return v;
}
// max / product over the 64 lanes.  The masked row_bcast steps leave 0 in the rows they skip,
// which is the identity of + only: here those steps are guarded by the lane's row instead.
__device__ __forceinline__ double wave_max_dpp(double v) {
  v = fmax(v, dpp_get<0xB1, 0xf>(v));
  v = fmax(v, dpp_get<0x4E, 0xf>(v));
  v = fmax(v, dpp_get<0x141, 0xf>(v));
  v = fmax(v, dpp_get<0x140, 0xf>(v));
  const int row = (threadIdx.x & 63) >> 4;
  const double b15 = dpp_get<0x142, 0xa>(v);
  v = (row & 1) ? fmax(v, b15) : v;
  const double b31 = dpp_get<0x143, 0xc>(v);
  v = (row & 2) ? fmax(v, b31) : v;
  return wave_last(v);
}
__device__ __forceinline__ double wave_prod_dpp(double v) {
  v *= dpp_get<0xB1, 0xf>(v);
  v *= dpp_get<0x4E, 0xf>(v);
  v *= dpp_get<0x141, 0xf>(v);
  v *= dpp_get<0x140, 0xf>(v);
  const int row = (threadIdx.x & 63) >> 4;
  const double b15 = dpp_get<0x142, 0xa>(v);
  v = (row & 1) ? v * b15 : v;
  const double b31 = dpp_get<0x143, 0xc>(v);
  v = (row & 2) ? v * b31 : v;
  return wave_last(v);
}

}  // namespace fm
