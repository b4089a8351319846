// Where the time of the draw generation goes: the full kernel, its arithmetic
// without the stores, the Philox rounds alone, the stores alone, and three store variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value ubench_gen.hip -o ubench_gen
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../pyvbmc_amd/csrc/philox.h"
__global__ __launch_bounds__(256) void k_full(GenSlice g) { gen_slice_block(g, blockIdx.x, threadIdx.x); }
__global__ __launch_bounds__(256) void k_compute(GenSlice g, double* sink) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double z0, z1; philox_normal_pair((uint64_t)t, (uint32_t)(t % 5), g.seed, z0, z1);
  if (z0 == 123.456) sink[0] = z1;   // never true: keeps the arithmetic, drops the stores
}
__global__ __launch_bounds__(256) void k_philox_only(GenSlice g, double* sink) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  Philox4 r = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), (uint32_t)(t % 5), 0u, (uint32_t)g.seed, (uint32_t)(g.seed >> 32));
  if (r.x[0] == 0x12345678u && r.x[1] == 77u) sink[0] = (double)r.x[2] + r.x[3];
}
// variant A: one 16-byte store per pair (D even, so every pair is 16-byte aligned)
__global__ __launch_bounds__(256) void k_vec(GenSlice g) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= g.item_count) return;
  const int np = (g.D + 1) / 2;
  const int p = (int)(t % np);
  const int64_t r = t / np;
  const int64_t j = r / g.rows, i = r - j * g.rows;
  const uint64_t grow = (uint64_t)j * (uint64_t)g.n_half + (uint64_t)(g.row_begin + i);
  double z0, z1;
  philox_normal_pair(grow, (uint32_t)p, g.seed, z0, z1);
  double2 v; v.x = z0; v.y = z1;
  *reinterpret_cast<double2*>(g.eps + r * g.D + 2 * p) = v;
}
// variant B: four items per thread, stores issued as soon as each pair is ready
__global__ __launch_bounds__(256) void k_multi(GenSlice g) {
  const int np = (g.D + 1) / 2;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t t = ((int64_t)blockIdx.x * 4 + u) * 256 + threadIdx.x;
    if (t >= g.item_count) return;
    const int p = (int)(t % np);
    const int64_t r = t / np;
    const int64_t j = r / g.rows, i = r - j * g.rows;
    const uint64_t grow = (uint64_t)j * (uint64_t)g.n_half + (uint64_t)(g.row_begin + i);
    double z0, z1;
    philox_normal_pair(grow, (uint32_t)p, g.seed, z0, z1);
    double2 v; v.x = z0; v.y = z1;
    *reinterpret_cast<double2*>(g.eps + r * g.D + 2 * p) = v;
  }
}
// variant C: compute, but store only z0 + z1 as one double per pair (half the bytes)
__global__ __launch_bounds__(256) void k_half(GenSlice g) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= g.item_count) return;
  double z0, z1;
  philox_normal_pair((uint64_t)t, (uint32_t)(t % 5), g.seed, z0, z1);
  g.eps[t] = z0 + z1;
}
__global__ __launch_bounds__(256) void k_write(double* eps, int64_t n2) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n2) { eps[2 * t] = (double)t; eps[2 * t + 1] = 1.0; }
}
int main() {
  const int K = 50, D = 10; const int64_t rows = 10000;
  const int64_t items = (int64_t)K * rows * 5;
  double* eps; hipMalloc(&eps, sizeof(double) * K * rows * D);
  double* sink; hipMalloc(&sink, 64);
  GenSlice g; g.eps = eps; g.K = K; g.D = D; g.rows = rows; g.n_half = rows; g.row_begin = 0; g.seed = 12345; g.seed_add = nullptr;
  g.item_begin = 0; g.item_count = items; g.n_blocks = (int)((items + 255) / 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-14s %.2f us per launch\n", name, ms * 1000 / 20);
  };
  time("full", [&] { hipLaunchKernelGGL(k_full, dim3(g.n_blocks), dim3(256), 0, 0, g); });
  time("compute only", [&] { hipLaunchKernelGGL(k_compute, dim3(g.n_blocks), dim3(256), 0, 0, g, sink); });
  time("philox only", [&] { hipLaunchKernelGGL(k_philox_only, dim3(g.n_blocks), dim3(256), 0, 0, g, sink); });
  time("vec store", [&] { hipLaunchKernelGGL(k_vec, dim3(g.n_blocks), dim3(256), 0, 0, g); });
  time("4 per thread", [&] { hipLaunchKernelGGL(k_multi, dim3((g.n_blocks + 3) / 4), dim3(256), 0, 0, g); });
  time("half bytes", [&] { hipLaunchKernelGGL(k_half, dim3(g.n_blocks), dim3(256), 0, 0, g); });
  time("write only", [&] { hipLaunchKernelGGL(k_write, dim3(g.n_blocks), dim3(256), 0, 0, eps, items); });
  return 0;
}
