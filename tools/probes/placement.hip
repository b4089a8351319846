// Where does the dispatcher put workgroup b of a 500-workgroup grid with two resident workgroups per CU?
// (70 KB of dynamic LDS per workgroup stands in for the entropy kernel's register footprint.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
__global__ void K(unsigned* out, int spin_us) {
  extern __shared__ double lds[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // XCC_ID
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  lds[threadIdx.x] = (double)t0;
  while (wall_clock64() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
}
int main() {
  const int NB = 500;
  unsigned* d; CHECK(hipMalloc(&d, 8 * NB));
  CHECK(hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(K, dim3(NB), dim3(256), 70 * 1024, 0, d, 30);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> h(2 * NB); CHECK(hipMemcpy(h.data(), d, 8 * NB, hipMemcpyDeviceToHost));
    printf("rep %d: b: xcc se sh cu (simd)\n", rep);
    std::map<unsigned, std::vector<int>> by_cu;
    for (int b = 0; b < NB; ++b) {
      const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
      const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      if (b < 40 || (b >= 256 && b < 272)) printf("  b=%3d xcc=%u se=%u sh=%u cu=%2u simd=%u\n", b, xcc, se, sh, cu, simd);
      by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    printf("  distinct CUs %zu; sample of co-resident pairs:", by_cu.size());
    int c = 0;
    for (auto& kv : by_cu) { if (c++ < 12) { printf(" ["); for (int b : kv.second) printf("%d ", b); printf("]"); } }
    printf("\n");
  }
  return 0;
}
