#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
__global__ void rd(const double* p, double* out, int n) {
  double s = 0; for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
  atomicAdd(out, s);
}
int main() {
  double* p = nullptr; double* out;
  hipError_t e = hipExtMallocWithFlags((void**)&p, 1 << 16, hipDeviceMallocFinegrained);
  printf("finegrained alloc: %s ptr=%p\n", hipGetErrorString(e), (void*)p);
  CHECK(hipMalloc(&out, 8));
  hipPointerAttribute_t at; CHECK(hipPointerGetAttributes(&at, p));
  printf("type=%d isManaged=%d host=%p dev=%p\n", (int)at.type, at.isManaged, at.hostPointer, at.devicePointer);
  fflush(stdout);
  const int n = 1280;
  std::vector<double> src(n, 1.0);
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < n; ++i) src[i] = rep + 1;
    auto t0 = std::chrono::steady_clock::now();
    memcpy(p, src.data(), n * 8);   // CPU stores into device memory through the BAR
    auto t1 = std::chrono::steady_clock::now();
    CHECK(hipMemset(out, 0, 8));
    hipLaunchKernelGGL(rd, dim3(1), dim3(256), 0, 0, p, out, n);
    double h = 0; CHECK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
    printf("rep %d: host memcpy %.2f us, kernel sum %.1f (expect %.1f)\n", rep, std::chrono::duration<double, std::micro>(t1 - t0).count(), h, (double)n * (rep + 1));
  }
  return 0;
}
