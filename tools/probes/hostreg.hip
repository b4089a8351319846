// Pageable 80 MB host buffer -> device: plain hipMemcpy vs hipHostRegister + async copy + unregister
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
int main() {
  const size_t n = 80ull << 20;
  char* h = (char*)malloc(n + 64); memset(h, 1, n + 64);
  char* d; CHECK(hipMalloc(&d, n));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  using clk = std::chrono::steady_clock;
  auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for (int rep = 0; rep < 4; ++rep) {
    auto t0 = clk::now();
    CHECK(hipMemcpy(d, h + 8, n, hipMemcpyHostToDevice));
    auto t1 = clk::now();
    CHECK(hipHostRegister(h + 8, n, hipHostRegisterDefault));
    auto t2 = clk::now();
    CHECK(hipMemcpyAsync(d, h + 8, n, hipMemcpyHostToDevice, st)); CHECK(hipStreamSynchronize(st));
    auto t3 = clk::now();
    CHECK(hipHostUnregister(h + 8));
    auto t4 = clk::now();
    printf("pageable hipMemcpy %.0f us | register %.0f + copy %.0f + unregister %.0f = %.0f us\n", us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4), us(t1, t4));
  }
  return 0;
}
