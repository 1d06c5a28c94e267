#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)
int main() {
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  const size_t in = 150u << 20, out = 260u << 20;
  char *d1, *d2, *h1, *h2;
  CK(hipMalloc((void**)&d1, in)); CK(hipMalloc((void**)&d2, out));
  CK(hipHostMalloc((void**)&h1, in, hipHostMallocDefault)); CK(hipHostMalloc((void**)&h2, out, hipHostMallocDefault));
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int mode = 0; mode < 3; ++mode) {
    double best = 1e9;
    for (int it = 0; it < 5; ++it) {
      CK(hipDeviceSynchronize());
      double t0 = now();
      if (mode != 1) CK(hipMemcpyAsync(d1, h1, in, hipMemcpyHostToDevice, a));
      if (mode != 0) CK(hipMemcpyAsync(h2, d2, out, hipMemcpyDeviceToHost, b));
      CK(hipDeviceSynchronize());
      double t = now() - t0;
      if (t < best) best = t;
    }
    printf("%s: %.3f ms (%.1f GB/s aggregate)\n", mode == 0 ? "H2D 150 MiB alone" : mode == 1 ? "D2H 260 MiB alone" : "both at once", best,
           ((mode != 1 ? in : 0) + (mode != 0 ? out : 0)) / best / 1e6);
  }
  return 0;
}
