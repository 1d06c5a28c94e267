// tools/sdma_probe.hip: does a host-to-device copy wait behind device-to-host copies queued on other streams?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void copy_k(uint4 *dst, const uint4 *src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
  const size_t out_bytes = 300u << 20, in_bytes = 30u << 20;
  const int n_out = 4;
  std::vector<void *> h_out(n_out), d_out(n_out);
  std::vector<hipStream_t> so(n_out);
  for (int i = 0; i < n_out; ++i) {
    CK(hipHostMalloc(&h_out[i], out_bytes, hipHostMallocDefault));
    CK(hipMalloc(&d_out[i], out_bytes));
    CK(hipStreamCreateWithFlags(&so[i], hipStreamNonBlocking));
  }
  void *h_in, *d_in;
  CK(hipHostMalloc(&h_in, in_bytes, hipHostMallocDefault));
  CK(hipMalloc(&d_in, in_bytes));
  hipStream_t si;
  CK(hipStreamCreateWithFlags(&si, hipStreamNonBlocking));
  std::vector<hipEvent_t> eo(n_out);
  hipEvent_t ei;
  for (auto &e : eo) CK(hipEventCreate(&e));
  CK(hipEventCreate(&ei));
  auto out_go = [&](int i, int how, int blocks) {
    if (how == 1) hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, so[i], (uint4 *)h_out[i], (const uint4 *)d_out[i], out_bytes / 16);
    else (void)hipMemcpyAsync(h_out[i], d_out[i], out_bytes, hipMemcpyDeviceToHost, so[i]);
    (void)hipEventRecord(eo[i], so[i]);
  };
  auto in_go = [&](int how) {
    if (how == 1) hipLaunchKernelGGL(copy_k, dim3(256), dim3(256), 0, si, (uint4 *)d_in, (const uint4 *)h_in, in_bytes / 16);
    else (void)hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, si);
    (void)hipEventRecord(ei, si);
  };
  struct Case { const char *name; int n_out, out_how, blocks, in_how, in_first; };
  const Case cases[] = {{"warm", 4, 0, 0, 0, 0}, {"1 out engine, then in engine", 1, 0, 0, 0, 0}, {"in engine first, then 1 out engine", 1, 0, 0, 0, 1},
                        {"in engine first, then 4 out engine", 4, 0, 0, 0, 1}, {"4 out engine, then in engine", 4, 0, 0, 0, 0},
                        {"1 out kernel 256 blocks, then in engine", 1, 1, 256, 0, 0}, {"1 out kernel 64 blocks, then in engine", 1, 1, 64, 0, 0},
                        {"1 out kernel 16 blocks, then in engine", 1, 1, 16, 0, 0}, {"1 out kernel 4 blocks, then in engine", 1, 1, 4, 0, 0},
                        {"1 out engine, then in kernel", 1, 0, 0, 1, 0}, {"1 out kernel 64, then in kernel", 1, 1, 64, 1, 0},
                        {"in alone engine", 0, 0, 0, 0, 1}, {"in alone kernel", 0, 0, 0, 1, 1}};
  for (const Case &c : cases) {
    CK(hipDeviceSynchronize());
    const double t0 = now_ms();
    if (c.in_first) in_go(c.in_how);
    for (int i = 0; i < c.n_out; ++i) out_go(i, c.out_how, c.blocks);
    if (!c.in_first) in_go(c.in_how);
    CK(hipEventSynchronize(ei));
    const double t_in = now_ms();
    printf("%-44s in done at %6.2f ms; outs done at", c.name, t_in - t0);
    for (int i = 0; i < c.n_out; ++i) {
      CK(hipEventSynchronize(eo[i]));
      printf(" %.2f", now_ms() - t0);
    }
    printf("\n");
  }
  {  // FEM map's pattern: one text out, the next batch's reads in on one stream, qualities / names / offsets on another
    void *h_q, *d_q;
    const size_t qb = 114u << 20, nb = 20u << 20, ob = 9u << 20;
    CK(hipHostMalloc(&h_q, qb + nb + ob, hipHostMallocDefault));
    CK(hipMalloc(&d_q, qb + nb + ob));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2, e3, e4, e5;
    for (hipEvent_t *e : {&e0, &e1, &e2, &e3, &e4, &e5}) CK(hipEventCreate(e));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, so[0]));
      CK(hipMemcpyAsync(h_out[0], d_out[0], out_bytes, hipMemcpyDeviceToHost, so[0]));
      CK(hipEventRecord(e1, so[0]));
      if (rep == 2) CK(hipMemcpyAsync(h_out[1], d_out[1], out_bytes, hipMemcpyDeviceToHost, so[1]));  // (a second text queued)
      CK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, si));
      CK(hipEventRecord(e2, si));
      CK(hipMemcpyAsync(d_q, h_q, qb, hipMemcpyHostToDevice, st));
      CK(hipEventRecord(e3, st));
      CK(hipMemcpyAsync((char *)d_q + qb, (char *)h_q + qb, nb, hipMemcpyHostToDevice, st));
      CK(hipEventRecord(e4, st));
      CK(hipMemcpyAsync((char *)d_q + qb + nb, (char *)h_q + qb + nb, ob, hipMemcpyHostToDevice, st));
      CK(hipEventRecord(e5, st));
      CK(hipDeviceSynchronize());
      float a, b, c, d, e;
      CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e0, e2)); CK(hipEventElapsedTime(&c, e0, e3));
      CK(hipEventElapsedTime(&d, e0, e4)); CK(hipEventElapsedTime(&e, e0, e5));
      printf("pattern rep %d: text out done %.2f; reads in done %.2f; qualities in %.2f, names %.2f, offsets %.2f\n", rep, a, b, c, d, e);
    }
  }
  return 0;
}
