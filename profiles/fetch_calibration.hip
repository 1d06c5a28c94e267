// fetch_calibration.hip — calibrates rocprofv3's FETCH_SIZE on the access patterns of the dense-index seed kernels
// (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Three kernels, each reading a KNOWN set of bytes once:
//   calib_runs   seed_join_kernel's list reads: a wave reads a run of `len` consecutive uint32 from a wave-uniform base
//                (lane * 4 bytes, lanes behind the run's end re-read its last entry), runs at random 4-byte-aligned
//                places of a 4 GiB buffer, five runs in flight per wave;
//   calib_gather seed_select_kernel's lookup[h] reads: one random 4-byte word per lane out of a 64 MiB table;
//   calib_pairs  seed_select_kernel's freq11 reads: lanes 2 i and 2 i + 1 read two words of one random 16-byte entry of a
//                64 MiB table.
// The program prints, per kernel, the bytes asked for and the bytes of the 64-byte sectors / 128-byte lines those
// touch (counted on the host from the same hash); profiles/make_profiles.py runs it under `rocprofv3 --pmc` and
// divides.  Build: hipcc -O3 --offload-arch=gfx950 profiles/fetch_calibration.hip -o gpurun_out/fetch_calibration
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__host__ __device__ inline uint64_t mix(uint64_t x) {  // splitmix64
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
constexpr uint64_t kBufWords = 1ull << 30;  // 4 GiB of uint32: the size of the 3 Gbp reference's occ32 table
constexpr uint32_t kTableWords = 1u << 24;  // 64 MiB
__host__ __device__ inline uint64_t run_start(uint64_t r) { return mix(r) % (kBufWords - 256); }
__host__ __device__ inline uint32_t run_len(uint64_t r) { return 38u + (uint32_t)(mix(r ^ 0xABCDEFull) % 27u); }  // 38..64, mean 51

__global__ void calib_runs(const uint32_t *buf, uint64_t n_runs, uint32_t *sink) {
  const uint32_t ln = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  uint32_t acc = 0;
  for (uint64_t r0 = wave * 5; r0 < n_runs; r0 += n_waves * 5) {
    uint32_t v[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const uint64_t r = r0 + t < n_runs ? r0 + t : r0;
      const uint32_t len = run_len(r);
      const uint32_t *bp = buf + run_start(r);
      v[t] = bp[ln < len ? ln : len - 1u];
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) acc ^= v[t];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void calib_gather(const uint32_t *table, uint64_t n_loads, uint32_t *sink) {
  uint32_t acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += stride) acc ^= table[mix(i) & (kTableWords - 1u)];
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void calib_pairs(const uint32_t *table, uint64_t n_loads, uint32_t *sink) {
  uint32_t acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += stride) {
    const uint32_t entry = (uint32_t)(mix(i >> 1) & (kTableWords / 4u - 1u));
    acc ^= table[entry * 4u + (uint32_t)(mix(i) & 3u)];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const uint64_t n_runs = 8ull << 20, n_loads = 256ull << 20;
  uint32_t *buf, *table, *sink;
  CHECK(hipMalloc(&buf, kBufWords * 4));
  CHECK(hipMalloc(&table, (size_t)kTableWords * 4));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 1, kBufWords * 4));
  CHECK(hipMemset(table, 1, (size_t)kTableWords * 4));
  // what the runs touch, counted on the host
  uint64_t asked = 0, sect64 = 0, line128 = 0;
  for (uint64_t r = 0; r < n_runs; ++r) {
    const uint64_t b0 = run_start(r) * 4, b1 = b0 + (uint64_t)run_len(r) * 4 - 1;
    asked += (uint64_t)run_len(r) * 4;
    sect64 += (b1 / 64 - b0 / 64 + 1) * 64;
    line128 += (b1 / 128 - b0 / 128 + 1) * 128;
  }
  // distinct sectors of the gathers (the table is 64 MiB = 2^20 sectors: nearly all loads of a pass are to a sector of
  // their own among the few million in flight, but over the whole launch every sector is re-read ~256 times: what
  // reaches the fabric depends on the caches, so the comparison is in units of L2 misses, printed by the profiler)
  printf("{\"runs\": {\"n\": %llu, \"asked_bytes\": %llu, \"sector64_bytes\": %llu, \"line128_bytes\": %llu},\n", (unsigned long long)n_runs,
         (unsigned long long)asked, (unsigned long long)sect64, (unsigned long long)line128);
  printf(" \"gather\": {\"lane_loads\": %llu, \"asked_bytes\": %llu, \"sector64_bytes_if_every_load_misses\": %llu},\n", (unsigned long long)n_loads,
         (unsigned long long)n_loads * 4, (unsigned long long)n_loads * 64);
  printf(" \"pairs\": {\"lane_loads\": %llu, \"asked_bytes\": %llu, \"sector64_bytes_if_every_pair_misses\": %llu}}\n", (unsigned long long)n_loads,
         (unsigned long long)n_loads * 4, (unsigned long long)n_loads / 2 * 64);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(calib_runs, dim3(256 * 8), dim3(256), 0, 0, buf, n_runs, sink);
    hipLaunchKernelGGL(calib_gather, dim3(256 * 8), dim3(256), 0, 0, table, n_loads, sink);
    hipLaunchKernelGGL(calib_pairs, dim3(256 * 8), dim3(256), 0, 0, table, n_loads, sink);
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
