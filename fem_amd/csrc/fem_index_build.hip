// fem_index_build.hip — construct_index (reference src/index.c:57-98) on the GPU.
//
// The reference hashes every step-th k-mer, radix-sorts (hash, location) by hash and then sorts every bucket by
// location, so its result is "CSR by hash, each bucket ascending by location".  Locations are generated here in
// ascending order, so one STABLE radix sort of the (hash, location) pairs on the 2k hash bits gives the same
// occurrence table, and a histogram + exclusive scan gives the same uint32 lookup table.
#include "fem_index_build.hip.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "../../include/fem_hip.h"

namespace femix {
namespace {

struct SeqTable {
  const uint64_t *entry_off;  // n_seq + 1: first entry of each sequence
  const uint64_t *seq_off;
  uint32_t n_seq;
};

// entry g -> (sequence, position); hash of the k-mer there (N -> A, src/utils.h:83-99)
__global__ void hash_entries_kernel(const uint8_t *ref, SeqTable t, uint64_t n, int k, int step, uint32_t *keys,
                                    uint64_t *vals, uint32_t *hist /* lookup + 1 */) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t mask = k == 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += stride) {
    uint32_t lo = 0, hi = t.n_seq;  // last sequence whose first entry is <= g
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (t.entry_off[mid] <= g)
        lo = mid;
      else
        hi = mid;
    }
    const uint64_t pos = (g - t.entry_off[lo]) * (uint64_t)step;
    const uint8_t *p = ref + t.seq_off[lo] + pos;
    uint32_t h = 0;
    for (int i = 0; i < k; ++i) {
      uint32_t c = p[i];
      h = ((h << 2) | (c < 4u ? c : 0u)) & mask;
    }
    keys[g] = h;
    vals[g] = ((uint64_t)lo << 32) | (uint32_t)pos;
    atomicAdd(&hist[h], 1u);
  }
}

#define IX_TRY(expr)                                                            \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      *err = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
      rc = e_ == hipErrorOutOfMemory ? FEM_ERR_NOMEM : FEM_ERR_HIP;             \
      goto done;                                                                \
    }                                                                           \
  } while (0)

}  // namespace

int build_index(const uint8_t *d_ref, const std::vector<uint64_t> &seq_off, const std::vector<uint32_t> &seq_len, int k,
                int step, int n_cu, uint32_t **d_lookup_out, uint64_t **d_occ_out, uint64_t *n_occ_out,
                std::string *err) {
  int rc = FEM_OK;
  const uint32_t n_seq = (uint32_t)seq_len.size();
  std::vector<uint64_t> entry_off(n_seq + 1, 0);
  for (uint32_t s = 0; s < n_seq; ++s)  // positions 0, step, 2*step, ... while pos + k - 1 < len (src/index.c:65)
    entry_off[s + 1] = entry_off[s] + (seq_len[s] >= (uint32_t)k ? (uint64_t)(seq_len[s] - k) / step + 1 : 0);
  const uint64_t n = entry_off[n_seq];
  const size_t n_lookup = ((size_t)1 << (2 * k)) + 1;
  uint64_t *d_entry_off = nullptr, *d_seq_off = nullptr, *d_vals = nullptr, *d_occ = nullptr;
  uint32_t *d_keys = nullptr, *d_keys2 = nullptr, *d_lookup = nullptr;
  void *d_tmp = nullptr;
  size_t tmp_bytes = 0, tmp2 = 0;
  if (n > 0xFFFFFFFFull) {
    *err = "reference yields more than 2^32 index entries (lookup table is uint32)";
    return FEM_ERR_UNSUPPORTED;
  }
  IX_TRY(hipMalloc((void **)&d_lookup, n_lookup * sizeof(uint32_t)));
  IX_TRY(hipMemset(d_lookup, 0, n_lookup * sizeof(uint32_t)));
  IX_TRY(hipMalloc((void **)&d_occ, (n ? n : 1) * sizeof(uint64_t)));
  if (n) {
    IX_TRY(hipMalloc((void **)&d_entry_off, (n_seq + 1) * sizeof(uint64_t)));
    IX_TRY(hipMalloc((void **)&d_seq_off, n_seq * sizeof(uint64_t)));
    IX_TRY(hipMemcpy(d_entry_off, entry_off.data(), (n_seq + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    IX_TRY(hipMemcpy(d_seq_off, seq_off.data(), n_seq * sizeof(uint64_t), hipMemcpyHostToDevice));
    IX_TRY(hipMalloc((void **)&d_keys, n * sizeof(uint32_t)));
    IX_TRY(hipMalloc((void **)&d_keys2, n * sizeof(uint32_t)));
    IX_TRY(hipMalloc((void **)&d_vals, n * sizeof(uint64_t)));
    SeqTable t{d_entry_off, d_seq_off, n_seq};
    hipLaunchKernelGGL(hash_entries_kernel, dim3((uint32_t)n_cu * 8u), dim3(256), 0, 0, d_ref, t, n, k, step, d_keys,
                       d_vals, d_lookup + 1);
    IX_TRY(hipGetLastError());
    // lookup[i] = number of entries with hash < i: inclusive scan over the shifted histogram
    IX_TRY(rocprim::inclusive_scan(nullptr, tmp_bytes, d_lookup, d_lookup, n_lookup, rocprim::plus<uint32_t>()));
    IX_TRY(rocprim::radix_sort_pairs(nullptr, tmp2, d_keys, d_keys2, d_vals, d_occ, (size_t)n, 0u, (unsigned)(2 * k)));
    tmp_bytes = tmp_bytes > tmp2 ? tmp_bytes : tmp2;
    IX_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
    IX_TRY(rocprim::inclusive_scan(d_tmp, tmp_bytes, d_lookup, d_lookup, n_lookup, rocprim::plus<uint32_t>()));
    IX_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_vals, d_occ, (size_t)n, 0u, (unsigned)(2 * k)));
    IX_TRY(hipDeviceSynchronize());
  }
done:
  for (void *p : {(void *)d_entry_off, (void *)d_seq_off, (void *)d_vals, (void *)d_keys, (void *)d_keys2, d_tmp})
    if (p) (void)hipFree(p);
  if (rc != FEM_OK) {
    if (d_lookup) (void)hipFree(d_lookup);
    if (d_occ) (void)hipFree(d_occ);
    return rc;
  }
  *d_lookup_out = d_lookup;
  *d_occ_out = d_occ;
  *n_occ_out = n;
  return FEM_OK;
}

}  // namespace femix
