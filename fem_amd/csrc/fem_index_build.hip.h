// fem_index_build.hip.h — device-side index construction (reference construct_index, src/index.c:57-98).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace femix {
// Builds lookup[4^k+1] (uint32 prefix sums) and occ[n] (seq<<32|pos, each bucket ascending) in device memory
// from the encoded reference (base codes 0..4).  On success the caller owns *d_lookup and *d_occ (hipFree).
int build_index(const uint8_t *d_ref_codes, const std::vector<uint64_t> &seq_off, const std::vector<uint32_t> &seq_len,
                int k, int step, int n_cu, uint32_t **d_lookup, uint64_t **d_occ, uint64_t *n_occ, std::string *err);
}  // namespace femix
