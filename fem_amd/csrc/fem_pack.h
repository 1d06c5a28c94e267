// fem_pack.h — the 2-bit form in which a batch of equal-length reads crosses the host link (host side; the device side is
// unpack_reads_kernel + scatter_chars_kernel in fem_kernels.hip.h).  Shared by libfemhip.so (fem_dev_stage_reads packs a
// caller's batch with it) and libfemhost.so (the FASTQ parser and the read generator write this form straight into the
// pinned staging that fem_dev_acquire_stage lends, for fem_dev_commit_stage_packed).
//
// Layout (include/fem_hip.h, fem_dev_packed_layout): read i takes bytes [i * bpr, (i + 1) * bpr), bpr = ceil(len / 4);
// base j of a read sits in bits 2 (j & 3) of its byte j / 4, A C G T = 0 1 2 3 (src/utils.h:72), unused bits zero.  Only the
// four upper-case letters are packed: every other byte (lower case, N, anything) becomes code 0 and is listed as an
// exception, batch-wide index << 8 | byte, so that the device gets the batch back byte for byte — seeding and verification
// see codes (src/utils.h:72-73), but the traceback compares characters (src/align.c:289-300, :344-366).
#pragma once
#include <stdint.h>

#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace fempack {

inline uint32_t code2(uint8_t c) { return ((c >> 1) ^ (c >> 2)) & 3u; }  // A C G T -> 0 1 2 3
inline bool is_acgt(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
inline uint32_t bytes_per_read(uint32_t len) { return (len + 3u) / 4u; }
// where the exception positions start behind the codes of n reads (8-byte aligned)
inline uint64_t code_bytes(uint64_t n_reads, uint32_t len) { return (n_reads * bytes_per_read(len) + 7u) & ~7ull; }

// `first_index` is the batch-wide index of src[0].
inline void pack_scalar(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t first_index, std::vector<uint64_t> &exc) {
  for (uint64_t i = 0; i < n; i += 4) {
    uint32_t b = 0;
    for (uint32_t q = 0; q < 4 && i + q < n; ++q) {
      const uint8_t c = src[i + q];
      if (is_acgt(c)) b |= code2(c) << (2u * q);
      else exc.push_back(((first_index + i + q) << 8) | c);
    }
    dst[i >> 2] = (uint8_t)b;
  }
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void pack_avx2(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t first_index,
                                                      std::vector<uint64_t> &exc) {
  const __m256i three = _mm256_set1_epi8(3);
  const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
  const __m256i w1 = _mm256_set1_epi16(0x0401), w2 = _mm256_set1_epi32(0x00100001);
  const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1,
                                        -1, -1, -1, -1, -1, -1, -1);
  const __m256i gather = _mm256_setr_epi32(0, 4, 1, 1, 1, 1, 1, 1);
  uint64_t i = 0;
  for (; i + 32 <= n; i += 32) {
    const __m256i v = _mm256_loadu_si256((const __m256i *)(src + i));
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, cA), _mm256_cmpeq_epi8(v, cC)),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(v, cG), _mm256_cmpeq_epi8(v, cT)));
    // (16-bit shifts: what crosses a byte border lands in bits 6-7 and is masked off)
    __m256i code = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), three);
    code = _mm256_and_si256(code, ok);
    const __m256i pairs = _mm256_maddubs_epi16(code, w1);  // c0 + 4 c1 per 16 bits
    const __m256i quads = _mm256_madd_epi16(pairs, w2);    // + 16 (c2 + 4 c3) per 32 bits: the packed byte
    const __m256i bytes = _mm256_permutevar8x32_epi32(_mm256_shuffle_epi8(quads, pick), gather);
    _mm_storel_epi64((__m128i *)(dst + (i >> 2)), _mm256_castsi256_si128(bytes));
    uint32_t bad = ~(uint32_t)_mm256_movemask_epi8(ok);
    while (bad) {
      const uint64_t at = i + (uint64_t)__builtin_ctz(bad);
      exc.push_back(((first_index + at) << 8) | src[at]);
      bad &= bad - 1;
    }
  }
  if (i < n) pack_scalar(src + i, n - i, dst + (i >> 2), first_index + i, exc);
}
#endif
// n characters at src (a whole number of reads when len % 4 == 0, else one read) -> ceil(n / 4) bytes at dst
inline void pack_bases(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t first_index, std::vector<uint64_t> &exc) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) return pack_avx2(src, n, dst, first_index, exc);
#endif
  pack_scalar(src, n, dst, first_index, exc);
}

}  // namespace fempack
