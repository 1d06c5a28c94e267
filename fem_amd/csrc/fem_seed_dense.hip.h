// fem_seed_dense.hip.h — what the two kernels of the DENSE-index path (long occurrence lists: a 3 Gbp reference has ~60
// entries per 12-mer bucket; what BASELINE configs C3-C5 run) share: seed_select_kernel (fem_seed_select.hip.h) and
// seed_join_kernel (fem_seed_join.hip.h), k = 12, step = 3, R = e + 1 + a at compile time.
//
//   * 32-BIT OCCURRENCE COORDINATES.  The index file's occurrence table is uint64 `seq << 32 | pos`
//     (src/index.c:67).  Next to it the library keeps a derived uint32 copy in ONE global coordinate,
//     G = goff[seq] + pos, with a gap of kDenseGap positions between sequences, so that a whole list is half the
//     bytes, every compare / subtract / LDS word of the join is one 32-bit operation, and candidates from different
//     sequences can never come within e of each other.  The reference drops an occurrence whose position is below
//     the seed's offset in the read (`(uint32_t)occ >= start`, src/filter.c:89,106).  Only entries with
//     pos < kDenseNear (= the longest read the device path takes) can ever be dropped that way; those are stored
//     in a separate code space (kDenseRemap | seq << 10 | pos) and resolved exactly on a rare path.
//   * ONE CANDIDATE PER STRAND IS THE COMMON CASE.  If every survivor of the strand's three phase groups lies
//     within e of the smallest, the staged greedy merge (src/filter.c:45-78, :209-213) leaves exactly that
//     smallest value: one wave min/max instead of sort + merge.  Anything else takes the exact general path
//     (dense_merge_group below).
//
// Reads the two kernels cannot finish (a list over 128 entries or a bucket of 65 535 and more, more than 64 flagged values in
// a group, a DP wider than 64 columns, a == 0) are queued for the generic seed_filter_kernel; results are identical
// either way.
#pragma once
#include "fem_seed_select.hip.h"

namespace femk {

constexpr uint32_t kDenseGap = 2048u;          // positions between two sequences in the global coordinate (> e + 1)
// kDenseNear = 1024 (entries with pos < this are stored remapped: >= the longest read) and kDenseRemap = 0xF0000000 (remapped
// entry: kDenseRemap | seq << 10 | pos, seq = index within its bank, < 2^18) live in fem_seed_select.hip.h: the selection looks
// at such entries too, where a reference is cut into banks.
constexpr uint32_t kDenseMaxSeq = 1u << 18;
// A reference whose sequences do not fit one 32-bit coordinate space is cut into up to kDenseMaxBanks BANKS of consecutive
// sequences, each with coordinates of its own (goff restarts at kDenseGap).  A bucket's list is sorted by (sequence,
// position), so it falls into one contiguous part per bank; values of different sequences are never within e of each
// other, so merge, window filter and de-duplication decompose per bank exactly and the candidates of bank after bank are
// ascending.  The one rule that looks across banks — the last run keeps values <= max(U) only (src/filter.c:85) —
// becomes, per unit and bank: keep all of it (U has entries in a higher bank), as usual (this is the highest bank with
// entries of U), or drop it (only lower banks have them); seed_select_kernel<R, true> writes that into what it hands over.
// (kDenseMaxBanks, kSelKeepAll: fem_seed_select.hip.h)
constexpr uint32_t kDenseLimit = 0xDFFFF000u;  // global coordinates stay below this (round 5: 0xEFFFF000 before; the room above is the pads')
constexpr uint32_t kDenseSent = 0xEFFFFFFFu;   // "no entry" in a lane: still above kDenseVLimit after the start is subtracted
constexpr uint32_t kDenseVLimit = 0xDFFFF800u; // v < this <=> the lane holds a real entry
// PADS of the strided table (round 5): behind bucket h's f entries its slots j = f .. 127 hold
// kDensePadBase + j * kDensePadStride + pad_shift(h) — above every coordinate (also once a seed's start, < 1 024, is
// subtracted), below kDenseRemap, consecutive lanes in different LDS banks (2 304 = 9 words of the join's bitmap).  What
// sent_a of seed_join_kernel was in round 4, but it comes with the load: no clamped lane offset, no compare, no select per
// chunk.  The join does not ask which lanes hold entries at all: a pad is a value like any other that can pair with no REAL
// value (they lie 2^28 below), and whatever survives the filter is tested for being a coordinate.  pad_shift — a multiple
// of 36 below the lane stride, by a hash of the bucket — keeps the pads of a unit's runs (different buckets, starts 12 or
// more apart) out of each other's neighbouring slots but for chance: without it the same lane's pads of two runs sat 12
// to 24 positions apart and flagged each other in every unit.
constexpr uint32_t kDensePadBase = 0xE0001000u, kDensePadStride = 2304u;
constexpr uint32_t kDensePadBuckets = 16u;  // buckets of nothing but pads behind the table: run t of a unit whose seed has no list reads bucket n_buckets + t
__host__ __device__ inline uint32_t dense_pad_shift(uint32_t h) { return ((h * 0x9E3779B1u) >> 26) * 36u; }
constexpr uint32_t kDenseBlkShift = 20;        // blkseq[] granularity: first sequence at or before a 1 Mi block
constexpr uint32_t kDenseMaxList = 128u;       // entries of one seed's list the join takes (two chunks)

// Flagged values one (strand, group) unit may have before the read goes to the generic kernel: one per lane.  Chance flags
// grow like 3 n^2 / slots: ~6 per unit at R = 5, ~20 at R >= 7 (n ~ 470), where rounds 3-4 took two per lane
// (FEM_DENSE_FLAGS_HI=128).  Round 4, last change: one per lane there too — the second row's code and registers cost every
// read of C5 more (join 13.46 -> 12.88 ms per 2.5 M reads) than the 0.4 % of reads that now go to the generic kernel (units of
// several long lists: 0.46 ms): 168.2 -> 170.5 Mreads/s.
#ifndef FEM_DENSE_FLAGS_HI
#define FEM_DENSE_FLAGS_HI 64u
#endif
constexpr uint32_t dense_flag_cap(int R) { return R >= 7 ? FEM_DENSE_FLAGS_HI : 64u; }

// ---- derived tables (built once per index upload) ----
// occ (uint64 seq << 32 | pos) -> global 32-bit coordinates; *bad is set if an entry names a sequence >= n_seq
struct BankFirst {
  uint32_t n, first[5];  // bank b = sequences [first[b], first[b + 1])
};
__global__ void dense_occ32_kernel(const uint64_t *occ, uint64_t n, const uint32_t *goff, uint32_t n_seq, BankFirst banks, uint32_t *out,
                                   uint32_t *bad) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t o = occ[i];
    const uint32_t seq = (uint32_t)(o >> 32), pos = (uint32_t)o;
    uint32_t v;
    if (seq >= n_seq) {
      v = kDenseSent;
      *bad = 1u;
    } else if (pos < kDenseNear) {
      uint32_t rel = seq;  // the sequence's index within its bank (< kDenseMaxSeq: refresh_dense cuts the banks so)
      for (uint32_t b = 1; b < banks.n; ++b) rel = seq >= banks.first[b] ? seq - banks.first[b] : rel;
      v = kDenseRemap | (rel << 10) | pos;
    } else {
      v = goff[seq] + pos;
    }
    out[i] = v;
  }
}

// THE STRIDED TABLE (round 4; references in one coordinate space): bucket h's first (up to) kDenseMaxList entries at
// out[h * kDenseMaxList ...] in the same 32-bit coordinates — 512 bytes per bucket, 8.6 GB at k = 12 whatever the reference
// (the compact table is 4 bytes per entry: 4 GB at 3 Gbp) out of 288 GB.  What it buys: a list starts on a 128-byte line
// (60 entries: two fabric requests instead of 2.56 on average) and the selection finds it by the hash alone — no lookup[h]
// read for the 6 R selected seeds of a read (30 of its 76 L2 misses).  seed_join_kernel does not know the difference: it is
// handed occ32 = this table and list bases h << 7.  Entries beyond the 128th are not stored: such a read goes to the generic
// kernel (which reads the 64-bit table) as before.  Blocks of 256 buckets: their entries are one contiguous stretch of occ.
constexpr uint32_t kDenseListShift = 7;
static_assert((1u << kDenseListShift) == kDenseMaxList, "a strided slot holds exactly the lists the join takes");
static_assert(kDensePadBase - kDenseNear >= kDenseVLimit && kDensePadBase + kDenseMaxList * kDensePadStride + 64u * 36u < kDenseRemap, "pads between the coordinates and the remapped entries");
static_assert(63u * 36u < kDensePadStride, "the shift stays inside a lane's stride");
// every slot of the strided table reads "pad" before the entries are written over it
__global__ void __launch_bounds__(256) dense_pad_kernel(uint4 *out, uint64_t n_quads) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_quads; i += stride) {
    const uint32_t j = ((uint32_t)i * 4u) & (kDenseMaxList - 1u);
    const uint32_t b = kDensePadBase + dense_pad_shift((uint32_t)(i >> (kDenseListShift - 2u)));
    out[i] = make_uint4(b + j * kDensePadStride, b + (j + 1u) * kDensePadStride, b + (j + 2u) * kDensePadStride, b + (j + 3u) * kDensePadStride);
  }
}
__global__ void __launch_bounds__(256) dense_occ32_strided_kernel(const uint64_t *occ, const uint32_t *lookup, uint32_t n_buckets, const uint32_t *goff,
                                                                  uint32_t n_seq, uint32_t *out, uint32_t *bad) {
  __shared__ uint32_t lk[257];
  for (uint32_t h0 = blockIdx.x * 256u; h0 < n_buckets; h0 += gridDim.x * 256u) {
    const uint32_t nb = n_buckets - h0 < 256u ? n_buckets - h0 : 256u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= nb; i += 256u) lk[i] = lookup[h0 + i];
    __syncthreads();
    const uint32_t lo = lk[0], hi = lk[nb];
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256u) {
      uint32_t a = 0, b = nb;  // the bucket of entry i: the last one whose list starts at or before i
      while (b - a > 1u) {
        const uint32_t m = (a + b) >> 1;
        if (lk[m] <= i) a = m; else b = m;
      }
      const uint32_t rank = i - lk[a];
      if (rank >= kDenseMaxList) continue;
      const uint64_t o = occ[i];
      const uint32_t seq = (uint32_t)(o >> 32), pos = (uint32_t)o;
      uint32_t v;
      if (seq >= n_seq) {
        v = kDenseSent;
        *bad = 1u;
      } else if (pos < kDenseNear) {
        v = kDenseRemap | (seq << 10) | pos;
      } else {
        v = goff[seq] + pos;
      }
      out[((size_t)(h0 + a) << kDenseListShift) + rank] = v;
    }
  }
}

// out[h] = index in occ of bucket h's first entry whose sequence is >= first_seq (lookup[h + 1] if none): where the
// bank starting at that sequence begins in the list.  One thread per bucket, binary search (the list ascends).
__global__ void bank_split_kernel(const uint64_t *occ, const uint32_t *lookup, uint32_t n_buckets, uint32_t first_seq, uint32_t *out) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < n_buckets; h += stride) {
    uint32_t lo = lookup[h], hi = lookup[h + 1u];
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2u;
      if ((uint32_t)(occ[mid] >> 32) < first_seq) lo = mid + 1u; else hi = mid;
    }
    out[h] = lo;
  }
}

// merge_kvec_t_uint64_t (src/filter.c:45-78) on 32-bit global coordinates: `fs` = nF sorted survivors in lanes
// 0..nF-1, `cv` = the nA candidates so far.  Returns the new count, 0xFFFFFFFF if the list outgrows the wave.
__device__ __forceinline__ uint32_t dense_merge_group(uint32_t &cv, uint32_t nA, uint32_t fs, uint32_t nF, uint32_t e) {
  const uint32_t ln = lane_id();
  uint32_t merged = 0, last_kept = 0, nB = 0, ia = 0, jf = 0;
  while (ia < nA || jf < nF) {
    const uint32_t xa = (uint32_t)__builtin_amdgcn_readlane((int)cv, (int)(ia < nA ? ia : 0));
    const uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)fs, (int)(jf < nF ? jf : 0));
    const bool take_a = ia < nA && (jf >= nF || xa < xf);
    const uint32_t x = take_a ? xa : xf;
    ia += take_a ? 1u : 0u;
    jf += take_a ? 0u : 1u;
    if (nB == 0 || x > last_kept + e) {  // (global coordinates stay far below 2^32 - e)
      if (nB >= (uint32_t)kWave) return 0xFFFFFFFFu;
      merged = ln == nB ? x : merged;
      ++nB;
      last_kept = x;
    }
  }
  cv = merged;
  return nB;
}

// LDS atomics of the join: relaxed, wavefront scope (only this wave touches its bitmap)
__device__ __forceinline__ uint32_t lds_or_rtn(uint32_t *w, uint32_t bits) {
  return __hip_atomic_fetch_or(w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

}  // namespace femk
