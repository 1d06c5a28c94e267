// fem_kernels.hip.h — gfx950 kernels of the FEM mapping hot path.
//
//   seed_filter_kernel : one wavefront per read.  Restates
//       generate_group_seeding_candidates (reference src/filter.c:146-223) for both
//       strands: 2-bit rolling hashes, CSR lookups, the seed-selection DP, the
//       merge of the shifted occurrence lists, the additional-q-gram window filter,
//       the staged greedy de-dup and the range clip.  Lists are staged in LDS;
//       reads whose lists do not fit take the same code path over a global arena.
//   verify_kernel      : one lane per candidate.  Restates banded_edit_distance /
//       vectorized_banded_edit_distance (src/align.c:102-277) and the accept test
//       of verify_candidates (src/align.c:22,40).
//   ref_encode_kernel  : reference characters -> base codes (src/utils.h:72).
//
// Integer / bit-parallel work only: nothing here is GEMM shaped, so no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fem_planes.hip.h"

namespace femk {

constexpr int kWave = 64;
constexpr uint32_t kFlagCandOverflow = 1u;   // candidate output buffer too small
constexpr uint32_t kFlagArenaOverflow = 2u;  // global arena too small
constexpr uint32_t kFlagTooLarge = 4u;       // a single group exceeds 2^31 entries
constexpr uint32_t kInvalidMeta = 0xFFFFFFFFu;  // padding slot in the candidate arrays (skipped by verify_kernel)
// bit 31 of a candidate's meta word: it belongs to a full group of 8 of its (read, strand) list, which the reference
// sends through the 16-bit SSE lanes (src/align.c:12-13); batches hold < 2^30 reads, so the bit is free
constexpr int kMaxR = 10;  // e <= 7, a <= 2
constexpr uint32_t kMeta16 = 0x80000000u;
constexpr uint32_t kSlotChunk = 256u;           // candidate slots a wave reserves per atomic on the shared cursor
constexpr uint32_t kFlagQueueOverflow = 8u;     // slow-read queue too small
constexpr uint32_t kQueueChunk = 32u;           // queue entries a wave reserves per atomic
constexpr uint32_t kInvalidRead = 0xFFFFFFFFu;  // padding entry of the slow-read queue

// Byte offsets (relative to a wave's LDS region) and capacities; filled by the host.
struct SeedLayout {
  uint32_t pkw, nkw;         // uint32[n_words]: 2-bit packed bases, and 2-bit N masks
  uint32_t sf;               // uint2[2][smax]: (lookup[h], frequency) per strand and seed
  uint32_t dp_rows;          // uint32[n_groups][2][cmax]
  uint32_t dp_bits;          // uint32[n_groups][R][cw]
  uint32_t picked;           // uint4[n_groups][R]: (start, lookup[h], frequency, -)
  uint32_t rb;               // uint32[2][R+1]: run bounds of X and of F
  uint32_t X, F, A, B;       // uint64 arrays
  uint32_t xcap, fcap, ccap; // capacities (entries) of X, F, A/B
  uint32_t smax, cmax, cw;
  uint32_t n_words;
  uint32_t blk, blk_bytes;   // seed_fast_kernel: the raw characters of one block of reads (0 bytes: not staged)
  uint32_t gq, gq_cap;       // seed_fast_kernel, lean form: queue of live phase groups (rows of gq_cap words, then descriptors)
  uint32_t strm, strm_words; // seed_fast_kernel, lean form: three 2-bit streams over one block of reads (forward bases,
                             // their reverse complement, N marks), strm_words words each, a pad word at either end included
  uint32_t fq, gstride, nb;  // seed_select_kernel: byte frequencies [nb reads][2][3][gstride], reads per sub-block
  uint32_t rinfo;            // seed_select_kernel: uint32[4][16] per-read place / length / flags / pre-filter count
  uint32_t wave_bytes;
};

struct SeedParams {
  const uint8_t *bases;
  const uint64_t *read_off;
  uint32_t n_reads;     // one past the last read of this launch
  uint32_t read_begin;  // first read of this launch (seed_fast_kernel; the generic kernel always covers its queue / all reads)
  uint32_t *work_cursor;  // seed_fast_kernel: reads handed out so far (own cache line, zeroed per launch)
  const uint32_t *lookup;
  const uint64_t *occ;
  uint32_t inf32;  // (uint32_t)occurrence_table_size, the DP's +inf (src/filter.c:9)
  const uint32_t *seq_len;
  int32_t e, a, R, k, step, lg;
  uint64_t *cand;
  uint32_t *cand_meta;  // per candidate: read*2 + strand, | kMeta16 when it sits in a full group of 8 of its list
  uint32_t cand_cap;
  uint32_t *cand_begin;  // [2*n_reads]
  uint32_t *cand_count;  // [2*n_reads]
  uint32_t *ctr;         // [0] candidate cursor, [1] flags, [2] slow-read queue cursor
  uint32_t n_seq;
  // reads the fast kernel cannot finish (lists too long for lanes, DP too wide) are queued for the generic kernel
  uint32_t *slow_queue;        // written by seed_fast_kernel
  uint32_t slow_cap;
  const uint32_t *work_queue;  // read by seed_filter_kernel; nullptr = process every read of the batch
  // Bucket summaries, one word per kSummaryBuckets = 24 consecutive hash buckets: bit r = bucket 24 q + r is non-empty;
  // bit 24 + m = one of the buckets 24 q + 3 m .. 3 m + 2 holds two or more entries.  A non-empty bucket whose second
  // bit is clear has frequency exactly 1 — all the seed-selection DP needs; lookup[h] is then fetched only for the few
  // such seeds that end up selected.  One load answers both questions; 2.7 MiB for k = 12, which each XCD's L2 keeps.
  // Sparse indexes only (nullptr otherwise): most of the lookups per read then never touch the 64 MiB table.
  const uint32_t *summary;
  // seed_join_kernel (fem_seed_dense.hip.h): the occurrence table in 32-bit global coordinates, the coordinate of
  // each sequence's first base (n_seq + 1 entries) and, per 2^20 coordinates, the last sequence starting at or before
  const uint32_t *occ32;
  // != 0: occ32 is the STRIDED table — bucket h's first (up to) 2^list_shift entries at occ32[h << list_shift], every list on a
  // line boundary, no lookup[h] to find it (fem_seed_dense.hip.h); 0: the compact table, bucket h's list at occ32[lookup[h]]
  uint32_t list_shift;
  const uint32_t *goff;
  const uint32_t *blkseq;
  // seed_select_kernel -> seed_join_kernel (fem_seed_select.hip.h): saturated byte frequencies per 11-mer; per read and
  // (strand, phase group) the R selected seeds in run order (lookup[h], start | frequency << 16); per read
  // (status | length << 8, pre-filter count)
  const uint32_t *freq11;
  uint2 *sel;
  uint2 *sel_hdr;
  // References beyond the 32-bit coordinate (fem_seed_dense.hip.h, "banks"): the sequences [bank_first[b], bank_first[b+1])
  // share one coordinate space; bank_lo[(b - 1) * n_buckets + h] = where bank b's part of bucket h's list starts in the
  // occurrence tables (b = 1 .. n_banks - 1); blkseq holds blk_stride entries per bank.  n_banks == 1: none of this is read.
  uint32_t n_banks;
  uint32_t bank_first[5];
  const uint32_t *bank_lo;
  uint32_t n_buckets;
  uint32_t blk_stride;
  unsigned long long *stats;  // [0] sum of pre-filter counts, [1] sum of candidates
  uint64_t *arena;
  unsigned long long arena_cap;   // entries
  unsigned long long *arena_ctr;  // [0] cursor, [1] total entries wanted (for the retry)
  SeedLayout lay;
};

// read offsets of a batch of equal-length reads (fem_dev_commit_stage_uniform): off[i] = i * len, i <= n
__global__ void uniform_offsets_kernel(uint64_t *off, uint64_t n, uint32_t len) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) off[i] = i * (uint64_t)len;
}

// ---- packed read transfer (fem_dev_stage_reads on batches of equal-length reads) ----
// The host sends two bits per base for the characters A C G T (four bases per byte, low bits first, every read padded
// to whole bytes) and, for every other byte of the batch, its position and the byte itself; the batch is rebuilt byte
// for byte.  One thread per packed byte.
__global__ void unpack_reads_kernel(const uint8_t *packed, uint64_t n_reads, uint32_t len, uint32_t bytes_per_read, uint8_t *bases) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, total = n_reads * bytes_per_read;
  const bool aligned = (len & 3u) == 0u && (((uintptr_t)bases) & 3u) == 0u;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const uint32_t b = packed[i];
    // "ACGT" as a little-endian word: the code selects its byte
    const uint32_t c0 = (0x54474341u >> (8u * (b & 3u))) & 0xFFu, c1 = (0x54474341u >> (8u * ((b >> 2) & 3u))) & 0xFFu;
    const uint32_t c2 = (0x54474341u >> (8u * ((b >> 4) & 3u))) & 0xFFu, c3 = (0x54474341u >> (8u * (b >> 6))) & 0xFFu;
    if (aligned) {  // (then bytes_per_read * 4 == len: the packed stream has no padding)
      ((uint32_t *)bases)[i] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    } else {
      const uint64_t r = i / bytes_per_read;
      const uint32_t k = (uint32_t)(i - r * bytes_per_read) * 4u;
      uint8_t *o = bases + r * len + k;
      o[0] = (uint8_t)c0;
      if (k + 1u < len) o[1] = (uint8_t)c1;
      if (k + 2u < len) o[2] = (uint8_t)c2;
      if (k + 3u < len) o[3] = (uint8_t)c3;
    }
  }
}
__global__ void scatter_chars_kernel(const uint32_t *at, const uint8_t *ch, uint64_t n, uint8_t *bases) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) bases[at[i]] = ch[i];
}

// bit r of bits[]: read r of a batch of equal-length reads has one of the characters the packed transfer sends separately
__global__ void mark_exception_reads_kernel(const uint32_t *at, uint64_t n, uint32_t len, uint32_t *bits) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t r = at[i] / len;
    atomicOr(&bits[r >> 5], 1u << (r & 31u));
  }
}

struct Picked {
  uint32_t start, lo, freq, pad;
};

struct Bufs {
  uint64_t *X;
  uint64_t *F;
  uint64_t *A;
  uint64_t *B;
  uint32_t xcap, fcap, ccap;
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Diagnostic build only (-DFEM_STAMPS, never the shipped library): per-phase cycle totals of the seed kernel.
#ifdef FEM_STAMPS
constexpr int kNumStamps = 12;
__device__ unsigned long long g_stamp_cycles[kNumStamps];
struct Prof {
  unsigned long long acc[kNumStamps] = {};
  unsigned long long last = 0;
  __device__ void start() { last = __builtin_amdgcn_s_memtime(); }
  __device__ void mark(int id) {
    unsigned long long t = __builtin_amdgcn_s_memtime();
    acc[id] += t - last;
    last = t;
  }
  __device__ void flush() {
    if (lane_id() == 0)
      for (int i = 0; i < kNumStamps; ++i) atomicAdd(&g_stamp_cycles[i], acc[i]);
  }
};
#define STAMP_START(p) (p).start()
#define STAMP(p, id) (p).mark(id)
#else
struct Prof {};
#define STAMP_START(p) ((void)0)
#define STAMP(p, id) ((void)0)
#endif

// Wave-synchronous ordering of LDS traffic: DS operations of one wave execute in
// order, so only the compiler has to be kept from moving them.
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Same for the global arena: make this wave's stores visible to its own later loads.
__device__ __forceinline__ void wave_sync_global() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <bool GLOBAL>
__device__ __forceinline__ void wave_sync() {
  if (GLOBAL)
    wave_sync_global();
  else
    wave_sync_lds();
}

__device__ __forceinline__ uint32_t bcast0(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// char -> base code (src/utils.h:72): A/a 0, C/c 1, G/g 2, T/t 3, anything else 4
__device__ __forceinline__ uint32_t base_code(uint32_t c) {
  uint32_t x = (c >> 1) & 3u;           // A 0, C 1, G 3, T 2
  uint32_t code = x ^ (x >> 1);         // A 0, C 1, G 2, T 3
  uint32_t u = c & 0xDFu;               // upper case
  bool acgt = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
  return acgt ? code : 4u;
}

// reverse the order of the 2-bit groups of a 2k-bit value
__device__ __forceinline__ uint32_t reverse_pairs(uint32_t x, int k) {
  uint32_t r = __brev(x) >> (32 - 2 * k);
  return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}

template <typename T>
__device__ __forceinline__ uint32_t lower_bound_u64(const T *x, uint32_t lo, uint32_t hi, uint64_t key) {
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (x[mid] < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}
template <typename T>
__device__ __forceinline__ uint32_t upper_bound_u64(const T *x, uint32_t lo, uint32_t hi, uint64_t key) {
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (x[mid] <= key)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------------------
// One strand: the three phase groups of src/filter.c:190-213 followed by the
// staged greedy de-dup.  Returns the number of entries left in the returned
// list (sorted, before the range clip), or 0xFFFFFFFF if a buffer was too small.
// All control flow is wave-uniform; lanes split the list entries.
// ---------------------------------------------------------------------------
template <bool GLOBAL>
__device__ uint32_t strand_lists(const SeedParams &p, const Picked *picked /* [step][R] of this strand */,
                                 uint32_t *rb, const Bufs &b, uint64_t **result, Prof &prof) {
  const uint32_t ln = lane_id();
  const uint64_t lt_mask = (1ull << ln) - 1ull;
  const int R = p.R;
  const uint64_t e64 = (uint64_t)p.e;
  uint32_t *fb = rb + (R + 1);
  uint64_t *A = b.A, *B = b.B;
  uint32_t nA = 0;

  for (int si = 0; si < p.step; ++si) {
    const Picked *pk = picked + si * R;
    // ---- merge_candidate_locations (src/filter.c:80-116) as a set: stage the R shifted runs ----
    uint32_t n = 0;
    bool too_big = false;
    if (ln == 0) rb[0] = 0, fb[0] = 0;
    for (int t = 0; t < R; ++t) {
      const uint32_t start = pk[t].start, lo = pk[t].lo, freq = pk[t].freq;
      uint64_t max_u = 0;
      bool last = (t == R - 1);
      if (last) {
        // the last seed is merged only while the accumulated list has elements
        // (src/filter.c:85): keep q <= max(U); nothing if U is empty.
        if (n == 0) {
          if (ln == 0) rb[t + 1] = n;
          continue;
        }
        wave_sync<GLOBAL>();
        for (int u = 0; u < R - 1; ++u) {
          uint32_t hi_ = rb[u + 1], lo_ = rb[u];
          if (hi_ > lo_ && hi_ <= b.xcap) {
            uint64_t v = b.X[hi_ - 1];
            max_u = v > max_u ? v : max_u;
          }
        }
      }
      for (uint32_t i0 = 0; i0 < freq; i0 += kWave) {
        uint32_t i = i0 + ln;
        bool ok = i < freq;
        uint64_t v = 0;
        if (ok) {
          uint64_t o = p.occ[(uint64_t)lo + i];
          ok = (uint32_t)o >= start;  // src/filter.c:89,106
          v = o - start;
          if (last) ok = ok && (v <= max_u);
        }
        uint64_t m = __ballot(ok);
        uint32_t pos = n + __popcll(m & lt_mask);
        if (ok && pos < b.xcap) b.X[pos] = v;
        n += __popcll(m);
        if (n > 0x7fffffffu) too_big = true;
      }
      if (ln == 0) rb[t + 1] = n;
    }
    if (too_big) return 0xFFFFFFFEu;
    if (n > b.xcap) return 0xFFFFFFFFu;
    STAMP(prof, 4);

    // ---- additional_qgram_filter (src/filter.c:118-131): v survives iff at least a+1 staged values lie
    //      in [v, v+e] (itself included); duplicates never survive the greedy pass below, so the set is enough.
    uint32_t nF = 0;
    if (n > (uint32_t)p.a) {
      wave_sync<GLOBAL>();
      const uint32_t need = (uint32_t)p.a + 1u;
      for (int t = 0; t < R; ++t) {
        uint32_t r_lo = rb[t], r_hi = rb[t + 1];
        for (uint32_t i0 = r_lo; i0 < r_hi; i0 += kWave) {
          uint32_t i = i0 + ln;
          bool act = i < r_hi;
          bool pass = false;
          uint64_t v = 0;
          if (act) {
            v = b.X[i];
            uint32_t cnt = 0;
            for (int u = 0; u < R && cnt < need; ++u) {
              uint32_t u_lo = rb[u], u_hi = rb[u + 1];
              if (u_lo == u_hi) continue;
              uint32_t j = (u == t) ? i : lower_bound_u64(b.X, u_lo, u_hi, v);
              if (u == t) {
                // own run: equal values before i also lie in [v, v+e]
                uint32_t jj = i;
                while (jj > u_lo && b.X[jj - 1] == v && cnt < need) {
                  ++cnt;
                  --jj;
                }
              }
              while (j < u_hi && cnt < need && b.X[j] <= v + e64) {
                ++cnt;
                ++j;
              }
            }
            pass = cnt >= need;
          }
          uint64_t m = __ballot(pass);
          uint32_t pos = nF + __popcll(m & lt_mask);
          if (pass && pos < b.fcap) b.F[pos] = v;
          nF += __popcll(m);
        }
        if (ln == 0) fb[t + 1] = nF;
      }
      if (nF > b.fcap) return 0xFFFFFFFFu;
    }
    STAMP(prof, 5);

    // ---- sort F (R sorted runs) into X by rank ----
    if (nF > 0) {
      wave_sync<GLOBAL>();
      for (int t = 0; t < R; ++t) {
        uint32_t r_lo = fb[t], r_hi = fb[t + 1];
        for (uint32_t i0 = r_lo; i0 < r_hi; i0 += kWave) {
          uint32_t i = i0 + ln;
          if (i < r_hi) {
            uint64_t v = b.F[i];
            uint32_t rank = i - r_lo;
            for (int u = 0; u < R; ++u) {
              uint32_t u_lo = fb[u], u_hi = fb[u + 1];
              if (u == t || u_lo == u_hi) continue;
              uint32_t j = (u < t) ? upper_bound_u64(b.F, u_lo, u_hi, v) : lower_bound_u64(b.F, u_lo, u_hi, v);
              rank += j - u_lo;
            }
            b.X[rank] = v;  // nF <= n <= xcap
          }
        }
      }
      wave_sync<GLOBAL>();
      // ---- merge_kvec_t_uint64_t (src/filter.c:45-78): merge with the candidates so far, keep x iff
      //      x > last kept + e.  Sequential by definition; one lane walks the two short lists.
      uint32_t nB = 0;
      if (ln == 0) {
        uint32_t i = 0, j = 0;
        uint64_t last_kept = 0;
        while (i < nA || j < nF) {
          uint64_t x;
          if (i < nA && (j >= nF || A[i] < b.X[j]))
            x = A[i++];
          else
            x = b.X[j++];
          if (nB == 0 || x > last_kept + e64) {
            if (nB < b.ccap) B[nB] = x;
            ++nB;
            last_kept = x;
          }
        }
      }
      nB = bcast0(nB);
      if (nB > b.ccap) return 0xFFFFFFFFu;
      uint64_t *tmp = A;
      A = B;
      B = tmp;
      nA = nB;
      wave_sync<GLOBAL>();
    }
    STAMP(prof, 6);
    // nF == 0: greedy(merge(cand, {})) == cand, because cand already satisfies the gap rule
  }
  *result = A;
  return nA;
}

// Two-level search in a sorted stretch x[lo, hi) of the global arena whose every 2^sh-th element (x[lo + (k << sh)], k < cnt)
// is kept in LDS as s[k]: the sample decides the piece, the piece is at most 2^sh elements of one or two cache lines (sh = 0:
// the whole stretch is in LDS and the arena is not read at all).
// (A plain binary search over 1 000 arena entries is ten dependent L2 round trips per element.)
typedef const __attribute__((address_space(3))) uint64_t *LdsU64;  // (a pointer that went through a call is "flat" to the compiler otherwise)
template <bool UPPER>
__device__ __forceinline__ uint32_t sampled_bound_u64(const uint64_t *x, uint32_t lo, uint32_t hi, const uint64_t *s_flat, uint32_t cnt, uint32_t sh,
                                                      uint64_t key) {
  const LdsU64 s = (LdsU64)s_flat;
  const uint32_t k = UPPER ? upper_bound_u64(s, 0u, cnt, key) : lower_bound_u64(s, 0u, cnt, key);
  if (k == 0) return lo;  // x[lo] itself is beyond the key (or the stretch is empty)
  const uint32_t lo2 = lo + ((k - 1u) << sh) + 1u, end = lo + (k << sh);
  const uint32_t hi2 = end < hi ? end : hi;
  return UPPER ? upper_bound_u64(x, lo2, hi2, key) : lower_bound_u64(x, lo2, hi2, key);
}
#ifndef FEM_SAMPLE_MIN
#define FEM_SAMPLE_MIN 256u
#endif
constexpr uint32_t kSampleMin = FEM_SAMPLE_MIN;  // stretches shorter than this are searched plainly (tests build it with 4)

// ---------------------------------------------------------------------------
// The same for a strand whose lists went to the global arena (they did not fit the wave's LDS: repeats, where a seed has
// hundreds of occurrences and a strand hundreds of candidates).  Same results as strand_lists<true>, organised so that
// nothing walks a long list with one lane (round 4: a read of a 1000-copy repeat cost ~1.2 ms of a wave):
//   * the staged runs are merged ONCE into sorted order (rank of an element = its place in its run + R - 1 binary
//     searches; ties by run, then by place: the order of the reference's merges, src/filter.c:91-110) — the
//     additional-q-gram filter is then what the reference states, x[i + a] <= x[i] + e on the sorted multiset
//     (src/filter.c:118-131), and its survivors come out sorted: no second round of searches;
//   * merge_kvec_t_uint64_t's greedy pass (src/filter.c:45-78; keep x iff x > last kept + e) by CHAINS: in the merged
//     order an element more than e above its predecessor is kept whatever came before (last kept <= predecessor) and
//     starts a chain; only inside a chain — consecutive gaps <= e — does the rule look back, and the chain's first
//     lane walks it.  Chains are a handful of elements (the values of one place's seeds), so all lanes work.
//   * every binary search goes through an LDS sample of the stretch it searches (sampled_bound_u64): `smp` is the wave's
//     LDS list space, unused on this path (smp_cap 64-bit entries).
// Buffers (arena slices of the caller): X, F >= the group's staged entries, A, B >= all groups' entries.
// ---------------------------------------------------------------------------
__device__ uint32_t strand_lists_big(const SeedParams &p, const Picked *picked, uint32_t *rb, const Bufs &b, uint64_t *smp, uint32_t smp_cap,
                                     uint64_t **result, Prof &prof) {
  const uint32_t ln = lane_id();
  const uint32_t cap_s = smp_cap > 64u ? smp_cap - 16u : 0u;  // samples; behind them the runs' sample offsets (kMaxR + 1 words)
  uint32_t *s_off = (uint32_t *)(smp + cap_s);
  const uint64_t lt_mask = (1ull << ln) - 1ull;
  const int R = p.R;
  const uint64_t e64 = (uint64_t)p.e;
  uint64_t *A = b.A, *B = b.B;
  uint32_t nA = 0;
  for (int si = 0; si < p.step; ++si) {
    const Picked *pk = picked + si * R;
    // ---- stage the R shifted runs (as strand_lists) ----
    uint32_t n = 0;
    bool too_big = false;
    if (ln == 0) rb[0] = 0;
    for (int t = 0; t < R; ++t) {
      const uint32_t start = pk[t].start, lo = pk[t].lo, freq = pk[t].freq;
      uint64_t max_u = 0;
      const bool last = (t == R - 1);
      if (last) {  // the last seed is merged only while the list has elements, and only up to its maximum (src/filter.c:85)
        if (n == 0) {
          if (ln == 0) rb[t + 1] = n;
          continue;
        }
        wave_sync_global();
        for (int u = 0; u < R - 1; ++u) {
          const uint32_t hi_ = rb[u + 1], lo_ = rb[u];
          if (hi_ > lo_ && hi_ <= b.xcap) {
            const uint64_t v = b.X[hi_ - 1];
            max_u = v > max_u ? v : max_u;
          }
        }
      }
      for (uint32_t i0 = 0; i0 < freq; i0 += kWave) {
        const uint32_t i = i0 + ln;
        bool ok = i < freq;
        uint64_t v = 0;
        if (ok) {
          const uint64_t o = p.occ[(uint64_t)lo + i];
          ok = (uint32_t)o >= start;  // src/filter.c:89,106
          v = o - start;
          if (last) ok = ok && (v <= max_u);
        }
        const uint64_t m = __ballot(ok);
        const uint32_t pos = n + __popcll(m & lt_mask);
        if (ok && pos < b.xcap) b.X[pos] = v;
        n += __popcll(m);
        if (n > 0x7fffffffu) too_big = true;
      }
      if (ln == 0) rb[t + 1] = n;
    }
    if (too_big) return 0xFFFFFFFEu;
    if (n > b.xcap || n > b.fcap) return 0xFFFFFFFFu;
    STAMP(prof, 4);
    if (n <= (uint32_t)p.a) continue;  // fewer than a + 1 values: nothing passes the filter
    wave_sync_global();
    // ---- the merged order: F[rank] = value.  (The R - 1 binary searches of an element in lockstep — one probe of every
    //      other run per round — were tried: the arrays cost more registers than the overlapped loads gave, 26.6 -> 32.6 ms
    //      per 50 k repeat reads.) ----
    const bool sampled = n >= kSampleMin && cap_s != 0u;
    uint32_t sh = 0;
    if (sampled) {  // every 2^sh-th element of every run into LDS: at most (n >> sh) + R samples (all of them where they fit)
      while ((n >> sh) + (uint32_t)R > cap_s) ++sh;
      if (ln == 0) {
        uint32_t o = 0;
        for (int t = 0; t < R; ++t) {
          const uint32_t len = rb[t + 1] - rb[t];
          s_off[t] = o;
          o += len ? ((len - 1u) >> sh) + 1u : 0u;
        }
        s_off[R] = o;
      }
      wave_sync_lds();
      for (int t = 0; t < R; ++t) {
        const uint32_t o = s_off[t], cnt = s_off[t + 1] - o, r_lo = rb[t];
        for (uint32_t k = ln; k < cnt; k += kWave) smp[o + k] = b.X[r_lo + (k << sh)];
      }
      wave_sync_lds();
    }
    for (int t = 0; t < R; ++t) {
      const uint32_t r_lo = rb[t], r_hi = rb[t + 1];
      for (uint32_t i0 = r_lo; i0 < r_hi; i0 += kWave) {
        const uint32_t i = i0 + ln;
        if (i < r_hi) {
          const uint64_t v = b.X[i];
          uint32_t rank = i - r_lo;
          for (int u = 0; u < R; ++u) {
            const uint32_t u_lo = rb[u], u_hi = rb[u + 1];
            if (u == t || u_lo == u_hi) continue;
            uint32_t j;
            if (sampled) {
              const uint32_t o = s_off[u], cnt = s_off[u + 1] - o;
              j = (u < t) ? sampled_bound_u64<true>(b.X, u_lo, u_hi, smp + o, cnt, sh, v) : sampled_bound_u64<false>(b.X, u_lo, u_hi, smp + o, cnt, sh, v);
            } else {
              j = (u < t) ? upper_bound_u64(b.X, u_lo, u_hi, v) : lower_bound_u64(b.X, u_lo, u_hi, v);
            }
            rank += j - u_lo;
          }
          b.F[rank] = v;
        }
      }
    }
    wave_sync_global();
    // ---- additional_qgram_filter on the sorted multiset (src/filter.c:118-131): survivors, still sorted, back into X ----
    uint32_t nF = 0;
    for (uint32_t k0 = 0; k0 < n; k0 += kWave) {
      const uint32_t k = k0 + ln;
      bool pass = false;
      uint64_t v = 0;
      if (k + (uint32_t)p.a < n) {
        v = b.F[k];
        pass = b.F[k + (uint32_t)p.a] <= v + e64;
      }
      const uint64_t m = __ballot(pass);
      if (pass) b.X[nF + __popcll(m & lt_mask)] = v;
      nF += __popcll(m);
    }
    STAMP(prof, 5);
    if (nF == 0) continue;  // greedy(merge(cand, {})) == cand
    wave_sync_global();
    // ---- merge with the candidates so far (A first unless X is smaller or equal: src/filter.c:52-60) into B ----
    const uint32_t nM = nA + nF;
    if (nM > b.ccap) return 0xFFFFFFFFu;
    const bool sampled2 = nA != 0u && nM >= kSampleMin && cap_s != 0u;
    uint32_t sh2 = 0, cX = 0, cA = 0;
    if (sampled2) {  // samples of the survivors, then of the candidates so far
      while ((nM >> sh2) + 2u > cap_s) ++sh2;
      cX = ((nF - 1u) >> sh2) + 1u, cA = ((nA - 1u) >> sh2) + 1u;
      wave_sync_lds();  // (the ranks above are done with their samples)
      for (uint32_t k = ln; k < cX; k += kWave) smp[k] = b.X[k << sh2];
      for (uint32_t k = ln; k < cA; k += kWave) smp[cX + k] = A[k << sh2];
      wave_sync_lds();
    }
    for (uint32_t i0 = 0; i0 < nA; i0 += kWave) {
      const uint32_t i = i0 + ln;
      if (i < nA) {
        const uint64_t v = A[i];
        const uint32_t at = sampled2 ? sampled_bound_u64<true>(b.X, 0u, nF, smp, cX, sh2, v) : upper_bound_u64(b.X, 0u, nF, v);
        B[i + at] = v;  // (an equal value of X goes first)
      }
    }
    for (uint32_t j0 = 0; j0 < nF; j0 += kWave) {
      const uint32_t j = j0 + ln;
      if (j < nF) {
        const uint64_t v = b.X[j];
        const uint32_t at = sampled2 ? sampled_bound_u64<false>(A, 0u, nA, smp + cX, cA, sh2, v) : lower_bound_u64(A, 0u, nA, v);
        B[j + at] = v;
      }
    }
    wave_sync_global();
    // ---- the greedy pass by chains; the keep flags are bytes over F (dead by now: 8 fcap bytes; nM <= 3 fcap at step <= 8 —
    //      beyond that, step 9..16, a strand may merge more than F holds as bytes: such a strand is declined like one that
    //      outgrows the arena, and the batch runs again with larger buffers) ----
    if (nM > 8u * b.fcap) return 0xFFFFFFFFu;
    uint8_t *keep = (uint8_t *)b.F;
    for (uint32_t k0 = 0; k0 < nM; k0 += kWave) {
      const uint32_t k = k0 + ln;
      if (k < nM) {
        const uint64_t v = B[k];
        if (k == 0 || v - B[k - 1] > e64) {  // a chain starts here: kept, and this lane walks it
          keep[k] = 1;
          uint64_t last = v, cur = v;
          for (uint32_t j = k + 1; j < nM; ++j) {
            const uint64_t nx = B[j];
            if (nx - cur > e64) break;  // (the next chain's first element: its own lane's)
            const bool kp = nx > last + e64;
            keep[j] = kp ? 1 : 0;
            last = kp ? nx : last;
            cur = nx;
          }
        }
      }
    }
    wave_sync_global();
    uint32_t nB = 0;
    for (uint32_t k0 = 0; k0 < nM; k0 += kWave) {
      const uint32_t k = k0 + ln;
      const bool kp = k < nM && keep[k] != 0;
      const uint64_t m = __ballot(kp);
      if (kp) A[nB + __popcll(m & lt_mask)] = B[k];  // (A's old content lives on in B)
      nB += __popcll(m);
    }
    nA = nB;
    wave_sync_global();
    STAMP(prof, 6);
  }
  *result = A;
  return nA;
}

// Candidate slots are handed out from one global cursor.  One returning atomic per (read, strand) on a single
// address tops out near 90 M/s on this chip, so each wave reserves kSlotChunk slots at a time and fills them
// locally; what is left of a chunk when the wave moves on is padded with kInvalidMeta.
struct SlotChunk {
  uint32_t next = 0, left = 0;
};

__device__ __forceinline__ void pad_chunk(const SeedParams &p, SlotChunk &ch) {
  for (uint32_t i = lane_id(); i < ch.left; i += kWave)
    if ((unsigned long long)ch.next + i < p.cand_cap) p.cand_meta[ch.next + i] = kInvalidMeta;
  ch.left = 0;
}

// remove_out_ranged_candidates (src/filter.c:133-144) + hand-over to the verify kernel
template <bool GLOBAL>
__device__ void clip_and_emit(const SeedParams &p, uint32_t read, uint32_t strand, uint32_t L, const uint64_t *list,
                              uint32_t n, uint64_t *tmp, unsigned long long &cand_sum, SlotChunk &ch) {
  const uint32_t ln = lane_id();
  const uint64_t lt_mask = (1ull << ln) - 1ull;
  uint32_t kept = 0;
  for (uint32_t i0 = 0; i0 < n; i0 += kWave) {
    uint32_t i = i0 + ln;
    bool ok = false;
    uint64_t x = 0;
    if (i < n) {
      x = list[i];
      uint32_t seq = (uint32_t)(x >> 32), pos = (uint32_t)x;
      uint32_t slen = p.seq_len[seq];
      ok = pos >= (uint32_t)p.e && pos + L + (uint32_t)p.e < slen;
    }
    uint64_t m = __ballot(ok);
    uint32_t pos = kept + __popcll(m & lt_mask);
    if (ok) tmp[pos] = x - (uint64_t)p.e;  // kept <= n <= capacity of tmp
    kept += __popcll(m);
  }
  uint32_t base = 0;
  if (kept > 0) {
    if (kept <= ch.left) {
      base = ch.next;
      ch.next += kept, ch.left -= kept;
    } else {
      // long lists get an exact reservation and leave the current chunk alone
      uint32_t want = kept > kSlotChunk / 2 ? kept : kSlotChunk;
      if (want == kSlotChunk) pad_chunk(p, ch);
      if (ln == 0) base = atomicAdd(&p.ctr[0], want);
      base = bcast0(base);
      if (want == kSlotChunk) ch.next = base + kept, ch.left = kSlotChunk - kept;
    }
    if ((unsigned long long)base + kept > p.cand_cap) {
      if (ln == 0) atomicOr(&p.ctr[1], kFlagCandOverflow);
    } else {
      wave_sync<GLOBAL>();
      for (uint32_t i = ln; i < kept; i += kWave) {
        p.cand[base + i] = tmp[i];
        p.cand_meta[base + i] = (read * 2u + strand) | (i < (kept & ~7u) ? kMeta16 : 0u);
      }
    }
  }
  if (ln == 0) {
    p.cand_begin[read * 2u + strand] = base;
    p.cand_count[read * 2u + strand] = kept;
  }
  cand_sum += kept;
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}

// ---------------------------------------------------------------------------
// Register-only form of one strand (src/filter.c:190-223) for the common case in
// which the step*R selected seeds hold at most 64 occurrences in total: every
// occurrence lives in one lane, the last-seed rule, the window filter and the
// staged greedy de-dup run on wave-uniform scalars (v_readlane), and nothing is
// staged in LDS except one 64-entry scatter.  Same results as strand_lists().
// Returns false (nothing done) when the strand does not qualify.
// ---------------------------------------------------------------------------
__device__ bool strand_small(const SeedParams &p, const Picked *pk /* [step][R] */, uint64_t *scatter /* LDS, 64 */,
                             uint32_t read, uint32_t strand, uint32_t L, unsigned long long &cand_sum,
                             SlotChunk &ch) {
  const uint32_t ln = lane_id();
  const int R = p.R;
  const uint32_t n_seeds = (uint32_t)(p.step * R);
  if (n_seeds > (uint32_t)kWave) return false;
  const uint64_t e64 = (uint64_t)p.e;
  // lane s < n_seeds holds seed s = (group, run)
  uint32_t s_start = 0, s_lo = 0, s_freq = 0, s_grp = 0, s_run = 0;
  if (ln < n_seeds) {
    Picked q = pk[ln];
    s_start = q.start, s_lo = q.lo, s_freq = q.freq;
    s_grp = ln / (uint32_t)R, s_run = ln - s_grp * (uint32_t)R;
  }
  const uint64_t nonempty = __ballot(s_freq > 0);
  uint32_t total = 0;
  for (uint64_t m = nonempty; m;) {
    int j = __builtin_ctzll(m);
    m &= m - 1;
    uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)s_freq, j);
    if (f > (uint32_t)kWave) return false;
    total += f;
  }
  if (total > (uint32_t)kWave) return false;

  uint32_t kept = 0;
  uint64_t cv = 0;  // lane i holds candidate i of the strand (sorted, before the range clip)
  if (total > (uint32_t)p.a) {
    // ---- one occurrence per lane, runs in (group, run) order = the order of the staged lists ----
    bool valid = false;
    uint64_t v = 0;
    uint32_t grp = 0, run = 0;
    {
      uint32_t at = 0;
      for (uint64_t m = nonempty; m;) {
        int j = __builtin_ctzll(m);
        m &= m - 1;
        uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)s_freq, j);
        uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)s_lo, j);
        uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)s_start, j);
        uint32_t gj = (uint32_t)__builtin_amdgcn_readlane((int)s_grp, j);
        uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)s_run, j);
        if (ln >= at && ln < at + f) {
          uint64_t o = p.occ[(uint64_t)lo + (ln - at)];
          valid = (uint32_t)o >= st;  // src/filter.c:89,106
          v = o - st;
          grp = gj, run = rj;
        }
        at += f;
      }
    }
    // ---- last seed of each group: only values <= max of the other runs survive (src/filter.c:85) ----
    for (uint32_t g = 0; g < (uint32_t)p.step; ++g) {
      bool is_last = valid && grp == g && run == (uint32_t)(R - 1);
      if (!__ballot(is_last)) continue;
      uint64_t mu = __ballot(valid && grp == g && run != (uint32_t)(R - 1));
      uint64_t max_u = 0;
      for (uint64_t m = mu; m;) {
        int j = __builtin_ctzll(m);
        m &= m - 1;
        uint64_t x = readlane64(v, j);
        max_u = x > max_u ? x : max_u;
      }
      if (is_last && (mu == 0 || v > max_u)) valid = false;
    }
    // ---- additional_qgram_filter (src/filter.c:118-131): >= a+1 values of the same group in [v, v+e] ----
    const uint64_t vm = __ballot(valid);
    bool pass = false;
    if ((uint32_t)__popcll(vm) > (uint32_t)p.a) {
      uint32_t cnt = 0;
      for (uint64_t m = vm; m;) {
        int j = __builtin_ctzll(m);
        m &= m - 1;
        uint64_t x = readlane64(v, j);
        uint32_t gj = (uint32_t)__builtin_amdgcn_readlane((int)grp, j);
        cnt += (uint32_t)(gj == grp && x >= v && x <= v + e64);
      }
      pass = valid && cnt > (uint32_t)p.a;
    }
    // ---- merge_kvec_t_uint64_t (src/filter.c:45-78), group after group ----
    if (__ballot(pass)) {
      uint32_t nA = 0;
      for (uint32_t g = 0; g < (uint32_t)p.step; ++g) {
        const bool mine = pass && grp == g;
        const uint64_t mf = __ballot(mine);
        const uint32_t nF = (uint32_t)__popcll(mf);
        if (nF == 0) continue;
        uint32_t rank = 0;  // position of v among this group's survivors
        for (uint64_t m = mf; m;) {
          int j = __builtin_ctzll(m);
          m &= m - 1;
          uint64_t x = readlane64(v, j);
          rank += (uint32_t)(x < v || (x == v && (uint32_t)j < ln));
        }
        wave_sync_lds();
        if (mine) scatter[rank] = v;
        wave_sync_lds();
        const uint64_t fs = ln < nF ? scatter[ln] : 0;
        uint64_t merged = 0, last_kept = 0;
        uint32_t nB = 0, ia = 0, jf = 0;
        while (ia < nA || jf < nF) {  // wave-uniform two-pointer merge + greedy gap rule
          uint64_t xa = readlane64(cv, (int)(ia < nA ? ia : 0));
          uint64_t xf = readlane64(fs, (int)(jf < nF ? jf : 0));
          bool take_a = ia < nA && (jf >= nF || xa < xf);
          uint64_t x = take_a ? xa : xf;
          ia += take_a ? 1u : 0u;
          jf += take_a ? 0u : 1u;
          if (nB == 0 || x > last_kept + e64) {
            merged = ln == nB ? x : merged;
            ++nB;
            last_kept = x;
          }
        }
        cv = merged;
        nA = nB;
      }
      kept = nA;
    }
  }
  // ---- remove_out_ranged_candidates (src/filter.c:133-144) + hand-over ----
  bool ok = false;
  if (ln < kept) {
    uint32_t seq = (uint32_t)(cv >> 32), pos = (uint32_t)cv;
    uint32_t slen = p.seq_len[seq];
    ok = pos >= (uint32_t)p.e && pos + L + (uint32_t)p.e < slen;
  }
  const uint64_t mo = __ballot(ok);
  const uint32_t n_out = (uint32_t)__popcll(mo);
  uint32_t base = 0;
  if (n_out > 0) {
    if (n_out <= ch.left) {
      base = ch.next;
      ch.next += n_out, ch.left -= n_out;
    } else {
      pad_chunk(p, ch);
      if (ln == 0) base = atomicAdd(&p.ctr[0], kSlotChunk);
      base = bcast0(base);
      ch.next = base + n_out, ch.left = kSlotChunk - n_out;
    }
    if ((unsigned long long)base + n_out > p.cand_cap) {
      if (ln == 0) atomicOr(&p.ctr[1], kFlagCandOverflow);
    } else if (ok) {
      const uint32_t rank = (uint32_t)__popcll(mo & ((1ull << ln) - 1ull)), at = base + rank;
      p.cand[at] = cv - e64;
      p.cand_meta[at] = (read * 2u + strand) | (rank < (n_out & ~7u) ? kMeta16 : 0u);
    }
  }
  if (ln == 0) {
    p.cand_begin[read * 2u + strand] = base;
    p.cand_count[read * 2u + strand] = n_out;
  }
  cand_sum += n_out;
  return true;
}

// ---------------------------------------------------------------------------
// Seed selection DP (src/filter.c:3-43) with the columns of a group spread over a
// lane segment of width W (16, 32 or 64, aligned): row r is
//     M[r][c] = min(M[r][c-1], M[r-1][c] + f)   with M[r][0] = +inf,
// i.e. an exclusive prefix-min of v[c] = M[r-1][c] + f inside the segment, done with
// DPP row shifts (+ row_bcast for W > 16): no LDS round trip per step.
// D[r][c] == 2 ("take") iff v[c] is strictly below that prefix-min; the take bits of
// a row are one ballot.  The traceback (src/filter.c:30-41) then needs R steps per
// group: the chosen column of row r is the highest take bit at or below the column
// chosen for row r+1.  Needs columns <= 64 per group and groups*R <= 64 lanes for the
// rank sort; otherwise the caller uses the one-lane-per-group form.
// Returns M[R][C-1] of group ln (lanes < 2*step).
// ---------------------------------------------------------------------------

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_min_step(uint32_t x, uint32_t inf) {
  uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)inf, (int)x, CTRL, ROW_MASK, 0xF, false);
  return t < x ? t : x;
}

// RT / STEPT / LGT > 0 fix R, step and ceil(k/step) at compile time (seed_fast_kernel); 0 = take them from p.
template <int RT, int STEPT, int LGT>
__device__ uint32_t select_seeds_dpp(const SeedParams &p, int S, const bool *strand_ok, const uint2 *sf, uint32_t smax,
                                     uint32_t W, unsigned long long *take_bits /* LDS [passes][R] */,
                                     Picked *picked) {
  const uint32_t ln = lane_id();
  const int R = RT ? RT : p.R, lg = LGT ? LGT : p.lg, step = STEPT ? STEPT : p.step;
  constexpr int kUnroll = RT ? RT : kMaxR;
  const uint32_t n_groups = 2u * (uint32_t)step;
  const uint32_t per_pass = (uint32_t)kWave / W;
  const uint32_t n_pass = (n_groups + per_pass - 1u) / per_pass;
  const uint32_t c = ln & (W - 1u), slot = ln / W;
  const uint32_t inf = p.inf32;
  uint32_t pre_mine = 0;
  // the group whose traceback this lane does (lanes < n_groups)
  const uint32_t own_strand = ln / (uint32_t)step, own_si = ln % (uint32_t)step;
  const uint32_t own_ncols = (ln < n_groups && strand_ok[own_strand & 1u])
                                 ? (uint32_t)((S - (int)own_si) / step - R * lg + 1) : 0u;  // C - 1
  for (uint32_t ps = 0; ps < n_pass; ++ps) {
    const uint32_t g = ps * per_pass + slot;
    const uint32_t strand = g / (uint32_t)step, si = g % (uint32_t)step;
    const bool g_ok = g < n_groups && strand_ok[strand & 1u];
    const uint32_t ncols = g_ok ? (uint32_t)((S - (int)si) / step - R * lg + 1) : 0u;
    const bool in_seg = c < ncols;
    const uint2 *sfs = sf + (strand & 1u) * smax;
    uint32_t f[kUnroll];
#pragma unroll
    for (int r = 1; r <= kUnroll; ++r)  // all LDS reads of the pass in flight together
      f[r - 1] = (r <= R && in_seg) ? sfs[si + (uint32_t)step * (c + (uint32_t)((r - 1) * lg))].y : 0u;
    uint32_t M = 0;  // M[0][c] = 0
#pragma unroll
    for (int r = 1; r <= kUnroll; ++r) {
      if (r <= R) {
        const uint32_t v = M + f[r - 1];  // uint32 wrap as in the reference
        uint32_t x = in_seg ? v : inf;    // lanes outside a segment must not disturb the min
        x = dpp_min_step<0x111, 0xF>(x, inf);  // row_shr:1
        x = dpp_min_step<0x112, 0xF>(x, inf);  // row_shr:2
        x = dpp_min_step<0x114, 0xF>(x, inf);  // row_shr:4
        x = dpp_min_step<0x118, 0xF>(x, inf);  // row_shr:8
        if (W > 16u) x = dpp_min_step<0x142, 0xA>(x, inf);  // row_bcast:15 into rows 1 and 3
        if (W > 32u) x = dpp_min_step<0x143, 0xC>(x, inf);  // row_bcast:31 into rows 2 and 3
        uint32_t ex = (uint32_t)__builtin_amdgcn_update_dpp((int)inf, (int)x, 0x138, 0xF, 0xF, false);  // wave_shr:1
        ex = (c == 0 || ex > inf) ? inf : ex;  // M[r][0] = (uint32)occurrence_table_size
        const bool take = in_seg && v < ex;    // strict: ties go horizontal (src/filter.c:20)
        M = take ? v : ex;
        const unsigned long long bits = __ballot(take);
        if (ln == 0) take_bits[ps * (uint32_t)R + (uint32_t)(r - 1)] = bits;
      }
    }
    // M[R][C-1] of the group this lane will trace
    const uint32_t own_pass = ln / per_pass, own_slot = ln % per_pass;
    const uint32_t src = own_slot * W + (own_ncols ? own_ncols - 1u : 0u);
    const uint32_t got = __shfl(M, (int)src);
    if (ln < n_groups && own_pass == ps) pre_mine = got;
  }
  wave_sync_lds();
  // traceback, one lane per group
  if (ln < n_groups) {
    Picked *out = picked + (size_t)ln * (uint32_t)R;
    const uint32_t own_pass = ln / per_pass, own_slot = ln % per_pass;
    unsigned long long rows[kUnroll];
#pragma unroll
    for (int r = 1; r <= kUnroll; ++r) rows[r - 1] = (r <= R && own_ncols) ? take_bits[own_pass * (uint32_t)R + (uint32_t)(r - 1)] : 0ull;
    const uint2 *sfo = sf + (own_strand & 1u) * smax;
    int col = (int)own_ncols - 1, n_out = 0;
    bool alive = own_ncols != 0;
    uint32_t sidx[kUnroll];
#pragma unroll
    for (int r = kUnroll; r >= 1; --r) {
      sidx[r - 1] = 0xFFFFFFFFu;
      if (r <= R && alive) {
        unsigned long long seg = (rows[r - 1] >> (own_slot * W)) & ((2ull << col) - 1ull);
        if (seg == 0) {
          alive = false;  // reached column 0 before taking R seeds (UB in reference): the rest stay zero
        } else {
          col = 63 - __builtin_clzll(seg);
          sidx[r - 1] = own_si + (uint32_t)step * (uint32_t)(col + (r - 1) * lg);
        }
      }
    }
#pragma unroll
    for (int r = kUnroll; r >= 1; --r) {  // picked in traceback order: row R first
      if (r <= R) {
        Picked q{0, 0, 0, 0};
        if (sidx[r - 1] != 0xFFFFFFFFu) {
          uint2 s2 = sfo[sidx[r - 1]];
          q = Picked{sidx[r - 1], s2.x, s2.y, 0};
        }
        // rows that were never reached come after the taken ones, as zeros
        int slot_out = sidx[r - 1] != 0xFFFFFFFFu ? n_out++ : -1;
        if (slot_out >= 0) out[slot_out] = q;
      }
    }
    for (int t = n_out; t < R; ++t) out[t] = Picked{0, 0, 0, 0};
  }
  wave_sync_lds();
  // qsort(compare_seed) (src/filter.c:204): stable by ascending frequency, as a rank sort with lane = (group, seed)
  {
    const uint32_t g = ln / (uint32_t)R, t = ln - g * (uint32_t)R;
    const bool act = ln < n_groups * (uint32_t)R;
    Picked mine{0, 0, 0, 0};
    uint32_t rank = 0;
    if (act) {
      const Picked *grp = picked + (size_t)g * (uint32_t)R;
      mine = grp[t];
      for (uint32_t u = 0; u < (uint32_t)R; ++u) {
        uint32_t fu = grp[u].freq;
        rank += (uint32_t)(fu < mine.freq || (fu == mine.freq && u < t));
      }
    }
    wave_sync_lds();
    if (act) picked[(size_t)g * (uint32_t)R + rank] = mine;
  }
  return pre_mine;
}

// ---------------------------------------------------------------------------
// seed + filter kernel: one wave per read, grid-stride over the batch
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) seed_filter_kernel(SeedParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t ln = lane_id();
  const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform, and known to be
  const uint32_t waves_per_block = blockDim.x >> 6;
  uint8_t *wbase = smem + (size_t)wave_in_block * p.lay.wave_bytes;
  uint32_t *pkw = (uint32_t *)(wbase + p.lay.pkw);
  uint32_t *nkw = (uint32_t *)(wbase + p.lay.nkw);
  uint2 *sf = (uint2 *)(wbase + p.lay.sf);
  uint32_t *dp_rows = (uint32_t *)(wbase + p.lay.dp_rows);
  uint32_t *dp_bits = (uint32_t *)(wbase + p.lay.dp_bits);
  Picked *picked = (Picked *)(wbase + p.lay.picked);
  uint32_t *rb = (uint32_t *)(wbase + p.lay.rb);
  Bufs lds;
  lds.X = (uint64_t *)(wbase + p.lay.X);
  lds.F = (uint64_t *)(wbase + p.lay.F);
  lds.A = (uint64_t *)(wbase + p.lay.A);
  lds.B = (uint64_t *)(wbase + p.lay.B);
  lds.xcap = p.lay.xcap, lds.fcap = p.lay.fcap, lds.ccap = p.lay.ccap;

  const int k = p.k, step = p.step, R = p.R, lg = p.lg;
  const uint32_t hash_mask = (k == 16) ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
  const uint32_t smax = p.lay.smax, cmax = p.lay.cmax, cw = p.lay.cw;
  unsigned long long pre_sum = 0, cand_sum = 0;
  SlotChunk chunk;
  Prof prof;

  const uint32_t wave_global = blockIdx.x * waves_per_block + wave_in_block;
  const uint32_t n_waves = gridDim.x * waves_per_block;

  const uint32_t n_items = p.work_queue ? min(p.ctr[2], p.slow_cap) : p.n_reads;
  for (uint32_t item = wave_global; item < n_items; item += n_waves) {
    const uint32_t read = p.work_queue ? p.work_queue[item] : item;
    if (read == kInvalidRead) continue;
    STAMP_START(prof);
    const uint64_t off = p.read_off[read];
    const uint32_t L = (uint32_t)(p.read_off[read + 1] - off);
    const uint8_t *seq = p.bases + off;
    const int S = (int)L - k + 1;  // num_seeds_in_read

    // ---- gates (src/filter.c:161-172) + the (L,e,a) shapes on which the reference DP is undefined ----
    bool shape_ok = S > 0 && R <= S / step;
    if (shape_ok) {
      int g_min = (S - (step - 1)) / step;
      shape_ok = g_min - R * lg + 2 >= 2;
    }
    if (!shape_ok || (uint32_t)S > smax) {
      // (S > smax cannot happen: the host sizes the layout from the longest read of the batch)
      if (ln < 2) {
        p.cand_begin[read * 2u + ln] = 0;
        p.cand_count[read * 2u + ln] = 0;
      }
      continue;
    }

    // ---- encode: lane handles bases 4*lane.. (+256 per round); 2 bits per base, first base in the top bits ----
    uint32_t n_fwd_amb = 0, n_rev_amb = 0;
    {
      const uint32_t n_words = (L + 15u) / 16u + 2u;
      for (uint32_t w = ln; w < n_words; w += kWave) {
        pkw[w] = 0;
        nkw[w] = 0;
      }
      wave_sync_lds();
      uint8_t *pkb = (uint8_t *)pkw, *nkb = (uint8_t *)nkw;
      for (uint32_t b0 = 0; b0 < L; b0 += 256u) {
        uint32_t idx = b0 + 4u * ln;
        if (idx < L) {
          uint32_t pv = 0, nv = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint32_t c = 0, isn = 0;
            if (idx + q < L) {
              uint32_t code = base_code(seq[idx + q]);
              isn = code >> 2;
              c = code & 3u & (isn - 1u);  // N -> A (src/utils.h:92)
              // hash_all_seeds_in_sequence counts ambiguous bases at offsets >= k only (src/utils.h:108-114);
              // on the reverse strand offset L-1-i >= k
              n_fwd_amb += isn & (uint32_t)(idx + q >= (uint32_t)k);
              n_rev_amb += isn & (uint32_t)(L - 1u - (idx + q) >= (uint32_t)k);
            }
            pv = (pv << 2) | c;
            nv = (nv << 2) | (isn * 3u);
          }
          uint32_t byte_addr = (idx >> 4) * 4u + (3u - ((idx >> 2) & 3u));  // big-endian inside each dword
          pkb[byte_addr] = (uint8_t)pv;
          nkb[byte_addr] = (uint8_t)nv;
        }
      }
      if (__any(n_fwd_amb | n_rev_amb)) {
        for (int d = 32; d >= 1; d >>= 1) {
          n_fwd_amb += __shfl_xor(n_fwd_amb, d);
          n_rev_amb += __shfl_xor(n_rev_amb, d);
        }
      }
      wave_sync_lds();
    }
    const bool strand_ok[2] = {n_fwd_amb <= (uint32_t)p.e, n_rev_amb <= (uint32_t)p.e};  // src/filter.c:180-182
    STAMP(prof, 0);

    // ---- hashes + CSR lookups (src/utils.h:101-117, src/index.h:22-28); lane j owns seed j of the forward
    //      strand and seed S-1-j of the reverse strand (the reverse complement of the same k-mer) ----
    for (int j0 = 0; j0 < S; j0 += kWave) {
      int j = j0 + (int)ln;
      if (j < S) {
        uint32_t w = (uint32_t)j >> 4, sh = 2u * ((uint32_t)j & 15u);
        uint64_t pw = ((uint64_t)pkw[w] << 32) | pkw[w + 1];
        uint64_t nw = ((uint64_t)nkw[w] << 32) | nkw[w + 1];
        uint32_t hf = (uint32_t)(pw >> (64 - 2 * k - sh)) & hash_mask;
        uint32_t nm = (uint32_t)(nw >> (64 - 2 * k - sh)) & hash_mask;
        uint32_t hr = reverse_pairs((~hf) & ~nm & hash_mask, k);
        if (strand_ok[0]) {
          uint32_t lo = p.lookup[hf], hi = p.lookup[hf + 1];
          sf[j] = make_uint2(lo, hi - lo);
        }
        if (strand_ok[1]) {
          uint32_t lo = p.lookup[hr], hi = p.lookup[hr + 1];
          sf[smax + (uint32_t)(S - 1 - j)] = make_uint2(lo, hi - lo);
        }
      }
    }
    wave_sync_lds();
    STAMP(prof, 1);

    // ---- seed selection DP (src/filter.c:3-43): columns over lanes when all groups fit in one wave,
    //      else one lane per (strand, phase group) ----
    uint32_t pre_g = 0;
    uint32_t dp_w = 0;  // lane segment per group: 16, 32 or 64; 0 = does not fit
    if (2u * (uint32_t)step * (uint32_t)R <= (uint32_t)kWave && R <= kMaxR) {
      uint32_t widest = (uint32_t)(S / step - R * lg + 1);  // phase 0 has the most columns
      dp_w = widest <= 16u ? 16u : widest <= 32u ? 32u : widest <= 64u ? 64u : 0u;
    }
    if (dp_w) {
      pre_g = select_seeds_dpp<0, 0, 0>(p, S, strand_ok, sf, smax, dp_w, (unsigned long long *)dp_bits, picked);
    } else if (ln < 2u * (uint32_t)step && strand_ok[ln / (uint32_t)step]) {
      const uint32_t strand = ln / (uint32_t)step, si = ln % (uint32_t)step;
      const int G = (S - (int)si) / step;
      const int C = G - R * lg + 2;  // num_columns
      uint32_t *rows = dp_rows + (size_t)ln * 2u * cmax;
      uint32_t *bits = dp_bits + (size_t)ln * (uint32_t)R * cw;
      const uint2 *sfs = sf + strand * smax;
      for (int c = 1; c < C; ++c) rows[c] = 0;  // M[0][c] = 0
      uint32_t left = p.inf32;
      for (int row = 1; row <= R; ++row) {
        const uint32_t *prev = rows + (size_t)((row - 1) & 1) * cmax;
        uint32_t *cur = rows + (size_t)(row & 1) * cmax;
        left = p.inf32;  // M[row][0]
        uint32_t acc = 0;
        for (int col = 1; col < C; ++col) {
          uint32_t pos = (uint32_t)(col + (row - 1) * lg - 1);
          uint32_t with_new = prev[col] + sfs[si + (uint32_t)step * pos].y;
          bool take = with_new < left;  // strict: ties go horizontal (src/filter.c:20)
          left = take ? with_new : left;
          cur[col] = left;
          acc |= (uint32_t)take << (col & 31);
          if ((col & 31) == 31 || col == C - 1) {
            bits[(uint32_t)(row - 1) * cw + ((uint32_t)col >> 5)] = acc;
            acc = 0;
          }
        }
      }
      pre_g = left;  // M[R][C-1]
      // traceback, right to left (src/filter.c:30-41); seeds never reached stay all-zero (UB in reference)
      Picked *out = picked + (size_t)ln * (uint32_t)R;
      for (int t = 0; t < R; ++t) out[t] = Picked{0, 0, 0, 0};
      int r = R, c = C - 1, n_out = 0;
      while (r > 0 && c > 0) {
        uint32_t bit = (bits[(uint32_t)(r - 1) * cw + ((uint32_t)c >> 5)] >> (c & 31)) & 1u;
        if (bit) {
          uint32_t sidx = si + (uint32_t)step * (uint32_t)(c + (r - 1) * lg - 1);
          uint2 s = sfs[sidx];
          out[n_out++] = Picked{sidx, s.x, s.y, 0};
          --r;
        } else {
          --c;
        }
      }
      // qsort(compare_seed) (src/filter.c:204): stable by ascending frequency
      for (int i = 1; i < R; ++i) {
        Picked t = out[i];
        int j = i;
        while (j > 0 && t.freq < out[j - 1].freq) {
          out[j] = out[j - 1];
          --j;
        }
        out[j] = t;
      }
    }
    wave_sync_lds();
    STAMP(prof, 2);

    // ---- per strand: lists, filter, de-dup, clip, emit ----
    for (uint32_t strand = 0; strand < 2; ++strand) {
      if (!strand_ok[strand]) {
        if (ln == 0) {
          p.cand_begin[read * 2u + strand] = 0;
          p.cand_count[read * 2u + strand] = 0;
        }
        continue;
      }
      // per-strand pre-filter count: uint32 sum of the groups' M[R][C-1] (src/filter.c:202)
      uint32_t pre = 0, pre_max = 0;
      unsigned long long pre_wide = 0;
      for (int si = 0; si < step; ++si) {
        uint32_t v = __shfl(pre_g, (int)(strand * (uint32_t)step) + si);
        pre += v;
        pre_wide += v;
        pre_max = v > pre_max ? v : pre_max;
      }
      pre_sum += pre;
      const Picked *pk = picked + (size_t)strand * (uint32_t)step * (uint32_t)R;
      uint64_t *list = nullptr;
      STAMP(prof, 3);
      if (strand_small(p, pk, lds.X, read, strand, L, cand_sum, chunk)) {
        STAMP(prof, 8);
        continue;
      }
      uint32_t n = strand_lists<false>(p, pk, rb, lds, &list, prof);
      if (n < 0xFFFFFFFEu) {
        clip_and_emit<false>(p, read, strand, L, list, n, list == lds.A ? lds.B : lds.A, cand_sum, chunk);
      } else if (n == 0xFFFFFFFEu) {
        if (ln == 0) atomicOr(&p.ctr[1], kFlagTooLarge);
      } else {
        // lists do not fit in LDS: same code over a slice of the global arena.
        // staged <= max group total, survivors <= staged, candidates <= sum of the groups
        unsigned long long want = 2ull * pre_max + 2ull * pre_wide + 8ull;
        unsigned long long base = 0;
        if (ln == 0) {
          base = atomicAdd(&p.arena_ctr[0], want);
          atomicAdd(&p.arena_ctr[1], want);
        }
        base = ((unsigned long long)bcast0((uint32_t)(base >> 32)) << 32) | bcast0((uint32_t)base);
        if (base + want > p.arena_cap) {
          if (ln == 0) atomicOr(&p.ctr[1], kFlagArenaOverflow);
          if (ln == 0) {
            p.cand_begin[read * 2u + strand] = 0;
            p.cand_count[read * 2u + strand] = 0;
          }
        } else {
          Bufs g;
          g.X = p.arena + base;
          g.F = g.X + pre_max + 2;
          g.A = g.F + pre_max + 2;
          g.B = g.A + pre_wide + 2;
          g.xcap = pre_max + 1u, g.fcap = pre_max + 1u;
          g.ccap = pre_wide > 0x7ffffff0ull ? 0x7ffffff0u : (uint32_t)pre_wide + 1u;
          // (the wave's LDS list space X, F, A, B — one contiguous piece, make_layout — holds the searches' samples there)
          uint32_t n2 = strand_lists_big(p, pk, rb, g, lds.X, p.lay.xcap + p.lay.fcap + 2u * p.lay.ccap, &list, prof);
          if (n2 < 0xFFFFFFFEu) {
            clip_and_emit<true>(p, read, strand, L, list, n2, list == g.A ? g.B : g.A, cand_sum, chunk);
          } else {
            if (ln == 0) atomicOr(&p.ctr[1], kFlagTooLarge);
          }
        }
      }
      wave_sync_lds();
      STAMP(prof, 7);
    }
  }
  pad_chunk(p, chunk);
#ifdef FEM_STAMPS
  prof.flush();
#endif
  if (ln == 0) {
    if (pre_sum) atomicAdd(&p.stats[0], pre_sum);
    if (cand_sum) atomicAdd(&p.stats[1], cand_sum);
  }
}

// ---------------------------------------------------------------------------
// verification: one lane per candidate (src/align.c:4-51, 102-277)
// ---------------------------------------------------------------------------
struct VerifyParams {
  const uint8_t *bases;       // 16 bytes of padding in front of the batch's characters, 64 behind
  const uint64_t *read_off;
  const uint8_t *planes;      // bit planes of the base codes 0..4, all sequences concatenated (plane_window; see verify_kernel)
  const uint64_t *seq_off;
  const uint64_t *cand;
  const uint32_t *cand_meta;
  const uint32_t *ctr;    // [0] = candidate slots handed out by the seed kernels, [1] = overflow flags
  uint32_t cand_cap;
  int32_t e;
  uint8_t *ed;
  int16_t *end;
  // MappingStats (src/map.c:37,48,51): accepted candidates per read (zeroed by the launcher), their sum and the reads
  // with at least one -> stats[2], stats[3]
  uint32_t *n_map;
  unsigned long long *stats;
};

__device__ __forceinline__ uint4 load_u128_unaligned(const uint8_t *p) {
  uint4 w;
  __builtin_memcpy(&w, p, 16);
  return w;
}

__device__ __forceinline__ uint4 plane_window(const uint8_t *planes, int q, uint64_t at) {  // fem_planes.hip.h
  return load_u128_unaligned(plane_addr(planes, q, at));
}
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p) {
  uint32_t w;
  __builtin_memcpy(&w, p, 4);  // gfx950 global loads take any byte address
  return w;
}

// ---------------------------------------------------------------------------------------------------------
// verify_candidates (src/align.c:4-51): one lane per candidate walks banded_edit_distance (src/align.c:102-147; the
// 16-bit SSE form of :149-277 differs only in the word width, `wm`).
// The kernel is bound by the number of divergent load instructions (every lane walks its own read and window), so
// both sides are fetched in the widest units that hold them:
//   * reference: three bit planes of the base codes (bit q of code(ref[i]) at bit i of plane q).  One unaligned
//     16-byte load per plane covers the windows of six 16-column steps; Peq[c] for a column is a three-way XNOR of
//     the planes, shifted;
//   * read: 16 characters per load, four loads back to back into the lane's LDS words (they share 64-byte sectors;
//     one load per step missed the L1 every time), decoded four at a time (SWAR), byte-reversed and complemented on
//     the reverse strand (prepare_negative_sequence_at, src/sequence_batch.h:90-98).
// MappingStats (src/map.c:37,48,51): an accepted candidate adds one to its read's n_map — the lane that finds it at zero
// counts the read as mapped — and a block adds its sums to the two counters once (round 1's returning atomics on the two
// COUNTERS, same address for every lane, had cost as much as the rest of the kernel; a separate count_mappings_kernel did
// it until round 3, 0.07 ms between one batch's join and the next's).
// ---------------------------------------------------------------------------------------------------------
struct MyersState {
  uint32_t VP, VN;
  int score;
};

// char -> 2-bit code for four bases at once: code per byte 0..3 (0 where the base is not A/C/G/T), nflag per byte 0/1
__device__ __forceinline__ void decode4(uint32_t chars, uint32_t complement, uint32_t &code, uint32_t &nflag) {
  const uint32_t t = (chars >> 1) & 0x03030303u;    // A 0, C 1, G 3, T 2
  const uint32_t c = t ^ ((t >> 1) & 0x01010101u);  // A 0, C 1, G 2, T 3
  const uint32_t upper = chars & 0xDFDFDFDFu;
  const uint32_t expect = __builtin_amdgcn_perm(0u, 0x54474341u /* "ACGT" */, c);
  const uint32_t z = upper ^ expect;  // zero byte <=> one of ACGT in either case (src/utils.h:72)
  nflag = ((((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) >> 7) & 0x01010101u;
  code = (c ^ complement) & ~(nflag * 3u);  // complement = 0x03030303 on the reverse strand (3 - code); N stays N
}

// The plane windows are consumed sixteen bits at a time: the 32 bits at bit `off` (< 32) of the window's head, and the
// window moving on by sixteen bits — four v_alignbit per plane and step.  (Indexing the window's words by the step made
// the compiler keep the windows in scratch memory: six scratch loads per step in the verify kernel.)
__device__ __forceinline__ uint32_t window_head(const uint4 &w, uint32_t off) { return __builtin_amdgcn_alignbit(w.y, w.x, off); }
__device__ __forceinline__ void window_advance16(uint4 &w) {
  w.x = __builtin_amdgcn_alignbit(w.y, w.x, 16u), w.y = __builtin_amdgcn_alignbit(w.z, w.y, 16u);
  w.z = __builtin_amdgcn_alignbit(w.w, w.z, 16u), w.w >>= 16;
}

// One column (src/align.c:118-133).  B0..B2: the step's plane windows; m0..m2: the read base's code bits as masks;
// j = column inside the step.
__device__ __forceinline__ void myers_column(MyersState &m, uint32_t B0, uint32_t B1, uint32_t B2, uint32_t m0, uint32_t m1,
                                             uint32_t m2, uint32_t j, uint32_t width, uint32_t wm) {
  const uint32_t eq = __builtin_amdgcn_ubfe(~((B0 ^ m0) | (B1 ^ m1) | (B2 ^ m2)), j, width);  // Peq[text[col]] over the band
  uint32_t X = eq | m.VN;
  const uint32_t D0 = ((((X & m.VP) + m.VP) ^ m.VP) | X) & wm;
  const uint32_t HN = m.VP & D0;
  const uint32_t HP = (m.VN | ~(m.VP | D0)) & wm;
  X = D0 >> 1;
  m.VN = X & HP;
  m.VP = (HN | ~(X | HP)) & wm;
  m.score += 1 - (int)(D0 & 1u);
}

constexpr int kStepsPerPlaneLoad = 6;  // 7 (bit offset) + 16 * 5 + 16 + 2 * 7 (band) bits <= 128

__global__ void __launch_bounds__(256) verify_kernel(VerifyParams p) {
  // a scratch buffer overflowed while seeding: slots may be unwritten, the host grows the buffer and re-runs the batch
  if (p.ctr[1] != 0) return;
  const uint32_t total = min(p.ctr[0], p.cand_cap);
  const uint32_t stride = gridDim.x * blockDim.x;
  const int e = p.e;
  const uint32_t width = 2u * (uint32_t)e + 1u;
  // Four 16-character chunks of the lane's read wait in LDS: fetched back to back they share their 64-byte sectors,
  // fetched one per step (16 columns = microseconds apart at eight waves per SIMD) every chunk missed the L1 again
  // (1.26 -> 1.14 ms at C2).  Eight chunks at once cost three waves per SIMD and were slower.
  constexpr int kStageChunks = 4;
  // (16 KB exactly, the block sums laid over it at the end)
  __shared__ uint4 stage[kStageChunks][256];
  uint32_t(*part)[4] = (uint32_t(*)[4]) & stage[0][0];
  uint32_t mappings = 0, mapped = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const uint32_t meta = p.cand_meta[i];
    if (meta == kInvalidMeta) {
      p.ed[i] = 0xFF, p.end[i] = 0;
      continue;
    }
    const uint32_t read = (meta & ~kMeta16) >> 1, strand = meta & 1u;
    const uint32_t wm = (meta & kMeta16) ? 0xFFFFu : 0xFFFFFFFFu;
    const uint64_t c = p.cand[i];
    const uint64_t pat = p.seq_off[(uint32_t)(c >> 32)] + (uint32_t)c;  // base index of pattern[0]
    const uint64_t off = p.read_off[read];
    const int L = (int)(p.read_off[read + 1] - off);
    const uint8_t *rd = p.bases + off;
    const uint32_t complement = strand ? 0x03030303u : 0u;
    MyersState m{0, 0, 0};
    bool rejected = false;
    const int n_steps = (L + 15) >> 4;
    // the reverse strand's chunk comes from the far end; the last, partial one may start in front of the read
    auto text_chunk = [&](int col) { return load_u128_unaligned(strand == 0 ? rd + col : rd + (L - 16 - col)); };
    auto plane_chunk = [&](int q, int col) { return plane_window(p.planes, q, (pat + (uint32_t)col) >> 3); };
    uint4 P0 = make_uint4(0, 0, 0, 0), P1 = P0, P2 = P0, P0n = P0, P1n = P0, P2n = P0;
    if (n_steps > 0) {
      P0n = plane_chunk(0, 0), P1n = plane_chunk(1, 0), P2n = plane_chunk(2, 0);
    }
    const uint32_t pat_bit = (uint32_t)pat & 7u;
    for (int step = 0; step < n_steps && !rejected; ++step) {
      const int col = step << 4, sub = step % kStepsPerPlaneLoad;
      if (step % kStageChunks == 0) {
        uint4 t4[kStageChunks];
#pragma unroll
        for (int c = 0; c < kStageChunks; ++c) t4[c] = step + c < n_steps ? text_chunk(col + 16 * c) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < kStageChunks; ++c) stage[c][threadIdx.x] = t4[c];
      }
      const uint4 r = stage[step % kStageChunks][threadIdx.x];
      if (sub == 0) {  // loads of the next stretch are issued one stretch ahead, those of the next step one step ahead
        P0 = P0n, P1 = P1n, P2 = P2n;
        if (step + kStepsPerPlaneLoad < n_steps) {
          const int nc = col + 16 * kStepsPerPlaneLoad;
          P0n = plane_chunk(0, nc), P1n = plane_chunk(1, nc), P2n = plane_chunk(2, nc);
        }
      } else {
        window_advance16(P0), window_advance16(P1), window_advance16(P2);
      }
      // window of this step: pattern[col .. col + 16 + 2e), bit j <-> pattern[col + j]; (pat + 96 k) & 7 == pat & 7
      const uint32_t b0 = window_head(P0, pat_bit), b1 = window_head(P1, pat_bit), b2 = window_head(P2, pat_bit);
      uint32_t cw[4], nw[4];
      {
        const uint32_t w0 = strand ? __builtin_bswap32(r.w) : r.x, w1 = strand ? __builtin_bswap32(r.z) : r.y;
        const uint32_t w2 = strand ? __builtin_bswap32(r.y) : r.z, w3 = strand ? __builtin_bswap32(r.x) : r.w;
        decode4(w0, complement, cw[0], nw[0]);
        decode4(w1, complement, cw[1], nw[1]);
        decode4(w2, complement, cw[2], nw[2]);
        decode4(w3, complement, cw[3], nw[3]);
      }
      if (col + 16 <= L) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)cw[q >> 2], 8 * (q & 3), 1);
          const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int)cw[q >> 2], 8 * (q & 3) + 1, 1);
          const uint32_t m2 = (uint32_t)__builtin_amdgcn_sbfe((int)nw[q >> 2], 8 * (q & 3), 1);
          myers_column(m, b0, b1, b2, m0, m1, m2, (uint32_t)q, width, wm);
        }
      } else {  // up to fifteen trailing columns
        const uint64_t clo = ((uint64_t)cw[1] << 32) | cw[0], chi = ((uint64_t)cw[3] << 32) | cw[2];
        const uint64_t nlo = ((uint64_t)nw[1] << 32) | nw[0], nhi = ((uint64_t)nw[3] << 32) | nw[2];
        for (int q = 0; q < L - col; ++q) {
          const uint32_t cb = (uint32_t)((q < 8 ? clo : chi) >> (8 * (q & 7)));
          const uint32_t nb = (uint32_t)((q < 8 ? nlo : nhi) >> (8 * (q & 7)));
          myers_column(m, b0, b1, b2, 0u - (cb & 1u), 0u - ((cb >> 1) & 1u), 0u - (nb & 1u), (uint32_t)q, width, wm);
        }
      }
      // the score along the band's lowest diagonal never decreases, so testing the early-reject threshold
      // (src/align.c:128-130) once per step rejects exactly the candidates a per-column test would
      rejected = m.score > 3 * e;
    }
    int score = m.score;
    int best = score, endp = L - 1;
    if (!rejected) {
      for (int j = 0; j < 2 * e; ++j) {  // first strict minimum (src/align.c:135-146)
        score += (int)((m.VP >> j) & 1u) - (int)((m.VN >> j) & 1u);
        if (score < best) {
          best = score;
          endp = L + j;
        }
      }
    }
    const bool accepted = !rejected && best <= e;
    p.ed[i] = accepted ? (uint8_t)best : (uint8_t)0xFF;
    p.end[i] = accepted ? (int16_t)endp : (int16_t)0;
    // The read's first accepted candidate (whichever atomic comes first) counts the read as mapped.  Neighbouring lanes that
    // accepted candidates of the SAME read add their count with one atomic (round 5): a read inside a repeat has a thousand
    // candidates in a row, and a thousand atomics on one address come one after the other (~10 ns each) — on the repeat-rich
    // 3 Gbp reference that was 7.4 of the kernel's 16.3 ms per 2.5 M reads.  Where every read has one candidate (C2, C3)
    // every lane is its own head and nothing changes but a shuffle and two ballots.
    {
      const uint32_t ln = lane_id();
      const uint64_t acc = __ballot(accepted);
      const uint32_t prev_read = __shfl_up(read, 1);
      const bool joins_prev = accepted && ln > 0u && ((acc >> (ln - 1u)) & 1ull) && prev_read == read;
      const uint64_t heads = __ballot(accepted && !joins_prev);
      if (accepted && !joins_prev) {
        const uint64_t stop = (heads | ~acc) >> ln >> 1;  // the next head, or the next lane that accepted nothing
        const uint32_t run = stop ? (uint32_t)__builtin_ctzll(stop) + 1u : 64u - ln;
        mapped += (uint32_t)(atomicAdd(&p.n_map[read], run) == 0u);
      }
      if (accepted) ++mappings;
    }
  }
  // one pair of atomics per block (same-address atomics complete at ~10 ns each: a pair per wave had cost more than the
  // verification itself)
  for (int d = 32; d >= 1; d >>= 1) mappings += __shfl_xor(mappings, d), mapped += __shfl_xor(mapped, d);
  __syncthreads();  // (every wave is through with its staged chunks)
  if (lane_id() == 0) part[0][threadIdx.x >> 6] = mappings, part[1][threadIdx.x >> 6] = mapped;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = part[0][0] + part[0][1] + part[0][2] + part[0][3], b = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    if (a) atomicAdd(&p.stats[2], (unsigned long long)a);
    if (b) atomicAdd(&p.stats[3], (unsigned long long)b);
  }
}

// ---- the batch's outcome in the form that crosses the link (fem_dev_fetch_packed, include/fem_hip.h) ----
// The seed kernels hand out candidate slots in chunks per wave, so a strand's candidates sit wherever its wave's chunk was
// (cand_begin / cand_count: 16 bytes per read) between padding slots.  Here every block of 256 strands claims one stretch of
// the packed arrays (one atomic per block) and lays its strands' candidates into it in strand order: what goes home is one
// byte per strand (the count; 255 = listed in big[]), one uint32 per 256 strands (where their stretch starts) and 11 bytes
// per candidate, none per padding slot — 11.4 instead of 30.8 bytes per read at BASELINE config 2 (0.855 candidates per read).
struct PackParams {
  const uint32_t *cand_begin, *cand_count;  // 2 n_reads each
  const uint64_t *cand;
  const uint8_t *ed;
  const int16_t *end;
  const uint32_t *ctr;      // [1] = overflow flags of the seed kernels (the batch is run again: nothing to pack)
  uint32_t n_strands;
  uint8_t *count8;
  uint32_t *seg_begin;      // ceil(n_strands / 256)
  uint64_t *pcand;
  uint8_t *ped;
  int16_t *pend;
  uint32_t pcap;            // entries the packed arrays hold (>= the candidate slots handed out)
  uint32_t *cursor;         // [0] packed candidates so far, [1] entries of big[] asked for
  uint2 *big;               // (strand, count) of the strands with 255 candidates or more
  uint32_t big_cap;
};
constexpr uint32_t kPackSegs = 16;  // segments of 256 strands one block lays out behind ONE claim of the packed arrays
__global__ void __launch_bounds__(256) pack_results_kernel(PackParams p) {
  // (A claim per segment was 19 532 atomics on one address per batch of 2.5 M reads — ~10 ns each, one after the other:
  //  0.23 ms, nine tenths of the kernel.  A block now claims for sixteen segments at once.)
  if (p.ctr[1] != 0) return;
  __shared__ uint32_t part[kPackSegs * 4u];  // [segment][wave]: candidates; then their exclusive prefix
  __shared__ uint32_t block_base;
  const uint32_t seg0 = blockIdx.x * kPackSegs;
  const uint32_t wave = threadIdx.x >> 6;
  uint32_t cnt[kPackSegs], beg[kPackSegs], incl[kPackSegs];
#pragma unroll
  for (uint32_t i = 0; i < kPackSegs; ++i) {
    const uint32_t s = (seg0 + i) * 256u + threadIdx.x;
    const bool in = s < p.n_strands;
    cnt[i] = in ? p.cand_count[s] : 0u;
    beg[i] = in ? p.cand_begin[s] : 0u;
  }
#pragma unroll
  for (uint32_t i = 0; i < kPackSegs; ++i) {
    uint32_t x = cnt[i];  // inclusive prefix inside the wave
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(x, d);
      if ((int)lane_id() >= d) x += o;
    }
    incl[i] = x;
    if (lane_id() == 63) part[i * 4u + wave] = x;
  }
  __syncthreads();
  if (threadIdx.x < 64u) {  // the 64 (segment, wave) sums -> exclusive prefix; one claim for the block
    const uint32_t v = part[threadIdx.x];
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(x, d);
      if ((int)threadIdx.x >= d) x += o;
    }
    part[threadIdx.x] = x - v;
    if (threadIdx.x == 63u) block_base = x ? atomicAdd(&p.cursor[0], x) : 0u;
  }
  __syncthreads();
  const uint32_t n_seg = (p.n_strands + 255u) / 256u;
#pragma unroll
  for (uint32_t i = 0; i < kPackSegs; ++i) {
    const uint32_t seg = seg0 + i, s = seg * 256u + threadIdx.x;
    if (threadIdx.x == 0 && seg < n_seg) p.seg_begin[seg] = block_base + part[i * 4u];
    if (s >= p.n_strands) continue;
    const uint32_t c = cnt[i], at = block_base + part[i * 4u + wave] + incl[i] - c;
    p.count8[s] = (uint8_t)(c < 255u ? c : 255u);
    if (c >= 255u) {
      const uint32_t j = atomicAdd(&p.cursor[1], 1u);
      if (j < p.big_cap) p.big[j] = make_uint2(s, c);
    }
    for (uint32_t k = 0; k < c; ++k) {
      if (at + k >= p.pcap) break;  // (cannot happen: the packed arrays hold every slot handed out)
      p.pcand[at + k] = p.cand[beg[i] + k];
      p.ped[at + k] = p.ed[beg[i] + k];
      p.pend[at + k] = p.end[beg[i] + k];
    }
  }
}

// bit q of code(text[i]) -> bit i of plane q (q = 0..2); plane 3: the character as uploaded is none of "ACGTN" (lower
// case, IUPAC codes: it then equals no read character the device traceback compares it with).  One thread per byte of
// the planes (eight bases).
__global__ void ref_planes_kernel(const uint8_t *codes, const uint8_t *raw, uint64_t n_bytes, uint8_t *planes) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_bytes; b += stride) {
    uint64_t w, r;
    __builtin_memcpy(&w, codes + 8 * b, 8);
    __builtin_memcpy(&r, raw + 8 * b, 8);
    uint32_t o[4] = {0, 0, 0, 0};
    for (int k = 0; k < 8; ++k) {
      const uint32_t c = (uint32_t)(w >> (8 * k)) & 0xFFu, ch = (uint32_t)(r >> (8 * k)) & 0xFFu;
      o[0] |= (c & 1u) << k, o[1] |= ((c >> 1) & 1u) << k, o[2] |= ((c >> 2) & 1u) << k;
      o[3] |= (uint32_t)(ch != ((0x4E54474341ull >> (8u * c)) & 0xFFu)) << k;  // "ACGTN"[code]
    }
    // byte b of a plane sits in group b / 16 and, as that group's look-ahead, behind the sixteen bytes of group b / 16 - 1
    const uint64_t g = b >> 4, r16 = b & 15u;
    for (int q = 0; q < 4; ++q) {
      planes[g * kPlaneGroup + (uint32_t)q * 32u + r16] = (uint8_t)o[q];
      if (g) planes[(g - 1) * kPlaneGroup + (uint32_t)q * 32u + 16u + r16] = (uint8_t)o[q];
    }
  }
}

constexpr uint32_t kSummaryBuckets = 24;  // buckets per summary word (SeedParams::summary)

// summary word and bit positions of bucket h: q = h / 24 by a float multiply (exact for h >> 3 < 2^22, checked
// exhaustively on the host: tests/test_host.py) — the integer multiplies are quarter rate
__device__ __forceinline__ void summary_slot(uint32_t h, uint32_t &q, uint32_t &r) {
  q = (uint32_t)((float)(h >> 3) * 0.33333334f);
  r = h - __umul24(q, kSummaryBuckets);  // (24-bit multiplies are full rate; q < 2^22)
}
__device__ __forceinline__ bool summary_nonempty(uint32_t w, uint32_t r) { return (w >> r) & 1u; }
__device__ __forceinline__ bool summary_multi(uint32_t w, uint32_t r) { return (w >> (kSummaryBuckets + (__umul24(r, 11u) >> 5))) & 1u; }  // r / 3

// one lane per bucket; `summary` zeroed by the caller (n_buckets / 24 + 2 words)
__global__ void bucket_summary_kernel(const uint32_t *lookup, uint64_t n_buckets, uint32_t *summary) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < n_buckets; h += stride) {
    const uint32_t f = lookup[h + 1] - lookup[h];
    if (f == 0u) continue;
    const uint32_t q = (uint32_t)(h / kSummaryBuckets), r = (uint32_t)(h % kSummaryBuckets);
    atomicOr(&summary[q], (1u << r) | (f >= 2u ? 1u << (kSummaryBuckets + r / 3u) : 0u));
  }
}

// reference characters -> codes, in place (src/utils.h:72)
__global__ void ref_encode_kernel(uint8_t *text, uint64_t n) {
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    text[i] = (uint8_t)base_code(text[i]);
}

}  // namespace femk
