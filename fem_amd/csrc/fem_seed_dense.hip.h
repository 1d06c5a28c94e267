// fem_seed_dense.hip.h — seed + filter kernel for DENSE indexes (long occurrence lists: a 3 Gbp reference has ~60
// entries per 12-mer bucket), k = 12, step = 3, R = e + 1 + a at compile time.  What BASELINE configs C3-C5 run.
//
// Same semantics as generate_group_seeding_candidates (reference src/filter.c:146-223); the data layout and the
// join are built for the machine:
//
//   * 32-BIT OCCURRENCE COORDINATES.  The index file's occurrence table is uint64 `seq << 32 | pos`
//     (src/index.c:67).  Next to it the library keeps a derived uint32 copy in ONE global coordinate,
//     G = goff[seq] + pos, with a gap of kDenseGap positions between sequences, so that a whole list is half the
//     bytes, every compare / subtract / LDS word of the join is one 32-bit operation, and candidates from different
//     sequences can never come within e of each other.  The reference drops an occurrence whose position is below
//     the seed's offset in the read (`(uint32_t)occ >= start`, src/filter.c:89,106).  Only entries with
//     pos < kDenseNear (= the longest read the device path takes) can ever be dropped that way; those are stored
//     in a separate code space (kDenseRemap | seq << 10 | pos) and resolved exactly on a rare path.
//   * PER-RUN CHUNKS ON SCALAR BASES.  A seed's list is read by consecutive lanes from a wave-uniform base
//     (coalesced; no per-lane run selection), at most two 64-entry chunks per seed in registers.  Longer lists
//     (repeats) send the read to the generic kernel.
//   * BITMAP JOIN IN LDS.  merge_candidate_locations + additional_qgram_filter (src/filter.c:80-131) keep a value
//     iff a+1 values of the multiset lie in [v, v+e].  Two bits per 8-position slot (present / hit twice), set
//     with one returning LDS atomic per entry: a value can only take part in a within-e pair if its slot was hit
//     twice or a neighbouring slot is present.  One two-word LDS read covers the three slots (the table wraps:
//     values in its first and last slot are always flagged).  The few flagged
//     values (true hits + ~3 n^2 / slots chance ones) are compacted and the filter is evaluated exactly on them.
//   * ONE CANDIDATE PER STRAND IS THE COMMON CASE.  If every survivor of the strand's three phase groups lies
//     within e of the smallest, the staged greedy merge (src/filter.c:45-78, :209-213) leaves exactly that
//     smallest value: one wave min/max instead of sort + merge.  Anything else takes the exact general path.
//
// Reads this kernel cannot finish (a list over 128 entries, more than 64 flagged values in a group, DP wider than
// a wave, a == 0) are queued for the generic seed_filter_kernel; results are identical either way.
#pragma once
#include "fem_seed_select.hip.h"

namespace femk {

constexpr uint32_t kDenseGap = 2048u;          // positions between two sequences in the global coordinate (> e + 1)
constexpr uint32_t kDenseNear = 1024u;         // entries with pos < this are stored remapped (>= the longest read)
constexpr uint32_t kDenseRemap = 0xF0000000u;  // remapped entry: kDenseRemap | seq << 10 | pos  (seq < 2^18)
constexpr uint32_t kDenseMaxSeq = 1u << 18;
constexpr uint32_t kDenseLimit = 0xEFFFF000u;  // global coordinates stay below this
constexpr uint32_t kDenseSent = 0xEFFFFFFFu;   // "no entry" in a lane: still above kDenseVLimit after the start is subtracted
constexpr uint32_t kDenseVLimit = 0xEFFFF800u; // v < this <=> the lane holds a real entry
constexpr uint32_t kDenseBlkShift = 20;        // blkseq[] granularity: first sequence at or before a 1 Mi block
constexpr uint32_t kDenseMaxList = 128u;       // entries of one seed's list this kernel takes (two chunks)

// bitmap geometry: two bits per slot of 8 positions; slots are offset by one so that a window never starts below 0
#ifndef FEM_DENSE_SLOTS_LO
#define FEM_DENSE_SLOTS_LO 16384u
#endif
#ifndef FEM_DENSE_SLOTS_HI
#define FEM_DENSE_SLOTS_HI 16384u
#endif
constexpr uint32_t dense_slots(int R) { return R >= 7 ? FEM_DENSE_SLOTS_HI : FEM_DENSE_SLOTS_LO; }
constexpr uint32_t dense_bitmap_words(int R) { return dense_slots(R) / 16u + 2u; }
// Flagged values one (strand, group) unit may have before the read goes to the generic kernel: chance flags grow like
// 3 n^2 / slots, and at R >= 7 (n ~ 470) a 16 Ki-slot bitmap gives ~40 of them: two per lane there, one otherwise.
#ifndef FEM_DENSE_FLAGS_HI
#define FEM_DENSE_FLAGS_HI 128u
#endif
constexpr uint32_t dense_flag_cap(int R) { return R >= 7 ? FEM_DENSE_FLAGS_HI : 64u; }

// ---- derived tables (built once per index upload) ----
// occ (uint64 seq << 32 | pos) -> global 32-bit coordinates; *bad is set if an entry names a sequence >= n_seq
__global__ void dense_occ32_kernel(const uint64_t *occ, uint64_t n, const uint32_t *goff, uint32_t n_seq, uint32_t *out,
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
      v = kDenseRemap | (seq << 10) | pos;
    } else {
      v = goff[seq] + pos;
    }
    out[i] = v;
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

// ---------------------------------------------------------------------------------------------------------
// Both strands of one read.  Lane strand * 3R + g * R + t holds run t of phase group g of that strand: (start,
// lookup[h], frequency), runs in the order of the stable frequency sort (src/filter.c:204); strands that failed the
// gates have frequency 0 everywhere.  The six (strand, group) units run one after the other in ONE rolled loop (the
// code stays small, the scalar registers few), and the first chunk of every run of unit u + 1 is requested before
// unit u is worked on: the occurrence loads of a read are one exposed round trip instead of six.  A unit is R
// chunks (entries 0..63 of each run) plus, when some list is longer, ONE extra chunk that packs the overflow
// (entries 64.. of all long runs).  Within a unit all LDS atomics are issued back to back, then all window reads.
// Leaves each strand's candidates (global coordinates, ascending, before the range clip) in
// cand_lds[strand * 64 + lane] and their counts in kept0/kept1; false = hand the read to the generic kernel.
// `bitmap` is all-zero (but for its padding bits) on entry and on exit.
// ---------------------------------------------------------------------------------------------------------
template <int R>
__device__ bool dense_join(const SeedParams &p, uint32_t s_start, uint32_t s_lo, uint32_t s_freq, uint32_t *bitmap,
                           uint32_t *flg /* LDS [3][dense_flag_cap + 1] */, uint32_t *scatter /* LDS [64] */, uint32_t *cand_lds,
                           uint32_t &kept0, uint32_t &kept1) {
  const uint32_t ln = lane_id();
  constexpr uint32_t kSlots = dense_slots(R);
  constexpr uint32_t kSlotBits = kSlots == 65536u ? 16u : kSlots == 32768u ? 15u : kSlots == 16384u ? 14u : 13u;
  constexpr uint32_t kFlagCap = dense_flag_cap(R);  // flagged values one unit may have (one or two per lane)
#ifndef FEM_DENSE_PROBE_R
#define FEM_DENSE_PROBE_R 1
#endif
#ifndef FEM_DENSE_PROBE_MIN
#define FEM_DENSE_PROBE_MIN 8u
#endif
  constexpr bool kSecondProbe = R >= FEM_DENSE_PROBE_R;  // weed the chance flags out before the exact filter
  constexpr uint32_t kProbeMin = FEM_DENSE_PROBE_MIN;    // ... when there are more flagged values than this
  constexpr uint32_t kFlgStride = kFlagCap + 1u;    // the entry behind a group's array takes the overflow writes
  constexpr uint32_t kUnits = 2u * (uint32_t)kStep;
  const uint32_t e = (uint32_t)p.e;
  const uint32_t *occ32 = p.occ32;
  kept0 = 0, kept1 = 0;
  if (__builtin_amdgcn_ballot_w64(s_freq > kDenseMaxList)) return false;  // a list beyond two chunks: generic kernel
  const uint32_t s_sf = s_start | (s_freq << 16);  // start < 1024, frequency <= 128: one readlane fetches both
  uint32_t nxt[R];     // first chunk of every run of the next unit (raw table entries)
  uint32_t nxt_sf[R];  // ... and its runs' (start | frequency << 16), wave-uniform: read once, used by the loads and the unit
  auto prefetch = [&](uint32_t u) {
#pragma unroll
    for (int t = 0; t < R; ++t) {
      nxt_sf[t] = (uint32_t)__builtin_amdgcn_readlane((int)s_sf, (int)(u * R + t));
      const uint32_t *bp = occ32 + (uint32_t)__builtin_amdgcn_readlane((int)s_lo, (int)(u * R + t));
      nxt[t] = kDenseSent;
      if (ln < (nxt_sf[t] >> 16)) nxt[t] = bp[ln];
    }
  };
  // own pair of a value: LDS word and bit of its "present" flag ("twice" is the next bit)
  auto pair_word = [&](uint32_t v) -> uint32_t * { return bitmap + ((__builtin_amdgcn_ubfe(v, 3u, kSlotBits) + 1u) >> 4); };
  auto pair_bit = [&](uint32_t v) -> uint32_t { return 1u << (((__builtin_amdgcn_ubfe(v, 3u, kSlotBits) + 1u) << 1) & 31u); };
  // window of a value: the word its left neighbour's pair sits in (the window's six bits span this word and the next)
  auto window_word = [&](uint32_t v) -> uint32_t * { return bitmap + __builtin_amdgcn_ubfe(v, 7u, kSlotBits - 4u); };
  prefetch(0);
  uint32_t cmin = 0xFFFFFFFFu, cmax = 0u;  // per lane: smallest / largest surviving value of this strand it has seen
  uint64_t pm0 = 0, pm1 = 0, pm2 = 0;      // survivors of the strand's groups (lanes of flg[g])
  uint32_t nf0 = 0, nf1 = 0, nf2 = 0;
  bool any_hi = false;  // a survivor sits in the second flagged value of some lane
#pragma unroll 1
  for (uint32_t u = 0; u < kUnits; ++u) {
    const uint32_t g = u >= (uint32_t)kStep ? u - (uint32_t)kStep : u;
    // ---- the unit's runs as wave-uniform scalars ----
    uint32_t f[R], st[R];
    uint32_t n_g = 0, n_ovf = 0;
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const uint32_t sf = nxt_sf[t];
      f[t] = sf >> 16, st[t] = sf & 0xFFFFu;
      n_g += f[t];
      n_ovf += f[t] > (uint32_t)kWave ? f[t] - (uint32_t)kWave : 0u;
    }
    uint32_t val[R];
#pragma unroll
    for (int t = 0; t < R; ++t) val[t] = nxt[t];
    if (u + 1u < kUnits) prefetch(u + 1u);
    if (n_ovf > (uint32_t)kWave) return false;  // (the bitmap is clean between units)
    // fewer than a+1 occurrences: nothing can pass the filter; no list but the last seed's: it is merged only while
    // the list has elements (src/filter.c:85)
    const bool skip = n_g <= (uint32_t)p.a || n_g == f[R - 1];
    uint32_t n_flag = 0;
    uint32_t *flg_g = flg + g * kFlgStride;
    if (!skip) {
      // ---- the extra chunk: entries 64.. of the long runs, packed; its lanes carry their own start ----
      uint32_t xval = kDenseSent, xst = 0;
      bool x_last = false;
      if (n_ovf) {
        uint32_t pre = 0, idx = 0;
#pragma unroll
        for (int t = 0; t < R; ++t) {
          const uint32_t o = f[t] > (uint32_t)kWave ? f[t] - (uint32_t)kWave : 0u;
          if (o && ln >= pre) {
            idx = (uint32_t)__builtin_amdgcn_readlane((int)s_lo, (int)(u * R + t)) + (uint32_t)kWave + (ln - pre);
            xst = st[t], x_last = t == R - 1;
          }
          pre += o;
        }
        if (ln < n_ovf) xval = occ32[idx];
      }
      uint64_t remap = 0;
      {
        uint32_t raw_max = val[0];  // the sentinel is below kDenseRemap: one compare for all the unit's first chunks
#pragma unroll
        for (int t = 1; t < R; ++t) raw_max = val[t] > raw_max ? val[t] : raw_max;
        remap = __builtin_amdgcn_ballot_w64(raw_max >= kDenseRemap);
      }
#pragma unroll
      for (int t = 0; t < R; ++t) val[t] -= st[t];  // (the sentinel stays above kDenseVLimit: start < 1024)
      if (n_ovf) {
        remap |= __builtin_amdgcn_ballot_w64(xval >= kDenseRemap);
        xval -= xst;
      }
      uint32_t max_u = 0;
      bool any_u = true;
      if (__builtin_expect(remap != 0 || n_ovf != 0, 0)) {
        // entries within kDenseNear of a sequence start are resolved exactly (pos >= start or dropped); then, as with
        // long lists, the maximum of U comes from a wave reduction (a dropped entry may sit at the end of a run)
        uint32_t mx = 0, have_u = 0;
        auto resolve = [&](uint32_t &v, uint32_t start) {
          const uint32_t raw = v + start;
          if (raw >= kDenseRemap) {
            const uint32_t sq = (raw - kDenseRemap) >> 10, pos = raw & (kDenseNear - 1u);
            v = pos >= start ? p.goff[sq] + pos - start : kDenseSent;
          }
        };
#pragma unroll
        for (int t = 0; t < R; ++t) {
          if (remap) resolve(val[t], st[t]);
          if (t < R - 1 && val[t] < kDenseVLimit) mx = val[t] > mx ? val[t] : mx, have_u = 1;
        }
        if (n_ovf) {
          if (remap) resolve(xval, xst);
          if (!x_last && xval < kDenseVLimit) mx = xval > mx ? xval : mx, have_u = 1;
        }
        any_u = __builtin_amdgcn_ballot_w64(have_u != 0) != 0;
        max_u = wave_max_u32(mx);
      } else {
        // every entry is real and lists ascend: the maximum of U is the largest last entry of runs 0..R-2
#pragma unroll
        for (int t = 0; t < R - 1; ++t) {
          const uint32_t lastv = (uint32_t)__builtin_amdgcn_readlane((int)val[t], (int)((f[t] - 1u) & 63u));
          max_u = f[t] && lastv > max_u ? lastv : max_u;
        }
      }
      if (any_u) {
        // the last run keeps values <= max(U) only (src/filter.c:85); everything dropped becomes the sentinel
        val[R - 1] = val[R - 1] <= max_u ? val[R - 1] : kDenseSent;
        if (n_ovf) xval = x_last && xval > max_u ? kDenseSent : xval;
        // ---- insert.  Slot s (8 positions) sits at bit pair s + 1: pair 0 and pair kSlots + 1 are padding whose
        //      "present" bits are permanently set, so values in the first / last slot are always flagged (the table
        //      wraps there; the exact filter below decides) ----
        uint32_t hit[R], xhit = 0;  // the own "present" bit if the slot already had a value, else 0
        uint64_t vmask[R], xvmask = 0;  // lanes that hold a real entry
#pragma unroll
        for (int t = 0; t < R; ++t) {
          hit[t] = 0;
          vmask[t] = __builtin_amdgcn_ballot_w64(val[t] < kDenseVLimit);
          if (val[t] < kDenseVLimit) {
            const uint32_t bit = pair_bit(val[t]);
            hit[t] = lds_or_rtn(pair_word(val[t]), bit) & bit;
          }
        }
        if (n_ovf) {
          xvmask = __builtin_amdgcn_ballot_w64(xval < kDenseVLimit);
          if (xval < kDenseVLimit) {
            const uint32_t bit = pair_bit(xval);
            xhit = lds_or_rtn(pair_word(xval), bit) & bit;
          }
        }
        uint32_t any_hit = xhit;
#pragma unroll
        for (int t = 0; t < R; ++t) any_hit |= hit[t];
        if (__builtin_amdgcn_ballot_w64(any_hit != 0u)) {  // some slot took a second value (every true hit does): mark "twice"
#pragma unroll
          for (int t = 0; t < R; ++t)
            if (hit[t]) (void)lds_or_rtn(pair_word(val[t]), hit[t] << 1);
          if (n_ovf && xhit) (void)lds_or_rtn(pair_word(xval), xhit << 1);
        }
        wave_sync_lds();
        // ---- flag: own slot hit twice, or a neighbouring slot present; compact the flagged values ----
        uint32_t w0[R], w1[R], xw0 = 0, xw1 = 0;
        uint32_t *wp[R], *xwp = bitmap;  // the windows' words: read here, cleared below
#pragma unroll
        for (int t = 0; t < R; ++t) {
          wp[t] = window_word(val[t]);  // (sentinel lanes read some word too: masked below)
          w0[t] = wp[t][0], w1[t] = wp[t][1];
        }
        if (n_ovf) {
          xwp = window_word(xval);
          xw0 = xwp[0], xw1 = xwp[1];
        }
        auto flag_chunk = [&](uint32_t v, uint32_t a0, uint32_t a1, uint64_t real) {
          const uint32_t x = __builtin_amdgcn_alignbit(a1, a0, (v >> 2) & 30u);  // bits 0..4: present/twice of slot-1, slot, slot+1
          const bool near = (x & 0x19u) != 0u;
          const uint64_t m = __builtin_amdgcn_ballot_w64(near) & real;
          uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, n_flag));
          pos = pos < kFlagCap ? pos : kFlagCap;
          if (near && v < kDenseVLimit) flg_g[pos] = v;
          n_flag += (uint32_t)__popcll(m);
        };
#pragma unroll
        for (int t = 0; t < R; ++t) flag_chunk(val[t], w0[t], w1[t], vmask[t]);
        if (n_ovf) flag_chunk(xval, xw0, xw1, xvmask);
        wave_sync_lds();
        // ---- leave the bitmap clean: every lane clears the two words of its window (its own bits are in one of
        //      them), then the padding pairs get their permanent bits back ----
#pragma unroll
        for (int t = 0; t < R; ++t) {
          if (val[t] < kDenseVLimit) wp[t][0] = 0u, wp[t][1] = 0u;
        }
        if (n_ovf && xval < kDenseVLimit) xwp[0] = 0u, xwp[1] = 0u;
        wave_sync_lds();
        if (ln == 0) bitmap[0] = 1u, bitmap[(kSlots + 1u) >> 4] = 1u << (((kSlots + 1u) << 1) & 31u);
        wave_sync_lds();
        if (n_flag > kFlagCap) return false;
      }
    }
    if (kSecondProbe && n_flag > kProbeMin) {
      // ---- second probe: most of the flagged values are chance flags — values whose slot or a
      //      neighbouring one was also hit by a value 2^17 k positions away.  The flagged values alone go through
      //      the (clean again) bitmap once more, with the slot shifted by a multiple of the value's bits above 17:
      //      a true pair (within e) lands in the same / adjacent slots again, chance partners scatter.  Values within
      //      e of a 2^17 boundary are kept unseen (their partner may sit under another shift).  What survives is a
      //      superset of every within-e pair, so the exact filter below gives the same result on far fewer values. ----
      const bool have0 = ln < n_flag, have1 = ln + (uint32_t)kWave < n_flag;
      const uint32_t v0 = have0 ? flg_g[ln] : kDenseSent, v1 = have1 ? flg_g[ln + (uint32_t)kWave] : kDenseSent;
      auto key2 = [&](uint32_t v) -> uint32_t {
        const uint32_t slot2 = (__builtin_amdgcn_ubfe(v, 3u, kSlotBits) + (v >> 17) * 0x9E5u) & (kSlots - 1u);
        return (slot2 << 3) | (v & 7u);
      };
      auto edge = [&](uint32_t v) -> bool {
        const uint32_t lo17 = v & 0x1FFFFu;
        return lo17 < e || lo17 + e >= 0x20000u;
      };
      const uint32_t k0 = key2(v0), k1 = key2(v1);
      uint32_t h0 = 0, h1 = 0;
      if (have0) {
        const uint32_t bit = pair_bit(k0);
        h0 = lds_or_rtn(pair_word(k0), bit) & bit;
      }
      if (have1) {
        const uint32_t bit = pair_bit(k1);
        h1 = lds_or_rtn(pair_word(k1), bit) & bit;
      }
      if (__builtin_amdgcn_ballot_w64((h0 | h1) != 0u)) {
        if (h0) (void)lds_or_rtn(pair_word(k0), h0 << 1);
        if (h1) (void)lds_or_rtn(pair_word(k1), h1 << 1);
      }
      wave_sync_lds();
      uint32_t *wq0 = window_word(k0), *wq1 = window_word(k1);
      const uint32_t a0 = wq0[0], a1 = wq0[1], b0 = wq1[0], b1 = wq1[1];
      const bool keep0 = have0 && (edge(v0) || (__builtin_amdgcn_alignbit(a1, a0, (k0 >> 2) & 30u) & 0x19u) != 0u);
      const bool keep1 = have1 && (edge(v1) || (__builtin_amdgcn_alignbit(b1, b0, (k1 >> 2) & 30u) & 0x19u) != 0u);
      wave_sync_lds();
      if (have0) wq0[0] = 0u, wq0[1] = 0u;
      if (have1) wq1[0] = 0u, wq1[1] = 0u;
      wave_sync_lds();
      if (ln == 0) bitmap[0] = 1u, bitmap[(kSlots + 1u) >> 4] = 1u << (((kSlots + 1u) << 1) & 31u);
      const uint64_t m0 = __builtin_amdgcn_ballot_w64(keep0), m1 = __builtin_amdgcn_ballot_w64(keep1);
      const uint32_t c0 = (uint32_t)__popcll(m0);
      if (keep0) flg_g[__builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))] = v0;
      if (keep1) flg_g[__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, c0))] = v1;
      n_flag = c0 + (uint32_t)__popcll(m1);
      wave_sync_lds();
    }
    if (n_flag > (uint32_t)p.a) {
      // ---- exact window filter on the flagged values: v stays iff a+1 of them lie in [v, v+e] (itself included) ----
      const bool have = ln < n_flag;
      const uint32_t fv = have ? flg_g[ln] : 0u;
      const uint32_t n_lo = n_flag < (uint32_t)kWave ? n_flag : (uint32_t)kWave;
      uint32_t cnt = 0;
      bool pass_hi = false;
      uint32_t fv_hi = 0;
      if (kFlagCap > (uint32_t)kWave && n_flag > (uint32_t)kWave) {
        // more than one flagged value per lane (long lists, small bitmap): the second goes through the same counts
        const bool have_hi = ln + (uint32_t)kWave < n_flag;
        fv_hi = have_hi ? flg_g[ln + (uint32_t)kWave] : 0u;
        uint32_t cnt_hi = 0;
        for (uint32_t j = 0; j < n_lo; ++j) {
          const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)fv, (int)j);
          cnt += (uint32_t)(x - fv <= e), cnt_hi += (uint32_t)(x - fv_hi <= e);
        }
        for (uint32_t j = (uint32_t)kWave; j < n_flag; ++j) {
          const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)fv_hi, (int)(j - (uint32_t)kWave));
          cnt += (uint32_t)(x - fv <= e), cnt_hi += (uint32_t)(x - fv_hi <= e);
        }
        pass_hi = have_hi && cnt_hi > (uint32_t)p.a;
      } else {
        for (uint32_t j = 0; j < n_lo; ++j) {
          const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)fv, (int)j);
          cnt += (uint32_t)(x - fv <= e);
        }
      }
      const bool pass = have && cnt > (uint32_t)p.a;
      const uint64_t pm = __builtin_amdgcn_ballot_w64(pass);
      if (__builtin_amdgcn_ballot_w64(pass_hi)) {
        // survivors among the second values: they only take part in the one-candidate shortcut below
        any_hi = true;
        cmin = pass_hi && fv_hi < cmin ? fv_hi : cmin;
        cmax = pass_hi && fv_hi > cmax ? fv_hi : cmax;
      }
      if (pm != 0) {
        if (g == 0) pm0 = pm, nf0 = n_flag;
        else if (g == 1) pm1 = pm, nf1 = n_flag;
        else pm2 = pm, nf2 = n_flag;
        cmin = pass && fv < cmin ? fv : cmin;
        cmax = pass && fv > cmax ? fv : cmax;
      }
    }
    if (g != (uint32_t)kStep - 1u) continue;
    // ---- the strand's three groups are done: its candidates ----
    uint32_t kept = 0, cv = 0;
    if ((pm0 | pm1 | pm2) != 0 || any_hi) {
      const uint32_t lo_all = wave_min_u32(cmin), hi_all = wave_max_u32(cmax);
      if (hi_all - lo_all <= e) {  // every survivor within e of the smallest: the greedy merges keep exactly that one
        cv = ln == 0 ? lo_all : 0u;
        kept = 1;
      } else if (any_hi) {
        return false;  // (the general path below takes one survivor per lane)
      } else {
        // general case: per group, survivors sorted into lanes and merged greedily (src/filter.c:45-78)
#pragma unroll 1
        for (uint32_t gg = 0; gg < (uint32_t)kStep; ++gg) {
          const uint64_t pm = gg == 0 ? pm0 : gg == 1 ? pm1 : pm2;
          const uint32_t nfl = gg == 0 ? nf0 : gg == 1 ? nf1 : nf2;
          if (pm == 0) continue;
          const uint32_t nF = (uint32_t)__popcll(pm);
          const bool mine = (pm >> ln) & 1ull;
          const uint32_t fv = ln < nfl ? flg[gg * kFlgStride + ln] : 0u;
          uint32_t rank = 0;
          for (uint64_t m = pm; m;) {  // rank among the survivors (ties by lane)
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)fv, j);
            rank += (uint32_t)(x < fv || (x == fv && (uint32_t)j < ln));
          }
          wave_sync_lds();
          if (mine) scatter[rank] = fv;
          wave_sync_lds();
          const uint32_t fs = ln < nF ? scatter[ln] : 0u;
          kept = dense_merge_group(cv, kept, fs, nF, e);
          if (kept == 0xFFFFFFFFu) return false;
        }
      }
    }
    cand_lds[(u >= (uint32_t)kStep ? (uint32_t)kWave : 0u) + ln] = cv;
    if (u >= (uint32_t)kStep) kept1 = kept; else kept0 = kept;
    cmin = 0xFFFFFFFFu, cmax = 0u, any_hi = false;
    pm0 = pm1 = pm2 = 0, nf0 = nf1 = nf2 = 0;
  }
  return true;
}

#ifndef FEM_DENSE_WAVES_LO
#define FEM_DENSE_WAVES_LO 5
#endif
#ifndef FEM_DENSE_WAVES_HI
#define FEM_DENSE_WAVES_HI 4
#endif
constexpr int dense_waves(int R) { return R <= 6 ? FEM_DENSE_WAVES_LO : FEM_DENSE_WAVES_HI; }

template <int R>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(dense_waves(R), 8))) seed_dense_kernel(SeedParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr uint32_t kSeeds = (uint32_t)(kStep * R);
  static_assert(2 * kStep * R <= kWave, "both strands' seeds must fit the lanes of one wave");
  const uint32_t ln = lane_id();
  const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t *wbase = smem + (size_t)wave_in_block * p.lay.wave_bytes;
  uint32_t *pkw = (uint32_t *)(wbase + p.lay.pkw);
  uint32_t *nkw = (uint32_t *)(wbase + p.lay.nkw);
  uint2 *sf = (uint2 *)(wbase + p.lay.sf);
  unsigned long long *take_bits = (unsigned long long *)(wbase + p.lay.dp_bits);
  uint32_t *scatter = (uint32_t *)(wbase + p.lay.X);
  uint32_t *flg = (uint32_t *)(wbase + p.lay.A);
  uint32_t *bitmap = (uint32_t *)(wbase + p.lay.F);
  uint2 *blk_entries = (uint2 *)(wbase + p.lay.B);
  uint32_t *cand_lds = (uint32_t *)(wbase + p.lay.sf);  // 2 x 64 candidates over the seed table (dead by then)
  uint64_t *boff = (uint64_t *)(wbase + p.lay.rb);      // offsets of the block's reads (kReadBlock + 1)
  uint2 *seqtab = (uint2 *)(wbase + p.lay.picked);      // (goff, length) of the first 64 sequences
  const bool small_ref = p.n_seq <= (uint32_t)kWave;
  seqtab[ln] = ln < p.n_seq ? make_uint2(p.goff[ln], p.seq_len[ln]) : make_uint2(0xFFFFFFFFu, 0u);
  for (uint32_t i = ln; i < dense_bitmap_words(R); i += kWave) bitmap[i] = 0;
  wave_sync_lds();
  if (ln == 0) bitmap[0] = 1u, bitmap[(dense_slots(R) + 1u) >> 4] = 1u << (((dense_slots(R) + 1u) << 1) & 31u);  // padding pairs: see dense_join
  const uint32_t smax = p.lay.smax;
  unsigned long long pre_sum = 0, cand_sum = 0;
  SlotChunk chunk, qchunk;

  auto queue_slow = [&](uint32_t read) {
    if (qchunk.left == 0) {
      uint32_t base = 0;
      if (ln == 0) base = atomicAdd(&p.ctr[2], kQueueChunk);
      qchunk.next = bcast0(base);
      qchunk.left = kQueueChunk;
    }
    if (qchunk.next < p.slow_cap) {
      if (ln == 0) p.slow_queue[qchunk.next] = read;
    } else if (ln == 0) {
      atomicOr(&p.ctr[1], kFlagQueueOverflow);
    }
    ++qchunk.next, --qchunk.left;
  };

  constexpr uint32_t kPullBlocks = 1;
  for (;;) {
    uint32_t pull = 0;
    if (ln == 0) pull = atomicAdd(p.work_cursor, kPullBlocks * kReadBlock);
    pull = bcast0(pull);
    if ((uint64_t)p.read_begin + pull >= p.n_reads) break;
    const uint32_t r0 = p.read_begin + pull;
    if (ln < 2u * kReadBlock) blk_entries[ln] = make_uint2(kBlkSkip, 0u);
    // the block's kReadBlock + 1 offsets come in with one load (lane i: read r0 + i) and sit in LDS; the first 256
    // characters of read rb + 1 are requested before read rb is worked on (one register per lane)
    {
      const uint32_t last = p.n_reads - r0 < kReadBlock ? p.n_reads - r0 : kReadBlock;
      if (ln <= last) boff[ln] = p.read_off[r0 + ln];
      wave_sync_lds();
    }
    uint32_t chars_next = 0;
    {
      const uint64_t o0 = boff[0];
      const uint32_t l0 = (uint32_t)(boff[1] - o0);
      if (4u * ln < l0) chars_next = load_u32_unaligned(p.bases + o0 + 4u * ln);
    }
    for (uint32_t rb = 0; rb < kReadBlock && r0 + rb < p.n_reads; ++rb) {
      const uint32_t read = r0 + rb;
      const uint64_t off = __builtin_amdgcn_readfirstlane((uint32_t)boff[rb]) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(boff[rb] >> 32)) << 32);
      const uint64_t off1 = __builtin_amdgcn_readfirstlane((uint32_t)boff[rb + 1u]) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(boff[rb + 1u] >> 32)) << 32);
      const uint32_t L = (uint32_t)(off1 - off);
      const int S = (int)L - kK + 1;  // num_seeds_in_read
      const uint32_t chars0 = chars_next;
      if (rb + 1u < kReadBlock && r0 + rb + 1u < p.n_reads) {
        const uint64_t off2 = __builtin_amdgcn_readfirstlane((uint32_t)boff[rb + 2u]) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(boff[rb + 2u] >> 32)) << 32);
        chars_next = 0;
        if (4u * ln < (uint32_t)(off2 - off1)) chars_next = load_u32_unaligned(p.bases + off1 + 4u * ln);
      }
      // ---- gates (src/filter.c:161-172) + the shapes on which the reference DP is undefined ----
      bool shape_ok = S > 0 && R <= S / kStep;
      if (shape_ok) shape_ok = (S - (kStep - 1)) / kStep - R * kLg + 2 >= 2;
      if (!shape_ok) {
        if (ln / 2u == rb) blk_entries[ln] = make_uint2(0u, 0u);
        continue;
      }
      const uint32_t widest = (uint32_t)(S / kStep - R * kLg + 1);  // columns of phase group 0
      if (widest > (uint32_t)kWave || (uint32_t)S > smax || p.a == 0) {
        queue_slow(read);
        continue;
      }
      // ---- encode: four characters per lane -> 2-bit codes packed big-endian into LDS ----
      bool strand_ok[2] = {true, true};
      uint32_t any_n = 0;
      for (uint32_t b0 = 0; b0 < L; b0 += 256u) {
        const uint32_t idx = b0 + 4u * ln;
        if (idx < L) {
          uint32_t code, nflag;
          encode4(b0 == 0u ? chars0 : load_u32_unaligned(p.bases + off + idx), code, nflag);  // may run up to 3 bytes past the read: masked below
          const uint32_t nb = L - idx;
          const uint32_t keep = nb >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nb)) - 1u);
          nflag &= keep;
          code &= keep & ~(nflag * 3u);  // N -> A (src/utils.h:92)
          const uint32_t byte_addr = (idx >> 4) * 4u + (3u - ((idx >> 2) & 3u));
          ((uint8_t *)pkw)[byte_addr] = (uint8_t)pack4(code);
          ((uint8_t *)nkw)[byte_addr] = (uint8_t)pack4(nflag * 3u);
          any_n |= nflag;
        }
      }
      const bool has_n = __any(any_n != 0);
      if (has_n) {  // rare: the ambiguous-base gate (src/utils.h:108-114, src/filter.c:180-182)
        uint32_t n_fwd_amb = 0, n_rev_amb = 0;
        for (uint32_t b0 = 0; b0 < L; b0 += 256u) {
          const uint32_t idx = b0 + 4u * ln;
          if (idx < L) {
            uint32_t code, nflag;
            encode4(load_u32_unaligned(p.bases + off + idx), code, nflag);
            const uint32_t nb = L - idx;
            nflag &= nb >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nb)) - 1u);
            for (uint32_t q = 0; q < 4u; ++q) {
              const uint32_t isn = (nflag >> (8u * q)) & 1u;
              n_fwd_amb += isn & (uint32_t)(idx + q >= (uint32_t)kK);
              n_rev_amb += isn & (uint32_t)(L - 1u - (idx + q) >= (uint32_t)kK);
            }
          }
        }
        for (int d = 32; d >= 1; d >>= 1) {
          n_fwd_amb += __shfl_xor(n_fwd_amb, d);
          n_rev_amb += __shfl_xor(n_rev_amb, d);
        }
        strand_ok[0] = n_fwd_amb <= (uint32_t)p.e;
        strand_ok[1] = n_rev_amb <= (uint32_t)p.e;
      }
      wave_sync_lds();
#if defined(FEM_ABLATE) && FEM_ABLATE == 0
      continue;
#endif
      // ---- hashes + CSR lookups: lane j owns seed j of the + strand and seed S-1-j of the - strand ----
      int last_used = 0;
      for (int si = 0; si < kStep; ++si) last_used = max(last_used, kStep * ((S - si) / kStep - kLg) + si);
      // (two rounds of 64 seeds go through the table together: one exposed round trip for reads up to 139 bases)
      for (int j0 = 0; j0 < S; j0 += 2 * kWave) {
        uint2 qf[2], qr[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = j0 + h * kWave + (int)ln;
          qf[h] = make_uint2(0u, 0u), qr[h] = make_uint2(0u, 0u);
          if (j < S) {
            const uint32_t w = (uint32_t)j >> 4, sh = 2u * ((uint32_t)j & 15u);
            const uint64_t pw = ((uint64_t)pkw[w] << 32) | pkw[w + 1];
            const uint32_t hf = (uint32_t)(pw >> (64 - 2 * kK - sh)) & kHashMask;
            uint32_t nm = 0;
            if (has_n) nm = (uint32_t)((((uint64_t)nkw[w] << 32) | nkw[w + 1]) >> (64 - 2 * kK - sh)) & kHashMask;
            const uint32_t r = __brev((~hf) & ~nm & kHashMask) >> (32 - 2 * kK);
            const uint32_t hr = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
            if (strand_ok[0] && j <= last_used) __builtin_memcpy(&qf[h], p.lookup + hf, 8);
            if (strand_ok[1] && S - 1 - j <= last_used) __builtin_memcpy(&qr[h], p.lookup + hr, 8);
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = j0 + h * kWave + (int)ln;
          if (j < S) {
            if (strand_ok[0]) sf[j] = make_uint2(qf[h].x, qf[h].y - qf[h].x);
            if (strand_ok[1]) sf[smax + (uint32_t)(S - 1 - j)] = make_uint2(qr[h].x, qr[h].y - qr[h].x);
          }
        }
      }
      wave_sync_lds();
#if defined(FEM_ABLATE) && FEM_ABLATE == 1
      continue;
#endif
      // ---- seed selection (src/filter.c:3-43 + the stable sort of :204) ----
      const uint32_t dp_w = widest <= 16u ? 16u : widest <= 32u ? 32u : 64u;
      uint32_t s_start, s_lo, s_freq;
      const uint32_t pre_g = select_seeds_lanes<R>(p, S, strand_ok, sf, smax, dp_w, take_bits, s_start, s_lo, s_freq);
      if (ln < 2u * kSeeds && !strand_ok[ln / kSeeds]) s_freq = 0;
      unsigned long long pre_read = 0;
      {
        const uint32_t t = pre_g + dpp_or_zero<0x111, 0xF>(pre_g) + dpp_or_zero<0x112, 0xF>(pre_g);  // lanes 2 and 5: strand sums
        if (strand_ok[0]) pre_read += (uint32_t)__builtin_amdgcn_readlane((int)t, 2);
        if (strand_ok[1]) pre_read += (uint32_t)__builtin_amdgcn_readlane((int)t, 5);
      }
#if defined(FEM_ABLATE) && FEM_ABLATE == 2
      if (s_freq != 0xFFFFFFFFu) continue;
#endif
      wave_sync_lds();  // the seed table is dead: its space takes the candidates
      // ---- lists -> candidates, one strand after the other ----
      uint32_t kept0 = 0, kept1 = 0;
      if (!dense_join<R>(p, s_start, s_lo, s_freq, bitmap, flg, scatter, cand_lds, kept0, kept1)) {
        queue_slow(read);
        continue;
      }
      pre_sum += pre_read;
      // ---- back to (sequence, position), remove_out_ranged_candidates (src/filter.c:133-144), hand-over ----
#pragma unroll 1
      for (uint32_t strand = 0; strand < 2u; ++strand) {
        const uint32_t kept = strand ? kept1 : kept0;
        uint64_t out = 0;
        bool ok = false;
        if (kept == 0) {
          if (ln == 0) blk_entries[2u * rb + strand] = make_uint2(0u, 0u);
          continue;
        }
        const uint32_t v = cand_lds[strand * (uint32_t)kWave + ln];  // written by this same lane
        uint32_t sq = 0, pos = 0, slen = 0;
        if (small_ref) {
          // at most 64 sequences: their coordinates sit in the lanes; one ballot per candidate finds its sequence
          const uint2 tab = seqtab[ln];
          for (uint32_t i = 0; i < kept; ++i) {
            const uint32_t vi = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)i);
            const uint32_t s_i = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(tab.x <= vi)) - 1u;  // (vi >= goff[0] always)
            const uint32_t g_i = (uint32_t)__builtin_amdgcn_readlane((int)tab.x, (int)s_i);
            const uint32_t l_i = (uint32_t)__builtin_amdgcn_readlane((int)tab.y, (int)s_i);
            if (ln == i) sq = s_i, pos = vi - g_i, slen = l_i;
          }
        } else if (ln < kept) {
          sq = p.blkseq[v >> kDenseBlkShift];
          while (sq + 1u < p.n_seq && p.goff[sq + 1u] <= v) ++sq;
          pos = v - p.goff[sq];
          slen = p.seq_len[sq];
        }
        if (ln < kept) {
          ok = pos >= (uint32_t)p.e && pos + L + (uint32_t)p.e < slen;
          out = (((uint64_t)sq << 32) | pos) - (uint64_t)p.e;
        }
        const uint64_t mo = __ballot(ok);
        const uint32_t n_out = (uint32_t)__popcll(mo);
        uint32_t base = 0;
        if (n_out > 0) {
          if (n_out <= chunk.left) {
            base = chunk.next;
            chunk.next += n_out, chunk.left -= n_out;
          } else {
            pad_chunk(p, chunk);
            if (ln == 0) base = atomicAdd(&p.ctr[0], kSlotChunk);
            base = bcast0(base);
            chunk.next = base + n_out, chunk.left = kSlotChunk - n_out;
          }
          if ((unsigned long long)base + n_out > p.cand_cap) {
            if (ln == 0) atomicOr(&p.ctr[1], kFlagCandOverflow);
          } else if (ok) {
            const uint32_t rank = (uint32_t)__popcll(mo & ((1ull << ln) - 1ull)), at = base + rank;
            p.cand[at] = out;
            p.cand_meta[at] = (read * 2u + strand) | (rank < (n_out & ~7u) ? kMeta16 : 0u);
          }
        }
        if (ln == 0) blk_entries[2u * rb + strand] = make_uint2(base, n_out);
        cand_sum += n_out;
      }
    }
    wave_sync_lds();
    const uint2 entry = blk_entries[ln];
    wave_sync_lds();
    if (ln < 2u * kReadBlock && r0 + ln / 2u < p.n_reads && entry.x != kBlkSkip) {
      __builtin_nontemporal_store(entry.x, &p.cand_begin[r0 * 2u + ln]);
      __builtin_nontemporal_store(entry.y, &p.cand_count[r0 * 2u + ln]);
    }
  }
  pad_chunk(p, chunk);
  for (uint32_t i = ln; i < qchunk.left; i += kWave)
    if (qchunk.next + i < p.slow_cap) p.slow_queue[qchunk.next + i] = kInvalidRead;
  if (ln == 0) {
    if (pre_sum) atomicAdd(&p.stats[0], pre_sum);
    if (cand_sum) atomicAdd(&p.stats[1], cand_sum);
  }
}


}  // namespace femk
