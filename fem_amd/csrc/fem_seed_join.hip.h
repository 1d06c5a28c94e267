// fem_seed_join.hip.h — seed_join_kernel<R>: merge_candidate_locations .. remove_out_ranged_candidates
// (reference src/filter.c:45-144,205-222) for the reads seed_select_kernel (fem_seed_select.hip.h) prepared, on dense
// indexes (k = 12, step = 3).  One wave per read, reads pulled in blocks of 16.  A read's 6 R selected seeds come in
// with ONE coalesced load (lane (strand * 3 + group) * R + run: list base, start | frequency << 16), requested while
// the previous read is joined.
//
// The join works on the 32-bit global coordinates of fem_seed_dense.hip.h (occ32, goff, the remapped near-start
// entries) and on the idea of round 2's fused seed_dense_kernel — a value survives merge + additional_qgram_filter iff a+1
// values of the unit's multiset lie in [v, v+e], so a bitmap over slots of 8 positions finds the few values that can
// have a partner and the filter is evaluated exactly on those — but is built to retire as few instructions as that
// idea allows (the round-2 join ran the vector, scalar and LDS ports at 73 / 74 / 58 % at once):
//   * ONE bit per slot, 32 Ki slots (FEM_JOIN_SLOTS_HI: 64 Ki at R >= 7 was slower): slot and bit of a value are two shifts, and the three slots a
//     within-e partner can sit in come out of one two-word read + v_alignbit.  A value is flagged when a neighbouring
//     slot is present.  A slot hit TWICE is seen by its second value only (the returning atomic); that lane then sets
//     both neighbours' bits, which flags the slot's first value like any other neighbour would (and, harmlessly,
//     whatever sits two slots away).  The word behind the table is all ones: values in the first or last slot are
//     always flagged instead of wrapping.
//   * lists are read with CLAMPED lane offsets (lanes behind a list's end re-read its last entry: no exec masking of the
//     loads, no extra traffic); a run of up to 128 entries is two such chunks, the second one only where some list of
//     the unit is that long;
//   * nothing in a chunk's insert / window / flag steps branches: a lane without an entry holds a sentinel, ORs a zero
//     into a word of its own and is masked out of the flagged set, so all of a unit's atomics issue back to back, then
//     all of its window reads;
//   * the bitmap is cleared by four 16-byte stores per lane and unit instead of two random word stores per entry.
#pragma once
#include "fem_seed_dense.hip.h"

namespace femk {

#ifndef FEM_JOIN_SLOTS_HI
#define FEM_JOIN_SLOTS_HI 32768u
#endif
// Round-4 instruction diet, bit by bit (all on in the product; FEM_JOIN_OPT builds the ablations of DESIGN.md 4.2):
//   1  exact filter as ONE all-pairs compare (8 x 8 or 16 x 16 lanes) instead of a readlane loop per flagged value, and the
//      second probe only beyond 16 flagged values
//   2  lanes behind a list's end hold a sentinel of their OWN (no two in one slot), so insert / window / flag test nothing:
//      validity is one scalar AND of masks the loads' compares left behind
//   4  a second chunk only for the runs that have one (was: for every run of a unit in which some list is long)
//   8  marks under `if (hit)` alone (the ballot around it cost three scalar instructions per chunk)
// At R >= 7 the kernel lives in the 80 registers of six waves per SIMD: 1 and 2 (their lane constants and masks) spill there
// and cost more than they save (C5, ms per 2.5 M reads: none 14.85, 1: 15.2, 1+2: 15.7, all four 16.1, 1+4+8: 14.8).
#ifndef FEM_JOIN_OPT
#define FEM_JOIN_OPT 15
#endif
#ifndef FEM_JOIN_OPT_HI
#define FEM_JOIN_OPT_HI 12
#endif
// Attribution builds (results WRONG, instruction counts and times only; scratch/abl_join.sh): bit 1 no flagged values kept,
// 2 no exact filter, 4 no marks, 8 bitmap not cleared, 16 no emission, 32 units skipped whole, 64 no insert / window / flag
#ifndef FEM_JOIN_ABL
#define FEM_JOIN_ABL 0
#endif
#ifndef FEM_JOIN_SLOTS_LO
#define FEM_JOIN_SLOTS_LO 32768u
#endif
constexpr uint32_t join_slots(int R, bool padded = false) { return R >= 7 ? FEM_JOIN_SLOTS_HI : padded ? FEM_JOIN_SLOTS_LO : 32768u; }
// Chunks whose LDS steps are issued together at R >= 7 (see join_read).  Round 3: one — with the 80 registers of six waves per
// SIMD five blocks of the join fit a CU beside the selection (C5: 120 -> 140 Mreads/s; three chunks at a time 134).  Round 4,
// after the diet freed registers: three fit the same 80 (C5 join, ms per 2.5 M reads: 1: 13.66, 2: 13.63, 3: 13.44, 4: 13.45,
// 5: 13.48, 9: 13.75).
#ifndef FEM_JOIN_BATCH_HI
#define FEM_JOIN_BATCH_HI 3
#endif

constexpr uint32_t join_bitmap_words(int R, bool padded = false) { return join_slots(R, padded) / 32u + 4u; }  // + the guard word (all ones), 16-byte padded

__device__ __forceinline__ void lds_or(uint32_t *w, uint32_t bits) {
  (void)__hip_atomic_fetch_or(w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// ---------------------------------------------------------------------------------------------------------
// Both strands of one read.  Lane strand * 3R + g * R + t holds run t of phase group g of that strand: (start,
// lookup[h], frequency), runs in the order of the stable frequency sort (src/filter.c:204); strands that failed the
// gates have frequency 0 everywhere.  The six (strand, group) units run one after the other in ONE rolled loop, and
// the first chunk of every run of unit u + 1 is requested before unit u is worked on.  Leaves each strand's candidates
// (global coordinates, ascending, before the range clip) in cand_lds[strand * 64 + lane] and their counts in
// kept0/kept1; false = hand the read to the generic kernel.  `bitmap` is all-zero (but for its guard word) on entry
// and on exit.
// ---------------------------------------------------------------------------------------------------------
template <int R, bool BANKED = false, bool PADDED = false>
__device__ bool join_read(const SeedParams &p, uint32_t s_start, uint32_t s_lo, uint32_t s_freq, uint32_t *bitmap,
                          uint32_t *flg /* LDS [3][dense_flag_cap + 1] */, uint32_t *scatter /* LDS [64] */,
                          uint32_t *cand_lds, uint32_t &kept0, uint32_t &kept1, uint32_t seq_base = 0u /* first sequence of the bank */) {
  const uint32_t ln = lane_id();
  constexpr uint32_t kSlots = join_slots(R, PADDED);
  constexpr uint32_t kSlotBits = kSlots == 65536u ? 16u : kSlots == 32768u ? 15u : 14u;
  static_assert(kSlots == (1u << kSlotBits), "a power of two of slots: 16, 32 or 64 Ki");
  constexpr uint32_t kWordBits = kSlotBits - 5u;     // words of 32 slots
  constexpr uint32_t kWords = kSlots / 32u;          // the guard word sits at bitmap[kWords]
  constexpr uint32_t kPeriodBits = kSlotBits + 3u;   // values this many bits apart share a slot
  constexpr uint32_t kFlagCap = dense_flag_cap(R);   // flagged values one unit may have (one or two per lane)
  constexpr int kOpt = R <= 6 ? FEM_JOIN_OPT : FEM_JOIN_OPT_HI;
  // (PADDED — the strided table with its pads, fem_seed_dense.hip.h: lanes behind a list's end hold a sentinel of their own by
  //  the load itself, so the forms of bit 2 cost nothing: no lane constant, the masks of valid lanes are scalar bit fields)
  constexpr bool kOptPairs = (kOpt & 1) != 0, kOptSent = (kOpt & 2) != 0 || (PADDED && R <= 6), kOptLong = (kOpt & 4) != 0, kOptHit = (kOpt & 8) != 0;
  constexpr bool kSecondProbe = true;  // weed the chance flags out before the exact filter ...
  constexpr uint32_t kProbeMin = kOptPairs ? 16u : 8u;  // ... when there are more flagged values than this (what the all-pairs filter takes)
  constexpr uint32_t kFlgStride = kFlagCap + 1u;     // the entry behind a group's array takes the overflow writes
  constexpr uint32_t kUnits = 2u * (uint32_t)kStep;
  const uint32_t e = (uint32_t)p.e;
  const uint32_t *occ32 = p.occ32;
  kept0 = 0, kept1 = 0;
  if (__builtin_amdgcn_ballot_w64(s_freq > kDenseMaxList)) return false;  // a list beyond two chunks: generic kernel
  // per seed, in its lane: (start | frequency << 16 | byte offset of the first chunk's last entry << 24) and the list's
  // address; a run then costs three readlanes and no scalar arithmetic to speak of
  auto pack_sf = [&](uint32_t start, uint32_t freq) -> uint32_t {
    const uint32_t f_c = freq < (uint32_t)kWave ? freq : (uint32_t)kWave;
    return start | (freq << 16) | ((f_c - (freq != 0u ? 1u : 0u)) << 26);  // (.. * 4) << 24; start < 1024, frequency <= 128
  };
  const uint32_t s_sf = pack_sf(s_start, s_freq);
  const uint64_t s_addr = (uint64_t)(uintptr_t)(occ32 + s_lo);
  const uint32_t s_alo = (uint32_t)s_addr, s_ahi = (uint32_t)(s_addr >> 32);
  uint32_t lane4 = ln * 4u;
  if (PADDED) asm("" : "+v"(lane4));  // (opaque: base + lane4 stays "scalar base + 32-bit lane offset", one global_load with an SGPR pair, not a 64-bit vector add per load)
  // Sentinels of a lane's own (kOptSent).  Slots repeat every 2^kPeriodBits positions; within one unit no two lanes
  // without an entry may land in one slot (a "second value of a slot" costs its marks): a lane behind a list's end holds
  // sent_a - start (lanes 2304 apart — starts are below 1024 and, within a unit, at least 12 apart; 2304 = 9 words of the
  // bitmap: consecutive lanes in different LDS banks, where 2048 put all of them into four), a dropped entry of the last
  // run sent_b (256 apart = one word, beyond the first family's range of 64 x 2304).  All of them are >= kDenseVLimit and
  // more than e from each other.
  const uint32_t sent_a = 0xF0000000u + ln * 2304u, sent_b = 0xF0000000u + 149504u + (ln << 8);
  static_assert(PADDED || (64u * 2304u + 2048u == 149504u && 149504u + 64u * 256u + 2048u < (1u << kPeriodBits)), "sentinel families inside one period");
  // all-pairs filter: which of four ballots holds lane i's row (i >> 2), and where in it (16 (i & 3))
  const uint64_t q_is1 = __builtin_amdgcn_ballot_w64((ln >> 2) == 1u), q_is2 = __builtin_amdgcn_ballot_w64((ln >> 2) == 2u),
                 q_is3 = __builtin_amdgcn_ballot_w64((ln >> 2) == 3u);
  const uint32_t sh16 = 16u * (ln & 3u);
  uint32_t nxt[R], nxt_sf[R];  // first chunk of every run of the next unit (raw table entries), its packed scalars
  typedef const __attribute__((address_space(1))) uint8_t *GlobalBytes;  // (an address rebuilt from integers is "flat" to the compiler otherwise)
  auto run_base_of = [&](uint32_t alo, uint32_t ahi, uint32_t lane) -> GlobalBytes {
    const uint64_t a = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)alo, (int)lane) |
                       ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)ahi, (int)lane) << 32);
    return (GlobalBytes)(uintptr_t)a;
  };
  auto run_base = [&](uint32_t lane) -> GlobalBytes { return run_base_of(s_alo, s_ahi, lane); };
  auto prefetch_of = [&](uint32_t sf, uint32_t alo, uint32_t ahi, uint32_t u) {
    GlobalBytes base[R];  // (all the readlanes first: a load may use a readlane's scalar only four wait states later)
#pragma unroll
    for (int t = 0; t < R; ++t) {
      nxt_sf[t] = (uint32_t)__builtin_amdgcn_readlane((int)sf, (int)(u * R + t));
      base[t] = run_base_of(alo, ahi, u * R + t);
    }
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const uint32_t last4 = nxt_sf[t] >> 24;
      if (PADDED)
        nxt[t] = *(const __attribute__((address_space(1))) uint32_t *)(base[t] + lane4);  // lanes behind the list's end: the table's pads
      else
        nxt[t] = *(const __attribute__((address_space(1))) uint32_t *)(base[t] + (lane4 < last4 ? lane4 : last4));  // lanes behind the list's end: its last entry again
    }
  };
  auto prefetch = [&](uint32_t u) { prefetch_of(s_sf, s_alo, s_ahi, u); };
  // ---- the steps of the join for one chunk.  A lane without an entry holds kDenseSent (>= kDenseVLimit); nothing here
  //      branches on that: such a lane ORs a zero into a word of its own (lane index: no bank conflict), reads some
  //      window and is masked out of the flagged set ----
  auto insert = [&](uint32_t v) -> uint32_t {  // -> own bit if the slot already held a value, else 0
    const bool valid = v < kDenseVLimit;
    const uint32_t bit = valid ? 1u << ((v >> 3) & 31u) : 0u;
    const uint32_t widx = valid ? __builtin_amdgcn_ubfe(v, 8u, kWordBits) : ln;
    return lds_or_rtn(bitmap + widx, bit) & bit;
  };
  // word of the bitmap that holds slot q >> 5 ... as v_bfe + v_lshl_add: left alone the compiler makes shift, and, add of it
  auto word_of = [&](uint32_t x, uint32_t from) -> uint32_t * {
    uint32_t idx = __builtin_amdgcn_ubfe(x, from, kWordBits);
    asm("" : "+v"(idx));
    return bitmap + idx;
  };
  auto insert_plain = [&](uint32_t v) -> uint32_t {  // the same where every lane holds a value or a sentinel of its own
    const uint32_t bit = 1u << ((v >> 3) & 31u);
    return lds_or_rtn(word_of(v, 8u), bit) & bit;
  };
  auto mark = [&](uint32_t v, uint32_t hit) {  // second value of a slot: both neighbours "present"
    if (FEM_JOIN_ABL & 4) {
      asm volatile("" ::"v"(hit));
      return;
    }
    if (hit) {
      const uint32_t qm = (v >> 3) - 1u, qp = (v >> 3) + 1u;
      lds_or(word_of(qm, 5u), 1u << (qm & 31u));
      lds_or(word_of(qp, 5u), 1u << (qp & 31u));
    }
  };
  auto window = [&](uint32_t v) -> uint32_t {  // bit 0: slot - 1 present, bit 1: own slot, bit 2: slot + 1
    const uint32_t qm = (v >> 3) - 1u;
    const uint32_t *w = word_of(qm, 5u);
    return __builtin_amdgcn_alignbit(w[1], w[0], qm & 31u);
  };
  // The same window with the three slots at the TOP of the word (bit 29: slot - 1, 30: own, 31: slot + 1): a lane's own slot is
  // always present, so "a neighbouring slot is present" is ONE unsigned compare, x >= kNearTop.  (Values in the table's first
  // 30 slots read the guard word there and are always flagged: 0.1 % of them.)
  constexpr uint32_t kNearTop = 0x60000000u;
  auto window_top = [&](uint32_t v) -> uint32_t {
    const uint32_t qs = (v >> 3) - 30u;
    const uint32_t *w = word_of(qs, 5u);
    return __builtin_amdgcn_alignbit(w[1], w[0], qs & 31u);
  };
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
    uint32_t n_g = 0, f_max = 0;
    auto smax = [](uint32_t a_, uint32_t b_) -> uint32_t {  // (wave-uniform operands: as a ternary the compiler builds v_max3 + v_readfirstlane of it)
      uint32_t m;
      asm("s_max_u32 %0, %1, %2" : "=s"(m) : "s"(a_), "s"(b_) : "scc");
      return m;
    };
    // (banks: U has entries in a higher bank — the last run is merged whole, also where this bank holds nothing else)
    const bool keep_all = BANKED && (nxt_sf[R - 1] & kSelKeepAll) != 0u;
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const uint32_t sf = nxt_sf[t];
      f[t] = (sf >> 16) & 0xFFu, st[t] = sf & (BANKED ? kSelKeepAll - 1u : 0xFFFFu);
      n_g += f[t];
      f_max = t == 0 ? f[0] : smax(f[t], f_max);
    }
    uint32_t val[R];
#pragma unroll
    for (int t = 0; t < R; ++t) val[t] = nxt[t];
    if (u + 1u < kUnits) prefetch(u + 1u);
    // fewer than a+1 occurrences: nothing can pass the filter; no list but the last seed's: it is merged only while
    // the list has elements (src/filter.c:85)
    const bool skip = (FEM_JOIN_ABL & 32) || n_g <= (uint32_t)p.a || (n_g == f[R - 1] && !keep_all);
    uint32_t n_flag = 0;
    uint32_t *flg_g = flg + g * kFlgStride;
    if (kOptPairs) {  // behind the unit's flagged values the array reads "no value": the all-pairs filter compares whole rows
      flg_g[ln] = 0xFFFFFFFFu;
      if (kFlagCap > (uint32_t)kWave) flg_g[ln + (uint32_t)kWave] = 0xFFFFFFFFu;
    }
    if (!skip) {
      const bool long_lists = f_max > (uint32_t)kWave;  // some list has a second chunk (entries 64..127)
      uint32_t hv[R];
#pragma unroll
      for (int t = 0; t < R; ++t) hv[t] = kDenseSent;
      if (long_lists) {
#pragma unroll
        for (int t = 0; t < R; ++t) {
          if (kOptLong && f[t] <= (uint32_t)kWave) continue;  // (wave-uniform: only the runs that have a second chunk)
          const uint32_t last4 = f[t] > (uint32_t)kWave ? (f[t] - 1u) * 4u : 0u;
          const uint32_t at4 = lane4 + 4u * (uint32_t)kWave;
          if (PADDED)
            hv[t] = *(const __attribute__((address_space(1))) uint32_t *)(run_base(u * R + t) + at4);
          else
            hv[t] = *(const __attribute__((address_space(1))) uint32_t *)(run_base(u * R + t) + (at4 < last4 ? at4 : last4));
        }
      }
      bool remap;
      {
        uint32_t raw_max = val[0];  // (a lane behind a list's end holds the list's last entry: one compare for the unit)
#pragma unroll
        for (int t = 1; t < R; ++t) raw_max = val[t] > raw_max ? val[t] : raw_max;
        if (long_lists) {
#pragma unroll
          for (int t = 0; t < R; ++t) raw_max = hv[t] > raw_max ? hv[t] : raw_max;  // (kDenseSent < kDenseRemap: a run without a second chunk says nothing)
        }
        remap = __builtin_amdgcn_ballot_w64(raw_max >= kDenseRemap) != 0;
      }
      uint32_t max_u = 0;
      bool any_u = true;
      uint64_t vm[R];  // lanes that hold an entry of run t's first chunk (kOptSent): scalar arithmetic on the run's length
#pragma unroll
      for (int t = 0; t < R; ++t) vm[t] = 0;
      if (__builtin_expect(remap, 0)) {
        // rare: entries within kDenseNear of a sequence start are resolved exactly (pos >= start or dropped); the maximum
        // of U then comes from a wave reduction (a dropped entry may sit at the end of a run)
        uint32_t mx = 0, have_u = 0;
        auto settle = [&](uint32_t &v, bool have, uint32_t start) {
          const uint32_t raw = v;
          v = kDenseSent;
          if (have) {
            v = raw - start;
            if (raw >= kDenseRemap) {
              const uint32_t sq = (raw - kDenseRemap) >> 10, pos = raw & (kDenseNear - 1u);
              v = pos >= start ? p.goff[seq_base + sq] + pos - start : kDenseSent;
            }
          }
        };
#pragma unroll 1
        for (int t = 0; t < R; ++t) {
          // (rolled, the arrays through selects: this path must stay small)
          uint32_t a_ = 0, b_ = 0;
#pragma unroll
          for (int q = 0; q < R; ++q) a_ = q == t ? val[q] : a_, b_ = q == t ? hv[q] : b_;
          uint32_t f_t = 0, st_t = 0;
#pragma unroll
          for (int q = 0; q < R; ++q) f_t = q == t ? f[q] : f_t, st_t = q == t ? st[q] : st_t;
          settle(a_, ln < f_t, st_t);
          settle(b_, long_lists && ln + (uint32_t)kWave < f_t, st_t);
          if (t < R - 1) {
            if (a_ < kDenseVLimit) mx = a_ > mx ? a_ : mx, have_u = 1;
            if (b_ < kDenseVLimit) mx = b_ > mx ? b_ : mx, have_u = 1;
          }
          if (kOptSent) a_ = a_ < kDenseVLimit ? a_ : sent_b;  // (a sentinel of the lane's own: no two dropped entries in one slot)
#pragma unroll
          for (int q = 0; q < R; ++q) val[q] = q == t ? a_ : val[q], hv[q] = q == t ? b_ : hv[q];
        }
        any_u = __builtin_amdgcn_ballot_w64(have_u != 0) != 0;
        max_u = wave_max_u32(mx);
        if (kOptSent) {
#pragma unroll
          for (int t = 0; t < R; ++t) vm[t] = __builtin_amdgcn_ballot_w64(val[t] < kDenseVLimit);
        }
      } else {
        // every entry is real and lists ascend: the maximum of U is the largest last entry of runs 0..R-2
        if (kOptSent) {
          // a lane behind the list's end takes a sentinel of its own (sent_a - start: no two in one slot, fem_seed_join.hip.h
          // top) instead of the list's last entry again; which lanes hold entries stays behind as a scalar mask
#pragma unroll
          for (int t = 0; t < R; ++t) {
            if (PADDED) {
              vm[t] = __builtin_amdgcn_ballot_w64(ln < f[t]);  // (one v_cmp into a scalar pair; as scalar arithmetic on f it is six instructions)
              val[t] -= st[t];
            } else {
              const bool in = ln < f[t];
              vm[t] = __builtin_amdgcn_ballot_w64(in);  // (the compare's own result: no instruction)
              val[t] = (in ? val[t] : sent_a) - st[t];
            }
          }
        } else {
#pragma unroll
          for (int t = 0; t < R; ++t) val[t] = PADDED ? val[t] - st[t] : ln < f[t] ? val[t] - st[t] : kDenseSent;  // (pads: sentinels already; the inserts test)
        }
        if (long_lists) {
#pragma unroll
          for (int t = 0; t < R; ++t) {
            if (PADDED && (!kOptLong || f[t] > (uint32_t)kWave))
              hv[t] -= st[t];  // (pads: "no entry" already)
            else
              hv[t] = ln + (uint32_t)kWave < f[t] ? hv[t] - st[t] : kDenseSent;
          }
#pragma unroll
          for (int t = 0; t < R - 1; ++t) {
            const uint32_t l_lo = (uint32_t)__builtin_amdgcn_readlane((int)val[t], (int)((f[t] - 1u) & 63u));
            const uint32_t l_hi = (uint32_t)__builtin_amdgcn_readlane((int)hv[t], (int)((f[t] - 65u) & 63u));
            const uint32_t lastv = f[t] > (uint32_t)kWave ? l_hi : l_lo;
            max_u = f[t] && lastv > max_u ? lastv : max_u;
          }
        } else {
#pragma unroll
          for (int t = 0; t < R - 1; ++t) {
            const uint32_t lastv = (uint32_t)__builtin_amdgcn_readlane((int)val[t], (int)((f[t] - 1u) & 63u));
            max_u = f[t] && lastv > max_u ? lastv : max_u;
          }
        }
      }
      if (keep_all) any_u = true, max_u = 0xFFFFFFFFu;
      if (any_u) {
        // the last run keeps values <= max(U) only (src/filter.c:85); everything dropped becomes a sentinel
        if (kOptSent) {
          const bool keep = val[R - 1] <= max_u;
          vm[R - 1] &= __builtin_amdgcn_ballot_w64(keep);
          val[R - 1] = keep ? val[R - 1] : sent_b;
        } else {
          val[R - 1] = val[R - 1] <= max_u ? val[R - 1] : kDenseSent;
        }
        if (long_lists) hv[R - 1] = hv[R - 1] <= max_u ? hv[R - 1] : kDenseSent;
        // ---- insert, then flag (a neighbouring slot is present: the flagged values are compacted into the group's array).
        //      In batches of kBatch chunks: all of a batch's atomics back to back, its marks, later all of a batch's window
        //      reads before the first is used.  At R <= 6 a batch is the whole unit; above, one chunk — what a batch holds
        //      in registers (at R = 5 batches of 1, 2, 3 or 5 chunks run within 3 % of each other: the kernel is bound by
        //      instruction issue, not by the LDS round trips) (hit bits, window words) decides whether the kernel fits the 80 registers of six waves per SIMD,
        //      i.e. whether five of its blocks or four sit on a CU beside seed_select_kernel ----
        constexpr int kBatch = R <= 6 ? R : FEM_JOIN_BATCH_HI;
        // `exact`: the lanes without an entry are tested one by one (v < kDenseVLimit).  Otherwise `valid` says which lanes count
        // and a flagged lane WITHOUT an entry — always above the run's lanes with one: lists ascend, both ends of a run are
        // cut from the top — stores its sentinel at the place the next flagged value will take, or behind the last one,
        // where the exact filter reads it as "no value" (it lies above every coordinate).
        auto flag_chunk = [&](uint32_t v, uint32_t xw, uint64_t valid, bool exact) {
          if (FEM_JOIN_ABL & 1) {
            asm volatile("" ::"v"(xw), "v"(v));
            return;
          }
          bool near = xw >= kNearTop;
          if (exact) near = near && v < kDenseVLimit;
          const uint64_t m = exact ? __builtin_amdgcn_ballot_w64(near) : __builtin_amdgcn_ballot_w64(near) & valid;
          if (!exact && m == 0) return;  // (wave-uniform)
          uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, n_flag));
          pos = pos < kFlagCap ? pos : kFlagCap;
          if (near) flg_g[pos] = v;
          n_flag += (uint32_t)__popcll(m);
        };
        auto insert_all = [&](uint32_t (&vals)[R], bool checked, bool second) {
#pragma unroll
          for (int t0 = 0; t0 < R; t0 += kBatch) {
            uint32_t hit[kBatch];
#pragma unroll
            for (int q = 0; q < kBatch; ++q)
              if (t0 + q < R) {
                hit[q] = 0;
                if (second && kOptLong && f[t0 + q] <= (uint32_t)kWave) continue;
                hit[q] = checked ? insert(vals[t0 + q]) : insert_plain(vals[t0 + q]);
              }
            // a slot that took a second value (every true hit does): chunk by chunk, only where some lane saw one
#pragma unroll
            for (int q = 0; q < kBatch; ++q)
              if (t0 + q < R) {
                if (second && kOptLong && f[t0 + q] <= (uint32_t)kWave) continue;
                if (kOptHit) {
                  mark(vals[t0 + q], hit[q]);
                } else if (__builtin_amdgcn_ballot_w64(hit[q] != 0u)) {
                  mark(vals[t0 + q], hit[q]);
                }
              }
          }
        };
        auto flag_all = [&](uint32_t (&vals)[R], bool second, bool exact) {
#pragma unroll
          for (int t0 = 0; t0 < R; t0 += kBatch) {
            uint32_t x[kBatch];
#pragma unroll
            for (int q = 0; q < kBatch; ++q)
              if (t0 + q < R) {
                if (second && kOptLong && f[t0 + q] <= (uint32_t)kWave) continue;
                x[q] = window_top(vals[t0 + q]);
              }
#pragma unroll
            for (int q = 0; q < kBatch; ++q)
              if (t0 + q < R) {
                if (second && kOptLong && f[t0 + q] <= (uint32_t)kWave) continue;
                flag_chunk(vals[t0 + q], x[q], vm[t0 + q], exact);
              }
          }
        };
        if (FEM_JOIN_ABL & 64) {
#pragma unroll
          for (int t = 0; t < R; ++t) asm volatile("" ::"v"(val[t]), "v"(hv[t]));
        } else {
        insert_all(val, !kOptSent, false);
        if (long_lists) insert_all(hv, true, true);
        wave_sync_lds();
        if (!kOptSent || __builtin_expect(remap, 0)) flag_all(val, false, true); else flag_all(val, false, false);
        if (long_lists) flag_all(hv, true, true);  // (second chunks keep the compare against the sentinel: they are the exception)
        wave_sync_lds();
        }
        if (!(FEM_JOIN_ABL & 8)) {  // leave the bitmap clean: every lane clears its 16-byte pieces (the guard word sits behind them)
          uint4 *b4 = (uint4 *)bitmap;
#pragma unroll
          for (uint32_t k = 0; k < kWords / 4u / (uint32_t)kWave; ++k) b4[k * (uint32_t)kWave + ln] = make_uint4(0u, 0u, 0u, 0u);
        }
        wave_sync_lds();
        if (n_flag > kFlagCap) return false;
      }
    }
    if (kSecondProbe && n_flag > kProbeMin) {
      // ---- second probe: most of the flagged values are chance flags — values whose slot or a neighbouring one was
      //      also hit by a value 2^kPeriodBits k positions away.  The flagged values alone go through the (clean again)
      //      bitmap once more, with the slot shifted by a multiple of the value's bits above the period: a true pair
      //      (within e) lands in the same / adjacent slots again, chance partners scatter.  Values within e of a period
      //      boundary are kept unseen (their partner may sit under another shift).  What survives is a superset of every
      //      within-e pair, so the exact filter below gives the same result on far fewer values. ----
      const bool have0 = ln < n_flag, have1 = ln + (uint32_t)kWave < n_flag;
      const uint32_t v0 = have0 ? flg_g[ln] : 0u, v1 = have1 ? flg_g[ln + (uint32_t)kWave] : 0u;
      auto key2 = [&](uint32_t v) -> uint32_t {
        const uint32_t slot2 = (__builtin_amdgcn_ubfe(v, 3u, kSlotBits) + (v >> kPeriodBits) * 0x9E5u) & (kSlots - 1u);
        return (slot2 << 3) | (v & 7u);
      };
      auto edge = [&](uint32_t v) -> bool {
        const uint32_t lo = v & ((1u << kPeriodBits) - 1u);
        return lo < e || lo + e >= (1u << kPeriodBits);
      };
      auto wipe = [&](uint32_t k) {  // the window's two words hold every bit this value set (its own and its marks)
        const uint32_t qm = (k >> 3) - 1u;
        uint32_t *w = bitmap + __builtin_amdgcn_ubfe(qm, 5u, kWordBits);
        w[0] = 0u, w[1] = 0u;
      };
      const uint32_t k0 = key2(v0), k1 = key2(v1);
      const uint32_t h0 = insert(have0 ? k0 : kDenseSent), h1 = insert(have1 ? k1 : kDenseSent);
      if (__builtin_amdgcn_ballot_w64((h0 | h1) != 0u)) mark(k0, h0), mark(k1, h1);
      wave_sync_lds();
      const bool keep0 = have0 && (edge(v0) || (window(k0) & 5u) != 0u);
      const bool keep1 = have1 && (edge(v1) || (window(k1) & 5u) != 0u);
      wave_sync_lds();
      if (have0) wipe(k0);
      if (have1) wipe(k1);
      wave_sync_lds();
      if (ln == 0) bitmap[kWords] = 0xFFFFFFFFu;  // (a window at the table's end reaches the guard word)
      const uint64_t m0 = __builtin_amdgcn_ballot_w64(keep0), m1 = __builtin_amdgcn_ballot_w64(keep1);
      const uint32_t c0 = (uint32_t)__popcll(m0);
      if (keep0) flg_g[__builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))] = v0;
      if (keep1) flg_g[__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, c0))] = v1;
      n_flag = c0 + (uint32_t)__popcll(m1);
      wave_sync_lds();
      if (kOptPairs) {  // what was dropped must read "no value" again
        if (ln >= n_flag) flg_g[ln] = 0xFFFFFFFFu;
        if (kFlagCap > (uint32_t)kWave && ln + (uint32_t)kWave >= n_flag) flg_g[ln + (uint32_t)kWave] = 0xFFFFFFFFu;
        wave_sync_lds();
      }
    }
    if (!(FEM_JOIN_ABL & 2) && n_flag > (uint32_t)p.a) {
      // ---- exact window filter on the flagged values: v stays iff a+1 of them lie in [v, v+e] (itself included) ----
      const bool have = ln < n_flag;
      uint32_t fv;
      uint32_t cnt = 0;
      bool pass_hi = false;
      uint32_t fv_hi = 0;
      if (kOptPairs && n_flag <= 8u) {
        // all pairs at once: lane 8 i + j compares value j with value i; row i of the ballot is byte i, which lane i counts
        const uint32_t xj = flg_g[ln & 7u], fi = flg_g[ln >> 3];
        const uint64_t m = __builtin_amdgcn_ballot_w64(xj - fi <= e);
        cnt = (uint32_t)__popc((uint32_t)(m >> (8u * (ln & 7u))) & 0xFFu);
        fv = have ? xj : 0u;
      } else if (kOptPairs && n_flag <= 16u) {
        // four passes of 16 x 4 pairs: pass q, lane 16 r + j: value j against value r + 4 q; row i = 16 bits of ballot i >> 2
        const uint32_t xj = flg_g[ln & 15u];
        const uint32_t *fp = flg_g + (ln >> 4);
        const uint64_t m0 = __builtin_amdgcn_ballot_w64(xj - fp[0] <= e), m1 = __builtin_amdgcn_ballot_w64(xj - fp[4] <= e);
        const uint64_t m2 = __builtin_amdgcn_ballot_w64(xj - fp[8] <= e), m3 = __builtin_amdgcn_ballot_w64(xj - fp[12] <= e);
        const uint32_t r0 = (uint32_t)(m0 >> sh16), r1 = (uint32_t)(m1 >> sh16), r2 = (uint32_t)(m2 >> sh16), r3 = (uint32_t)(m3 >> sh16);
        uint32_t row = r0;  // (three conditional moves on lane masks that never change: as selects the compiler turns them into branches)
        asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(row) : "v"(r1), "s"(q_is1));
        asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(row) : "v"(r2), "s"(q_is2));
        asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(row) : "v"(r3), "s"(q_is3));
        cnt = (uint32_t)__popc(row & 0xFFFFu);
        fv = have ? xj : 0u;
      } else {
        fv = have ? flg_g[ln] : 0u;
        const uint32_t n_lo = n_flag < (uint32_t)kWave ? n_flag : (uint32_t)kWave;
        if (kFlagCap > (uint32_t)kWave && n_flag > (uint32_t)kWave) {
          // more than one flagged value per lane (long lists): the second goes through the same counts
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

#ifndef FEM_JOIN_WAVES_LO
#define FEM_JOIN_WAVES_LO 7
#endif
#ifndef FEM_JOIN_WAVES_HI
#define FEM_JOIN_WAVES_HI 6
#endif

constexpr int join_waves(int R) { return R <= 6 ? FEM_JOIN_WAVES_LO : FEM_JOIN_WAVES_HI; }
// Registers: six blocks of this kernel (R <= 6) share a CU with one block of the next batch's seed_select_kernel (80
// registers): 6 x 72 + 80 = 512 per lane and SIMD.  (amdgpu_num_vgpr is not honoured by this compiler; the budget follows
// from the waves per SIMD asked for, and that attribute wants a literal: one kernel per R instead of a template.)

template <int R, bool BANKED = false, bool PADDED = false>
__device__ __forceinline__ void seed_join_body(const SeedParams &p, uint8_t *smem) {
  constexpr uint32_t kSeeds = (uint32_t)(kStep * R);
  static_assert(2 * kStep * R <= kWave, "both strands' seeds must fit the lanes of one wave");
  const uint32_t ln = lane_id();
  const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t *wbase = smem + (size_t)wave_in_block * p.lay.wave_bytes;
  uint32_t *scatter = (uint32_t *)(wbase + p.lay.X);
  uint32_t *flg = (uint32_t *)(wbase + p.lay.A);
  uint32_t *bitmap = (uint32_t *)(wbase + p.lay.F);
  uint2 *blk_entries = (uint2 *)(wbase + p.lay.B);
  uint32_t *cand_lds = (uint32_t *)(wbase + p.lay.sf);  // 2 x 64 candidates
  uint2 *seqtab = (uint2 *)(smem + p.lay.picked);       // (goff, length) of the first 64 sequences: one table per block
  const bool small_ref = p.n_seq <= (uint32_t)kWave;
  if (wave_in_block == 0) seqtab[ln] = ln < p.n_seq ? make_uint2(p.goff[ln], p.seq_len[ln]) : make_uint2(0xFFFFFFFFu, 0u);
  __syncthreads();
  for (uint32_t i = ln; i < join_bitmap_words(R, PADDED); i += kWave) bitmap[i] = i < join_slots(R, PADDED) / 32u ? 0u : 0xFFFFFFFFu;  // (guard word)
  wave_sync_lds();
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

  for (;;) {
    uint32_t pull = 0;
    if (ln == 0) pull = atomicAdd(p.work_cursor, kReadBlock);
    pull = bcast0(pull);
    if ((uint64_t)p.read_begin + pull >= p.n_reads) break;
    const uint32_t r0 = p.read_begin + pull;
    const uint32_t n_blk = p.n_reads - r0 < kReadBlock ? p.n_reads - r0 : kReadBlock;
    if (ln < 2u * kReadBlock) blk_entries[ln] = make_uint2(kBlkSkip, 0u);
    // lane i: header of read r0 + i (status | length << 8, pre-filter count)
    uint2 hdr = make_uint2(kSelSlow, 0u);
    if (ln < n_blk) hdr = p.sel_hdr[r0 + ln];
    uint2 sel_next = make_uint2(0u, 0u);
    if (!BANKED && ln < 2u * kSeeds) sel_next = p.sel[(size_t)r0 * (2u * kSeeds) + ln];
    for (uint32_t rb = 0; rb < n_blk; ++rb) {
      const uint32_t read = r0 + rb;
      const uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane((int)hdr.x, (int)rb);
      const uint32_t status = h0 & 3u, L = h0 >> 8;
      const uint2 sel = sel_next;
      if (!BANKED && rb + 1u < n_blk && ln < 2u * kSeeds) sel_next = p.sel[(size_t)(read + 1u) * (2u * kSeeds) + ln];
      if (status == kSelSlow) continue;  // queued by seed_select_kernel
      if (status == kSelNone) {
        if (ln / 2u == rb) blk_entries[ln] = make_uint2(0u, 0u);
        continue;
      }
      if constexpr (BANKED) {
        // ---- a reference in banks (fem_seed_dense.hip.h): the join once per bank on that bank's parts of the lists; a
        //      strand's candidates of bank after bank gather in LDS (ascending: banks are runs of sequences) and go out
        //      together — the 16-bit-lane flag of full groups of eight is a function of the strand's total ----
        constexpr uint32_t kStash = (uint32_t)kWave;  // candidates one strand may have over all banks (as in one bank: a wave's lanes)
        uint64_t *stash = (uint64_t *)(wbase + p.lay.gq);  // [2][kStash]
        uint32_t acc[2] = {0u, 0u};
        bool failed = false;
        for (uint32_t b = 0; b < p.n_banks && !failed; ++b) {
          uint2 sel_b = make_uint2(0u, 0u);
          if (ln < 2u * kSeeds) sel_b = p.sel[((size_t)read * p.n_banks + b) * (2u * kSeeds) + ln];
          uint32_t kept0 = 0, kept1 = 0;
          if (!join_read<R, true>(p, sel_b.y & 0xFFFFu, sel_b.x, sel_b.y >> 16, bitmap, flg, scatter, cand_lds, kept0, kept1, p.bank_first[b])) {
            failed = true;
            break;
          }
          const uint32_t sq_lo = p.bank_first[b], sq_hi = p.bank_first[b + 1u];
#pragma unroll 1
          for (uint32_t strand = 0; strand < 2u; ++strand) {
            const uint32_t kept = strand ? kept1 : kept0;
            if (kept == 0) continue;
            const uint32_t v = cand_lds[strand * (uint32_t)kWave + ln];
            uint32_t sq = 0, pos = 0, slen = 0;
            if (small_ref) {
              const uint2 tab = seqtab[ln];
              const bool mine = ln >= sq_lo && ln < sq_hi;  // the bank's sequences
              for (uint32_t i = 0; i < kept; ++i) {
                const uint32_t vi = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)i);
                const uint32_t s_i = sq_lo + (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(mine && tab.x <= vi)) - 1u;
                const uint32_t g_i = (uint32_t)__builtin_amdgcn_readlane((int)tab.x, (int)s_i);
                const uint32_t l_i = (uint32_t)__builtin_amdgcn_readlane((int)tab.y, (int)s_i);
                if (ln == i) sq = s_i, pos = vi - g_i, slen = l_i;
              }
            } else if (ln < kept) {
              sq = p.blkseq[(size_t)b * p.blk_stride + (v >> kDenseBlkShift)];
              while (sq + 1u < sq_hi && p.goff[sq + 1u] <= v) ++sq;
              pos = v - p.goff[sq];
              slen = p.seq_len[sq];
            }
            const bool ok = ln < kept && pos >= (uint32_t)p.e && pos + L + (uint32_t)p.e < slen;
            const uint64_t mo = __ballot(ok);
            const uint32_t n_ok = (uint32_t)__popcll(mo);
            if (acc[strand] + n_ok > kStash) {
              failed = true;
              break;
            }
            if (ok) stash[strand * kStash + acc[strand] + (uint32_t)__popcll(mo & ((1ull << ln) - 1ull))] = (((uint64_t)sq << 32) | pos) - (uint64_t)p.e;
            acc[strand] += n_ok;
          }
        }
        if (failed) {
          queue_slow(read);
          continue;
        }
        pre_sum += (uint32_t)__builtin_amdgcn_readlane((int)hdr.y, (int)rb);
        wave_sync_lds();
#pragma unroll 1
        for (uint32_t strand = 0; strand < 2u; ++strand) {
          const uint32_t n_out = acc[strand];
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
            } else {
              for (uint32_t i = ln; i < n_out; i += (uint32_t)kWave) {
                p.cand[base + i] = stash[strand * kStash + i];
                p.cand_meta[base + i] = (read * 2u + strand) | (i < (n_out & ~7u) ? kMeta16 : 0u);
              }
            }
          }
          if (ln == 0) blk_entries[2u * rb + strand] = make_uint2(base, n_out);
          cand_sum += n_out;
        }
        wave_sync_lds();
        continue;
      }
      const uint32_t s_lo = sel.x, s_start = sel.y & 0xFFFFu, s_freq = sel.y >> 16;
      // ---- lists -> candidates, one strand after the other ----
      uint32_t kept0 = 0, kept1 = 0;
      if (!join_read<R, false, PADDED>(p, s_start, s_lo, s_freq, bitmap, flg, scatter, cand_lds, kept0, kept1)) {
        queue_slow(read);
        continue;
      }
      if (FEM_JOIN_ABL & 16) kept0 = kept1 = 0;
      pre_sum += (uint32_t)__builtin_amdgcn_readlane((int)hdr.y, (int)rb);
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


#define FEM_JOIN_KERNEL(R, WAVES)                                                                                               \
  __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) seed_join_kernel_r##R(SeedParams p) {                         \
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];                                                               \
    seed_join_body<R>(p, smem);                                                                                                  \
  }                                                                                                                              \
  __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FEM_JOIN_WAVES_HI, 8))) seed_join_banked_kernel_r##R(SeedParams p) { /* references in banks */ \
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];                                                               \
    seed_join_body<R, true>(p, smem);                                                                                            \
  }                                                                                                                              \
  __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) seed_join_kernel_padded_r##R(SeedParams p) { /* the strided table with pads */ \
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];                                                               \
    seed_join_body<R, false, true>(p, smem);                                                                                     \
  }
FEM_JOIN_KERNEL(1, FEM_JOIN_WAVES_LO)
FEM_JOIN_KERNEL(2, FEM_JOIN_WAVES_LO)
FEM_JOIN_KERNEL(3, FEM_JOIN_WAVES_LO)
FEM_JOIN_KERNEL(4, FEM_JOIN_WAVES_LO)
FEM_JOIN_KERNEL(5, FEM_JOIN_WAVES_LO)
FEM_JOIN_KERNEL(6, FEM_JOIN_WAVES_LO)
FEM_JOIN_KERNEL(7, FEM_JOIN_WAVES_HI)
FEM_JOIN_KERNEL(8, FEM_JOIN_WAVES_HI)
FEM_JOIN_KERNEL(9, FEM_JOIN_WAVES_HI)
FEM_JOIN_KERNEL(10, FEM_JOIN_WAVES_HI)
#undef FEM_JOIN_KERNEL

}  // namespace femk
