// fem_seed_fast.hip.h — the seed + filter kernel for the parameters `FEM map` actually runs:
// k = 12, step = 3 (reference src/FEM_map.c:67-68) and R = e + 1 + a fixed at compile time (1..10).
//
// A wavefront takes blocks of 16 consecutive reads (pulled from a cursor), in one of two forms:
//   HASH (dense index, long occurrence lists), per read: encode (four characters per lane, SWAR char -> 2-bit code,
//     packed into LDS), 24-bit k-mer hash per lane (reverse strand = reversed complement of the same hash), frequency
//     of every seed on both strands (src/index.h:22-28), seed-selection DP (DPP prefix-min, clz traceback,
//     src/filter.c:3-43), lists in lanes or the hash-join form, staged greedy de-dup, range clip, hand-over
//     (src/filter.c:45-144);
//   lean (sparse index), per block: the 16 reads' characters are staged and encoded at once into forward /
//     reverse-complement / N streams (a seed's hash on either strand is a 24-bit window of one stream), the reads'
//     parameters are worked out in lanes, and round 1 of the lookups (columns 0 and 1 of every row, against the bucket
//     summaries) finds each read's live phase groups; then per read: round 2 looks up the live groups' seeds straight
//     into the group queue; select_flush runs DP + traceback for a dozen groups of several reads at once, flush_small
//     finishes all queued reads' lists in one pass over the lanes.
// Reads that do not fit (more than 64 occurrences selected, a DP group wider than 64 columns, ...) are appended to a
// queue and done by the generic seed_filter_kernel.  Results are identical.
#pragma once
#include "fem_kernels.hip.h"

namespace femk {

constexpr int kK = 12, kStep = 3, kLg = 4;
constexpr uint32_t kHashMask = (1u << (2 * kK)) - 1u;
constexpr uint32_t kReadBlock = 16;  // consecutive reads one wave takes at a time (seed_fast_kernel)
constexpr uint32_t kBlkSkip = 0xFFFFFFFFu;  // (begin) entry of a read the fast kernel did not handle
// Lean form, queue of selected seeds: the non-empty seeds of several reads (at most kQueueSeeds, 64 occurrences) wait in
// LDS until flush_small finishes all of them in one pass.  Entry: lookup[h] and start | frequency << 10 | tag << 17,
// tag = read-in-block << 7 | strand << 6 | group << 4 | run.
// Lean form, queue of live phase groups: up to kGroupQueue of them, from several reads, are selected at once.
constexpr uint32_t kGroupQueue = 12;
constexpr uint32_t kQueueSeeds = 64;
constexpr uint32_t kQueueBytes = kQueueSeeds * 8u + kReadBlock * 4u + kReadBlock * 8u;  // seeds, the reads' lengths and pre-filter counts
// LDS scratch of flush_small (bytes from the group queue's offset: its groups are all in lanes by then)
// Arrays whose lifetimes do not overlap share words: gmax -> sv, own -> firstp, nval -> slot_n / slot_first, slen -> lastp.
constexpr uint32_t kFlSv = 0, kFlMax = 0, kFlEv = 544, kFlFirst = 1056, kFlNval = 1312, kFlSlotN = 1312, kFlSlotFirst = 1440,
                   kFlLen = 1568, kFlLast = 1568, kFlSlotLast = 1824, kFlushScratchBytes = 1952;

// char -> 2-bit code for four bases at once.  code: per byte 0..3; nflag: per byte 1 where the base is not
// A/C/G/T in either case (src/utils.h:72).
__device__ __forceinline__ void encode4(uint32_t chars, uint32_t &code, uint32_t &nflag) {
  uint32_t t = (chars >> 1) & 0x03030303u;              // A 0, C 1, G 3, T 2
  code = t ^ ((t >> 1) & 0x01010101u);                   // A 0, C 1, G 2, T 3
  uint32_t upper = chars & 0xDFDFDFDFu;
  uint32_t expect = __builtin_amdgcn_perm(0u, 0x54474341u /* "ACGT" */, code);
  uint32_t z = upper ^ expect;                           // zero byte <=> the base is one of ACGT
  nflag = ((((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) >> 7) & 0x01010101u;
}
// bytes b0..b3 (each < 4) -> (b0 << 6) | (b1 << 4) | (b2 << 2) | b3
__device__ __forceinline__ uint32_t pack4(uint32_t b) {
  return ((b & 3u) << 6) | ((b >> 4) & 0x30u) | ((b >> 14) & 0xCu) | (b >> 24);
}

// ---------------------------------------------------------------------------------------------------------
// merge_kvec_t_uint64_t (src/filter.c:45-78) for one group: `fs` holds the group's nF survivors SORTED in lanes
// 0..nF-1, `cv` the nA candidates so far; two-pointer merge on wave-uniform scalars with the greedy gap rule.
// Returns the new count (the list may outgrow the wave: 0xFFFFFFFF).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t merge_group(uint64_t &cv, uint32_t nA, uint64_t fs, uint32_t nF, uint64_t e64) {
  const uint32_t ln = lane_id();
  uint64_t merged = 0, last_kept = 0;
  uint32_t nB = 0, ia = 0, jf = 0;
  while (ia < nA || jf < nF) {
    const uint64_t xa = readlane64(cv, (int)(ia < nA ? ia : 0));
    const uint64_t xf = readlane64(fs, (int)(jf < nF ? jf : 0));
    const bool take_a = ia < nA && (jf >= nF || xa < xf);
    const uint64_t x = take_a ? xa : xf;
    ia += take_a ? 1u : 0u;
    jf += take_a ? 0u : 1u;
    if (nB == 0 || x > last_kept + e64) {
      if (nB >= (uint32_t)kWave) return 0xFFFFFFFFu;
      merged = ln == nB ? x : merged;
      ++nB;
      last_kept = x;
    }
  }
  cv = merged;
  return nB;
}

// Sort the nF (<= 64) values flagged `mine` (any lanes) into lanes 0..nF-1: rank by all-pairs compare on scalars,
// then one scatter through LDS.
__device__ __forceinline__ uint64_t sort_into_lanes(bool mine, uint64_t v, uint64_t mf, uint32_t nF, uint64_t *scatter) {
  const uint32_t ln = lane_id();
  uint32_t rank = 0;
  for (uint64_t m = mf; m;) {
    const int j = __builtin_ctzll(m);
    m &= m - 1;
    const uint64_t x = readlane64(v, j);
    rank += (uint32_t)(x < v || (x == v && (uint32_t)j < ln));
  }
  wave_sync_lds();
  if (mine) scatter[rank] = v;
  wave_sync_lds();
  return ln < nF ? scatter[ln] : 0;
}

// wave-wide inclusive scans over the 64 lanes (DPP: row_shr 1/2/4/8, then row_bcast 15 and 31); zero is the identity
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x) {
  x += dpp_or_zero<0x111, 0xF>(x), x += dpp_or_zero<0x112, 0xF>(x), x += dpp_or_zero<0x114, 0xF>(x);
  x += dpp_or_zero<0x118, 0xF>(x), x += dpp_or_zero<0x142, 0xA>(x), x += dpp_or_zero<0x143, 0xC>(x);
  return x;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x) {
  x = max(x, dpp_or_zero<0x111, 0xF>(x)), x = max(x, dpp_or_zero<0x112, 0xF>(x)), x = max(x, dpp_or_zero<0x114, 0xF>(x));
  x = max(x, dpp_or_zero<0x118, 0xF>(x)), x = max(x, dpp_or_zero<0x142, 0xA>(x)), x = max(x, dpp_or_zero<0x143, 0xC>(x));
  return x;
}
__device__ __forceinline__ uint32_t lane_pull(uint32_t v, uint32_t from_lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from_lane * 4u), (int)v);
}

// LDS scratch of lists_in_lanes (bytes, per wave)
constexpr uint32_t kListEv = 0, kListEg = 68 * 8, kListOwn = kListEg + 68 * 4, kListMax = kListOwn + 64 * 4;
constexpr uint32_t kListScratchBytes = kListMax + 4 * 8;

// ---------------------------------------------------------------------------------------------------------
// One strand whose 3*R selected seeds hold `total` <= 64 occurrences: one occurrence per lane, vector work only —
// the scalar unit is the busiest port of this kernel, so nothing here loops over lanes with readlane.
// Seeds live one per lane (lane0 + group*R + run; `at` = exclusive prefix sum of their frequencies).
//   1. entry lane i finds its seed (owner marks in LDS + prefix max), loads occ[lookup + i - at]   src/filter.c:89,106
//   2. last run of each group: only values <= max of the other runs survive (LDS atomic max)        src/filter.c:85
//   3. entries ranked by (group, value) — the order X of merge_candidate_locations — all pairs by rotation
//   4. additional_qgram_filter as the reference states it: X[i] stays iff X[i+a] <= X[i] + e        src/filter.c:118-131
//   5. per group, survivors (already sorted) are merged greedily into the candidates              src/filter.c:45-78
// Returns the number of candidates (left in cv), 0xFFFFFFFF if they outgrow the wave.
// ---------------------------------------------------------------------------------------------------------
template <int R>
__device__ uint32_t lists_in_lanes(const SeedParams &p, uint32_t lane0, uint32_t at, uint32_t total, uint32_t s_start,
                                   uint32_t s_lo, uint32_t s_freq, uint8_t *ls, uint64_t *scatter, uint64_t &cv) {
  const uint32_t ln = lane_id();
  constexpr uint32_t kSeeds = (uint32_t)(kStep * R);
  const uint64_t e64 = (uint64_t)p.e;
  uint64_t *ev = (uint64_t *)(ls + kListEv);
  uint32_t *eg = (uint32_t *)(ls + kListEg);
  uint32_t *own = (uint32_t *)(ls + kListOwn);
  unsigned long long *gmax = (unsigned long long *)(ls + kListMax);
  // 1. expand the seeds' lists into entry lanes 0 .. total-1
  own[ln] = 0;
  if (ln < 4u) gmax[ln] = 0;
  wave_sync_lds();
  if (ln >= lane0 && ln < lane0 + kSeeds && s_freq > 0) own[at] = ln + 1u;  // at < total <= 64
  wave_sync_lds();
  const uint32_t owner = wave_scan_max(own[ln]);  // seeds come in lane order, so the last mark at or before i owns i
  const bool have = ln < total;
  const uint32_t src = have ? owner - 1u : lane0;
  const uint32_t e_at = lane_pull(at, src), e_lo = lane_pull(s_lo, src), e_st = lane_pull(s_start, src);
  const uint32_t within = src - lane0;
  const uint32_t e_grp = within / (uint32_t)R, e_run = within % (uint32_t)R;
  bool valid = false;
  uint64_t v = 0;
  if (have) {
    const uint64_t o = p.occ[(uint64_t)e_lo + (ln - e_at)];
    valid = (uint32_t)o >= e_st;  // src/filter.c:89,106
    v = o - e_st;
  }
  // 2. last seed of each group
  const bool is_last = valid && e_run == (uint32_t)(R - 1);
  if (__ballot(is_last)) {
    if (valid && !is_last) atomicMax(&gmax[e_grp], (unsigned long long)v + 1ull);  // 0 = no other run has a value
    wave_sync_lds();
    if (is_last) {
      const unsigned long long mu = gmax[e_grp];
      if (mu == 0 || v > mu - 1ull) valid = false;
    }
  }
  const uint32_t n_valid = (uint32_t)__popcll(__ballot(valid));
  if (n_valid <= (uint32_t)p.a) return 0;
  // 3. rank by (group, value, lane); dropped entries sort behind everything (group 3)
  const uint32_t g = valid ? e_grp : 3u;
  ev[ln] = v;
  eg[ln] = have ? g : 3u;
  if (ln < 4u) eg[64u + ln] = 3u;
  wave_sync_lds();
  uint32_t rank = 0;
  for (uint32_t s = 1; s < total; ++s) {
    uint32_t j = ln + s;
    j = j >= total ? j - total : j;
    j = have ? j : 0u;
    const uint64_t vj = ev[j];
    const uint32_t gj = eg[j];
    rank += (uint32_t)(gj < g || (gj == g && (vj < v || (vj == v && j < ln))));
  }
  wave_sync_lds();
  if (have) ev[rank] = v, eg[rank] = g;
  wave_sync_lds();
  // 4. X[i] stays iff X[i + a] exists in the same group and is <= X[i] + e
  const uint32_t my_g = eg[ln];
  const uint64_t my_v = ev[ln];
  const uint32_t far = ln + (uint32_t)p.a;  // < 64 + 3: eg is padded with "dropped", ev's padding is never compared
  const bool pass = ln < n_valid && eg[far] == my_g && ev[far] <= my_v + e64;
  if (!__ballot(pass)) return 0;
  // 5. staged merge, group by group
  uint32_t nA = 0;
  for (uint32_t grp = 0; grp < (uint32_t)kStep; ++grp) {
    const bool mine = pass && my_g == grp;
    const uint64_t mf = __ballot(mine);
    const uint32_t nF = (uint32_t)__popcll(mf);
    if (nF == 0) continue;
    wave_sync_lds();
    if (mine) scatter[__popcll(mf & ((1ull << ln) - 1ull))] = my_v;  // survivors are in ascending order already
    wave_sync_lds();
    const uint64_t fs = ln < nF ? scatter[ln] : 0;
    if (nA == 0 && readlane64(fs, (int)nF - 1) <= readlane64(fs, 0) + e64) {
      cv = ln == 0 ? fs : 0;  // everything within e of the first: the greedy rule keeps the first only
      nA = 1;
    } else {
      nA = merge_group(cv, nA, fs, nF, e64);
      if (nA == 0xFFFFFFFFu) return nA;
    }
  }
  return nA;
}

// ---------------------------------------------------------------------------------------------------------
// One strand with long occurrence lists (large references).  Per phase group:
//   * the R sorted runs are read as ONE flat index space (consecutive lanes -> consecutive entries of a run:
//     coalesced HBM walks, every chunk of the group in flight together) and kept in registers;
//   * pre-filter: an LDS bitmap with two bits per key slot (key = v >> 3: present / seen-twice).  A value can
//     have a neighbour in [v-e, v+e] (e <= 7) only if its own slot was hit twice or an adjacent slot is present,
//     so everything that takes part in ANY within-e pair gets flagged; slot aliasing only adds false positives;
//   * the few flagged values (true hits + ~n^2/slots chance ones) are compacted into lanes and the
//     additional-q-gram filter is evaluated exactly on them: a value passes iff >= a+1 flagged values lie in
//     [v, v+e] — every value in that range is itself flagged, so this equals the count over the whole list
//     (closed form of src/filter.c:80-131);
//   * survivors are sorted in lanes and merged greedily (src/filter.c:45-78).
// Returns the number of candidates (in cv), or 0xFFFFFFFF if a group does not fit (the read is then queued).
// ---------------------------------------------------------------------------------------------------------
// a group may select up to chunks * 64 occurrences: more seeds per group (larger e) mean longer lists
constexpr int bloom_chunks(int R) { return R >= 7 ? 12 : 8; }
// key slots of the pre-filter bitmap (two bits each): chance flags grow like n^2 / slots, so the long lists of
// large R get twice the slots (4 KiB / 8 KiB of LDS per wave)
constexpr uint32_t bloom_slots(int R) { return R >= 7 ? 32768u : 16384u; }

template <int R>
__device__ uint32_t lists_bloom_join(const SeedParams &p, uint32_t lane0, const uint32_t *group_total, uint32_t s_start,
                                     uint32_t s_lo, uint32_t s_freq, uint64_t *scatter, uint32_t *bloom, uint64_t &cv) {
  const uint32_t ln = lane_id();
  const uint64_t e64 = (uint64_t)p.e;
  constexpr int kMaxChunks = bloom_chunks(R);
  constexpr uint32_t kMask = bloom_slots(R) - 1u;
  uint32_t nA = 0;
  for (uint32_t g = 0; g < (uint32_t)kStep; ++g) {
    const uint32_t n_g = group_total[g];
    if (n_g <= (uint32_t)p.a) continue;
    // per-run (first entry, read offset) as wave-uniform scalars; prefix sums of the frequencies
    uint32_t pf[R + 1], base[R], start[R];
    pf[0] = 0;
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const int j = (int)(lane0 + g * (uint32_t)R) + t;
      const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)s_freq, j);
      pf[t + 1] = pf[t] + f;
      base[t] = (uint32_t)__builtin_amdgcn_readlane((int)s_lo, j) - pf[t];  // entry idx of run t = occ[base[t] + idx]
      start[t] = (uint32_t)__builtin_amdgcn_readlane((int)s_start, j);
    }
    const uint32_t n_u = pf[R - 1];  // entries of runs 0..R-2 (the set U); the last run follows
    if (n_u == 0) continue;          // the last seed is merged only while the list has elements (src/filter.c:85)
    // ---- load: chunk c holds flat entries c*64 + lane ----
    uint64_t val[kMaxChunks];
    uint32_t valid = 0;  // bit c: this lane's entry of chunk c takes part
    uint64_t max_u = 0;
    uint32_t any_u = 0;
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
      val[c] = 0;
      const uint32_t idx = (uint32_t)c * kWave + ln;
      if ((uint32_t)c * kWave < n_g && idx < n_g) {
        uint32_t b = base[0], st = start[0];
#pragma unroll
        for (int t = 1; t < R; ++t)
          if (idx >= pf[t]) b = base[t], st = start[t];
        const uint64_t o = p.occ[(uint32_t)(b + idx)];  // base may have wrapped: 32-bit sum
        if ((uint32_t)o >= st) {                         // src/filter.c:89,106
          const uint64_t v = o - st;
          val[c] = v;
          valid |= 1u << c;
          if (idx < n_u) {
            max_u = v > max_u ? v : max_u;
            any_u = 1;
          }
        }
      }
    }
    if (!__any(any_u)) continue;
    for (int d = 32; d >= 1; d >>= 1) {
      const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(max_u >> 32), d) << 32) |
                             (uint32_t)__shfl_xor((int)(uint32_t)max_u, d);
      max_u = other > max_u ? other : max_u;
    }
    // ---- insert (last-run values above max(U) are dropped first) ----
    uint32_t slot[kMaxChunks];
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
      slot[c] = 0;
      if ((uint32_t)c * kWave < n_g) {
        const uint32_t idx = (uint32_t)c * kWave + ln;
        if (((valid >> c) & 1u) && idx >= n_u && val[c] > max_u) valid &= ~(1u << c);
        if ((valid >> c) & 1u) {
          const uint64_t key = val[c] >> 3;
          const uint32_t sl = ((uint32_t)key + (uint32_t)(key >> 29) * 0x9E3779B1u) & kMask;  // neighbours stay adjacent
          slot[c] = sl;
          const uint32_t bit = 1u << (2u * (sl & 15u));
          const uint32_t old = atomicOr(&bloom[sl >> 4], bit);
          if (old & bit) atomicOr(&bloom[sl >> 4], bit << 1);  // second value on this slot
        }
      }
    }
    wave_sync_lds();
    // ---- flag: own slot seen twice, or a neighbouring slot present ----
    uint32_t n_flag = 0;
    bool too_many = false;
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
      if ((uint32_t)c * kWave < n_g) {
        bool flag = false;
        if ((valid >> c) & 1u) {
          const uint32_t sl = slot[c], lf = (sl - 1u) & kMask, rt = (sl + 1u) & kMask;
          const uint32_t wc = bloom[sl >> 4], wl = bloom[lf >> 4], wr = bloom[rt >> 4];
          // a == 0: the filter keeps every value (src/filter.c:120-128 with num_additional_qgrams 0)
          flag = p.a == 0 || (((wc >> (2u * (sl & 15u) + 1u)) & 1u) | ((wl >> (2u * (lf & 15u))) & 1u) | ((wr >> (2u * (rt & 15u))) & 1u));
        }
        const uint64_t m = __ballot(flag);
        const uint32_t pos = n_flag + (uint32_t)__popcll(m & ((1ull << ln) - 1ull));
        if (flag && pos < (uint32_t)kWave) scatter[pos] = val[c];
        n_flag += (uint32_t)__popcll(m);
        too_many = too_many || n_flag > (uint32_t)kWave;
      }
    }
    wave_sync_lds();
    // ---- leave the bitmap clean for the next group (only the touched words) ----
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c)
      if ((uint32_t)c * kWave < n_g && ((valid >> c) & 1u)) bloom[slot[c] >> 4] = 0;
    if (too_many) return 0xFFFFFFFFu;
    if (n_flag <= (uint32_t)p.a) continue;
    // ---- exact window filter on the flagged values ----
    const bool have = ln < n_flag;
    const uint64_t fv = have ? scatter[ln] : 0;
    uint32_t cnt = 0;
    for (uint32_t j = 0; j < n_flag; ++j) {
      const uint64_t x = readlane64(fv, (int)j);
      cnt += (uint32_t)(x >= fv && x <= fv + e64);
    }
    const bool pass = have && cnt > (uint32_t)p.a;
    const uint64_t mf = __ballot(pass);
    const uint32_t nF = (uint32_t)__popcll(mf);
    if (nF == 0) continue;
    const uint64_t fs = sort_into_lanes(pass, fv, mf, nF, scatter);
    nA = merge_group(cv, nA, fs, nF, e64);
    if (nA == 0xFFFFFFFFu) return nA;
  }
  return nA;
}

// ---------------------------------------------------------------------------------------------------------
// Seed selection for seed_fast_kernel: the DP of select_seeds_dpp (same table, same tie rule) with R, step and
// ceil(k/step) fixed, followed by a traceback in which EVERY (group, row) lane walks the chain of "highest take
// bit at or below the previous column" down to its own row, so each selected seed ends up in its own lane
// without a round trip through LDS.  The stable frequency sort (src/filter.c:204) is a rank among the R lanes
// of a group (R shuffles) and one forward permute puts the seeds in run order:
//     lane = (strand * 3 + phase) * R + run  holds  (start, lookup[h], frequency) of that run.
// Returns M[R][C-1] of group ln in lanes < 6 (for the "candidates before the filter" counter).
// ---------------------------------------------------------------------------------------------------------
template <int R>
__device__ uint32_t select_seeds_lanes(const SeedParams &p, int S, const bool *strand_ok, const uint2 *sf, uint32_t smax,
                                       uint32_t W, unsigned long long *take_bits /* LDS [passes][R] */,
                                       uint32_t &s_start, uint32_t &s_lo, uint32_t &s_freq) {
  const uint32_t ln = lane_id();
  constexpr uint32_t n_groups = 2u * (uint32_t)kStep;
  constexpr uint32_t kFill = 0xFFFFFFFFu;  // identity of min: lets the DPP move fold into v_min_u32
  const uint32_t per_pass = (uint32_t)kWave / W;
  const uint32_t c = ln & (W - 1u), slot = ln / W;
  const uint32_t inf = p.inf32;
  uint32_t pre_mine = 0;
  // A group whose seeds at column 0 of every row (seed 0, Lg, 2 Lg, ...) all have frequency 0 has a zero-cost
  // selection: its minimum is 0, every seed the traceback takes has frequency 0 and the group contributes nothing
  // (M[R][C-1] = 0 as well).  On a sparse index that is most groups away from the read's true locus; the DP runs
  // for the others only, packed into as few passes as they need.
  uint32_t live = 0;
  {
    const uint32_t tg = ln / (uint32_t)R, tr = ln % (uint32_t)R;
    const bool t_in = ln < n_groups * (uint32_t)R && strand_ok[(tg / (uint32_t)kStep) & 1u];
    const uint32_t t_idx = ((tg / (uint32_t)kStep) & 1u) * smax + tg % (uint32_t)kStep + (uint32_t)kStep * (uint32_t)kLg * tr;
    const uint64_t zero = __ballot(t_in && sf[t_in ? t_idx : 0u].y == 0u);
    const uint64_t full = (1ull << R) - 1ull;
    const bool is_live = ln < n_groups && strand_ok[(ln / (uint32_t)kStep) & 1u] && ((zero >> (ln * (uint32_t)R)) & full) != full;
    live = (uint32_t)__ballot(is_live);
  }
  uint32_t packed = 0, n_live = 0;  // group ids of the live groups, three bits each
  for (uint32_t m = live; m; m &= m - 1u) packed |= (uint32_t)__builtin_ctz(m) << (3u * n_live++);
  const uint32_t n_pass_live = (n_live + per_pass - 1u) / per_pass;
  for (uint32_t ps = 0; ps < n_pass_live; ++ps) {
    const uint32_t k = ps * per_pass + slot;
    const uint32_t g = (packed >> (3u * k)) & 7u;
    const uint32_t strand = g / (uint32_t)kStep, si = g % (uint32_t)kStep;
    const bool g_ok = k < n_live;
    const uint32_t ncols = g_ok ? (uint32_t)((S - (int)si) / kStep - R * kLg + 1) : 0u;  // C - 1
    const bool in_seg = c < ncols;
    const uint2 *sfs = sf + (strand & 1u) * smax + si;
    const uint32_t c_safe = in_seg ? c : 0u;  // keeps the unconditional LDS reads inside the array
    uint32_t f[R];
#pragma unroll
    for (int r = 1; r <= R; ++r) f[r - 1] = sfs[(uint32_t)kStep * (c_safe + (uint32_t)((r - 1) * kLg))].y;
    uint32_t M = 0;  // M[0][c] = 0
#pragma unroll
    for (int r = 1; r <= R; ++r) {
      const uint32_t v = M + f[r - 1];  // uint32 wrap as in the reference
      uint32_t x = in_seg ? v : kFill;
      x = dpp_min_step<0x111, 0xF>(x, kFill);  // row_shr:1
      x = dpp_min_step<0x112, 0xF>(x, kFill);  // row_shr:2
      x = dpp_min_step<0x114, 0xF>(x, kFill);  // row_shr:4
      x = dpp_min_step<0x118, 0xF>(x, kFill);  // row_shr:8
      if (W > 16u) x = dpp_min_step<0x142, 0xA>(x, kFill);  // row_bcast:15 into rows 1 and 3
      if (W > 32u) x = dpp_min_step<0x143, 0xC>(x, kFill);  // row_bcast:31 into rows 2 and 3
      uint32_t ex = (uint32_t)__builtin_amdgcn_update_dpp((int)kFill, (int)x, 0x138, 0xF, 0xF, false);  // wave_shr:1
      ex = (c == 0 || ex > inf) ? inf : ex;  // M[r][0] = (uint32)occurrence_table_size
      const bool take = in_seg && v < ex;    // strict: ties go horizontal (src/filter.c:20)
      M = take ? v : ex;
      const unsigned long long bits = __ballot(take);
      if (ln == 0) take_bits[ps * (uint32_t)R + (uint32_t)(r - 1)] = bits;
    }
    // M[R][C-1] of group ln: last column of that group's segment, if it ran in this pass
    const uint32_t own_si = ln % (uint32_t)kStep;
    const bool own_live = ln < n_groups && ((live >> ln) & 1u);
    const uint32_t own_k = (uint32_t)__popc(live & ((1u << (ln & 31u)) - 1u));  // slot index of group ln among the live ones
    const uint32_t own_ncols = own_live ? (uint32_t)((S - (int)own_si) / kStep - R * kLg + 1) : 0u;
    const uint32_t got = __shfl(M, (int)((own_k % per_pass) * W + (own_ncols ? own_ncols - 1u : 0u)));
    if (own_live && own_k / per_pass == ps && own_ncols) pre_mine = got;
  }
  wave_sync_lds();
  // ---- traceback: lane (g, t) finds the seed taken at row R - t ----
  const uint32_t g = ln / (uint32_t)R, t = ln % (uint32_t)R;
  const bool act = ln < n_groups * (uint32_t)R && ((live >> g) & 1u);
  const uint32_t g_c = act ? g : 0u;
  const uint32_t si = g_c % (uint32_t)kStep;
  const uint32_t ncols = (uint32_t)((S - (int)si) / kStep - R * kLg + 1);
  const uint32_t k_c = (uint32_t)__popc(live & ((1u << g_c) - 1u));
  const uint32_t pass_of = k_c / per_pass, slot_of = k_c % per_pass;
  int col = (int)ncols - 1;
  bool alive = act;
  uint32_t sidx = 0xFFFFFFFFu;
  if (W <= 32u) {  // a group's take bits fit one word (every read up to ~170 bases): 32-bit arithmetic
#pragma unroll
    for (int r = R; r >= 1; --r) {
      if (alive && (uint32_t)(R - r) <= t) {
        const uint32_t seg = (uint32_t)(take_bits[pass_of * (uint32_t)R + (uint32_t)(r - 1)] >> (slot_of * W)) & ((2u << col) - 1u);
        if (seg == 0) {
          alive = false;  // column 0 reached before R seeds were taken (UB in reference): the rest stay zero
        } else {
          col = 31 - __builtin_clz(seg);
          if ((uint32_t)(R - r) == t) sidx = si + (uint32_t)kStep * (uint32_t)(col + (r - 1) * kLg);
        }
      }
    }
  } else {
#pragma unroll
    for (int r = R; r >= 1; --r) {
      if (alive && (uint32_t)(R - r) <= t) {
        const unsigned long long seg = take_bits[pass_of * (uint32_t)R + (uint32_t)(r - 1)] & ((2ull << col) - 1ull);  // one group per pass
        if (seg == 0) {
          alive = false;
        } else {
          col = 63 - __builtin_clzll(seg);
          if ((uint32_t)(R - r) == t) sidx = si + (uint32_t)kStep * (uint32_t)(col + (r - 1) * kLg);
        }
      }
    }
  }
  const bool have = alive && sidx != 0xFFFFFFFFu;
  const uint2 q = sf[((g_c / (uint32_t)kStep) & 1u) * smax + (have ? sidx : 0u)];
  uint32_t start = have ? sidx : 0u, lo = have ? q.x : 0u, freq = have ? q.y : 0u;
  // ---- qsort(compare_seed): stable by ascending frequency; traceback order t breaks ties ----
  uint32_t rank = 0;
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const uint32_t fu = (uint32_t)__shfl((int)freq, (int)(g * (uint32_t)R) + u);
    rank += (uint32_t)(fu < freq || (fu == freq && (uint32_t)u < t));
  }
  const int dst = (int)((g * (uint32_t)R + rank) * 4u);  // ds_permute: byte address of the destination lane
  s_start = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)start);
  s_lo = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)lo);
  s_freq = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)freq);
  if (ln >= n_groups * (uint32_t)R) s_start = 0, s_lo = 0, s_freq = 0;
  return pre_mine;
}

// Register budget (measured): the lean form is bound by instruction issue and gains from every extra wave per SIMD as
// long as nothing hot is spilled: six waves (80 VGPRs) with the batched small-read path, whose flush needs more live
// values than the rest (seven waves spill into the read loop and lose 20 %).  The hash-join form wants ~120-150
// registers: with few seeds per group five waves (96 VGPRs) beat four, its LDS allows no more; with many seeds per
// group the spills cost more than the fifth wave brings.
#ifndef FEM_LEAN_WAVES
#define FEM_LEAN_WAVES 7
#endif
constexpr int lean_waves(int R, bool hash) { return hash ? (R <= 6 ? 5 : 1) : FEM_LEAN_WAVES; }

template <int R, bool HASH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(lean_waves(R, HASH), 8))) seed_fast_kernel(SeedParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr uint32_t kSeeds = (uint32_t)(kStep * R);  // selected seeds per strand
  static_assert(2 * kStep * R <= kWave, "both strands' seeds must fit the lanes of one wave");
  const uint32_t ln = lane_id();
  const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform, and known to be
  uint8_t *wbase = smem + (size_t)wave_in_block * p.lay.wave_bytes;
  uint32_t *pkw = (uint32_t *)(wbase + p.lay.pkw);
  uint32_t *nkw = (uint32_t *)(wbase + p.lay.nkw);
  uint2 *sf = (uint2 *)(wbase + p.lay.sf);
  unsigned long long *take_bits = (unsigned long long *)(wbase + p.lay.dp_bits);
  uint64_t *scatter = (uint64_t *)(wbase + p.lay.X);
  uint32_t *bloom = (uint32_t *)(wbase + p.lay.F);  // HASH only: two bits per key slot, kept all-zero between groups
  uint8_t *blk_chars = wbase + p.lay.blk;
  uint32_t *strm_fwd = (uint32_t *)(wbase + p.lay.strm), *strm_rev = strm_fwd + p.lay.strm_words, *strm_n = strm_rev + p.lay.strm_words;
  uint8_t *list_scratch = wbase + p.lay.A;
  uint2 *blk_entries = (uint2 *)(wbase + p.lay.B);
  uint64_t *cand_lds = (uint64_t *)(wbase + p.lay.sf);  // 2 x 64 candidates; the seed table's region holds at least that
  if (HASH)
    for (uint32_t i = ln; i < bloom_slots(R) / 16u; i += kWave) bloom[i] = 0;
  const uint32_t smax = p.lay.smax;
  const uint64_t e64 = (uint64_t)p.e;
  unsigned long long pre_sum = 0, cand_sum = 0;
  SlotChunk chunk;   // candidate slots
  SlotChunk qchunk;  // slow-read queue entries
  [[maybe_unused]] Prof prof;


  // ---- lists -> candidates -> clip + emit for ONE read whose selected seeds sit in the lanes (lane strand * kSeeds +
  //      group * R + run).  Returns false if the read has to go to the generic kernel (nothing emitted then). ----
  auto finish_read = [&](uint32_t read, uint32_t rb, uint32_t L, bool ok0, bool ok1, uint32_t s_start, uint32_t s_lo,
                         uint32_t s_freq, uint32_t s_at, uint32_t total0, uint32_t total1, uint64_t nonempty,
                         unsigned long long pre_read) -> bool {
    // candidates of the two strands wait in LDS (over the seed table, which is dead by now) until both are known to
    // fit: registers are what limits the waves per SIMD
    uint32_t kept0 = 0, kept1 = 0;
    bool slow = false;
    for (uint32_t strand = 0; strand < 2u && !slow; ++strand) {
      // (selects, not array indexing: an array indexed by the loop variable ends up in scratch memory)
      const bool s_ok = strand ? ok1 : ok0;
      const uint32_t s_total = strand ? total1 : total0;
      uint32_t kept = 0;
      if (!s_ok) continue;
      if (s_total <= (uint32_t)p.a) continue;  // fewer than a+1 occurrences: nothing can pass the filter
      const uint32_t lane0 = strand * kSeeds;
      if (s_total <= (uint32_t)kWave) {
        uint64_t cv = 0;
        kept = lists_in_lanes<R>(p, lane0, s_at, s_total, s_start, s_lo, s_freq, list_scratch, scatter, cv);
        cand_lds[strand * (uint32_t)kWave + ln] = cv;
      } else if (HASH) {
        uint32_t group_total[kStep];  // xcap = most occurrences of one group the join takes
        for (uint32_t g = 0; g < (uint32_t)kStep && !slow; ++g) {
          uint32_t n_g = 0;
          for (uint64_t m = (nonempty >> (lane0 + g * (uint32_t)R)) & ((1ull << R) - 1ull); m;) {
            const int j = __builtin_ctzll(m) + (int)(lane0 + g * (uint32_t)R);
            m &= m - 1;
            const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)s_freq, j);
            n_g = f > 0x7fffffffu - n_g ? 0x7fffffffu : n_g + f;
          }
          group_total[g] = n_g;
          slow = n_g > p.lay.xcap;
        }
        if (slow) break;
        uint64_t cv = 0;
        kept = lists_bloom_join<R>(p, lane0, group_total, s_start, s_lo, s_freq, scatter, bloom, cv);
        cand_lds[strand * (uint32_t)kWave + ln] = cv;
      }
      if (kept == 0xFFFFFFFFu) slow = true;
      if (strand) kept1 = kept; else kept0 = kept;
    }
    if (slow) return false;
    // ---- remove_out_ranged_candidates (src/filter.c:133-144) + hand-over to the verify kernel ----
    pre_sum += pre_read;
#pragma unroll 1
    for (uint32_t strand = 0; strand < 2u; ++strand) {
      const uint32_t kept = strand ? kept1 : kept0;
      const uint64_t cv = kept ? cand_lds[strand * (uint32_t)kWave + ln] : 0;  // written by this same lane
      bool ok = false;
      if (ln < kept) {
        const uint32_t sq = (uint32_t)(cv >> 32), pos = (uint32_t)cv;
        const uint32_t slen = p.seq_len[sq];
        ok = pos >= (uint32_t)p.e && pos + L + (uint32_t)p.e < slen;
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
          p.cand[at] = cv - e64;
          p.cand_meta[at] = (read * 2u + strand) | (rank < (n_out & ~7u) ? kMeta16 : 0u);
        }
      }
      if (ln == 0) blk_entries[2u * rb + strand] = make_uint2(base, n_out);
      cand_sum += n_out;
    }
    return true;
  };

  // hands a read to the generic kernel (nothing has been emitted for it)
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

  // ---- small-read batch (lean form only) ----
  uint32_t *q_lo = (uint32_t *)(wbase + p.lay.F), *q_info = q_lo + kQueueSeeds, *q_len = q_info + kQueueSeeds;
  uint8_t *fl = wbase + p.lay.gq;  // flush_small's scratch: over the group queue, whose groups are all in lanes by then
  unsigned long long *q_pre = (unsigned long long *)(q_len + kReadBlock);
  uint32_t q_seeds = 0, q_entries = 0;  // seeds / occurrences queued
  // Finishes every queued read at once: one occurrence per lane over all of them, segments = (read, strand, group).
  // Same steps as lists_in_lanes, per segment.  A (read, strand) whose survivors lie in one group and within e of
  // the first gets that first value as its only candidate (what the staged greedy merge of src/filter.c:45-78 leaves);
  // anything else sends its read to the generic kernel.
  auto flush_small = [&](uint32_t r0) {
    if (q_seeds == 0) return;
    uint64_t *ev = (uint64_t *)(fl + kFlEv), *sv = (uint64_t *)(fl + kFlSv);
    unsigned long long *gmax = (unsigned long long *)(fl + kFlMax);
    uint32_t *own = (uint32_t *)(fl + kFlFirst), *firstp = own, *nval = (uint32_t *)(fl + kFlNval);
    uint32_t *slen = (uint32_t *)(fl + kFlLen), *lastp = (uint32_t *)(fl + kFlLast);
    uint32_t *slot_n = (uint32_t *)(fl + kFlSlotN), *slot_first = (uint32_t *)(fl + kFlSlotFirst);
    uint32_t *slot_last = (uint32_t *)(fl + kFlSlotLast);
    const uint32_t n_seeds = q_seeds, total = q_entries;
    q_seeds = 0, q_entries = 0;
    uint32_t lo = 0, info = 0, f = 0;
    if (ln < n_seeds) lo = q_lo[ln], info = q_info[ln], f = (info >> 10) & 127u;
    const uint32_t at = wave_scan_add(f) - f;
    own[ln] = 0, gmax[ln] = 0, nval[ln] = 0, slen[ln] = 0;
    wave_sync_lds();
    if (ln < n_seeds) own[at] = ln + 1u;
    wave_sync_lds();
    const uint32_t owner = wave_scan_max(own[ln]);
    const bool have = ln < total;
    const uint32_t src = have ? owner - 1u : 0u;
    const uint32_t e_at = lane_pull(at, src), e_lo = lane_pull(lo, src), e_info = lane_pull(info, src);
    const uint32_t e_st = e_info & 1023u, tag = e_info >> 17, run = tag & 15u;
    const uint32_t seg_tag = have ? tag >> 4 : 0xFFFFu;  // read-in-block << 3 | strand << 2 | group
    bool valid = false;
    uint64_t v = ~0ull;  // dropped entries sort behind everything
    if (have) {
      const uint64_t o = p.occ[(uint64_t)e_lo + (ln - e_at)];
      valid = (uint32_t)o >= e_st;  // src/filter.c:89,106
      if (valid) v = o - e_st;
    }
    // segments are runs of equal tags: dense id, first lane
    const uint32_t prev_tag = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)seg_tag, 0x138, 0xF, 0xF, false);  // wave_shr:1
    const bool is_start = have && (ln == 0 || prev_tag != seg_tag);
    const uint32_t seg = have ? wave_scan_add((uint32_t)is_start) - 1u : 63u;
    const uint32_t seg_start = wave_scan_max(is_start ? ln : 0u);
    wave_sync_lds();  // own[] is read; its words become firstp[]
    firstp[ln] = 0xFFFFFFFFu;
    // last seed of each group: only values <= max of the other runs survive (src/filter.c:85)
    const bool is_last = run == (uint32_t)(R - 1);
    if (valid && !is_last) atomicMax(&gmax[seg], (unsigned long long)v + 1ull);
    if (have) atomicAdd(&slen[seg], 1u);
    wave_sync_lds();
    if (valid && is_last) {
      const unsigned long long mu = gmax[seg];
      if (mu == 0 || v > mu - 1ull) valid = false, v = ~0ull;
    }
    if (valid) atomicAdd(&nval[seg], 1u);
    ev[ln] = v;
    wave_sync_lds();
    const uint32_t my_len = have ? slen[seg] : 0u, my_nval = have ? nval[seg] : 0u;
    wave_sync_lds();  // slen[] and nval[] are read; their words become lastp[] and the slot arrays
    lastp[ln] = 0;
    if (ln < 32u) slot_n[ln] = 0;
    // rank inside the segment by (value, lane)
    const uint32_t pos = ln - seg_start;
    uint32_t rank = 0;
    for (uint32_t s_ = 1; __ballot(s_ < my_len); ++s_) {
      const bool act = s_ < my_len;
      uint32_t j = pos + s_;
      j = (j >= my_len ? j - my_len : j) + seg_start;
      const uint64_t vj = ev[act ? j : ln];
      rank += (uint32_t)(act && (vj < v || (vj == v && j < ln)));
    }
    wave_sync_lds();
    if (have) sv[seg_start + rank] = v;
    wave_sync_lds();
    // additional_qgram_filter on the sorted order (lane = sorted position): X[i] stays iff X[i+a] <= X[i] + e
    const bool pass = have && pos + (uint32_t)p.a < my_nval && sv[ln + (uint32_t)p.a] <= sv[ln] + e64;
    if (pass) atomicMin(&firstp[seg], ln), atomicMax(&lastp[seg], ln);
    wave_sync_lds();
    if (is_start && firstp[seg] != 0xFFFFFFFFu) {  // one lane per segment with survivors reports to its (read, strand)
      const uint32_t slot = seg_tag >> 2;
      atomicAdd(&slot_n[slot], 1u);
      slot_first[slot] = firstp[seg], slot_last[slot] = lastp[seg];
    }
    wave_sync_lds();
    // ---- (read, strand) lanes: candidate, clip, emit ----
    const uint32_t n_grp = ln < 32u ? slot_n[ln] : 0u;
    uint64_t cv = 0;
    bool simple = false;
    if (n_grp == 1u) {
      cv = sv[slot_first[ln]];
      simple = sv[slot_last[ln]] <= cv + e64;
    }
    const uint32_t cm = (uint32_t)__ballot(n_grp >= 2u || (n_grp == 1u && !simple));  // complex (read, strand) slots
    const bool read_complex = ln < 32u && ((cm >> (ln & ~1u)) & 3u);
    bool ok = false;
    if (n_grp == 1u && simple && !read_complex) {
      const uint32_t sq = (uint32_t)(cv >> 32), cpos = (uint32_t)cv;
      const uint32_t slen_ref = p.seq_len[sq], L = q_len[ln >> 1];
      ok = cpos >= (uint32_t)p.e && cpos + L + (uint32_t)p.e < slen_ref;  // src/filter.c:133-144
    }
    const uint64_t mo = __ballot(ok);
    const uint32_t n_out = (uint32_t)__popcll(mo);
    if (n_out > 0) {
      // every entry is one candidate: the batch fills what is left of the wave's chunk and goes on in a new one (no
      // slots are left unused in between: unused slots are lanes the verify kernel wastes)
      const uint32_t left = chunk.left, old_next = chunk.next;
      uint32_t new_base = 0;
      if (n_out > left) {
        if (ln == 0) new_base = atomicAdd(&p.ctr[0], kSlotChunk);
        new_base = bcast0(new_base);
        chunk.next = new_base + (n_out - left), chunk.left = kSlotChunk - (n_out - left);
      } else {
        chunk.next += n_out, chunk.left -= n_out;
      }
      const uint32_t rank = (uint32_t)__popcll(mo & ((1ull << ln) - 1ull));
      const uint32_t at_ = rank < left ? old_next + rank : new_base + (rank - left);
      const unsigned long long last = n_out > left ? (unsigned long long)new_base + (n_out - left) : (unsigned long long)old_next + n_out;
      if (last > p.cand_cap) {
        if (ln == 0) atomicOr(&p.ctr[1], kFlagCandOverflow);
      } else if (ok) {
        // (written once, read by the next kernel: non-temporal, like the block's characters — the L2 is the summary table's)
        __builtin_nontemporal_store(cv - e64, &p.cand[at_]);
        __builtin_nontemporal_store((r0 + (ln >> 1)) * 2u + (ln & 1u), &p.cand_meta[at_]);  // a list of one: never a full group of 8
        blk_entries[ln] = make_uint2(at_, 1u);
      }
      cand_sum += n_out;
    }
    // reads with a complex slot go to the generic kernel, which starts them over (rare: survivors in two phase groups,
    // or further apart than e)
    for (uint32_t m = cm; m;) {
      const uint32_t rb_c = (uint32_t)__builtin_ctz(m) >> 1;
      m &= ~(3u << (2u * rb_c));
      const unsigned long long pv = q_pre[rb_c];  // the generic kernel counts the read again
      pre_sum -= ((unsigned long long)bcast0((uint32_t)(pv >> 32)) << 32) | bcast0((uint32_t)pv);
      if (ln / 2u == rb_c) blk_entries[ln] = make_uint2(kBlkSkip, 0u);
      queue_slow(r0 + rb_c);
    }
  };

  // ---- lean form: queue of live phase groups ----
  // gq_row: per group its G - Lg + 1 seeds as hash << 8 | frequency (< 255); gq_desc: (first word, words, DP columns,
  // tag = read-in-block << 3 | strand << 2 | phase).  Totals per (read, strand) slot behind the descriptors.
  uint32_t *gq_row = (uint32_t *)(wbase + p.lay.gq);
  uint4 *gq_desc = (uint4 *)(gq_row + p.lay.gq_cap);
  uint32_t *slot_total = (uint32_t *)(gq_desc + kGroupQueue), *slot_pre = slot_total + 32;
  uint32_t gq_groups = 0, gq_entries = 0, gq_maxcols = 0, gq_reads = 0;
  constexpr uint32_t kGroups = kGroupQueue * (uint32_t)R <= (uint32_t)kWave ? kGroupQueue : (uint32_t)kWave / (uint32_t)R;  // one traceback lane per (group, row)
  static_assert(kGroups >= 2u * (uint32_t)kStep, "a read's six groups must fit the queue");
  // Selects the seeds of every queued group at once — the DP of select_seeds_lanes with the groups of several reads in
  // the lanes, the traceback with one lane per (group, row) — and hands each read's seeds to the small-read queue
  // (flushing that when it is full, and at the end of the block).  Reads with more than 64 occurrences go to the
  // generic kernel.
  auto select_flush = [&](uint32_t r0, bool tail) {
    const uint32_t K = gq_groups, reads = gq_reads;
    const uint32_t W = gq_maxcols <= 16u ? 16u : gq_maxcols <= 32u ? 32u : 64u;
    const uint32_t per_pass = (uint32_t)kWave / W;
    uint32_t g_start = 0, g_lo = 0, g_freq = 0, g_tag = 0;  // lane k * R + run: that run's seed of group k
    gq_groups = 0, gq_entries = 0, gq_maxcols = 0, gq_reads = 0;
    if (K) {
      constexpr uint32_t kFill = 0xFFFFFFFFu;
      const uint32_t inf = p.inf32;
      if (ln < 32u) slot_total[ln] = 0, slot_pre[ln] = 0;
      wave_sync_lds();
      const uint32_t c = ln & (W - 1u), slot = ln / W;
      for (uint32_t ps = 0; ps * per_pass < K; ++ps) {
        const uint32_t k = ps * per_pass + slot;
        const bool g_ok = k < K;
        const uint4 d = gq_desc[g_ok ? k : 0u];
        const uint32_t ncols = g_ok ? d.z : 0u;
        const bool in_seg = c < ncols;
        const uint32_t *row = gq_row + d.x + (in_seg ? c : 0u);
        uint32_t f[R];
#pragma unroll
        for (int r = 1; r <= R; ++r) f[r - 1] = row[(r - 1) * kLg] & 255u;
        uint32_t M = 0;  // M[0][c] = 0
#pragma unroll
        for (int r = 1; r <= R; ++r) {
          const uint32_t v = M + f[r - 1];
          uint32_t x = in_seg ? v : kFill;
          x = dpp_min_step<0x111, 0xF>(x, kFill);  // row_shr:1
          x = dpp_min_step<0x112, 0xF>(x, kFill);  // row_shr:2
          x = dpp_min_step<0x114, 0xF>(x, kFill);  // row_shr:4
          x = dpp_min_step<0x118, 0xF>(x, kFill);  // row_shr:8
          if (W > 16u) x = dpp_min_step<0x142, 0xA>(x, kFill);  // row_bcast:15 into rows 1 and 3
          if (W > 32u) x = dpp_min_step<0x143, 0xC>(x, kFill);  // row_bcast:31 into rows 2 and 3
          uint32_t ex = (uint32_t)__builtin_amdgcn_update_dpp((int)kFill, (int)x, 0x138, 0xF, 0xF, false);  // wave_shr:1
          ex = (c == 0 || ex > inf) ? inf : ex;  // M[r][0] = (uint32)occurrence_table_size
          const bool take = in_seg && v < ex;    // strict: ties go horizontal (src/filter.c:20)
          M = take ? v : ex;
          const unsigned long long bits = __ballot(take);
          if (ln == 0) take_bits[ps * (uint32_t)R + (uint32_t)(r - 1)] = bits;
        }
        // M[R][C-1]: uint32 sum per (read, strand) as the reference forms it (src/filter.c:202)
        if (in_seg && c == ncols - 1u) atomicAdd(&slot_pre[d.w >> 2], M);
      }
      wave_sync_lds();
      // ---- traceback: lane (k, t) finds the seed taken at row R - t ----
      const uint32_t k = ln / (uint32_t)R, t = ln % (uint32_t)R;
      const bool act = k < K;
      const uint4 d = gq_desc[act ? k : 0u];
      const uint32_t pass_of = k / per_pass, slot_of = k % per_pass;
      int col = (int)d.z - 1;
      bool alive = act;
      uint32_t idx = 0xFFFFFFFFu;
      if (W <= 32u) {  // a group's take bits fit one word: the chain runs on 32-bit values
#pragma unroll
        for (int r = R; r >= 1; --r) {
          if (alive && (uint32_t)(R - r) <= t) {
            const uint32_t row_bits = (uint32_t)(take_bits[pass_of * (uint32_t)R + (uint32_t)(r - 1)] >> (slot_of * W));
            const uint32_t seg = row_bits & ((2u << col) - 1u);
            if (seg == 0) {
              alive = false;  // column 0 reached before R seeds were taken (UB in reference): the rest stay zero
            } else {
              col = 31 - __builtin_clz(seg);
              if ((uint32_t)(R - r) == t) idx = (uint32_t)(col + (r - 1) * kLg);
            }
          }
        }
      } else {
#pragma unroll
        for (int r = R; r >= 1; --r) {
          if (alive && (uint32_t)(R - r) <= t) {
            const unsigned long long row_bits = take_bits[pass_of * (uint32_t)R + (uint32_t)(r - 1)];
            const unsigned long long seg = row_bits & ((2ull << col) - 1ull);
            if (seg == 0) {
              alive = false;
            } else {
              col = 63 - __builtin_clzll(seg);
              if ((uint32_t)(R - r) == t) idx = (uint32_t)(col + (r - 1) * kLg);
            }
          }
        }
      }
      const bool have = alive && idx != 0xFFFFFFFFu;
      const uint32_t word = gq_row[d.x + (have ? idx : 0u)];
      const uint32_t start = have ? (d.w & 3u) + (uint32_t)kStep * idx : 0u, hash = have ? word >> 8 : 0u;
      const uint32_t freq = have ? word & 255u : 0u;
      // qsort(compare_seed): stable by ascending frequency; traceback order t breaks ties
      uint32_t rank = 0;
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const uint32_t fu = (uint32_t)__shfl((int)freq, (int)(k * (uint32_t)R) + u);
        rank += (uint32_t)(fu < freq || (fu == freq && (uint32_t)u < t));
      }
      const int dst = (int)((k * (uint32_t)R + rank) * 4u);
      g_start = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)start);
      g_freq = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)freq);
      const uint32_t g_hash = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)hash);
      if (!act) g_start = 0, g_freq = 0;
      g_tag = (d.w << 4) | t;  // read << 7 | strand << 6 | phase << 4 | run
      if (g_freq) {
        g_lo = p.lookup[g_hash];
        atomicAdd(&slot_total[d.w >> 2], g_freq);
      }
      wave_sync_lds();
    }
    // ---- into the small-read queue, as many reads at a time as fit ----
    // What that needs per read is worked out for all of them first, (read, strand) slot ln in lane ln < 32.
    bool mine_any = false;   // seed lanes: the seed has occurrences and its strand can pass the filter
    uint32_t cum = 0;        // lanes 2 rb: occurrences of the reads up to rb that stay here and can pass the filter
    uint32_t rem = 0, go_slow = 0;  // bit 2 rb: read rb waits to be pushed / has to go to the generic kernel
    if (K) {
      const uint32_t t_own = ln < 2u * kReadBlock ? slot_total[ln] : 0u, pre_own = ln < 2u * kReadBlock ? slot_pre[ln] : 0u;
      const uint32_t t_other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t_own, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
      const uint32_t pre_other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre_own, 0xB1, 0xF, 0xF, false);
      const bool stays = t_own <= (uint32_t)kWave && t_other <= (uint32_t)kWave && t_own + t_other <= (uint32_t)kWave;
      const uint32_t use_own = t_own > (uint32_t)p.a ? t_own : 0u, use_other = t_other > (uint32_t)p.a ? t_other : 0u;
      const bool of_read = ln < 2u * kReadBlock && !(ln & 1u) && ((reads >> (ln >> 1)) & 1u);
      const bool counts = of_read && stays;
      rem = (uint32_t)__ballot(counts), go_slow = (uint32_t)__ballot(of_read && !stays);
      cum = wave_scan_add(counts ? use_own + use_other : 0u);
      // "candidates before the filter" of the reads that stay: uint32 per strand (src/filter.c:202), widened
      const unsigned long long pre_read = (unsigned long long)pre_own + pre_other;
      if (counts) q_pre[ln >> 1] = pre_read;
      const uint32_t lo16 = counts ? (uint32_t)(pre_read & 0xFFFFu) : 0u, hi = counts ? (uint32_t)(pre_read >> 16) : 0u;  // (hi < 2^17)
      pre_sum += (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)wave_scan_add(lo16), kWave - 1) +
                 ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)wave_scan_add(hi), kWave - 1) << 16);
      const uint32_t use_mine = (uint32_t)__shfl((int)use_own, (int)((g_tag >> 6) & 31u));  // slot = read << 1 | strand
      mine_any = g_freq > 0 && use_mine != 0u;
    }
    for (uint32_t m = go_slow; m; m &= m - 1u) {  // (nothing has been emitted or counted for these)
      const uint32_t rb = (uint32_t)__builtin_ctz(m) >> 1;
      if (ln / 2u == rb) blk_entries[ln] = make_uint2(kBlkSkip, 0u);
      queue_slow(r0 + rb);
    }
    for (uint32_t pushed = 0;;) {  // pushed: occurrences of the reads pushed so far
      // the reads that still fit the queue: a prefix of the waiting ones (cum is monotone)
      const uint32_t now = (uint32_t)__ballot(ln < 2u * kReadBlock && ((rem >> ln) & 1u) && q_entries + (cum - pushed) <= (uint32_t)kWave);
      if (now) {
        const bool mine = mine_any && ((now >> (2u * (g_tag >> 7))) & 1u);
        const uint64_t mm = __ballot(mine);
        if (mine) {
          const uint32_t at_ = q_seeds + (uint32_t)__popcll(mm & ((1ull << ln) - 1ull));
          q_lo[at_] = g_lo;
          q_info[at_] = g_start | (g_freq << 10) | (g_tag << 17);
        }
        const uint32_t upto = (uint32_t)__builtin_amdgcn_readlane((int)cum, 31 - __builtin_clz(now));
        q_seeds += (uint32_t)__popcll(mm), q_entries += upto - pushed;
        pushed = upto, rem &= ~now;
        wave_sync_lds();
      }
      if (rem == 0 && !tail) break;
      flush_small(r0);  // the next read does not fit, or the block ends
      if (rem == 0) break;
    }
  };

  // Each wave takes blocks of kReadBlock consecutive reads: its loads of read bases and its stores of the
  // per-(read, strand) begin/count entries then cover whole cache lines instead of one word per line and XCD.
  // The blocks are handed out dynamically, kPullBlocks at a time from one cursor (the CUs do not all run at the same
  // pace: a fixed stride needed six times the resident waves to even that out, and every extra wave pads a chunk of
  // candidate slots).  One returning atomic per 64 reads and wave.
  constexpr uint32_t kPullBlocks = 4;  // (one block per pull: 7.8 ms, the cursor's atomics serialize; two 5.55, four 5.47, eight 5.55)
  for (;;) {
  uint32_t pull = 0;
  if (ln == 0) pull = atomicAdd(p.work_cursor, kPullBlocks * kReadBlock);
  pull = bcast0(pull);
  if ((uint64_t)p.read_begin + pull >= p.n_reads) break;
  const uint32_t pull_first = p.read_begin + pull;
  const uint32_t pull_end = p.n_reads - pull_first > kPullBlocks * kReadBlock ? pull_first + kPullBlocks * kReadBlock : p.n_reads;
  for (uint32_t r0 = pull_first; r0 < pull_end; r0 += kReadBlock) {
  // begin / count of the block's 2 * kReadBlock (read, strand) entries gather in LDS and go out in one vector store;
  // kBlkSkip marks reads left to the generic kernel (it writes their entries)
  if (HASH && ln < 2u * kReadBlock) blk_entries[ln] = make_uint2(kBlkSkip, 0u);
  const uint64_t blk_base = p.read_off[r0];
  // Lean form.  The block's characters are contiguous: one coalesced copy into LDS pays the HBM latency once for the
  // whole block, and the whole block is encoded at once (every lane busy, no per-read loop) into three 2-bit streams
  // of 16 bases per word, first base in the top bits: the bases (N as A, src/utils.h:92), their reverse complement
  // (the block read backwards; N as A again) and the N marks.  A seed's hash on either strand is then a 24-bit window
  // of one stream.  Blocks too long for the staging space go to the generic kernel.
  bool blk_ok = false, blk_has_n = false;
  uint32_t n_stream = 0;  // bases per stream
  if (!HASH) {
    const uint32_t r_hi = r0 + kReadBlock < p.n_reads ? r0 + kReadBlock : p.n_reads;
    const uint64_t nbytes64 = p.read_off[r_hi] - blk_base;
    blk_ok = nbytes64 + 32u <= (uint64_t)p.lay.blk_bytes;
    wave_sync_lds();  // the previous block's last reader is done (the staging space is the queues', both empty now)
    if (blk_ok) {
      const uint32_t nbytes = (uint32_t)nbytes64;
#pragma unroll 1
      for (uint32_t i = ln * 16u; i < nbytes + 16u; i += (uint32_t)kWave * 16u)  // 64 bytes of slack behind the batch's bases
      {  // read once: non-temporal, so that the bytes do not push the summary table out of L2
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
        const u32x4 t4 = __builtin_nontemporal_load((const u32x4_unaligned *)(p.bases + blk_base + i));
        *(uint4 *)(blk_chars + i) = make_uint4(t4.x, t4.y, t4.z, t4.w);
      }
      wave_sync_lds();
      const uint32_t nw = nbytes / 16u + 1u;
      n_stream = 16u * nw;
      uint32_t any_n = 0;
#pragma unroll 1
      for (uint32_t m = ln; m < 4u * nw; m += (uint32_t)kWave) {  // four characters -> one byte of each stream
        uint32_t code, nflag;
        encode4(*(const uint32_t *)(blk_chars + 4u * m), code, nflag);
        const uint32_t nb = 4u * m < nbytes ? nbytes - 4u * m : 0u;
        nflag &= nb >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nb)) - 1u);
        code &= ~(nflag * 3u);
        const uint32_t byte_addr = 4u + (m & ~3u) + (3u - (m & 3u));  // big-endian inside each word; word 0 is padding
        ((uint8_t *)strm_fwd)[byte_addr] = (uint8_t)pack4(code);
        ((uint8_t *)strm_n)[byte_addr] = (uint8_t)pack4(nflag * 3u);
        any_n |= nflag;
      }
      blk_has_n = __any(any_n != 0);
      wave_sync_lds();
#pragma unroll 1
      for (uint32_t w = ln; w < nw; w += (uint32_t)kWave) {
        const uint32_t r = __brev(~strm_fwd[nw - w] & ~strm_n[nw - w]);  // base order reversed; the pairs' bits swap back below
        strm_rev[1u + w] = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
      }
      wave_sync_lds();
    }
  }
  // Lean form, continued.  What each read of the block needs later is worked out for all of them at once, read rb in
  // lanes 2 rb and 2 rb + 1 (the block's begin/count entries are two per read), and fetched with v_readlane when the
  // read's turn comes: no scalar arithmetic, no dependent load of read_off per read.
  uint32_t rd_flags = 0, rd_live = 0, rd_total = 0;  // (rd_total: seeds of the live groups | columns of the widest group << 16)
  // ... and per (read, group), lanes 6 (rb & 7) + g, [0] for reads 0..7 and [1] for 8..15: where the group's seeds go in
  // its read's stretch of the group queue (kPg* fields), and the stream position of its seed 0 minus step * that start
  uint32_t pg_word[2] = {0, 0}, pg_base[2] = {0, 0};
  constexpr uint32_t kPgFirstBits = 10, kPgUsedShift = 10, kPgOrdShift = 18, kPgOnBit = 21, kPgTagShift = 22;
  constexpr uint32_t kRdFast = 1u, kRdSlow = 2u;  // rd_flags; bits 2.. = column 1 exists in the groups of phase 0, 1, 2
  if (!HASH) {
    uint32_t rd_org0 = 0, rd_org1 = 0, rd_used = 0;
    const uint32_t my_rb = ln >> 1, my_read = r0 + my_rb;
    const bool in = my_rb < kReadBlock && my_read < p.n_reads;
    uint64_t o0 = 0, o1 = 0;
    if (in) o0 = p.read_off[my_read], o1 = p.read_off[my_read + 1];
    const uint32_t L = (uint32_t)(o1 - o0);
    const int S = (int)L - kK + 1;  // num_seeds_in_read
    // gates (src/filter.c:161-172) + the shapes on which the reference DP is undefined: both entries stay 0
    bool shape_ok = in && S > 0 && R <= S / kStep;
    if (shape_ok) shape_ok = (S - (kStep - 1)) / kStep - R * kLg + 2 >= 2;
    // DP wider than a wave (columns of phase group 0), or no streams: the generic kernel's
    const bool too_wide = (uint32_t)(S / kStep - R * kLg + 1) > (uint32_t)kWave || (uint32_t)S > smax || !blk_ok;
    const uint32_t a = (uint32_t)(o0 - blk_base);
    // Stream position of seed 0 of either strand: the reverse strand's seed j is the window that starts j bases
    // behind the reversed read's start; its stream lies strm_words words behind the forward one.
    rd_org0 = a, rd_org1 = 16u * p.lay.strm_words + n_stream - a - L;
    if (shape_ok && !too_wide) {
      rd_flags = kRdFast;
#pragma unroll
      for (int si = 0; si < kStep; ++si) {
        const uint32_t used = (uint32_t)((S - si) / kStep - kLg + 1);  // seeds the DP of a phase group looks at: G - Lg + 1
        rd_used |= used << (8 * si);                                   // (<= 64 + (R - 1) Lg < 256)
        rd_flags |= (uint32_t)(used - (uint32_t)((R - 1) * kLg) > 1u) << (2 + si);
      }
    } else if (shape_ok) {
      rd_flags = kRdSlow;
    }
    if (ln < 2u * kReadBlock) {
      // begin / count stay 0 unless flush_small finds a candidate; kBlkSkip: the generic kernel (or nobody) writes them
      blk_entries[ln] = make_uint2(in && !(rd_flags & kRdSlow) ? 0u : kBlkSkip, 0u);
      if (in && !(ln & 1u)) q_len[my_rb] = L;
    }
    wave_sync_lds();
    // ---- round 1 of the lookups, for every read of the block: which phase groups can contribute at all ----
    // A group with a zero-cost selection among the seeds at columns 0 and 1 of every row contributes nothing (its
    // minimum is 0: whatever the traceback takes has no occurrences, and M[R][C-1] = 0) — rows 1..s free at column 0
    // and rows s+1..R free at column 1 for some s.  Only those seeds are tested here; round 2, at the read's turn,
    // looks up all seeds of the other ("live") groups.  Every probe is a 64-byte fill from L2 and those fills are what
    // bounds this phase, so column 1 is probed only for groups that column 0 left undecided (a group
    // whose rows are all free at column 0 is dead already).  One lane per (read, group, row): kRpo reads per
    // wave-instruction, kRound1Batch reads' probes in flight together.
    constexpr uint32_t kPer = 2u * (uint32_t)kStep * (uint32_t)R;  // (group, row) lanes of one read
    constexpr uint32_t kFit = (uint32_t)kWave / kPer, kRpo = kFit >= 4u ? 4u : kFit >= 2u ? 2u : 1u;  // reads per wave-instruction
    constexpr uint32_t kRound1Batch = 4u, kOps = kRound1Batch / kRpo;
    static_assert(kRpo >= 1u && kRound1Batch % kRpo == 0u && kReadBlock % kRound1Batch == 0u, "round-1 batching");
    const uint32_t t_rd = ln / kPer, t_gl = ln % kPer;             // read within the instruction (idle lanes: >= kRpo)
    const uint32_t t_g = t_gl / (uint32_t)R, t_row = t_gl % (uint32_t)R, t_strand = t_g / (uint32_t)kStep, t_si = t_g % (uint32_t)kStep;
    const bool t_lane = ln < kRpo * kPer;
    const uint32_t t_ofs = t_si + (uint32_t)(kStep * kLg) * t_row;  // seed offset on its strand, column 0
    const uint32_t t_shift = kPer * t_rd + (uint32_t)R * t_g;        // where the group's rows sit in an instruction's ballot
    constexpr uint32_t kFull = (1u << R) - 1u;
    // the 24 bits that end 2 (pos + k) bits into the streams -> summary word and bit of that bucket
    auto probe = [&](uint32_t pos, uint32_t &r) -> uint32_t {
      const uint32_t end2 = 2u * pos + 2u * (uint32_t)kK;
      const uint32_t *st = strm_fwd + ((end2 - 1u) >> 5);
      uint32_t q;
      summary_slot(__builtin_amdgcn_alignbit(st[0], st[1], 0u - end2) & kHashMask, q, r);
      return p.summary[q];
    };
#pragma unroll 1
    for (uint32_t b0 = 0; b0 < kReadBlock && r0 + b0 < p.n_reads; b0 += kRound1Batch) {
      uint32_t okm[kRound1Batch];  // groups the ambiguous-base gate leaves (src/utils.h:108-114, src/filter.c:180-182)
#pragma unroll
      for (uint32_t u = 0; u < kRound1Batch; ++u) {
        const int sel = (int)(2u * (b0 + u));
        const uint32_t flags = (uint32_t)__builtin_amdgcn_readlane((int)rd_flags, sel);
        okm[u] = (flags & kRdFast) ? 63u : 0u;
        if ((flags & kRdFast) && blk_has_n) {  // N at offsets >= k, counted on either strand
          const uint32_t len = bcast0(q_len[b0 + u]), org0 = (uint32_t)__builtin_amdgcn_readlane((int)rd_org0, sel);
          uint32_t n_fwd_amb = 0, n_rev_amb = 0;
          for (uint32_t c0 = 0; c0 < len; c0 += (uint32_t)kWave) {
            const uint32_t c = c0 + ln, pos = org0 + c;
            const bool is_n = c < len && ((strm_n[1u + (pos >> 4)] >> (30u - 2u * (pos & 15u))) & 1u);
            n_fwd_amb += (uint32_t)__popcll(__ballot(is_n && c >= (uint32_t)kK));
            n_rev_amb += (uint32_t)__popcll(__ballot(is_n && len - 1u - c >= (uint32_t)kK));
          }
          okm[u] = (n_fwd_amb <= (uint32_t)p.e ? 7u : 0u) | (n_rev_amb <= (uint32_t)p.e ? 56u : 0u);
        }
      }
      uint32_t pos0[kOps], w[kOps], sr[kOps];
      bool col1[kOps];  // the lane's group has a column 1
      uint64_t z0[kOps], z1[kOps];
#pragma unroll
      for (uint32_t op = 0; op < kOps; ++op) {  // column 0
        const uint32_t u_lane = op * kRpo + (t_lane ? t_rd : 0u);  // the lane's read of the batch
        const int src = (int)(2u * (b0 + u_lane));
        const uint32_t flags = (uint32_t)__shfl((int)rd_flags, src);
        const uint32_t o0 = (uint32_t)__shfl((int)rd_org0, src), o1 = (uint32_t)__shfl((int)rd_org1, src);
        uint32_t ok = okm[op * kRpo];
#pragma unroll
        for (uint32_t k2 = 1; k2 < kRpo; ++k2) ok = t_rd == k2 ? okm[op * kRpo + k2] : ok;
        const bool t_in = p.summary && t_lane && ((ok >> t_g) & 1u);
        col1[op] = t_in && ((flags >> (2u + t_si)) & 1u);
        pos0[op] = (t_strand ? o1 : o0) + t_ofs;
        w[op] = 0xFFFFFFFFu, sr[op] = 0;  // (not tested: not empty)
        if (t_in) w[op] = probe(pos0[op], sr[op]);
      }
#pragma unroll
      for (uint32_t op = 0; op < kOps; ++op) {  // column 1, where column 0 did not settle it
        z0[op] = __ballot(!summary_nonempty(w[op], sr[op]));
        // rows 1..free0 of the lane's group are free at column 0: only the rows behind them matter at column 1 (s =
        // free0 asks the least of column 1), and none if that is all of them
        const uint32_t free0 = (uint32_t)__builtin_ctz(~((uint32_t)(z0[op] >> t_shift) & kFull));
        w[op] = 0xFFFFFFFFu, sr[op] = 0;
        if (col1[op] && t_row >= free0) w[op] = probe(pos0[op] + (uint32_t)kStep, sr[op]);
      }
#pragma unroll
      for (uint32_t op = 0; op < kOps; ++op) z1[op] = __ballot(!summary_nonempty(w[op], sr[op]));
#pragma unroll
      for (uint32_t u = 0; u < kRound1Batch; ++u) {  // lane g < 6: group g of read u of the batch
        const uint32_t op = u / kRpo, at = kPer * (u % kRpo) + ln * (uint32_t)R;
        const uint32_t a0 = (uint32_t)(z0[op] >> at) & kFull, a1 = (uint32_t)(z1[op] >> at) & kFull;
        const uint32_t free0 = (uint32_t)__builtin_ctz(~a0);                                   // rows 1..free0 are free at column 0
        const uint32_t from1 = a1 == kFull ? 0u : 32u - (uint32_t)__builtin_clz(~a1 & kFull);  // rows from1+1..R are free at column 1
        const uint32_t live = (uint32_t)__ballot(ln < 2u * (uint32_t)kStep && !(from1 <= free0)) & okm[u];
        rd_live = my_rb == b0 + u ? live : rd_live;
      }
    }
    // ---- the live groups' places in their reads' stretches of the group queue, for the whole block ----
    {
      constexpr uint32_t kPhase0 = 1u | (1u << kStep);  // groups of phase 0, one per strand
      uint32_t tot = 0;
#pragma unroll
      for (int si = 0; si < kStep; ++si) tot += ((rd_used >> (8 * si)) & 255u) * (uint32_t)__popc(rd_live & (kPhase0 << si));
      rd_total = tot | (((rd_used & 255u) - (uint32_t)((R - 1) * kLg)) << 16);  // columns of phase group 0: no group has more
      const uint32_t pl = ln < 8u * 2u * (uint32_t)kStep ? ln : 0u, prb = (pl * 43u) >> 8, g = pl - 2u * (uint32_t)kStep * prb;  // pl / 6
      static_assert(kStep == 3, "the division above");
      const uint32_t strand = g >= (uint32_t)kStep ? 1u : 0u, si = g - strand * (uint32_t)kStep;
#pragma unroll
      for (uint32_t half = 0; half < 2u; ++half) {
        const int src = (int)(2u * (prb + 8u * half));
        const uint32_t live_r = (uint32_t)__shfl((int)rd_live, src), used_w = (uint32_t)__shfl((int)rd_used, src);
        const uint32_t o0 = (uint32_t)__shfl((int)rd_org0, src), o1 = (uint32_t)__shfl((int)rd_org1, src);
        const uint32_t org = strand ? o1 : o0;
        const uint32_t below = live_r & ((1u << g) - 1u);
        uint32_t first = 0;
#pragma unroll
        for (int s2 = 0; s2 < kStep; ++s2) first += ((used_w >> (8 * s2)) & 255u) * (uint32_t)__popc(below & (kPhase0 << s2));
        const bool on = ln < 8u * 2u * (uint32_t)kStep && ((live_r >> g) & 1u);
        const uint32_t used = (used_w >> (8u * si)) & 255u;
        pg_word[half] = on ? first | (used << kPgUsedShift) | ((uint32_t)__popc(below) << kPgOrdShift) | (1u << kPgOnBit) | (((strand << 2) | si) << kPgTagShift) : 0u;
        pg_base[half] = org + si - (uint32_t)kStep * first;
      }
    }
  }
  // HASH form: four characters of a read at any byte offset
  auto chars_at = [&](uint64_t off, uint32_t idx) -> uint32_t { return load_u32_unaligned(p.bases + off + idx); };
  // One extra turn after the block's last read flushes the queue of small reads; flush_small and finish_read are
  // called from one place each (they are large, and inlined).
  for (uint32_t rb = 0;; ++rb) {
    const bool tail_turn = rb >= kReadBlock || r0 + rb >= p.n_reads;
    const uint32_t read = r0 + rb;
    uint64_t off = 0;
    uint32_t L = 0;
    if (HASH && !tail_turn) off = p.read_off[read], L = (uint32_t)(p.read_off[read + 1] - off);
    const int S = (int)L - kK + 1;  // num_seeds_in_read
    if (!HASH) {
      const int sel = (int)(2u * (rb & (kReadBlock - 1u)));  // a lane with the read's parameters
      const uint32_t live = tail_turn ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)rd_live, sel);
      const uint32_t total_w = (uint32_t)__builtin_amdgcn_readlane((int)rd_total, sel);
      const uint32_t total = tail_turn ? 0u : total_w & 0xFFFFu;  // seeds of all live groups
      // the queues are flushed while no read is in flight: before a read whose live groups do not fit, and at the
      // block's end
      if (tail_turn || gq_groups + (uint32_t)__popc(live) > kGroups || gq_entries + total > p.lay.gq_cap) select_flush(r0, tail_turn);
      if (tail_turn) break;
      if ((uint32_t)__builtin_amdgcn_readlane((int)rd_flags, sel) & kRdSlow) {
        queue_slow(read);
        continue;
      }
      if (!live) continue;  // (begin / count stay 0)
      // hash of the seed at a stream position: the 24 bits that end 2 (pos + k) bits into the stream
      auto stream_hash = [&](uint32_t pos) -> uint32_t {
        const uint32_t end2 = 2u * pos + 2u * (uint32_t)kK;
        const uint32_t *st = strm_fwd + ((end2 - 1u) >> 5);  // st[0], st[1]: the word before the one the window ends in, and that one
        return __builtin_amdgcn_alignbit(st[0], st[1], 0u - end2) & kHashMask;
      };
      // ---- round 2: the live groups' seeds, straight into the group queue as hash << 8 | frequency.  Empty buckets
      // (summary) have 0, non-empty ones without the "two or more" mark exactly 1; only the others read the table.
      const uint32_t groups_before = gq_groups, entries_before = gq_entries, maxcols_before = gq_maxcols;
      uint32_t big_freq = 0;  // some bucket holds more than the queue's 8-bit field takes
      // The live groups' seeds, one after the other, are the read's stretch of the queue; its lanes of pg_word / pg_base
      // say where each group starts.  Lane t of the stretch finds its seed from them: position = base + step * t.
      // kStreams x 64 of them go through the dependent levels (summary test, table read) together.
      const uint32_t pg_lane0 = 2u * (uint32_t)kStep * (rb & 7u);
      const uint32_t pw = rb < 8u ? pg_word[0] : pg_word[1], pb = rb < 8u ? pg_base[0] : pg_base[1];
      if (ln - pg_lane0 < 2u * (uint32_t)kStep && ((pw >> kPgOnBit) & 1u)) {
        const uint32_t used = (pw >> kPgUsedShift) & 255u;
        gq_desc[gq_groups + ((pw >> kPgOrdShift) & 7u)] =
            make_uint4(gq_entries + (pw & ((1u << kPgFirstBits) - 1u)), used, used - (uint32_t)((R - 1) * kLg), (rb << 3) | ((pw >> kPgTagShift) & 7u));
      }
      gq_groups += (uint32_t)__popc(live);
      {
        const uint32_t widest = total_w >> 16;
        gq_maxcols = widest > gq_maxcols ? widest : gq_maxcols;
      }
#ifndef FEM_LEAN_STREAMS
#define FEM_LEAN_STREAMS 1
#endif
      constexpr int kStreams = FEM_LEAN_STREAMS;
      for (uint32_t t0 = 0; t0 < total; t0 += (uint32_t)(kStreams * kWave)) {
        uint32_t hh[kStreams], fq[kStreams], w1[kStreams];
        bool act[kStreams], ne[kStreams];
#pragma unroll
        for (int v = 0; v < kStreams; ++v) {
          const uint32_t t = t0 + (uint32_t)(v * kWave) + ln;
          act[v] = t < total;
          uint32_t base = 0;
          for (uint32_t m = live; m; m &= m - 1u) {  // (the first live group starts at 0)
            const int gl = (int)(pg_lane0 + (uint32_t)__builtin_ctz(m));
            const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)pw, gl) & ((1u << kPgFirstBits) - 1u);
            const uint32_t gbase = (uint32_t)__builtin_amdgcn_readlane((int)pb, gl);
            base = t >= first ? gbase : base;
          }
          hh[v] = act[v] ? stream_hash(base + (uint32_t)kStep * t) : 0u;
          fq[v] = 0;
        }
        if (p.summary) {
          uint32_t sr[kStreams];
#pragma unroll
          for (int v = 0; v < kStreams; ++v) {
            uint32_t q;
            summary_slot(hh[v], q, sr[v]);
            w1[v] = act[v] ? p.summary[q] : 0u;
          }
#pragma unroll
          for (int v = 0; v < kStreams; ++v) {
            const bool some = summary_nonempty(w1[v], sr[v]);
            fq[v] = some ? 1u : 0u;
            ne[v] = some && summary_multi(w1[v], sr[v]);  // the table has to be read
          }
        } else {
#pragma unroll
          for (int v = 0; v < kStreams; ++v) ne[v] = act[v];
        }
        uint2 tb[kStreams];
#pragma unroll
        for (int v = 0; v < kStreams; ++v) {
          tb[v] = make_uint2(0u, 0u);
          if (ne[v]) __builtin_memcpy(&tb[v], p.lookup + hh[v], 8);  // plain load: `nt` was measured 40 % slower here
        }
#pragma unroll
        for (int v = 0; v < kStreams; ++v) {
          if (ne[v]) fq[v] = tb[v].y - tb[v].x;
          big_freq |= (uint32_t)(fq[v] > 254u);
          if (act[v]) gq_row[gq_entries + t0 + (uint32_t)(v * kWave) + ln] = (hh[v] << 8) | (fq[v] & 255u);
        }
      }
      gq_entries += total;
      if (__any(big_freq != 0)) {  // the generic kernel takes the read: take its groups out of the queue again
        gq_groups = groups_before, gq_entries = entries_before, gq_maxcols = maxcols_before;
        if (ln / 2u == rb) blk_entries[ln] = make_uint2(kBlkSkip, 0u);
        queue_slow(read);
        continue;
      }
      if (live) gq_reads |= 1u << rb;
      wave_sync_lds();
      continue;
    }
    if (tail_turn) break;
    bool slow = false, selected = false;  // selected: seeds are in the lanes (or the read is `slow`)
    bool strand_ok[2] = {true, true};
    uint32_t pre_g = 0;
    unsigned long long pre_read = 0;
    uint32_t s_start = 0, s_lo = 0, s_freq = 0;
    uint64_t nonempty = 0;
    uint32_t strand_total[2] = {0, 0}, s_at = 0;
    do {
    STAMP_START(prof);
    // ---- gates (src/filter.c:161-172) + the shapes on which the reference DP is undefined ----
    bool shape_ok = S > 0 && R <= S / kStep;
    if (shape_ok) shape_ok = (S - (kStep - 1)) / kStep - R * kLg + 2 >= 2;
    if (!shape_ok) {
      if (ln / 2u == rb) blk_entries[ln] = make_uint2(0u, 0u);  // both entries stay 0
      break;
    }
    const uint32_t widest = (uint32_t)(S / kStep - R * kLg + 1);  // columns of phase group 0
    slow = widest > (uint32_t)kWave || (uint32_t)S > smax;
    selected = true;
    if (!slow) {
      // ---- encode ----
      uint32_t any_n = 0;
      for (uint32_t b0 = 0; b0 < L; b0 += 256u) {
        const uint32_t idx = b0 + 4u * ln;
        if (idx < L) {
          uint32_t code, nflag;
          encode4(chars_at(off, idx), code, nflag);  // may run up to 3 bytes past the read: masked below
          const uint32_t nb = L - idx;
          const uint32_t keep = nb >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nb)) - 1u);
          nflag &= keep;
          code &= keep & ~(nflag * 3u);  // N -> A (src/utils.h:92); bases past the end never enter a window
          const uint32_t byte_addr = (idx >> 4) * 4u + (3u - ((idx >> 2) & 3u));  // big-endian inside each dword
          ((uint8_t *)pkw)[byte_addr] = (uint8_t)pack4(code);
          any_n |= nflag;
        }
      }
      uint32_t n_fwd_amb = 0, n_rev_amb = 0;
      const bool has_n = __any(any_n != 0);
      if (has_n) {  // rare: N masks for the hashes + the ambiguous-base gate (src/utils.h:108-114, src/filter.c:180-182)
        for (uint32_t b0 = 0; b0 < L; b0 += 256u) {
          const uint32_t idx = b0 + 4u * ln;
          if (idx < L) {
            uint32_t code, nflag;
            encode4(chars_at(off, idx), code, nflag);
            const uint32_t nb = L - idx;
            nflag &= nb >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nb)) - 1u);
            const uint32_t byte_addr = (idx >> 4) * 4u + (3u - ((idx >> 2) & 3u));
            ((uint8_t *)nkw)[byte_addr] = (uint8_t)pack4(nflag * 3u);
            for (uint32_t q = 0; q < 4u; ++q) {
              const uint32_t isn = (nflag >> (8u * q)) & 1u;
              n_fwd_amb += isn & (uint32_t)(idx + q >= (uint32_t)kK);             // offsets >= k only
              n_rev_amb += isn & (uint32_t)(L - 1u - (idx + q) >= (uint32_t)kK);  // same rule on the other strand
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
      STAMP(prof, 0);
#if defined(FEM_ABLATE) && FEM_ABLATE == 0
      selected = false;
      break;
#endif

      // ---- hashes + CSR lookups: lane j owns seed j of the + strand and seed S-1-j of the - strand ----
      // The DP of phase group si only ever looks at its first G - Lg + 1 seeds, G = (S - si) / step (row r spans columns
      // c + (r-1) Lg, c < C - 1): seeds behind the last one any group uses are not looked up.
      int last_used = 0;
      for (int si = 0; si < kStep; ++si) last_used = max(last_used, kStep * ((S - si) / kStep - kLg) + si);
      // hash of seed j and of its reverse-strand partner S-1-j (bit-reversed complement: no second encode)
      auto seed_hashes = [&](int j, uint32_t &hf, uint32_t &hr) {
        const uint32_t w = (uint32_t)j >> 4, sh = 2u * ((uint32_t)j & 15u);
        const uint64_t pw = ((uint64_t)pkw[w] << 32) | pkw[w + 1];
        hf = (uint32_t)(pw >> (64 - 2 * kK - sh)) & kHashMask;
        uint32_t nm = 0;
        if (has_n) nm = (uint32_t)((((uint64_t)nkw[w] << 32) | nkw[w + 1]) >> (64 - 2 * kK - sh)) & kHashMask;
        const uint32_t r = __brev((~hf) & ~nm & kHashMask) >> (32 - 2 * kK);  // pair order restored below
        hr = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
      };
      {
        for (int j0 = 0; j0 < S; j0 += kWave) {
          const int j = j0 + (int)ln;
          if (j < S) {
            uint32_t hf, hr;
            seed_hashes(j, hf, hr);
            uint2 qf = make_uint2(0u, 0u), qr = make_uint2(0u, 0u);
            if (strand_ok[0] && j <= last_used) __builtin_memcpy(&qf, p.lookup + hf, 8);
            if (strand_ok[1] && S - 1 - j <= last_used) __builtin_memcpy(&qr, p.lookup + hr, 8);
            if (strand_ok[0]) sf[j] = make_uint2(qf.x, qf.y - qf.x);
            if (strand_ok[1]) sf[smax + (uint32_t)(S - 1 - j)] = make_uint2(qr.x, qr.y - qr.x);
          }
        }
      }
      wave_sync_lds();
      STAMP(prof, 1);
#if defined(FEM_ABLATE) && FEM_ABLATE == 1
      selected = false;
      break;
#endif

      // ---- seed selection ----
      const uint32_t dp_w = widest <= 16u ? 16u : widest <= 32u ? 32u : 64u;
      pre_g = select_seeds_lanes<R>(p, S, strand_ok, sf, smax, dp_w, take_bits, s_start, s_lo, s_freq);
      // lane s = strand * kSeeds + group * R + run now holds that run's seed
      if (ln < 2u * kSeeds && !strand_ok[ln / kSeeds]) s_freq = 0;
      STAMP(prof, 2);
#if defined(FEM_ABLATE) && FEM_ABLATE == 2
      selected = false;
      break;
#endif
      // "candidates before the filter": uint32 sum of the strand's three M[R][C-1] (src/filter.c:202), widened; added
      // to the counter only once the read is known to stay in this kernel
      {
        const uint32_t t = pre_g + dpp_or_zero<0x111, 0xF>(pre_g) + dpp_or_zero<0x112, 0xF>(pre_g);  // lanes 2 and 5: strand sums
        pre_read = 0;
        if (strand_ok[0]) pre_read += (uint32_t)__builtin_amdgcn_readlane((int)t, 2);
        if (strand_ok[1]) pre_read += (uint32_t)__builtin_amdgcn_readlane((int)t, 5);
      }
      nonempty = __ballot(s_freq > 0);
      // occurrences selected per strand: an inclusive scan over the seed lanes (frequencies clamped so that the sum
      // cannot wrap).  A strand with <= 64 in total is done in lanes; more needs the hash-join form (HASH).
      {
        const uint32_t incl = wave_scan_add(s_freq < 65u ? s_freq : 65u);
        const uint32_t mid = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)kSeeds - 1);
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 2 * (int)kSeeds - 1);
        strand_total[0] = mid, strand_total[1] = all - mid;
        s_at = incl - (s_freq < 65u ? s_freq : 65u) - (ln >= kSeeds ? mid : 0u);
        if (!HASH) slow = strand_total[0] > (uint32_t)kWave || strand_total[1] > (uint32_t)kWave;
      }
    }

    } while (false);
    STAMP(prof, 3);
    if (selected) {
      if (!slow) slow = !finish_read(read, rb, L, strand_ok[0], strand_ok[1], s_start, s_lo, s_freq, s_at, strand_total[0], strand_total[1], nonempty, pre_read);
    }
    if (slow) queue_slow(read);
    STAMP(prof, 5);
  }
  wave_sync_lds();
  const uint2 entry = blk_entries[ln];
  wave_sync_lds();
  if (ln < 2u * kReadBlock && entry.x != kBlkSkip) {
    __builtin_nontemporal_store(entry.x, &p.cand_begin[r0 * 2u + ln]);
    __builtin_nontemporal_store(entry.y, &p.cand_count[r0 * 2u + ln]);
  }
  }
  }
#ifdef FEM_STAMPS
  prof.flush();
#endif
  pad_chunk(p, chunk);
  for (uint32_t i = ln; i < qchunk.left; i += kWave)
    if (qchunk.next + i < p.slow_cap) p.slow_queue[qchunk.next + i] = kInvalidRead;
  if (ln == 0) {
    if (pre_sum) atomicAdd(&p.stats[0], pre_sum);
    if (cand_sum) atomicAdd(&p.stats[1], cand_sum);
  }
}

}  // namespace femk
