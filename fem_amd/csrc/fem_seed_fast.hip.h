// fem_seed_fast.hip.h — the lean form of the seed + filter kernel for the parameters `FEM map` actually runs:
// k = 12, step = 3 (reference src/FEM_map.c:67-68) and R = e + 1 + a fixed at compile time (1..10).
//
// One wavefront per read:
//   encode     one unaligned dword (4 bases) per lane, SWAR char -> 2-bit code, packed into LDS
//   hash       24-bit k-mer hash per lane from the packed words; reverse strand = reversed complement of the same hash
//   lookup     one 8-byte load of lookup[h], lookup[h+1] per seed and strand            (src/index.h:22-28)
//   select     seed-selection DP with DPP prefix-min + clz traceback                    (src/filter.c:3-43)
//   lists      every occurrence of the selected seeds in one lane; last-seed rule, window filter and staged
//              greedy de-dup on wave-uniform scalars; range clip; hand-over              (src/filter.c:80-144)
// Reads that do not fit this shape (more than 64 occurrences selected on a strand, a DP group wider than 64
// columns) are appended to a queue and done by the generic seed_filter_kernel.  Results are identical.
#pragma once
#include "fem_kernels.hip.h"

namespace femk {

constexpr int kK = 12, kStep = 3, kLg = 4;
constexpr uint32_t kHashMask = (1u << (2 * kK)) - 1u;

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

template <int R>
__global__ void __launch_bounds__(256) seed_fast_kernel(SeedParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr uint32_t kSeeds = (uint32_t)(kStep * R);  // selected seeds per strand
  static_assert(2 * kStep * R <= kWave, "both strands' seeds must fit the lanes of one wave");
  const uint32_t ln = lane_id();
  const uint32_t wave_in_block = threadIdx.x >> 6;
  const uint32_t waves_per_block = blockDim.x >> 6;
  uint8_t *wbase = smem + (size_t)wave_in_block * p.lay.wave_bytes;
  uint32_t *pkw = (uint32_t *)(wbase + p.lay.pkw);
  uint32_t *nkw = (uint32_t *)(wbase + p.lay.nkw);
  uint2 *sf = (uint2 *)(wbase + p.lay.sf);
  unsigned long long *take_bits = (unsigned long long *)(wbase + p.lay.dp_bits);
  Picked *picked = (Picked *)(wbase + p.lay.picked);
  uint64_t *scatter = (uint64_t *)(wbase + p.lay.X);
  const uint32_t smax = p.lay.smax;
  const uint64_t e64 = (uint64_t)p.e;
  unsigned long long pre_sum = 0, cand_sum = 0;
  SlotChunk chunk;   // candidate slots
  SlotChunk qchunk;  // slow-read queue entries

  const uint32_t wave_global = blockIdx.x * waves_per_block + wave_in_block;
  const uint32_t n_waves = gridDim.x * waves_per_block;

  for (uint32_t read = wave_global; read < p.n_reads; read += n_waves) {
    const uint64_t off = p.read_off[read];
    const uint32_t L = (uint32_t)(p.read_off[read + 1] - off);
    const uint8_t *seq = p.bases + off;
    const int S = (int)L - kK + 1;  // num_seeds_in_read

    // ---- gates (src/filter.c:161-172) + the shapes on which the reference DP is undefined ----
    bool shape_ok = S > 0 && R <= S / kStep;
    if (shape_ok) shape_ok = (S - (kStep - 1)) / kStep - R * kLg + 2 >= 2;
    if (!shape_ok) {
      if (ln < 2) {
        p.cand_begin[read * 2u + ln] = 0;
        p.cand_count[read * 2u + ln] = 0;
      }
      continue;
    }
    const uint32_t widest = (uint32_t)(S / kStep - R * kLg + 1);  // columns of phase group 0
    bool slow = widest > (uint32_t)kWave || (uint32_t)S > smax;

    bool strand_ok[2] = {true, true};
    uint32_t pre_g = 0;
    uint32_t s_start = 0, s_lo = 0, s_freq = 0, s_grp = 0, s_run = 0;
    uint64_t nonempty = 0;
    if (!slow) {
      // ---- encode ----
      uint32_t any_n = 0;
      for (uint32_t b0 = 0; b0 < L; b0 += 256u) {
        const uint32_t idx = b0 + 4u * ln;
        if (idx < L) {
          uint32_t code, nflag;
          encode4(load_u32_unaligned(seq + idx), code, nflag);  // may run up to 3 bytes past the read: masked below
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
            encode4(load_u32_unaligned(seq + idx), code, nflag);
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

      // ---- hashes + CSR lookups: lane j owns seed j of the + strand and seed S-1-j of the - strand ----
      for (int j0 = 0; j0 < S; j0 += kWave) {
        const int j = j0 + (int)ln;
        if (j < S) {
          const uint32_t w = (uint32_t)j >> 4, sh = 2u * ((uint32_t)j & 15u);
          const uint64_t pw = ((uint64_t)pkw[w] << 32) | pkw[w + 1];
          const uint32_t hf = (uint32_t)(pw >> (64 - 2 * kK - sh)) & kHashMask;
          uint32_t nm = 0;
          if (has_n) nm = (uint32_t)((((uint64_t)nkw[w] << 32) | nkw[w + 1]) >> (64 - 2 * kK - sh)) & kHashMask;
          uint32_t r = __brev((~hf) & ~nm & kHashMask) >> (32 - 2 * kK);  // reversed complement, pair order restored below
          const uint32_t hr = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
          if (strand_ok[0]) {
            uint2 q;
            __builtin_memcpy(&q, p.lookup + hf, 8);
            sf[j] = make_uint2(q.x, q.y - q.x);
          }
          if (strand_ok[1]) {
            uint2 q;
            __builtin_memcpy(&q, p.lookup + hr, 8);
            sf[smax + (uint32_t)(S - 1 - j)] = make_uint2(q.x, q.y - q.x);
          }
        }
      }
      wave_sync_lds();

      // ---- seed selection ----
      const uint32_t dp_w = widest <= 16u ? 16u : widest <= 32u ? 32u : 64u;
      pre_g = select_seeds_dpp<R, kStep, kLg>(p, S, strand_ok, sf, smax, dp_w, take_bits, picked);
      wave_sync_lds();

      // ---- the selected seeds of both strands, one per lane: lane s = strand * kSeeds + group * R + run ----
      if (ln < 2u * kSeeds) {
        const Picked q = picked[ln];
        const uint32_t within = ln % kSeeds;
        s_start = q.start, s_lo = q.lo, s_freq = q.freq;
        s_grp = within / (uint32_t)R, s_run = within % (uint32_t)R;
        if (!strand_ok[ln / kSeeds]) s_freq = 0;
      }
      nonempty = __ballot(s_freq > 0);
      // a strand fits the lanes if its seeds hold at most 64 occurrences in total
      for (uint32_t strand = 0; strand < 2u && !slow; ++strand) {
        uint32_t total = 0;
        for (uint64_t m = (nonempty >> (strand * kSeeds)) & ((1ull << kSeeds) - 1ull); m;) {
          const int j = __builtin_ctzll(m) + (int)(strand * kSeeds);
          m &= m - 1;
          total += (uint32_t)__builtin_amdgcn_readlane((int)s_freq, j);
          if (total > (uint32_t)kWave) break;  // (also keeps the sum from wrapping)
        }
        slow = total > (uint32_t)kWave;
      }
    }

    if (slow) {  // hand the whole read to the generic kernel
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
      continue;
    }

    // ---- per strand: lists in registers (same steps as strand_small() of the generic kernel) ----
    for (uint32_t strand = 0; strand < 2u; ++strand) {
      uint32_t n_out = 0, base = 0;
      bool ok = false;
      uint64_t cv = 0;  // lane i holds candidate i (sorted, before the range clip)
      uint64_t mo = 0;
      if (strand_ok[strand]) {
        uint32_t pre = 0;  // uint32 sum of the groups' M[R][C-1] (src/filter.c:202)
        for (int si = 0; si < kStep; ++si) pre += (uint32_t)__builtin_amdgcn_readlane((int)pre_g, (int)(strand * kStep) + si);
        pre_sum += pre;
        const uint32_t lane0 = strand * kSeeds;
        const uint64_t ne = (nonempty >> lane0) & ((1ull << kSeeds) - 1ull);
        uint32_t total = 0;
        for (uint64_t m = ne; m;) {
          const int j = __builtin_ctzll(m) + (int)lane0;
          m &= m - 1;
          total += (uint32_t)__builtin_amdgcn_readlane((int)s_freq, j);
        }
        uint32_t kept = 0;
        if (total > (uint32_t)p.a) {
          // one occurrence per lane, in (group, run) order = the order of the merged lists
          bool valid = false;
          uint64_t v = 0;
          uint32_t grp = 0, run = 0;
          uint32_t at = 0;
          for (uint64_t m = ne; m;) {
            const int j = __builtin_ctzll(m) + (int)lane0;
            m &= m - 1;
            const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)s_freq, j);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)s_lo, j);
            const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)s_start, j);
            const uint32_t gj = (uint32_t)__builtin_amdgcn_readlane((int)s_grp, j);
            const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)s_run, j);
            if (ln >= at && ln < at + f) {
              const uint64_t o = p.occ[(uint64_t)lo + (ln - at)];
              valid = (uint32_t)o >= st;  // src/filter.c:89,106
              v = o - st;
              grp = gj, run = rj;
            }
            at += f;
          }
          // last seed of each group: only values <= max of the other runs survive (src/filter.c:85)
          for (uint32_t g = 0; g < (uint32_t)kStep; ++g) {
            const bool is_last = valid && grp == g && run == (uint32_t)(R - 1);
            if (!__ballot(is_last)) continue;
            const uint64_t mu = __ballot(valid && grp == g && run != (uint32_t)(R - 1));
            uint64_t max_u = 0;
            for (uint64_t m = mu; m;) {
              const int j = __builtin_ctzll(m);
              m &= m - 1;
              const uint64_t x = readlane64(v, j);
              max_u = x > max_u ? x : max_u;
            }
            if (is_last && (mu == 0 || v > max_u)) valid = false;
          }
          // additional_qgram_filter (src/filter.c:118-131): >= a+1 values of the same group in [v, v+e]
          const uint64_t vm = __ballot(valid);
          bool pass = false;
          if ((uint32_t)__popcll(vm) > (uint32_t)p.a) {
            uint32_t cnt = 0;
            for (uint64_t m = vm; m;) {
              const int j = __builtin_ctzll(m);
              m &= m - 1;
              const uint64_t x = readlane64(v, j);
              const uint32_t gj = (uint32_t)__builtin_amdgcn_readlane((int)grp, j);
              cnt += (uint32_t)(gj == grp && x >= v && x <= v + e64);
            }
            pass = valid && cnt > (uint32_t)p.a;
          }
          // merge_kvec_t_uint64_t (src/filter.c:45-78), group after group
          if (__ballot(pass)) {
            uint32_t nA = 0;
            for (uint32_t g = 0; g < (uint32_t)kStep; ++g) {
              const bool mine = pass && grp == g;
              const uint64_t mf = __ballot(mine);
              const uint32_t nF = (uint32_t)__popcll(mf);
              if (nF == 0) continue;
              uint32_t rank = 0;  // position of v among this group's survivors
              for (uint64_t m = mf; m;) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const uint64_t x = readlane64(v, j);
                rank += (uint32_t)(x < v || (x == v && (uint32_t)j < ln));
              }
              wave_sync_lds();
              if (mine) scatter[rank] = v;
              wave_sync_lds();
              const uint64_t fs = ln < nF ? scatter[ln] : 0;
              uint64_t merged = 0, last_kept = 0;
              uint32_t nB = 0, ia = 0, jf = 0;
              while (ia < nA || jf < nF) {  // wave-uniform two-pointer merge + greedy gap rule
                const uint64_t xa = readlane64(cv, (int)(ia < nA ? ia : 0));
                const uint64_t xf = readlane64(fs, (int)(jf < nF ? jf : 0));
                const bool take_a = ia < nA && (jf >= nF || xa < xf);
                const uint64_t x = take_a ? xa : xf;
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
        // remove_out_ranged_candidates (src/filter.c:133-144)
        if (ln < kept) {
          const uint32_t sq = (uint32_t)(cv >> 32), pos = (uint32_t)cv;
          const uint32_t slen = p.seq_len[sq];
          ok = pos >= (uint32_t)p.e && pos + L + (uint32_t)p.e < slen;
        }
        mo = __ballot(ok);
        n_out = (uint32_t)__popcll(mo);
      }
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
          const uint32_t at = base + (uint32_t)__popcll(mo & ((1ull << ln) - 1ull));
          p.cand[at] = cv - e64;
          p.cand_meta[at] = read * 2u + strand;
        }
      }
      if (ln == 0) {
        p.cand_begin[read * 2u + strand] = base;
        p.cand_count[read * 2u + strand] = n_out;
      }
      cand_sum += n_out;
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
