// fem_seed_select.hip.h — seed selection for DENSE indexes as a kernel of its own (k = 12, step = 3, R at compile time):
// hash_all_seeds_in_sequence + the frequency lookups + generate_optimal_prefix_qgram_for_group_seeding + the stable
// qsort of the selected seeds (reference src/utils.h:83-117, src/index.h:22-28, src/filter.c:3-43,146-204), for both
// strands of every read.  Its output — per read and (strand, phase group) the R selected seeds in run order as
// (lookup[h], start | frequency << 16) — is what seed_join_kernel (fem_seed_join.hip.h) walks the lists of.
//
// Why a kernel of its own: inside the wave-per-read kernel this front end used 6..60 of the 64 lanes and 178 random
// 8-byte table reads per read, each of which moves a 64-byte sector across the fabric.  Here the work is laid out so
// that every lane is busy, over blocks of reads:
//   * FREQUENCY PAIRS BY 11-MER.  The DP only needs frequencies, and the seeds at read offsets j and j + 1 share eleven
//     bases X.  The derived table freq11 holds, per 11-mer X, sixteen saturated byte frequencies: of the four 12-mers
//     a.X and the four X.b, and of the reverse complements of those eight (which are again extensions of rc(X)).  Lane
//     j reads ONE dword of it — X = the seed's last eleven bases when j is even, its first eleven when j is odd — and
//     has the frequency of seed j on the forward strand and of its reverse complement (seed S-1-j of the reverse
//     strand).  89 four-byte lane-loads in 45 sectors per 100-base read instead of 178 eight-byte ones in 178 sectors.
//     lookup[h] itself is fetched for the 6 R selected seeds only.  A byte that reads 255 ("255 or more": never on
//     BASELINE's references, whose buckets hold 60 +- 8 entries; a real genome's repeats) has the lane fetch the exact
//     frequency from the lookup table; the DP works on 16-bit frequencies, and only a bucket of 65 535 entries or more
//     sends the read to the generic kernel.
//   * ONE LANE PER PHASE GROUP.  The DP table of a group is R rows by C - 1 <= 64 columns.  A lane walks it column by
//     column with the R running row values in registers; the frequencies it needs for a column are R 16-bit LDS reads
//     off one address (seed index = column + 4 (row - 1)).  Five vector instructions per cell, 64 groups (ten reads)
//     at a time, against one DPP prefix-min chain per row and group before.  Take bits are shifted into per-row masks; the traceback, the
//     frequency sort (a sorting network on frequency << 14 | traceback order << 10 | start: stable by construction)
//     and the lookup of the selected seeds' list bases stay in the lane.
#pragma once
#include <type_traits>

#include "fem_seed_fast.hip.h"

namespace femk {

constexpr uint32_t kDenseMaxBanks = 4;       // banks of sequences with coordinates of their own (fem_seed_dense.hip.h)
constexpr uint32_t kDenseNear = 1024u;         // entries of the 32-bit occurrence table with pos < this are stored remapped ...
constexpr uint32_t kDenseRemap = 0xF0000000u;  // ... as kDenseRemap | seq << 10 | pos (fem_seed_dense.hip.h)
constexpr uint32_t kSelKeepAll = 1u << 15;  // in a last run's (start | frequency << 16) word: no truncation at max(U) in this bank
constexpr uint32_t kSelOk = 0u, kSelNone = 1u, kSelSlow = 2u;  // sel_hdr[read].x & 3: joined / no candidates / generic kernel
constexpr uint32_t kSelMaxCols = 128u;                         // DP columns a lane's take masks hold (1, 2 or 4 words per row)
constexpr uint32_t kSelMaxList = 128u;                         // longest list seed_join_kernel takes (kDenseMaxList)
constexpr uint32_t kX11 = 1u << 22;                            // number of 11-mers

// wave-wide min / max of a 32-bit value (DPP scans; the result is wave-uniform)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
  constexpr uint32_t kFill = 0xFFFFFFFFu;
  x = dpp_min_step<0x111, 0xF>(x, kFill), x = dpp_min_step<0x112, 0xF>(x, kFill), x = dpp_min_step<0x114, 0xF>(x, kFill);
  x = dpp_min_step<0x118, 0xF>(x, kFill), x = dpp_min_step<0x142, 0xA>(x, kFill), x = dpp_min_step<0x143, 0xC>(x, kFill);
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_max(x), 63);
}

// reverse complement of an 11-mer (22 bits)
__device__ __forceinline__ uint32_t rc11(uint32_t x) {
  const uint32_t r = __brev(~x & (kX11 - 1u)) >> 10;
  return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}

// freq11[X * 4 + i], bytes: 0 = f(i.X)   1 = f(rc(i.X))   2 = f(X.i)   3 = f(rc(X.i)),  f saturated at 255
__global__ void freq11_kernel(const uint32_t *lookup, uint32_t *freq11) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < kX11 * 4u; t += stride) {
    const uint32_t X = t >> 2, i = t & 3u, rx = rc11(X);
    auto f = [&](uint32_t h) -> uint32_t {
      const uint32_t d = lookup[h + 1u] - lookup[h];
      return d < 255u ? d : 255u;
    };
    const uint32_t h_l = (i << 22) | X, h_r = (X << 2) | i;
    const uint32_t h_lrc = (rx << 2) | (3u - i), h_rrc = ((3u - i) << 22) | rx;
    freq11[t] = f(h_l) | (f(h_lrc) << 8) | (f(h_r) << 16) | (f(h_rrc) << 24);
  }
}

// 32 consecutive bits of a big-endian packed 2-bit stream, starting at base `pos`.  The word pair is taken one base
// early — words (pos + 15) / 16 - 1 and the next —, so that the funnel shift is by -2 pos mod 32 in every case: where the
// window starts on a word boundary the shift is 0 and the first word of the pair (strm[-1] at pos 0: any readable word
// of the wave's LDS) drops out.  No branch and no multiply (a 32-bit integer multiply issues at a quarter of the rate).
__device__ __forceinline__ uint32_t stream_window(const uint32_t *strm, uint32_t pos) {
  const uint32_t *w = strm + ((pos + 15u) >> 4) - 1;
  uint32_t twice = pos << 1;
  asm("" : "+v"(twice));  // (left to itself the compiler makes -2 pos mod 32 a multiplication by 30)
  return __builtin_amdgcn_alignbit(w[0], w[1], 0u - twice);
}
// x / 3 and x / 6 for x < 4096 with the full-rate 24-bit multiply (the compiler's division by a constant is a v_mul_hi_u32)
__device__ __forceinline__ uint32_t div3_small(uint32_t x) { return __umul24(x, 21846u) >> 16; }
__device__ __forceinline__ uint32_t div6_small(uint32_t x) { return __umul24(x, 10923u) >> 16; }
// 3 x as one shift-add (written as a product the compiler folds it into a v_mul_lo_u32 / v_mad_u64_u32 with its neighbours)
__device__ __forceinline__ uint32_t times3(uint32_t x) {
  uint32_t r;
  asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(r) : "v"(x));
  return r;
}
// hash of the reverse complement of a 12-mer whose forward hash is hf (N counted as A on both strands: nm marks them)
__device__ __forceinline__ uint32_t rc_hash(uint32_t hf, uint32_t nm) {
  const uint32_t r = __brev((~hf) & ~nm & kHashMask) >> (32 - 2 * kK);
  return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}

// ---------------------------------------------------------------------------------------------------------
// The seed-selection DP of one phase group in one lane (src/filter.c:3-28).  F = the frequencies of the strand's seeds from
// the group's first one on, 16 bits per seed, in read order: the group's seed g sits at F[3 g] (round 5: one linear array
// per strand — round 4 kept an array per phase group, and the lanes that fill them paid a division by three per seed);
// ncols = C - 1 of this lane's group (0: idle lane), maxcols = the largest in the wave.  Column c (0-based) of row r
// (0-based) uses the group's seed c + 4 r.  take[r] receives the take bits: column c at bit iters - 1 - c.
// ---------------------------------------------------------------------------------------------------------
template <int R, int W>
__device__ __forceinline__ void select_dp(const uint16_t *F, uint32_t ncols, uint32_t maxcols, uint32_t inf, uint32_t (&take)[R][W],
                                          uint32_t &m_last, uint32_t &iters) {
  uint32_t M[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    M[r] = inf;
#pragma unroll
    for (int w = 0; w < W; ++w) take[r][w] = 0;
  }
  for (uint32_t c = 0; c < maxcols; ++c) {
    uint32_t f[R];
#pragma unroll
    for (int r = 0; r < R; ++r) f[r] = F[3u * c + 12u * (uint32_t)r];  // (one address, R immediate offsets)
    const bool in = c < ncols;
    uint32_t up = 0;  // M[0][c] = 0
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t v = up + f[r];      // uint32 wrap as in the reference
      const bool tk = in && v < M[r];    // strict: ties go horizontal (src/filter.c:20); M[r][0] = inf (src/filter.c:9)
      M[r] = tk ? v : M[r];
      up = M[r];
#pragma unroll
      for (int w = W - 1; w > 0; --w) take[r][w] = (take[r][w] << 1) | (take[r][w - 1] >> 31);
      take[r][0] = (take[r][0] << 1) | (uint32_t)tk;
    }
  }
  m_last = M[R - 1];
  iters = maxcols;
}
// lowest set bit at or above position `from` of a W-word mask (word 0 = bits 0..31); -1 if none
template <int W>
__device__ __forceinline__ int first_set_from(const uint32_t (&m)[W], uint32_t from) {
  int best = -1;
#pragma unroll
  for (int w = W - 1; w >= 0; --w) {  // (high words first: the lowest hit is written last)
    const uint32_t lo = 32u * (uint32_t)w;
    uint32_t x = m[w];
    if (from > lo) x = from - lo >= 32u ? 0u : x & (0xFFFFFFFFu << (from - lo));
    if (x) best = (int)lo + __builtin_ctz(x);
  }
  return best;
}

#ifndef FEM_SELECT_WAVES
#define FEM_SELECT_WAVES 6
#endif
#ifndef FEM_SELECT_UNROLL
#define FEM_SELECT_UNROLL 2
#endif

template <int R, bool BANKED = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FEM_SELECT_WAVES, 8))) seed_select_kernel(SeedParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t ln = lane_id();
  const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t *wbase = smem + (size_t)wave_in_block * p.lay.wave_bytes;
  uint64_t *boff = (uint64_t *)(wbase + p.lay.rb);    // offsets of the block's reads (kReadBlock + 1)
  uint32_t *fw = (uint32_t *)(wbase + p.lay.strm);    // the sub-block's bases, 2 bits each (N as A), 16 per big-endian word
  uint32_t *nw = fw + p.lay.strm_words;               // ... and its N marks (3 = not one of ACGT)
  uint16_t *fq = (uint16_t *)(wbase + p.lay.fq);      // [read][strand][sstride] frequencies, 16 bits per seed, in the strand's seed order
  uint32_t *r_base = (uint32_t *)(wbase + p.lay.rinfo), *r_len = r_base + kReadBlock, *r_flag = r_len + kReadBlock,
           *r_pre = r_flag + kReadBlock;              // per read of the sub-block
  constexpr uint32_t kOk0 = 1u, kOk1 = 2u, kShape = 4u, kSlow = 8u;
  const uint32_t sstride = p.lay.gstride, nb = p.lay.nb;  // (elements per strand: even, rows on 4-byte boundaries)
  const uint32_t inf = p.inf32;
  SlotChunk qchunk;

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
    wave_sync_lds();
    if (ln <= n_blk) boff[ln] = p.read_off[r0 + ln];
    wave_sync_lds();
    for (uint32_t sb = 0; sb < n_blk; sb += nb) {
      const uint32_t cnt = n_blk - sb < nb ? n_blk - sb : nb;
      const uint32_t rd0 = r0 + sb;
      const uint64_t o_first = boff[sb];
      const uint64_t blk_off = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)o_first) |
                               ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(o_first >> 32)) << 32);
      // ---- per read (lane i): place in the stream, length, gates (src/filter.c:161-172) and the shapes on which the
      //      reference DP is undefined (src/filter.c:5-7) ----
      uint32_t my_S = 0, my_flag = 0;
      {
        uint32_t L = 0, base = 0;
        if (ln < cnt) {
          const uint64_t o0 = boff[sb + ln], o1 = boff[sb + ln + 1u];
          L = (uint32_t)(o1 - o0), base = (uint32_t)(o0 - blk_off);
          const int S = (int)L - kK + 1;
          bool shape_ok = S > 0 && R <= S / kStep;
          if (shape_ok) shape_ok = (S - (kStep - 1)) / kStep - R * kLg + 2 >= 2;
          if (shape_ok) {
            my_S = (uint32_t)S;
            my_flag = kShape | kOk0 | kOk1;
            const uint32_t widest = (uint32_t)(S / kStep - R * kLg + 1);  // columns of phase group 0
            if (widest > kSelMaxCols || p.a == 0) my_flag |= kSlow;
          }
        }
        if (ln < kReadBlock) r_base[ln] = base, r_len[ln] = L, r_flag[ln] = my_flag, r_pre[ln] = 0;
      }
      const uint32_t total_chars = (uint32_t)(boff[sb + cnt] - blk_off);
      // ---- encode the sub-block: four characters per lane -> one byte of each stream ----
      uint32_t any_n = 0;
      {
        const uint8_t *src = p.bases + blk_off;
        for (uint32_t q = ln; 4u * q < total_chars; q += (uint32_t)kWave) {
          uint32_t code, nflag;
          encode4(load_u32_unaligned(src + 4u * q), code, nflag);  // may run up to 3 bytes past the sub-block: masked
          const uint32_t left = total_chars - 4u * q;
          const uint32_t keep = left >= 4u ? 0xFFFFFFFFu : ((1u << (8u * left)) - 1u);
          nflag &= keep;
          code &= keep & ~(nflag * 3u);  // N -> A (src/utils.h:92)
          const uint32_t byte_addr = (q >> 2) * 4u + (3u - (q & 3u));
          ((uint8_t *)fw)[byte_addr] = (uint8_t)pack4(code);
          ((uint8_t *)nw)[byte_addr] = (uint8_t)pack4(nflag * 3u);
          any_n |= nflag;
        }
      }
      const bool has_n = __builtin_amdgcn_ballot_w64(any_n != 0) != 0;
      wave_sync_lds();
      if (has_n) {
        // rare: the ambiguous-base gate per read and strand (src/utils.h:108-114, src/filter.c:180-182); bases at
        // offsets >= k count on the forward strand, at offsets <= L - 1 - k on the reverse strand
        for (uint32_t i = 0; i < cnt; ++i) {
          const uint32_t L = r_len[i], base = r_base[i];
          uint32_t n_f = 0, n_r = 0;
          for (uint32_t b0 = 0; b0 < L; b0 += (uint32_t)kWave) {
            const uint32_t at = b0 + ln;
            if (at < L) {
              const uint32_t pos = base + at;
              const uint32_t isn = (nw[pos >> 4] >> (30u - 2u * (pos & 15u))) & 1u;
              n_f += isn & (uint32_t)(at >= (uint32_t)kK);
              n_r += isn & (uint32_t)(L - 1u - at >= (uint32_t)kK);
            }
          }
          for (int d = 32; d >= 1; d >>= 1) n_f += __shfl_xor(n_f, d), n_r += __shfl_xor(n_r, d);
          if (ln == 0) {
            uint32_t fl = r_flag[i];
            if (n_f > (uint32_t)p.e) fl &= ~kOk0;
            if (n_r > (uint32_t)p.e) fl &= ~kOk1;
            r_flag[i] = fl;
          }
        }
        wave_sync_lds();
      }
      // ---- frequencies of every seed on both strands: lane (read i, seeds 2 q and 2 q + 1) — the two share eleven bases X,
      //      so their freq11 dwords sit in one 16-byte entry (one sector): one place in the stream, one window of it, two
      //      loads; the four bytes they bring (seed 2 q and 2 q + 1 on the forward strand, their reverse complements =
      //      seeds S - 1 - 2 q and S - 2 - 2 q of the reverse strand) go out as one aligned 32-bit store and two 16-bit ones.
      //      (Round 4: a lane per seed, the place worked out twice — before the load and behind it —, and a division by
      //      three per store for the per-group arrays: 122 of the kernel's 253 vector instructions per read.) ----
      const uint32_t SP = wave_max_u32(my_S);
      if (SP != 0) {
        const uint32_t SPp = (SP + 1u) >> 1;             // pairs per read
        const uint32_t magic = 0xFFFFFFFFu / SPp + 1u;   // w / SPp = umulhi(w, magic) for w * SPp < 2^32
        const uint32_t total = cnt * SPp;
        constexpr int U = FEM_SELECT_UNROLL;
        constexpr uint32_t kActBit = 1u << 25, kTwoBit = 1u << 24;
        // U rounds of loads are issued before the first one is used; a round's place (read | first seed << 4 | S << 14 | flags)
        // waits in one register.  The kernel shares the chip with seed_join_kernel: it is built to be small.
        for (uint32_t w0 = 0; w0 < total; w0 += (uint32_t)(U * kWave)) {
          uint32_t d0[U], d1[U], where[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t w = w0 + (uint32_t)(u * kWave) + ln;
            const uint32_t i = __umulhi(w, magic);
            const uint32_t j = (w - __umul24(i, SPp)) << 1;
            const uint32_t i_c = i < cnt ? i : 0u;
            const uint32_t fl = r_flag[i_c];
            const uint32_t S = r_len[i_c] - (uint32_t)(kK - 1);
            const bool act = w < total && (fl & kShape) && !(fl & kSlow) && j < S;
            const bool two = act && j + 1u < S;
            const uint32_t win = stream_window(fw, act ? r_base[i_c] + j : 0u);  // bases j .. j + 15: both seeds
            const uint32_t hf0 = win >> 8;
            // seed j: its last eleven bases X + its first base; seed j + 1 = X . its last base: the hash itself
            const uint32_t at0 = ((hf0 & (kX11 - 1u)) << 2) | (hf0 >> 22), at1 = (win >> 6) & kHashMask;
            d0[u] = p.freq11[act ? at0 : 0u];
            d1[u] = p.freq11[two ? at1 : 0u];
            where[u] = i_c | (j << 4) | (S << 14) | (two ? kTwoBit : 0u) | (act ? kActBit : 0u);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (where[u] & kActBit) {
              const uint32_t i = where[u] & 15u, j = (where[u] >> 4) & 1023u, S = (where[u] >> 14) & 1023u;
              const bool two = (where[u] & kTwoBit) != 0u;
              // bytes: f(seed j), f(its reverse complement), f(seed j + 1), f(its reverse complement)
              uint32_t x = (d0[u] & 0x0000FFFFu) | (d1[u] & 0xFFFF0000u);
              if (!two) x &= 0x0000FFFFu;
              uint32_t ffwd = x & 0x00FF00FFu, frev = (x >> 8) & 0x00FF00FFu;  // (seed j | seed j + 1 << 16)
              const uint32_t is255 = ((x & 0x7F7F7F7Fu) + 0x01010101u) & x & 0x80808080u;  // some byte reads 255
              if (has_n || is255 != 0u) {
                // rare: "255 or more" is looked up exactly; the reverse strand counts N as A after complementing (its own hash)
                uint32_t f2[2] = {0u, 0u}, r2[2] = {0u, 0u};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                  if (q == 1 && !two) continue;
                  uint32_t f_fwd = (x >> (16 * q)) & 255u, f_rev = (x >> (16 * q + 8)) & 255u;
                  const uint32_t pos = r_base[i] + j + (uint32_t)q;
                  const uint32_t hf = stream_window(fw, pos) >> 8;
                  const uint32_t nm = has_n ? stream_window(nw, pos) >> 8 : 0u;
                  if (f_fwd == 255u) f_fwd = p.lookup[hf + 1u] - p.lookup[hf];
                  if (nm != 0u || f_rev == 255u) {
                    const uint32_t hr = rc_hash(hf, nm);
                    f_rev = p.lookup[hr + 1u] - p.lookup[hr];
                  }
                  if (f_fwd >= 0xFFFFu || f_rev >= 0xFFFFu) atomicOr(&r_flag[i], kSlow);  // (16 bits do not hold it: generic kernel)
                  f2[q] = f_fwd < 0xFFFFu ? f_fwd : 0xFFFFu, r2[q] = f_rev < 0xFFFFu ? f_rev : 0xFFFFu;
                }
                ffwd = f2[0] | (f2[1] << 16), frev = r2[0] | (r2[1] << 16);
              }
              uint16_t *row_f = fq + __umul24(i, 2u * sstride), *row_r = row_f + sstride;
              *(uint32_t *)(row_f + j) = ffwd;  // (j is even; without a second seed a zero lands behind the strand's last one)
              const uint32_t jr = S - 1u - j;
              row_r[jr] = (uint16_t)frev;
              if (two) row_r[jr - 1u] = (uint16_t)(frev >> 16);
            }
          }
        }
      }
      wave_sync_lds();
      // ---- seed selection: lane (read i, strand, phase) ----
      const uint32_t n_gl = cnt * 6u;
      for (uint32_t gl0 = 0; gl0 < n_gl; gl0 += (uint32_t)kWave) {
        const uint32_t gl = gl0 + ln;
        const uint32_t i = div6_small(gl), u = gl - __umul24(i, 6u);
        const uint32_t i_c = gl < n_gl ? i : 0u;
        const uint32_t strand = u >= 3u ? 1u : 0u, si = u - times3(strand);
        const uint32_t fl = r_flag[i_c], S = r_len[i_c] - (uint32_t)(kK - 1);
        const bool read_ok = gl < n_gl && (fl & kShape) && !(fl & kSlow);
        const bool valid = read_ok && ((fl >> strand) & 1u);
        const uint32_t ncols = valid ? div3_small(S - si) - (uint32_t)(R * kLg) + 1u : 0u;
        const uint32_t maxcols = wave_max_u32(ncols);
        const uint16_t *F = fq + __umul24(i_c * 2u + strand, sstride) + si;
        uint32_t key[R];  // frequency << 14 | traceback order << 10 | start, per selected seed
#pragma unroll
        for (int t = 0; t < R; ++t) key[t] = (uint32_t)t << 10;  // a seed that was never taken is all zero (see below)
        if (maxcols != 0) {
          // take masks of 1, 2 or 4 words per row: up to 32 columns (every BASELINE shape), 64, 128 (reads of ~450 bases at e = 3)
          auto select_group = [&](auto words_c) {
            constexpr int W = decltype(words_c)::value;
            uint32_t take[R][W], m_last = 0, iters = 0;
            select_dp<R, W>(F, ncols, maxcols, inf, take, m_last, iters);
            if (valid) atomicAdd(&r_pre[i_c], m_last);  // M[R][C-1] (src/filter.c:202)
            // ---- traceback (src/filter.c:30-41): row R first; the highest taken column at or below the previous one.
            //      If column 0 is reached before R seeds were taken the reference's remaining seeds are uninitialised
            //      (src/filter.c:33-41): all zero here, as in the oracle ----
            int col = (int)ncols - 1;
            bool alive = valid;
#pragma unroll
            for (int r = R - 1; r >= 0; --r) {
              const int t = R - 1 - r;
              if (alive) {
                const uint32_t shift = iters - 1u - (uint32_t)col;  // bits >= shift are the columns <= col
                const int at = first_set_from<W>(take[r], shift);
                if (at < 0) {
                  alive = false;
                } else {
                  col -= at - (int)shift;
                  const uint32_t idx3 = times3((uint32_t)col + (uint32_t)(4 * r));
                  key[t] = ((uint32_t)F[idx3] << 14) | ((uint32_t)t << 10) | (si + idx3);  // (65 534 << 14 fits)
                }
              }
            }
          };
          if (maxcols <= 32u) select_group(std::integral_constant<int, 1>{});
          else if (maxcols <= 64u) select_group(std::integral_constant<int, 2>{});
          else select_group(std::integral_constant<int, 4>{});
        }
        // ---- qsort(compare_seed) (src/filter.c:204): ascending frequency, stable (glibc's merge sort):
        //      the traceback order in the key breaks ties, so any sorting network gives the stable order ----
#pragma unroll
        for (int ps = 0; ps < R; ++ps) {
#pragma unroll
          for (int x = ps & 1; x + 1 < R; x += 2) {
            const uint32_t lo = key[x] < key[x + 1] ? key[x] : key[x + 1], hi = key[x] < key[x + 1] ? key[x + 1] : key[x];
            key[x] = lo, key[x + 1] = hi;
          }
        }
        // ---- the selected seeds' list bases; hand-over ----
        if (read_ok) {
          const uint32_t base = r_base[i_c];
          uint32_t lo[R], hs[R];
          bool too_long = false;
#pragma unroll
          for (int t = 0; t < R; ++t) {
            const uint32_t f = key[t] >> 14, sidx = key[t] & 1023u;
            // (a seed that was never taken, or whose bucket is empty: in the strided table its "list" is the stretch of pads
            //  behind the last bucket, one bucket of them per run — seed_join_kernel reads a run's first chunk whatever its length)
            lo[t] = !BANKED && p.list_shift ? (p.n_buckets + (uint32_t)t) << p.list_shift : 0u, hs[t] = 0;
            if (f != 0u) {
              const uint32_t j = strand ? S - 1u - sidx : sidx;
              const uint32_t hf = stream_window(fw, base + j) >> 8;
              uint32_t h = hf;
              if (strand) h = rc_hash(hf, has_n ? stream_window(nw, base + j) >> 8 : 0u);
              // where the seed's list starts: by the hash alone in the strided table (no read), lookup[h] in the compact one
              lo[t] = !BANKED && p.list_shift ? h << p.list_shift : p.lookup[h], hs[t] = h;
            }
            too_long |= f > kSelMaxList;
          }
          if constexpr (!BANKED) {
            uint2 *out = p.sel + ((size_t)(rd0 + i_c) * 6u + u) * (uint32_t)R;
#pragma unroll
            for (int t = 0; t < R; ++t) out[t] = make_uint2(lo[t], (key[t] & 1023u) | ((key[t] >> 14) << 16));
          } else {
            // ---- banks (fem_seed_dense.hip.h): every list cut at the banks' boundaries; per bank the part's base and
            //      length, and for the last run what becomes of "values <= max(U) only" (src/filter.c:85) ----
            const uint32_t n_banks = p.n_banks;
            uint32_t cut[R];  // where the current bank's part of list t starts
#pragma unroll
            for (int t = 0; t < R; ++t) cut[t] = lo[t];
            uint32_t u_banks = 0;  // bit b: some run of U (t < R - 1) has entries in bank b
            uint32_t part[kDenseMaxBanks][R];
            bool unsure = false;
            for (uint32_t b = 0; b < n_banks; ++b) {
#pragma unroll
              for (int t = 0; t < R; ++t) {
                const uint32_t f = key[t] >> 14;
                uint32_t end = lo[t] + f;
                if (f != 0u && b + 1u < n_banks) end = p.bank_lo[(size_t)b * p.n_buckets + hs[t]];
                part[b][t] = f != 0u ? end - cut[t] : 0u;
                if (t < R - 1 && part[b][t] != 0u) {
                  u_banks |= 1u << b;
                  // "U has entries in this bank" must mean entries the reference keeps: one with pos < start is dropped
                  // (src/filter.c:89,106), and if that is all the bank holds of U, max(U) lies in a lower bank.  A part is
                  // ascending, so it is dropped whole only if its first AND its last entry are (both near a sequence
                  // start, stored remapped): such a read goes to the generic kernel (review of round 3).
                  const uint32_t start = key[t] & 1023u;
                  const uint32_t first = p.occ32[cut[t]], last = p.occ32[end - 1u];
                  if (first >= kDenseRemap && (first & (kDenseNear - 1u)) < start && last >= kDenseRemap && (last & (kDenseNear - 1u)) < start) unsure = true;
                }
              }
#pragma unroll
              for (int t = 0; t < R; ++t) cut[t] += part[b][t];
            }
            too_long |= unsure;
#pragma unroll
            for (int t = 0; t < R; ++t) cut[t] = lo[t];
            for (uint32_t b = 0; b < n_banks; ++b) {
              uint2 *out = p.sel + (((size_t)(rd0 + i_c) * n_banks + b) * 6u + u) * (uint32_t)R;
              const bool above = (u_banks >> (b + 1u)) != 0u;                 // U has entries in a higher bank: keep all of the last run
              const bool gone = !above && !((u_banks >> b) & 1u) && u_banks;  // ... only in lower banks: the last run is dropped here
#pragma unroll
              for (int t = 0; t < R; ++t) {
                uint32_t f_b = part[b][t];
                uint32_t word = key[t] & 1023u;
                if (t == R - 1) {
                  if (gone) f_b = 0;
                  if (above) word |= kSelKeepAll;
                }
                out[t] = make_uint2(cut[t], word | (f_b << 16));
                cut[t] += part[b][t];
              }
            }
          }
          if (too_long) atomicOr(&r_flag[i_c], kSlow);
        }
      }
      wave_sync_lds();
      // ---- per read: what the join kernel is to do with it ----
      {
        const uint32_t fl = ln < cnt ? r_flag[ln] : 0u;
        const uint32_t status = !(fl & kShape) ? kSelNone : (fl & kSlow) ? kSelSlow : kSelOk;
        if (ln < cnt) p.sel_hdr[rd0 + ln] = make_uint2(status | (r_len[ln] << 8), r_pre[ln]);
        for (uint64_t m = __builtin_amdgcn_ballot_w64(ln < cnt && status == kSelSlow); m; m &= m - 1) queue_slow(rd0 + (uint32_t)__builtin_ctzll(m));
      }
      wave_sync_lds();
    }
  }
  for (uint32_t i = ln; i < qchunk.left; i += kWave)
    if (qchunk.next + i < p.slow_cap) p.slow_queue[qchunk.next + i] = kInvalidRead;
}

}  // namespace femk
