// fem_tail.hip — device mapping tail: kernels + the host-side driver behind fem_dev_fetch_records.
// See fem_tail.hip.h for what is computed and the reference lines it follows.
#include "fem_tail.hip.h"
#include "fem_planes.hip.h"

#include <cstring>  // before rocprim: its headers use memcpy unqualified

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/fem_hip.h"

namespace femt {
namespace {

constexpr uint32_t kOpM = 0, kOpI = 1, kOpD = 2, kOpS = 3;  // S: the traceback's pseudo-run (src/align.c:342), never emitted
constexpr uint32_t kOpsCap = 16, kMdCap = 32;  // first-pass staging per record; anything longer goes to the overflow pass
constexpr uint32_t kSortLdsHits = 2048;         // hits of one read the ordering kernel keeps in LDS
constexpr uint16_t kFlagBroken = 0x8000;

// ---- hit = one accepted candidate: Mapping (src/utils.h:44-49) ----
// misc = end_position_offset (low 16) | edit_distance << 16 | direction << 24
__device__ __forceinline__ uint64_t hit_key(uint64_t cand, uint32_t misc) {  // MappingSortKey, src/align.c:53
  const int64_t end = (int16_t)(misc & 0xFFFFu);
  return ((uint64_t)((misc >> 16) & 0xFFu) << 60) | ((uint64_t)((misc >> 24) & 1u) << 59) | (cand + (uint64_t)end);
}

struct Params {
  // mapping outcome
  const uint8_t *bases;
  const uint64_t *read_off;
  uint32_t n_reads;
  const uint8_t *ref_raw;
  uint64_t ref_bytes;
  const uint8_t *planes;
  const uint8_t *packed;     // TailInput::packed, packed_bpr, exc_bits
  uint32_t packed_bpr;
  const uint32_t *exc_bits;
  const uint64_t *seq_off;
  const uint64_t *cand;
  const uint8_t *ed;
  const int16_t *end;
  const uint32_t *cand_begin, *cand_count;
  int32_t e;
  uint32_t n_records;
  const uint32_t *rec_begin;  // n_reads + 1
  // hits in verify order (u_) and in record order (s_)
  uint64_t *u_cand;
  uint32_t *u_misc;
  uint64_t *s_cand;
  uint32_t *s_misc, *s_read;
  uint32_t *queue;  // reads with three or more hits
  uint32_t *ctl;    // [0] queue length, [1] records of the overflow pass, [2] staging too small even there, [3] length of rec_list
  uint64_t *g_keys;  // ordering scratch for reads with more hits than fit LDS (n_records each)
  uint32_t *g_idx;
  // traceback
  uint32_t lanes, text_words, pat_words, max_len;  // LDS plan of one block of the general kernel
  uint32_t fast_lanes, fast_ops;                   // ... and of the first-pass kernel (lanes, runs kept per lane)
  uint32_t *t_ops;
  uint8_t *t_md;
  uint32_t ops_cap, md_cap;
  const uint32_t *ovf_queue;  // overflow pass: the records to redo, staged at index * cap
  uint32_t *ovf_out;          // first pass: where overflowing records are queued
  uint32_t *rec_list;         // records trace_ident_kernel left to the walking kernels (ctl[3] of them); nullptr = all
  uint32_t *src_slot;         // per record: 0 = first-pass staging, kSlotDiagonal | L = `L M` (MD in the first-pass staging), else 1 + index in the overflow staging
  uint32_t *n_ops, *n_md;
  uint16_t *flag;
  uint32_t *tid, *pos0;
  uint8_t *nm;
};

// ---------------------------------------------------------------------------------------------------------
// Ordering.  One lane per read rebuilds the read's Mapping list in verify_candidates' order (+ strand first,
// candidates ascending, src/map.c:31-49); lists of one or two are ordered on the spot, longer ones are queued.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_kernel(Params p) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.n_reads) return;
  const uint32_t b = p.rec_begin[r], n = p.rec_begin[r + 1] - b;
  if (n == 0) return;
  uint32_t j = 0;
  uint64_t c0 = 0, c1 = 0;
  uint32_t m0 = 0, m1 = 0;
  for (uint32_t dir = 0; dir < 2u; ++dir) {
    const uint32_t s = 2u * r + dir, cb = p.cand_begin[s], cn = p.cand_count[s];
    for (uint32_t i = cb; i < cb + cn; ++i) {
      const uint32_t ed = p.ed[i];
      if (ed == 0xFFu) continue;
      const uint64_t c = p.cand[i];
      const uint32_t m = (uint32_t)(uint16_t)p.end[i] | (ed << 16) | (dir << 24);
      if (j == 0) c0 = c, m0 = m;
      if (j == 1) c1 = c, m1 = m;
      if (j < n) p.u_cand[b + j] = c, p.u_misc[b + j] = m;
      ++j;
    }
  }
  if (n <= 2u) {
    const bool swap = n == 2u && hit_key(c1, m1) < hit_key(c0, m0);  // stable: equal keys keep verify order
    p.s_cand[b] = swap ? c1 : c0, p.s_misc[b] = swap ? m1 : m0, p.s_read[b] = r;
    if (n == 2u) p.s_cand[b + 1] = swap ? c0 : c1, p.s_misc[b + 1] = swap ? m0 : m1, p.s_read[b + 1] = r;
  } else {
    p.queue[atomicAdd(&p.ctl[0], 1u)] = r;
  }
}

// klib's radix sort on (key, original index) pairs, run by one lane: KRADIX_SORT_INIT(mapping, Mapping,
// MappingSortKey, 8) (src/ksort.h:101-151).  <= 64 records are never sent here.  Each level is an in-place
// cycle-leader permutation into 256 buckets (not stable); buckets of <= 64 are finished by insertion sort, larger
// ones recurse on the next byte.  Equal keys exist, so the exact permutation decides which record is the primary.
__device__ void insertion_by_key(uint64_t *keys, uint32_t *idx, uint32_t beg, uint32_t end) {
  for (uint32_t i = beg + 1; i < end; ++i) {
    if (keys[i] < keys[i - 1]) {
      const uint64_t tk = keys[i];
      const uint32_t ti = idx[i];
      uint32_t j = i;
      for (; j > beg && tk < keys[j - 1]; --j) keys[j] = keys[j - 1], idx[j] = idx[j - 1];
      keys[j] = tk, idx[j] = ti;
    }
  }
}

__device__ void radix_permute_level(uint64_t *keys, uint32_t *idx, uint32_t beg, uint32_t end, int shift, uint32_t *bin_b,
                                    uint32_t *bin_e) {
  for (int k = 0; k < 256; ++k) bin_e[k] = 0;
  for (uint32_t i = beg; i < end; ++i) ++bin_e[(keys[i] >> shift) & 255u];
  uint32_t run = beg;
  for (int k = 0; k < 256; ++k) {
    const uint32_t c = bin_e[k];
    bin_b[k] = run;
    run += c;
    bin_e[k] = run;
  }
  for (int k = 0; k < 256;) {
    if (bin_b[k] == bin_e[k]) {
      ++k;
      continue;
    }
    uint32_t at = bin_b[k];
    int dst = (int)((keys[at] >> shift) & 255u);
    if (dst == k) {
      ++bin_b[k];
      continue;
    }
    uint64_t ck = keys[at];
    uint32_t ci = idx[at];
    do {  // follow the cycle until an element of bucket k comes back
      const uint32_t to = bin_b[dst]++;
      const uint64_t dk = keys[to];
      const uint32_t di = idx[to];
      keys[to] = ck, idx[to] = ci;
      ck = dk, ci = di;
      dst = (int)((ck >> shift) & 255u);
    } while (dst != k);
    keys[bin_b[k]] = ck, idx[bin_b[k]] = ci;
    ++bin_b[k];
  }
}

struct RadixFrame {
  uint32_t beg, end;
  int shift, k;  // k < 0: level not permuted yet
};

__device__ void klib_radix_sort(uint64_t *keys, uint32_t *idx, uint32_t n, uint32_t *bin_b, uint32_t *bin_e_levels,
                                RadixFrame *frames) {
  int sp = 0;
  frames[0] = RadixFrame{0u, n, 56, -1};
  while (sp >= 0) {
    RadixFrame &f = frames[sp];
    uint32_t *bin_e = bin_e_levels + sp * 256;
    if (f.k < 0) {
      radix_permute_level(keys, idx, f.beg, f.end, f.shift, bin_b, bin_e);
      f.k = 0;
      if (f.shift == 0) {
        --sp;
        continue;
      }
    }
    bool pushed = false;
    while (f.k < 256) {
      const int k = f.k++;
      const uint32_t lo = k ? bin_e[k - 1] : f.beg, hi = bin_e[k];
      if (hi - lo > 64u) {
        frames[sp + 1] = RadixFrame{lo, hi, f.shift > 8 ? f.shift - 8 : 0, -1};
        ++sp;
        pushed = true;
        break;
      }
      if (hi - lo > 1u) insertion_by_key(keys, idx, lo, hi);
    }
    if (!pushed) --sp;
  }
}

__global__ void __launch_bounds__(64) sort_kernel(Params p) {
  __shared__ uint32_t bin_b[256];
  __shared__ uint32_t bin_e[8 * 256];
  __shared__ RadixFrame frames[8];
  __shared__ uint64_t l_keys[kSortLdsHits];
  __shared__ uint32_t l_idx[kSortLdsHits];
  const uint32_t ln = threadIdx.x;
  const uint32_t n_queue = p.ctl[0];
  for (uint32_t q = blockIdx.x; q < n_queue; q += gridDim.x) {
    const uint32_t r = p.queue[q];
    const uint32_t b = p.rec_begin[r], n = p.rec_begin[r + 1] - b;
    if (n <= 64u) {  // insertion sort == any stable sort: rank = keys that must precede
      const bool have = ln < n;
      const uint64_t c = have ? p.u_cand[b + ln] : 0;
      const uint32_t m = have ? p.u_misc[b + ln] : 0;
      const uint64_t key = have ? hit_key(c, m) : ~0ull;
      uint32_t rank = 0;
      for (uint32_t u = 0; u < n; ++u) {
        const uint64_t ku = __shfl(key, (int)u);
        rank += (uint32_t)(ku < key || (ku == key && u < ln));
      }
      if (have) p.s_cand[b + rank] = c, p.s_misc[b + rank] = m, p.s_read[b + rank] = r;
    } else {
      uint64_t *keys = n <= kSortLdsHits ? l_keys : p.g_keys + b;
      uint32_t *idx = n <= kSortLdsHits ? l_idx : p.g_idx + b;
      for (uint32_t i = ln; i < n; i += 64u) keys[i] = hit_key(p.u_cand[b + i], p.u_misc[b + i]), idx[i] = i;
      __threadfence();
      __syncthreads();
      if (ln == 0) klib_radix_sort(keys, idx, n, bin_b, bin_e, frames);
      __threadfence();
      __syncthreads();
      for (uint32_t i = ln; i < n; i += 64u) {
        const uint32_t src = idx[i];
        p.s_cand[b + i] = p.u_cand[b + src], p.s_misc[b + i] = p.u_misc[b + src], p.s_read[b + i] = r;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// Traceback.  One lane per record; nothing crosses lanes.  Each lane keeps, word-interleaved in LDS (word w of
// lane l at [w * lanes + l]): the read as aligned (raw characters, or the canonical reverse complement of
// prepare_negative_sequence_at), the reference window pattern[0 .. L + 2e), and D0 / HP of every column.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t base_code(uint32_t c) {  // src/utils.h:72
  const uint32_t x = (c >> 1) & 3u;
  const uint32_t code = x ^ (x >> 1);
  const uint32_t u = c & 0xDFu;
  return ((u == 'A') | (u == 'C') | (u == 'G') | (u == 'T')) ? code : 4u;
}
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p) {
  uint32_t w;
  __builtin_memcpy(&w, p, 4);
  return w;
}
__device__ __forceinline__ uint4 load_u128_unaligned(const uint8_t *p) {
  uint4 w;
  __builtin_memcpy(&w, p, 16);
  return w;
}
// four characters -> their complements in canonical upper case, anything but ACGT -> 'N' (src/sequence_batch.h:90-98)
__device__ __forceinline__ uint32_t complement4(uint32_t chars) {
  const uint32_t t = (chars >> 1) & 0x03030303u;
  const uint32_t code = t ^ ((t >> 1) & 0x01010101u);
  const uint32_t upper = chars & 0xDFDFDFDFu;
  const uint32_t expect = __builtin_amdgcn_perm(0u, 0x54474341u /* "ACGT" */, code);
  const uint32_t z = upper ^ expect;
  const uint32_t nflag = ((((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) >> 7) & 0x01010101u;
  return __builtin_amdgcn_perm(0x4E4E4E4Eu /* "NNNN" */, 0x54474341u, (code ^ 0x03030303u) | (nflag << 2));
}

// four characters at once (SWAR): code per byte 0..3 (0 where the base is not A/C/G/T), complemented on the reverse
// strand; nflag per byte 0/1 (not A/C/G/T in either case, src/utils.h:72); odd = 0x80 in the bytes that are none of "ACGTN"
__device__ __forceinline__ void decode4(uint32_t chars, uint32_t complement, uint32_t &code, uint32_t &nflag, uint32_t &odd) {
  const uint32_t t = (chars >> 1) & 0x03030303u;
  const uint32_t c = t ^ ((t >> 1) & 0x01010101u);
  const uint32_t expect = __builtin_amdgcn_perm(0u, 0x54474341u /* "ACGT" */, c);
  const uint32_t z = (chars & 0xDFDFDFDFu) ^ expect;
  nflag = ((((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) >> 7) & 0x01010101u;
  code = (c ^ complement) & ~(nflag * 3u);
  const uint32_t ze = chars ^ expect, zn = chars ^ 0x4E4E4E4Eu;
  odd = (((ze & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | ze) & (((zn & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | zn) & 0x80808080u;
}
// The plane windows are consumed sixteen bits at a time: the 32 bits at bit `off` (< 32) of the window's head, and the
// window moving on by sixteen bits — four v_alignbit per plane and step.  (Indexing the window's words by the step made
// the compiler keep the windows in scratch memory: six scratch loads per step in the verify kernel.)
__device__ __forceinline__ uint32_t window_head(const uint4 &w, uint32_t off) { return __builtin_amdgcn_alignbit(w.y, w.x, off); }
__device__ __forceinline__ void window_advance16(uint4 &w) {
  w.x = __builtin_amdgcn_alignbit(w.y, w.x, 16u), w.y = __builtin_amdgcn_alignbit(w.z, w.y, 16u);
  w.z = __builtin_amdgcn_alignbit(w.w, w.z, 16u), w.w >>= 16;
}
constexpr int kStepsPerPlaneLoad = 6;  // 7 (bit offset) + 16 * 5 + 16 + 2 * 7 (band) bits <= 128

struct LaneView {
  const uint32_t *text, *pat;
  uint32_t *d0, *hp;
  uint32_t nl, ln;
  __device__ __forceinline__ uint32_t text_at(uint32_t i) const { return (text[(i >> 2) * nl + ln] >> (8u * (i & 3u))) & 0xFFu; }
  __device__ __forceinline__ uint32_t pat_at(uint32_t i) const { return (pat[(i >> 2) * nl + ln] >> (8u * (i & 3u))) & 0xFFu; }
};

struct Staging {  // where one record's CIGAR runs and MD characters are written
  uint32_t *ops;
  uint8_t *md;
  uint32_t ops_cap, md_cap;
  uint32_t n_ops = 0, n_md = 0;
  bool overflow = false;
  __device__ __forceinline__ void push_op(uint32_t op, uint32_t len) {
    if (n_ops < ops_cap)
      ops[n_ops] = (len << 4) | op;
    else
      overflow = true;
    ++n_ops;
  }
  __device__ __forceinline__ void push_md(uint32_t ch) {
    if (n_md < md_cap)
      md[n_md] = (uint8_t)ch;
    else
      overflow = true;
    ++n_md;
  }
  __device__ void push_number(uint32_t v) {
    uint32_t digits = 1;
    for (uint32_t t = v; t >= 10u; t /= 10u) ++digits;
    uint32_t div = 1;
    for (uint32_t i = 1; i < digits; ++i) div *= 10u;
    for (; div; div /= 10u) push_md('0' + (v / div) % 10u);
  }
};

// generate_alignment (src/align.c:279-499) + generate_MD_tag (src/align.c:501-544) for one record.
// Returns the start offset inside pattern, or -1 where the reference would have tripped one of its asserts.
// pattern = the window's first byte in HBM; [at_min, at_max] = offsets from it that stay inside the reference buffer.
__device__ int trace_record(const LaneView &v, const uint8_t *pattern, int64_t at_min, int64_t at_max, int L, int e, int ed,
                            int end, Staging &st) {
  int start = end - L + 1;
  if (start < 0) return -1;
  const int pat_len = L + 2 * e;
  bool identical = true;
  for (int i = 0; i < L; ++i) {
    if (v.text_at((uint32_t)i) != v.pat_at((uint32_t)(start + i))) {
      identical = false;
      break;
    }
  }
  if (identical) {  // src/align.c:294-300
    st.push_op(kOpM, (uint32_t)L);
    st.push_number((uint32_t)L);
    return start;
  }
  // ---- the recurrence again, D0 and HP of every column kept (src/align.c:303-338) ----
  {
    uint32_t B0 = 0, B1 = 0, B2 = 0;  // bit planes of the pattern window: Peq[c] is a three-way XNOR
    for (int j = 0; j < 2 * e; ++j) {
      const uint32_t pc = base_code(v.pat_at((uint32_t)j));
      B0 |= (pc & 1u) << j, B1 |= ((pc >> 1) & 1u) << j, B2 |= ((pc >> 2) & 1u) << j;
    }
    const int sh = 2 * e;
    const uint32_t band = (2u << sh) - 1u;
    uint32_t vp = 0, vn = 0;
    for (int i = 0; i < L; ++i) {
      const uint32_t pc = base_code(v.pat_at((uint32_t)(i + sh))), tc = base_code(v.text_at((uint32_t)i));
      B0 |= (pc & 1u) << sh, B1 |= ((pc >> 1) & 1u) << sh, B2 |= ((pc >> 2) & 1u) << sh;
      const uint32_t m0 = 0u - (tc & 1u), m1 = 0u - ((tc >> 1) & 1u), m2 = 0u - ((tc >> 2) & 1u);
      uint32_t x = (~((B0 ^ m0) | (B1 ^ m1) | (B2 ^ m2)) & band) | vn;
      const uint32_t d0 = ((vp + (x & vp)) ^ vp) | x;
      const uint32_t hn = vp & d0;
      const uint32_t hp = vn | ~(vp | d0);
      x = d0 >> 1;
      vn = x & hp;
      vp = hn | ~(x | hp);
      v.d0[(uint32_t)i * v.nl + v.ln] = d0;
      v.hp[(uint32_t)i * v.nl + v.ln] = hp;
      B0 >>= 1, B1 >>= 1, B2 >>= 1;
    }
  }
  // ---- walk back from (last read base, end) until `ed` errors are accounted for (src/align.c:340-440) ----
  enum Move { MATCH, MISMATCH, INSERT, DELETE };
  int bit = end - L + 1, t = L - 1, pe = end, n_err = 0;
  auto classify = [&]() -> Move {
    const bool d = (v.d0[(uint32_t)t * v.nl + v.ln] >> bit) & 1u;
    // pe never exceeds `end`, which lies inside the staged window
    if (d && v.pat_at((uint32_t)pe) == v.text_at((uint32_t)t)) return MATCH;
    if (!d) return MISMATCH;
    if ((v.hp[(uint32_t)t * v.nl + v.ln] >> bit) & 1u) return INSERT;
    return DELETE;
  };
  // The pseudo-run 'S' collects the errors at the read's 3' end and is finally added to the run that follows it
  // (src/align.c:398-399,413-414,466-469); here its length rides along in s_len and is added when that run starts.
  uint32_t cur_op = kOpS, cur_n = 1;
  switch (classify()) {  // the first step replaces the initial pseudo-run (src/align.c:345-368)
    case MATCH: --t, --pe, cur_op = kOpM; break;
    case MISMATCH: --t, --pe, ++n_err; break;
    case INSERT: --t, ++bit, ++n_err, ++start; break;
    case DELETE: return -1;  // assert(1 == 0)
  }
  auto extend = [&](uint32_t op) {
    if (cur_op == op) {
      ++cur_n;
    } else if (cur_op == kOpS) {
      cur_op = op, cur_n += 1;  // S(n) followed by op(1) ends up as op(1 + n)
    } else {
      st.push_op(cur_op, cur_n);
      cur_op = op, cur_n = 1;
    }
  };
  while (t >= 0 && n_err != ed) {
    if (bit < 0 || bit > 31 || pe < 0) return -1;
    switch (classify()) {
      case MATCH: --t, --pe, extend(kOpM); break;
      case MISMATCH:
        --t, --pe, ++n_err;
        if (cur_op == kOpS) ++cur_n; else extend(kOpM);
        break;
      case INSERT:
        --t, ++bit, ++n_err, ++start;
        if (cur_op == kOpS) ++cur_n; else extend(kOpI);
        break;
      case DELETE: --bit, --pe, ++n_err, --start, extend(kOpD); break;
    }
  }
  if (t >= 0) {  // everything left of the last error matches (src/align.c:445-455)
    if (cur_op == kOpM || cur_op == kOpS)
      cur_op = kOpM, cur_n += (uint32_t)(t + 1);
    else
      st.push_op(cur_op, cur_n), cur_op = kOpM, cur_n = (uint32_t)(t + 1);
  }
  if (cur_op == kOpS) return -1;  // nothing but the pseudo-run: the reference indexes past its run list
  st.push_op(cur_op, cur_n);
  if (st.overflow) return start;  // redone with room in the overflow pass
  // runs were produced right to left: reverse in place (src/align.c:470-479)
  for (uint32_t i = 0, j = st.n_ops - 1u; i < j; ++i, --j) {
    const uint32_t a = st.ops[i], b = st.ops[j];
    st.ops[i] = b, st.ops[j] = a;
  }
  // ---- generate_MD_tag over pattern + start (src/align.c:501-544) ----
  uint32_t run = 0, tp = 0;
  int64_t rp = start;  // offset from pattern[0]; the 'S' fold can push a run beyond the staged window
  auto ref_char = [&](int64_t at) -> uint32_t {
    if (at >= 0 && at < pat_len) return v.pat_at((uint32_t)at);
    return pattern[at < at_min ? at_min : at > at_max ? at_max : at];  // only walks the reference itself would reject
  };
  for (uint32_t k = 0; k < st.n_ops; ++k) {
    const uint32_t op = st.ops[k] & 0xFu, n = st.ops[k] >> 4;
    if (op == kOpM) {
      for (uint32_t i = 0; i < n; ++i, ++rp, ++tp) {
        const uint32_t rc = ref_char(rp);
        if (rc == v.text_at(tp)) {
          ++run;
        } else {
          if (run) st.push_number(run), run = 0;
          st.push_md(rc);
        }
      }
    } else if (op == kOpI) {
      tp += n;
    } else {
      if (run) st.push_number(run), run = 0;
      st.push_md('^');
      for (uint32_t i = 0; i < n; ++i, ++rp) st.push_md(ref_char(rp));
    }
  }
  if (run) st.push_number(run);
  return start;
}

// General form of the traceback: redoes the records the first pass queued (p.ovf_queue, p.ctl[1] of them).
__global__ void __launch_bounds__(64) trace_kernel(Params p) {
  extern __shared__ uint32_t lds[];
  const uint32_t nl = p.lanes, ln = threadIdx.x;
  uint32_t *text_w = lds;
  uint32_t *pat_w = text_w + p.text_words * nl;
  uint32_t *d0_w = pat_w + p.pat_words * nl;
  uint32_t *hp_w = d0_w + p.max_len * nl;
  const uint32_t n_items = p.ctl[1];
  for (uint32_t base = blockIdx.x * nl; base < n_items; base += gridDim.x * nl) {
    const uint32_t item = base + ln;
    if (ln >= nl || item >= n_items) continue;
    const uint32_t rec = p.ovf_queue[item];
    const uint32_t read = p.s_read[rec], misc = p.s_misc[rec];
    const uint64_t cand = p.s_cand[rec];
    const int end = (int16_t)(misc & 0xFFFFu), ed = (int)((misc >> 16) & 0xFFu);
    const uint32_t dir = (misc >> 24) & 1u;
    const uint64_t off = p.read_off[read];
    const int L = (int)(p.read_off[read + 1] - off);
    const uint8_t *fwd = p.bases + off;
    const uint32_t tid = (uint32_t)(cand >> 32);
    const uint64_t pat_abs = p.seq_off[tid] + (uint32_t)cand;
    const uint8_t *pattern = p.ref_raw + pat_abs;
    // ---- stage the read as aligned ----
    for (int c = 0; c < L; c += 16) {
      uint32_t w[4];
      if (dir == 0) {
        const uint4 q = load_u128_unaligned(fwd + c);  // may run past the read: those bytes are never looked at
        w[0] = q.x, w[1] = q.y, w[2] = q.z, w[3] = q.w;
      } else if (L - 16 - c >= 0) {  // text[c + i] = complement(fwd[L - 1 - c - i])
        const uint4 q = load_u128_unaligned(fwd + (L - 16 - c));
        w[0] = __builtin_bswap32(complement4(q.w)), w[1] = __builtin_bswap32(complement4(q.z));
        w[2] = __builtin_bswap32(complement4(q.y)), w[3] = __builtin_bswap32(complement4(q.x));
      } else {
        for (int k = 0; k < 4; ++k) {
          const int p0 = L - 4 - c - 4 * k;  // fwd offset of the word's last character
          uint32_t raw = 0;
          if (p0 >= 0)
            raw = load_u32_unaligned(fwd + p0);
          else if (p0 > -4)
            raw = load_u32_unaligned(fwd) << (8 * -p0);
          w[k] = __builtin_bswap32(complement4(raw));
        }
      }
      for (int k = 0; k < 4; ++k)
        if ((uint32_t)(c / 4 + k) < p.text_words) text_w[(uint32_t)(c / 4 + k) * nl + ln] = w[k];
    }
    for (int c = 0; c < L + 2 * p.e; c += 16) {  // the reference buffer has 64 bytes of slack behind its last base
      const uint4 q = load_u128_unaligned(pattern + c);
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
      for (int k = 0; k < 4; ++k)
        if ((uint32_t)(c / 4 + k) < p.pat_words) pat_w[(uint32_t)(c / 4 + k) * nl + ln] = w[k];
    }
    LaneView v{text_w, pat_w, d0_w, hp_w, nl, ln};
    Staging st;
    st.ops = p.t_ops + (size_t)item * p.ops_cap;
    st.md = p.t_md + (size_t)item * p.md_cap;
    st.ops_cap = p.ops_cap, st.md_cap = p.md_cap;
    int start = trace_record(v, pattern, -(int64_t)pat_abs, (int64_t)p.ref_bytes - 1 - (int64_t)pat_abs, L, p.e, ed, end, st);
    if (st.overflow) {
      atomicAdd(&p.ctl[2], 1u);  // cannot happen: this staging holds the longest possible walk
      start = -1;
    }
    const uint32_t rank = rec - p.rec_begin[read];
    uint16_t flag = (uint16_t)((dir ? 16u : 0u) | (rank ? 256u : 0u));  // BAM_FREVERSE, BAM_FSECONDARY (src/align.c:82-84)
    if (start < 0) flag |= kFlagBroken, start = 0, st.n_ops = 0, st.n_md = 0;
    p.flag[rec] = flag;
    p.tid[rec] = tid;
    p.pos0[rec] = (uint32_t)start + (uint32_t)cand;  // src/align.c:80
    p.nm[rec] = (uint8_t)ed;
    p.n_ops[rec] = st.n_ops, p.n_md[rec] = st.n_md;
    p.src_slot[rec] = item + 1u;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Traceback, first pass: the form nearly every record takes.  Same walk as trace_record, but the lane keeps one
// packed word per column in LDS — per diagonal of the band (0 .. 2e) two bits that say what the walk would do in
// that cell: they fold D0, HP and whether the reference character on that diagonal EQUALS the read character (the
// traceback and the MD tag compare characters, not codes, src/align.c:355,523) — and reads the sequences straight
// from HBM, sixteen columns per load.
// Character equality is derived from the code-equality mask Peq: for characters of the canonical alphabet
// "ACGTN" it is the same thing, a reference character outside it (lower case, IUPAC) equals no canonical one, and
// a read with such characters is left to the general kernel.  So is every walk that leaves the band, and every
// record whose CIGAR or MD outgrows the staging; the general kernel (trace_kernel) redoes those from scratch.
// ---------------------------------------------------------------------------------------------------------
// The packed column word (2 x (2e+1) bits) is kept in as few LDS bytes as hold it — LDS per block is what limits the
// waves per CU here, and the walk is a chain of dependent LDS reads that only more waves can hide: a low plane of
// P0 words and, where needed, a high plane of P1 words; column c of lane l sits at [c * lanes + l] of each.
// ---------------------------------------------------------------------------------------------------------
// Traceback, pass zero: records whose alignment is the end position's diagonal — no recurrence, no walk.
//  * Edit distance 0: generate_alignment first compares the read with the reference at its end position character by
//    character (src/align.c:285-300) and, with no mismatch, emits `L M`.
//  * Edit distance ed > 0 with exactly ed mismatching columns on that diagonal (every record of a read whose errors
//    are substitutions: most of a real sequencer's).  The walk of src/align.c:340-479 then never leaves the diagonal:
//    ed is the least cost of any path into (L-1, end), so no path reaches a cell (t, j) of the diagonal for less than
//    the mismatches before it (it would continue down the diagonal for less than ed), D[t][j] is that count, D0 — "the
//    diagonal step costs nothing" — is set exactly in the columns that match, and the walk tests match, then
//    mismatch, before it looks at HP.  It emits M runs only (the 'S' pseudo-run of mismatches at the read's end folds
//    into the M run behind it; L > ed keeps that run from being the only one): CIGAR `L M`, start = end - L + 1, MD
//    from the mismatching columns.  tests/test_traceback_model.py checks that claim on the matrix model.
// Both need character equality to be code equality: no reference character of the window outside "ACGTN" (plane[3]),
// no non-canonical read character on the forward strand (the reverse strand's complement table turns them into N).
// One lane per record, no LDS: read characters sixteen at a time (decoded to code-bit masks), the reference from its
// bit planes, compared on the one diagonal; MD written as the columns go by (a record that turns out to have more
// mismatches than ed is walked later and its MD written again).  Every other record is appended to rec_list for the
// walking kernels.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lsb_mask4(uint32_t bytes01) {  // bytes of 0/1 -> their four bits, byte 0 in bit 0
  return ((bytes01 & 0x01010101u) * 0x01020408u) >> 24;
}
__device__ __forceinline__ uint32_t md_number(uint8_t *md, uint32_t at, uint32_t v) {  // decimal of v < 10000 -> md[at ...]
  const uint32_t digits = v >= 1000u ? 4u : v >= 100u ? 3u : v >= 10u ? 2u : 1u;
  for (uint32_t k = digits; k-- > 0; v /= 10u) md[at + k] = (uint8_t)('0' + v % 10u);
  return at + digits;
}
// ---- the read's bases from the packed form of the batch (two bits per base, low bits first; A C G T = 0 1 2 3) ----
// bases j0 .. j0 + 15 of the read whose row starts at `row` (j0 >= 0), base j0 in bits 0-1
__device__ __forceinline__ uint32_t packed16(const uint8_t *row, int j0) {
  const uint8_t *a = row + (j0 >> 2);
  return __builtin_amdgcn_alignbit(load_u32_unaligned(a + 4), load_u32_unaligned(a), 2u * ((uint32_t)j0 & 3u));
}
__device__ __forceinline__ uint32_t reverse_fields2(uint32_t x) {  // the sixteen 2-bit fields in reverse order
  x = __builtin_bitreverse32(x);
  return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
__device__ __forceinline__ uint32_t even_bits16(uint32_t x) {  // bits 0, 2, 4 .. 30 -> bits 0 .. 15
  x &= 0x55555555u;
  x = (x | (x >> 1)) & 0x33333333u;
  x = (x | (x >> 2)) & 0x0F0F0F0Fu;
  x = (x | (x >> 4)) & 0x00FF00FFu;
  return (x | (x >> 8)) & 0xFFFFu;
}
constexpr uint32_t kSlotDiagonal = 0x80000000u;
constexpr uint32_t kIdentChunk = 1024;  // records one block classifies at a time (four per thread)
__global__ void __launch_bounds__(256) trace_ident_kernel(Params p) {
  __shared__ uint32_t walk[kIdentChunk];
  __shared__ uint32_t n_walk, walk_base;
  const int sh = 2 * p.e;
  for (uint32_t base = blockIdx.x * kIdentChunk; base < p.n_records; base += gridDim.x * kIdentChunk) {
    __syncthreads();
    if (threadIdx.x == 0) n_walk = 0;
    __syncthreads();
#pragma unroll 1
    for (uint32_t k = 0; k < kIdentChunk / 256u; ++k) {
      const uint32_t rec = base + k * 256u + threadIdx.x;
      if (rec >= p.n_records) continue;
      bool done = false;
      const uint32_t misc = p.s_misc[rec];
      const uint32_t read = p.s_read[rec];
      const uint64_t cand = p.s_cand[rec];
      const int end = (int16_t)(misc & 0xFFFFu);
      const uint32_t ed = (misc >> 16) & 0xFFu;
      const uint32_t dir = (misc >> 24) & 1u;
      const uint64_t off = p.read_off[read];
      const int L = (int)(p.read_off[read + 1] - off);
      const uint8_t *fwd = p.bases + off;
      const uint32_t tid = (uint32_t)(cand >> 32);
      const int start = end - L + 1;
      const uint32_t digits = L >= 1000 ? 4u : L >= 100 ? 3u : L >= 10 ? 2u : 1u;
      // MD: at most ed reference characters and ed + 1 numbers
      if (start >= 0 && start <= sh && L > (int)ed && L < 10000 && p.ops_cap >= 1u && (ed + 1u) * digits + ed <= p.md_cap) {
        const uint64_t ref0 = p.seq_off[tid] + (uint32_t)cand + (uint32_t)start;  // compared with text[0]
        const uint32_t complement = dir ? 0x03030303u : 0u;
        const uint32_t bit0 = (uint32_t)ref0 & 7u;
        uint8_t *md = p.t_md + (size_t)rec * p.md_cap;
        uint32_t odd_ref = 0, odd_text = 0, n_mm = 0, n_md = 0;
        int last = -1;  // the column of the last mismatch
        uint4 W0{}, W1{}, W2{}, W3{};
        // a read of A C G T only, in a batch that arrived packed: its code bits come out of the packed words with a
        // handful of shifts (sixteen characters decoded to the same masks are a quarter of this kernel's instructions)
        const bool from_packed = p.packed != nullptr && !((p.exc_bits[read >> 5] >> (read & 31u)) & 1u);
        const uint8_t *row = p.packed + (size_t)read * p.packed_bpr;
        for (int col = 0; col < L; col += 16) {
          const int sub = (col >> 4) % 7;  // 7 (bit offset) + 16 * 7 <= 128: one load per plane covers seven steps
          if (sub == 0) {
            const uint64_t at = (ref0 + (uint32_t)col) >> 3;
            W0 = load_u128_unaligned(femk::plane_addr(p.planes, 0, at)), W1 = load_u128_unaligned(femk::plane_addr(p.planes, 1, at));
            W2 = load_u128_unaligned(femk::plane_addr(p.planes, 2, at)), W3 = load_u128_unaligned(femk::plane_addr(p.planes, 3, at));
          } else {
            window_advance16(W0), window_advance16(W1), window_advance16(W2), window_advance16(W3);
          }
          const int ncol = L - col < 16 ? L - col : 16;
          uint32_t m0 = 0, m1 = 0, m2 = 0;
          if (from_packed) {
            uint32_t t16;  // text[col .. col + 16), two bits each
            if (dir == 0) {
              t16 = packed16(row, col);
            } else if (ncol == 16) {  // text[col + i] = complement(fwd[L - 1 - col - i])
              t16 = ~reverse_fields2(packed16(row, L - 16 - col));
            } else {  // the last, partial step: the read's first ncol bases
              t16 = ~(reverse_fields2(packed16(row, 0)) >> (2u * (uint32_t)(16 - ncol)));
            }
            m0 = even_bits16(t16), m1 = even_bits16(t16 >> 1);
          } else {
            const uint4 r = load_u128_unaligned(dir == 0 ? fwd + col : fwd + (L - 16 - col));
            const uint32_t w[4] = {dir ? __builtin_bswap32(r.w) : r.x, dir ? __builtin_bswap32(r.z) : r.y,
                                   dir ? __builtin_bswap32(r.y) : r.z, dir ? __builtin_bswap32(r.x) : r.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              uint32_t cw, nw, odd;
              decode4(w[q], complement, cw, nw, odd);
              const int nb = ncol - 4 * q;
              odd_text |= nb >= 4 ? odd : nb > 0 ? odd & ((1u << (8 * nb)) - 1u) : 0u;
              m0 |= lsb_mask4(cw) << (4 * q), m1 |= lsb_mask4(cw >> 1) << (4 * q), m2 |= lsb_mask4(nw) << (4 * q);
            }
          }
          const uint32_t colmask = ncol == 16 ? 0xFFFFu : ((1u << ncol) - 1u);  // ((ref0 + 112 k) & 7 == ref0 & 7)
          odd_ref |= window_head(W3, bit0) & colmask;
          const uint32_t b0 = window_head(W0, bit0), b1 = window_head(W1, bit0), b2 = window_head(W2, bit0);
          uint32_t diff = ((b0 ^ m0) | (b1 ^ m1) | (b2 ^ m2)) & colmask;
          while (diff) {  // generate_MD_tag over an M run (src/align.c:515-529): the matches counted, the reference's character
            const uint32_t i = (uint32_t)__builtin_ctz(diff);
            const int at = col + (int)i;
            diff &= diff - 1u;
            if (++n_mm > ed) break;
            if (at - last > 1) n_md = md_number(md, n_md, (uint32_t)(at - last - 1));
            // the character from its code (it is one of "ACGTN", or odd_ref sends the record to the walk): no load
            const uint32_t code = ((b0 >> i) & 1u) | (((b1 >> i) & 1u) << 1);
            md[n_md++] = (uint8_t)((b2 >> i) & 1u ? 'N' : 0x54474341u >> (8u * code));
            last = at;
          }
          if (n_mm > ed) break;
        }
        if (n_mm == ed && odd_ref == 0u && !(dir == 0 && odd_text)) {
          if (L - 1 - last > 0) n_md = md_number(md, n_md, (uint32_t)(L - 1 - last));
          const uint32_t rank = rec - p.rec_begin[read];
          p.flag[rec] = (uint16_t)((dir ? 16u : 0u) | (rank ? 256u : 0u));
          p.tid[rec] = tid;
          p.pos0[rec] = (uint32_t)start + (uint32_t)cand;
          p.nm[rec] = (uint8_t)ed;
          p.n_ops[rec] = 1u, p.n_md[rec] = n_md;
          p.src_slot[rec] = kSlotDiagonal | (uint32_t)L;  // CIGAR `L M`: compact_kernel writes it from this word
          done = true;
        }
      }
      if (!done) walk[atomicAdd(&n_walk, 1u)] = rec;
    }
    // Records for the walking kernels gather in LDS and go out with ONE atomic on the list's cursor per chunk (one per
    // wave cost 1.3 ms per 8.6 M records: same-address atomics complete at ~10 ns each).
    __syncthreads();
    if (threadIdx.x == 0) walk_base = n_walk ? atomicAdd(&p.ctl[3], n_walk) : 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_walk; i += 256u) p.rec_list[walk_base + i] = walk[i];
  }
}

struct NoPlane {};
template <typename P0, typename P1>
struct ColumnHistory {
  P0 *lo;
  P1 *hi;
  uint32_t nl, ln;
  static constexpr bool kTwo = !__is_same(P1, NoPlane);
  __device__ __forceinline__ void put(uint32_t col, uint64_t v) const {
    lo[col * nl + ln] = (P0)v;
    if constexpr (kTwo) hi[col * nl + ln] = (P1)(v >> (8 * sizeof(P0)));
  }
  __device__ __forceinline__ uint64_t get(uint32_t col) const {
    uint64_t v = lo[col * nl + ln];
    if constexpr (kTwo) v |= (uint64_t)hi[col * nl + ln] << (8 * sizeof(P0));
    return v;
  }
};

template <typename P0, typename P1>
__global__ void __launch_bounds__(64) trace_fast_kernel(Params p) {
  extern __shared__ uint32_t lds[];
  const uint32_t nl = p.fast_lanes, ln = threadIdx.x;
  using HistT = uint64_t;
  ColumnHistory<P0, P1> hist;
  hist.lo = (P0 *)lds;
  hist.hi = (P1 *)(hist.lo + (size_t)p.max_len * nl);
  hist.nl = nl, hist.ln = ln;
  // run k of this lane: ops[k * nl + ln] = len << 2 | op
  uint16_t *ops = (uint16_t *)((uint8_t *)lds + (((size_t)p.max_len * nl * (sizeof(P0) + (ColumnHistory<P0, P1>::kTwo ? sizeof(P1) : 0)) + 3u) & ~(size_t)3u));
  const int e = p.e, sh = 2 * e, W = 2 * e + 1;
  const uint32_t band = (1u << W) - 1u;
  const uint32_t n_list = p.rec_list ? p.ctl[3] : p.n_records;
  for (uint32_t base = blockIdx.x * nl; base < n_list; base += gridDim.x * nl) {
    if (ln >= nl || base + ln >= n_list) continue;
    const uint32_t rec = p.rec_list ? p.rec_list[base + ln] : base + ln;
    const uint32_t read = p.s_read[rec], misc = p.s_misc[rec];
    const uint64_t cand = p.s_cand[rec];
    const int end = (int16_t)(misc & 0xFFFFu), ed = (int)((misc >> 16) & 0xFFu);
    const uint32_t dir = (misc >> 24) & 1u;
    const uint64_t off = p.read_off[read];
    const int L = (int)(p.read_off[read + 1] - off);
    const uint8_t *fwd = p.bases + off;
    const uint32_t tid = (uint32_t)(cand >> 32);
    const uint64_t pat_abs = p.seq_off[tid] + (uint32_t)cand;
    const uint8_t *pattern = p.ref_raw + pat_abs;
    int start = end - L + 1;
    bool punt = start < 0 || start > sh;  // (never: end lies in [L-1, L-1+2e])
    uint32_t odd_ref = 0;  // some reference character of the window is none of "ACGTN"

    // ---- the recurrence (src/align.c:303-338), one packed word per column ----
    // Sixteen columns per step.  Reference side: bit planes of the base codes plus the "none of ACGTN" plane; one
    // unaligned 16-byte load per plane covers the windows pattern[col .. col + 16 + 2e) of six steps.  Read side: 16
    // characters per load, decoded four at a time; on the reverse strand the chunk comes from the read's far end,
    // byte-reversed and complemented (src/sequence_batch.h:90-98; the batch's characters have 16 bytes of padding
    // in front).  Loads are issued one step (text) / one stretch (planes) ahead of their use.
    const uint32_t complement = dir ? 0x03030303u : 0u;
    auto text_chunk = [&](int c) { return load_u128_unaligned(dir == 0 ? fwd + c : fwd + (L - 16 - c)); };
    auto plane_chunk = [&](int q, int c) { return load_u128_unaligned(femk::plane_addr(p.planes, q, (pat_abs + (uint32_t)c) >> 3)); };
    const uint32_t pat_bit = (uint32_t)pat_abs & 7u;
    const int n_steps = (L + 15) >> 4;
    uint32_t vp = 0, vn = 0, odd_text = 0, dm = 0;
    int last_bad = -1;  // the last column whose characters differ on the end position's diagonal (-1: the read is identical there)
    uint4 rw = make_uint4(0, 0, 0, 0), W0 = rw, W1 = rw, W2 = rw, W3 = rw, W0n = rw, W1n = rw, W2n = rw, W3n = rw;
    if (n_steps > 0) {
      rw = text_chunk(0);
      W0n = plane_chunk(0, 0), W1n = plane_chunk(1, 0), W2n = plane_chunk(2, 0), W3n = plane_chunk(3, 0);
    }
    auto column = [&](uint32_t col, uint32_t q, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t bw, uint32_t m0, uint32_t m1,
                      uint32_t m2) {
      const uint32_t eq = __builtin_amdgcn_ubfe(~((b0 ^ m0) | (b1 ^ m1) | (b2 ^ m2)), q, (uint32_t)W);  // Peq[text[col]]
      uint32_t x = eq | vn;
      const uint32_t d0 = ((vp + (x & vp)) ^ vp) | x;
      const uint32_t hn = vp & d0;
      const uint32_t hp = vn | ~(vp | d0);
      x = d0 >> 1;
      vn = x & hp;
      vp = hn | ~(x | hp);
      const uint32_t same = eq & ~__builtin_amdgcn_ubfe(bw, q, (uint32_t)W);  // the characters themselves are equal
      dm |= __builtin_amdgcn_ubfe(same, (uint32_t)start, 1u) << q;  // this step's columns that are equal on the end position's diagonal
      // What the walk asks of a cell is one of four things: match (D0 and equal characters), mismatch (not D0),
      // insertion (D0, unequal, HP), deletion (D0, unequal, not HP) — two bits per diagonal, as planes
      // P = match | deletion and Q = match | insertion (so: D0 = P | Q, equal characters = P & Q, HP where it matters = Q).
      hist.put(col + q, (HistT)(d0 & (same | ~hp) & band) | ((HistT)(d0 & (same | hp) & band) << W));
    };
    for (int step = 0; step < n_steps; ++step) {
      const int col = step << 4, sub = step % kStepsPerPlaneLoad;
      const uint4 r = rw;
      if (sub == 0) {
        W0 = W0n, W1 = W1n, W2 = W2n, W3 = W3n;
        if (step + kStepsPerPlaneLoad < n_steps) {
          const int nc = col + 16 * kStepsPerPlaneLoad;
          W0n = plane_chunk(0, nc), W1n = plane_chunk(1, nc), W2n = plane_chunk(2, nc), W3n = plane_chunk(3, nc);
        }
      } else {
        window_advance16(W0), window_advance16(W1), window_advance16(W2), window_advance16(W3);
      }
      if (step + 1 < n_steps) rw = text_chunk(col + 16);
      // ((pat_abs + 96 k) & 7 == pat_abs & 7)
      const uint32_t b0 = window_head(W0, pat_bit), b1 = window_head(W1, pat_bit), b2 = window_head(W2, pat_bit), bw = window_head(W3, pat_bit);
      odd_ref |= bw;
      const uint32_t w[4] = {dir ? __builtin_bswap32(r.w) : r.x, dir ? __builtin_bswap32(r.z) : r.y,
                             dir ? __builtin_bswap32(r.y) : r.z, dir ? __builtin_bswap32(r.x) : r.w};
      const int ncol = L - col < 16 ? L - col : 16;
      uint32_t cw[4], nw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t odd;
        decode4(w[k], complement, cw[k], nw[k], odd);
        const int nb = ncol - 4 * k;  // characters of this word that belong to the read
        odd_text |= nb >= 4 ? odd : nb > 0 ? odd & ((1u << (8 * nb)) - 1u) : 0u;
      }
      if (ncol == 16) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
          column((uint32_t)col, (uint32_t)q, b0, b1, b2, bw, (uint32_t)__builtin_amdgcn_sbfe((int)cw[q >> 2], 8 * (q & 3), 1),
                 (uint32_t)__builtin_amdgcn_sbfe((int)cw[q >> 2], 8 * (q & 3) + 1, 1),
                 (uint32_t)__builtin_amdgcn_sbfe((int)nw[q >> 2], 8 * (q & 3), 1));
      } else {  // up to fifteen trailing columns
        const uint64_t clo = ((uint64_t)cw[1] << 32) | cw[0], chi = ((uint64_t)cw[3] << 32) | cw[2];
        const uint64_t nlo = ((uint64_t)nw[1] << 32) | nw[0], nhi = ((uint64_t)nw[3] << 32) | nw[2];
        for (int q = 0; q < ncol; ++q) {
          const uint32_t cb = (uint32_t)((q < 8 ? clo : chi) >> (8 * (q & 7)));
          const uint32_t nb = (uint32_t)((q < 8 ? nlo : nhi) >> (8 * (q & 7)));
          column((uint32_t)col, (uint32_t)q, b0, b1, b2, bw, 0u - (cb & 1u), 0u - ((cb >> 1) & 1u), 0u - (nb & 1u));
        }
      }
      const uint32_t bad = ~dm & (ncol == 16 ? 0xFFFFu : (1u << ncol) - 1u);
      if (bad) last_bad = col + 31 - (int)__builtin_clz(bad);
      dm = 0;
    }
    if (dir == 0 && odd_text) punt = true;  // a read character outside "ACGTN": character equality is not code equality

    uint32_t n_ops = 0, n_md = 0;
    uint8_t *md = p.t_md + (size_t)rec * p.md_cap;
    auto push_op = [&](uint32_t op, uint32_t len) {
      if (n_ops < p.ops_cap && n_ops < p.fast_ops && len < (1u << 14)) ops[n_ops * nl + ln] = (uint16_t)((len << 2) | op); else punt = true;
      ++n_ops;
    };
    auto push_md = [&](uint32_t ch) {
      if (n_md < p.md_cap) md[n_md] = (uint8_t)ch; else punt = true;
      ++n_md;
    };
    auto push_number = [&](uint32_t v) {  // divisions by constants only (multiply + shift)
      if (v >= 10000u) {
        uint32_t div = 10000u;
        while (v / div >= 10u) div *= 10u;
        for (; div >= 10000u; div /= 10u) push_md('0' + (v / div) % 10u);
        v %= 10000u;
        push_md('0' + v / 1000u), push_md('0' + (v / 100u) % 10u), push_md('0' + (v / 10u) % 10u), push_md('0' + v % 10u);
        return;
      }
      if (v >= 1000u) push_md('0' + v / 1000u);
      if (v >= 100u) push_md('0' + (v / 100u) % 10u);
      if (v >= 10u) push_md('0' + (v / 10u) % 10u);
      push_md('0' + v % 10u);
    };
    bool broken = false;
    uint32_t lead = 0;  // leading read bases the walk never visited: exact matches when no odd reference character is near
    if (!punt && last_bad < 0) {  // src/align.c:294-300
      push_op(kOpM, (uint32_t)L);
      push_number((uint32_t)L);
    } else if (!punt) {
      // ---- walk back (src/align.c:340-440); pe == t + bit throughout ----
      int bit = start, t = L - 1, n_err = 0;
      uint32_t cur_op = kOpS, cur_n = 1;
      // The read positions of the walk's mismatch steps, the leftmost in the low byte: what the MD tag needs to know
      // about an M run besides its length (every other base of it is a match).  Where that does not say it all — see
      // use_hist below — the MD loop reads the history instead.
      uint64_t mm = 0;
      uint32_t n_mm = 0;
      bool use_hist = L > 255;  // (a position is kept in eight bits)
      // Behind the last column that differs on the end position's diagonal the walk only matches (equal characters
      // set D0, and match is the first thing it tests): it starts at that column with the M run already counted.
      const uint32_t trail = (uint32_t)(L - 1 - last_bad);
      if (trail) {
        t = last_bad, cur_op = kOpM, cur_n = trail;
      } else {  // the first step replaces the initial pseudo-run (src/align.c:345-368)
        const HistT h = hist.get((uint32_t)t) >> bit;
        const bool pp = (uint32_t)h & 1u, qq = (uint32_t)(h >> W) & 1u;
        const bool d = pp || qq, same = pp && qq, horiz = qq;
        if (d && same) --t, cur_op = kOpM;
        else if (!d) mm = (uint64_t)(uint32_t)t, n_mm = 1, --t, ++n_err;
        else if (horiz) --t, ++bit, ++n_err, ++start, use_hist = true;  // a read-end insertion folds into the run that follows
        else broken = true;  // assert(1 == 0)
      }
      while (!broken && !punt && t >= 0 && n_err != ed) {
        if (bit < 0) {  // the reference's own guard; above the band it would read bits this pass does not keep
          broken = true;
          break;
        }
        if (bit > sh) {
          punt = true;
          break;
        }
        const HistT h = hist.get((uint32_t)t) >> bit;
        const bool pp = (uint32_t)h & 1u, qq = (uint32_t)(h >> W) & 1u;
        const bool d = pp || qq, same = pp && qq, horiz = qq;
        const bool is_match = d && same, is_ins = d && !same && horiz, is_del = d && !same && !horiz;  // else: mismatch
        const uint32_t op = is_del ? kOpD : is_ins ? kOpI : kOpM;
        const bool absorbed = cur_op == kOpS && !is_match && !is_del;  // read-end errors pile up in the pseudo-run
        // the pseudo-run takes an insertion, or ends in a deletion: the run it folds into is not what the walk stepped through
        use_hist |= cur_op == kOpS && (is_ins || is_del);
        mm = d ? mm : (mm << 8) | (uint64_t)(uint32_t)t;
        n_mm += (uint32_t)!d;
        t -= (int)!is_del;
        bit += (int)is_ins - (int)is_del;
        start += (int)is_ins - (int)is_del;
        n_err += (int)!is_match;
        if (absorbed || op == cur_op) {
          ++cur_n;
        } else if (cur_op == kOpS) {
          cur_op = op, cur_n += 1;  // S(n) followed by op(1) ends up as op(1 + n)
        } else {
          push_op(cur_op, cur_n);
          cur_op = op, cur_n = 1;
        }
      }
      if (!broken && !punt) {
        if (t >= 0 && n_err == ed && odd_ref == 0u) lead = (uint32_t)(t + 1);
        if (t >= 0) {
          if (cur_op == kOpM || cur_op == kOpS)
            cur_op = kOpM, cur_n += (uint32_t)(t + 1);
          else
            push_op(cur_op, cur_n), cur_op = kOpM, cur_n = (uint32_t)(t + 1);
        }
        if (cur_op == kOpS) broken = true; else push_op(cur_op, cur_n);
      }
      // leading bases the walk never visited match on codes; on characters too unless an odd reference character is near
      use_hist |= (t >= 0 && odd_ref != 0u) || n_mm > 8u;
      if (!broken && !punt && !use_hist) {
        // ---- MD (src/align.c:501-544) from the runs (produced right to left) and the mismatch positions: every base of
        //      an M run the walk stepped through as a match is a match of characters, as are the bases in front of the
        //      walk's last step (all ed edits found) and behind its first (`trail`) ----
        uint32_t run = 0, tp = 0;
        int rp = start;
        for (uint32_t k = n_ops; k-- > 0 && !punt;) {
          const uint32_t o = ops[k * nl + ln], op = o & 3u, n = o >> 2;
          if (op == kOpM) {
            const uint32_t end = tp + n;
            while (n_mm && ((uint32_t)mm & 0xFFu) < end) {
              const uint32_t at = (uint32_t)mm & 0xFFu;
              mm >>= 8, --n_mm;
              run += at - tp;
              if (run) push_number(run), run = 0;
              rp += (int)(at - tp);
              push_md(pattern[rp]);
              ++rp, tp = at + 1u;
            }
            run += end - tp, rp += (int)(end - tp), tp = end;
          } else if (op == kOpI) {
            tp += n;
          } else {
            if (rp < 0) {
              punt = true;
              break;
            }
            if (run) push_number(run), run = 0;
            push_md('^');
            for (uint32_t i = 0; i < n; ++i, ++rp) push_md(pattern[rp]);
          }
        }
        if (run) push_number(run);
      } else if (!broken && !punt) {
        // ---- MD over pattern + start, character equality read off the history ----
        uint32_t run = 0, tp = 0;
        int rp = start;
        for (uint32_t k = n_ops; k-- > 0 && !punt;) {
          const uint32_t o = ops[k * nl + ln], op = o & 3u, n = o >> 2;
          if (op == kOpM) {
            const int diag = rp - (int)tp;  // constant along the run
            if (diag < 0 || diag > sh) {
              punt = true;
              break;
            }
            uint32_t i = 0;
            if (tp == 0 && lead) {
              // The walk stopped with all ed edits found: the alignment's cost is ed, so these bases match on codes;
              // with only canonical characters around, on characters too (the history is not read for them).
              i = lead < n ? lead : n;
              run += i, rp += (int)i, tp += i;
            }
            // the rightmost run ends in `trail` matches the walk never visited either (k == 0: it is that run)
            const uint32_t n_walked = k == 0 && trail ? (n > trail ? n - trail : 0u) : n;
            for (; i < n_walked; ++i, ++rp, ++tp) {
              const HistT hd = hist.get(tp) >> diag;
              if ((uint32_t)hd & (uint32_t)(hd >> W) & 1u) {  // equal characters = P & Q
                ++run;
              } else {
                if (run) push_number(run), run = 0;
                push_md(pattern[rp]);
              }
            }
            if (i < n) run += n - i, rp += (int)(n - i), tp += n - i;
          } else if (op == kOpI) {
            tp += n;
          } else {
            if (rp < 0) {
              punt = true;
              break;
            }
            if (run) push_number(run), run = 0;
            push_md('^');
            for (uint32_t i = 0; i < n; ++i, ++rp) push_md(pattern[rp]);
          }
        }
        if (run) push_number(run);
      }
    }
    if (punt) {
      p.ovf_out[atomicAdd(&p.ctl[1], 1u)] = rec;
      p.n_ops[rec] = 0, p.n_md[rec] = 0;
      continue;
    }
    const uint32_t rank = rec - p.rec_begin[read];
    uint16_t flag = (uint16_t)((dir ? 16u : 0u) | (rank ? 256u : 0u));
    if (broken) flag |= kFlagBroken, start = 0, n_ops = 0, n_md = 0;
    uint32_t *out_ops = p.t_ops + (size_t)rec * p.ops_cap;
    for (uint32_t k = 0; k < n_ops; ++k) {
      const uint32_t o = ops[(n_ops - 1u - k) * nl + ln];
      out_ops[k] = ((o >> 2) << 4) | (o & 3u);
    }
    p.flag[rec] = flag;
    p.tid[rec] = tid;
    p.pos0[rec] = (uint32_t)start + (uint32_t)cand;
    p.nm[rec] = (uint8_t)ed;
    p.n_ops[rec] = n_ops, p.n_md[rec] = n_md;
    p.src_slot[rec] = 0u;
  }
}

struct CompactParams {
  uint32_t n_records;
  const uint32_t *src_slot, *n_ops, *n_md, *cigar_off, *md_off;
  const uint32_t *t_ops, *o_ops;  // first-pass and overflow staging
  const uint8_t *t_md, *o_md;
  uint32_t ops_cap, md_cap, o_ops_cap, o_md_cap;
  uint32_t *cigar;
  uint8_t *md;
};

__global__ void __launch_bounds__(256) compact_kernel(CompactParams p) {
  const uint32_t rec = blockIdx.x * blockDim.x + threadIdx.x;
  if (rec >= p.n_records) return;
  uint32_t slot = p.src_slot[rec];
  const uint32_t no = p.n_ops[rec], nm = p.n_md[rec], co = p.cigar_off[rec], mo = p.md_off[rec];
  if (slot & kSlotDiagonal) {  // trace_ident_kernel's records: one M run over the whole read
    p.cigar[co] = (slot & ~kSlotDiagonal) << 4;
    slot = 0;
  } else {
    const uint32_t *ops = slot ? p.o_ops + (size_t)(slot - 1u) * p.o_ops_cap : p.t_ops + (size_t)rec * p.ops_cap;
    for (uint32_t i = 0; i < no; ++i) p.cigar[co + i] = ops[i];
  }
  const uint8_t *md = slot ? p.o_md + (size_t)(slot - 1u) * p.o_md_cap : p.t_md + (size_t)rec * p.md_cap;
  for (uint32_t i = 0; i < nm; ++i) p.md[mo + i] = md[i];
}

struct PairPlus {  // component-wise sum of (run count, MD length) pairs
  __host__ __device__ rocprim::tuple<uint32_t, uint32_t> operator()(const rocprim::tuple<uint32_t, uint32_t> &a,
                                                                   const rocprim::tuple<uint32_t, uint32_t> &b) const {
    return rocprim::make_tuple(rocprim::get<0>(a) + rocprim::get<0>(b), rocprim::get<1>(a) + rocprim::get<1>(b));
  }
};

// ---- host-side buffer helpers ----
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t need(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr, cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 256);
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr, cap = 0;
  }
  template <typename T>
  T *as() const { return (T *)p; }
};
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t need(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr, cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 256);
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr, cap = 0;
  }
  template <typename T>
  T *as() const { return (T *)p; }
};


// ---------------------------------------------------------------------------------------------------------
// SAM text.  One line per record, the fields generate_bam1_t packs (src/align.c:546-632) as htslib's sam_format1 prints
// them: QNAME FLAG RNAME POS 255 CIGAR * 0 0 SEQ QUAL NM:i MD:Z.  SEQ is the read as it came (src/align.c:79) through
// the 4-bit round trip of the BAM record (seq_nt16_table / seq_nt16_str: IUPAC letters upper-cased, anything else N);
// only a read's first record carries SEQ and QUAL (src/align.c:83-88).  Same bytes as fem_records_sam (fem_host.cc).
// ---------------------------------------------------------------------------------------------------------
__device__ const uint8_t kSamSeqLut[256] = {
    78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78,
    78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 65, 67, 71, 84, 78, 78, 78, 78, 78, 78, 78, 78, 78, 61, 78, 78,
    78, 65, 66, 67, 68, 78, 78, 71, 72, 78, 78, 75, 78, 77, 78, 78, 78, 78, 82, 83, 84, 78, 86, 87, 78, 89, 78, 78, 78, 78, 78, 78,
    78, 65, 66, 67, 68, 78, 78, 71, 72, 78, 78, 75, 78, 77, 78, 78, 78, 78, 82, 83, 84, 78, 86, 87, 78, 89, 78, 78, 78, 78, 78, 78,
    78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78,
    78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78,
    78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78,
    78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78, 78,
};

struct SamParams {
  uint32_t n_records;
  const uint32_t *rec_begin, *s_read;
  const uint16_t *flag;
  const uint32_t *tid, *pos0;
  const uint8_t *nm;
  const uint32_t *cigar_off, *cigar, *md_off;
  const uint8_t *md;
  const uint8_t *bases;
  const uint64_t *read_off;
  const uint8_t *quals, *names;
  const uint64_t *name_off;
  const uint8_t *ref_names;
  const uint32_t *ref_name_off;
  unsigned long long *line_len;        // n_records + 1 (the last one zero)
  const unsigned long long *line_off;  // exclusive scan of line_len
  uint8_t *text;
  uint32_t *asserted;
  unsigned long long *qual_at;  // qual_hole: n_reads entries, set for the reads that have a record
  uint32_t qual_hole;           // the QUAL field of a primary record is sized but not written (quals == nullptr)
};

__device__ __forceinline__ uint32_t dec_digits(uint32_t v) {
  return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u
       : v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
__device__ __forceinline__ uint8_t *put_dec(uint8_t *w, uint32_t v) {
  const uint32_t n = dec_digits(v);
  for (uint32_t i = n; i-- > 0;) {
    w[i] = (uint8_t)('0' + v % 10u);
    v /= 10u;
  }
  return w + n;
}

__global__ void __launch_bounds__(256) sam_len_kernel(SamParams p) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j <= p.n_records; j += stride) {
    if (j == p.n_records) {
      p.line_len[j] = 0;
      continue;
    }
    const uint32_t r = p.s_read[j];
    const bool primary = p.rec_begin[r] == j;
    const uint32_t L = (uint32_t)(p.read_off[r + 1] - p.read_off[r]);
    const uint32_t name_len = (uint32_t)(p.name_off[r + 1] - p.name_off[r]);
    const uint32_t t = p.tid[j], rname_len = p.ref_name_off[t + 1] - p.ref_name_off[t];
    const uint16_t flag = p.flag[j];
    if (flag & 0x8000u) atomicAdd(p.asserted, 1u);
    uint32_t cig = 0;
    const uint32_t c0 = p.cigar_off[j], c1 = p.cigar_off[j + 1];
    for (uint32_t c = c0; c < c1; ++c) cig += dec_digits(p.cigar[c] >> 4) + 1u;
    if (c1 == c0) cig = 1;  // '*'
    const uint32_t md_len = p.md_off[j + 1] - p.md_off[j];
    const uint32_t seq_qual = primary && L > 0 ? L + 1u + (p.quals || p.qual_hole ? L : 1u) : 3u;
    p.line_len[j] = (unsigned long long)name_len + 1u + dec_digits(flag & 0x7FFFu) + 1u + rname_len + 1u + dec_digits(p.pos0[j] + 1u) + 5u + cig +
                    7u + seq_qual + 6u + dec_digits(p.nm[j]) + 6u + md_len + 1u;
  }
}

// One wave per 64 consecutive records.  First every lane takes a record of its own: its numbers (two levels of small loads —
// 64 records' worth in flight where one record per wave had one, which made the kernel wait for memory 1.4 ms per million
// records), where each field of its line starts, and the short fields (numbers, separators, tags), written by the lane itself.
// Then the wave goes through its records two at a time, the long fields (QNAME, RNAME, SEQ, QUAL, MD) a byte per lane, every
// load of a pair requested before the first store; what a lane knows of record i reaches the others by v_readlane.
__global__ void __launch_bounds__(256) sam_write_kernel(SamParams p) {
  __shared__ uint8_t lut[256];
  lut[threadIdx.x] = kSamSeqLut[threadIdx.x];
  __syncthreads();
  const uint32_t ln = threadIdx.x & 63u;
  const uint32_t j0 = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64u;
  if (j0 >= p.n_records) return;
  const uint32_t n_here = p.n_records - j0 < 64u ? p.n_records - j0 : 64u;
  const bool mine = ln < n_here;
  const uint32_t j = j0 + (mine ? ln : n_here - 1u);  // (lanes behind the last record repeat its loads and write nothing)
  // level 1
  const uint32_t r = p.s_read[j];
  const uint32_t t = p.tid[j];
  const uint32_t flag = p.flag[j] & 0x7FFFu, pos1 = p.pos0[j] + 1u, nm = p.nm[j];
  const uint32_t c0 = p.cigar_off[j], c1 = p.cigar_off[j + 1], m0 = p.md_off[j], md_len = p.md_off[j + 1] - m0;
  const unsigned long long at = p.line_off[j];
  // level 2
  const bool primary = p.rec_begin[r] == j;
  const uint64_t ro = p.read_off[r];
  const uint32_t L = (uint32_t)(p.read_off[r + 1] - ro);
  const uint64_t no = p.name_off[r];
  const uint32_t name_len = (uint32_t)(p.name_off[r + 1] - no);
  const uint32_t rn0 = p.ref_name_off[t], rname_len = p.ref_name_off[t + 1] - rn0;
  uint32_t op_a = 0, op_b = 0, op_c = 0;  // the first CIGAR operations (most records have one to three)
  const uint32_t n_ops = c1 - c0;
  if (n_ops > 0) op_a = p.cigar[c0];
  if (n_ops > 1) op_b = p.cigar[c0 + 1];
  if (n_ops > 2) op_c = p.cigar[c0 + 2];
  const bool seq = primary && L > 0;
  uint32_t cig = 0;
  if (n_ops <= 3u) {
    if (n_ops > 0) cig += dec_digits(op_a >> 4) + 1u;
    if (n_ops > 1) cig += dec_digits(op_b >> 4) + 1u;
    if (n_ops > 2) cig += dec_digits(op_c >> 4) + 1u;
  } else {
    for (uint32_t c = c0; c < c1; ++c) cig += dec_digits(p.cigar[c] >> 4) + 1u;
  }
  if (n_ops == 0) cig = 1;
  // where the line's fields start (offsets from its first byte)
  const uint32_t o_flag = name_len + 1u;
  const uint32_t o_rname = o_flag + dec_digits(flag) + 1u;
  const uint32_t o_pos = o_rname + rname_len + 1u;
  const uint32_t o_cig = o_pos + dec_digits(pos1) + 5u;
  const uint32_t o_seq = o_cig + cig + 7u;
  const uint32_t o_nm = o_seq + (seq ? L + 1u + (p.quals || p.qual_hole ? L : 1u) : 3u) + 6u;
  const uint32_t o_md = o_nm + dec_digits(nm) + 6u;
  if (mine) {  // the short fields of the lane's own record
    uint8_t *w = p.text + at;
    w[name_len] = '\t';
    put_dec(w + o_flag, flag)[0] = '\t';
    w[o_rname + rname_len] = '\t';
    uint8_t *q = put_dec(w + o_pos, pos1);
    q[0] = '\t', q[1] = '2', q[2] = '5', q[3] = '5', q[4] = '\t';
    q = w + o_cig;
    if (n_ops == 0) *q++ = '*';
    if (n_ops <= 3u) {
      if (n_ops > 0) q = put_dec(q, op_a >> 4), *q++ = (uint8_t)"MIDNSHP=XB"[op_a & 0xFu];
      if (n_ops > 1) q = put_dec(q, op_b >> 4), *q++ = (uint8_t)"MIDNSHP=XB"[op_b & 0xFu];
      if (n_ops > 2) q = put_dec(q, op_c >> 4), *q++ = (uint8_t)"MIDNSHP=XB"[op_c & 0xFu];
    } else {
      for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t op = p.cigar[c];
        q = put_dec(q, op >> 4);
        *q++ = (uint8_t)"MIDNSHP=XB"[op & 0xFu];
      }
    }
    q[0] = '\t', q[1] = '*', q[2] = '\t', q[3] = '0', q[4] = '\t', q[5] = '0', q[6] = '\t';
    uint8_t *w_seq = w + o_seq;
    if (seq) {
      w_seq[L] = '\t';
      if (p.qual_hole) p.qual_at[r] = at + o_seq + L + 1u;  // (the caller has the qualities: it writes them here)
      else if (!p.quals) w_seq[L + 1u] = '*';
    } else {
      w_seq[0] = '*', w_seq[1] = '\t', w_seq[2] = '*';
    }
    q = w + o_nm - 6u;
    q[0] = '\t', q[1] = 'N', q[2] = 'M', q[3] = ':', q[4] = 'i', q[5] = ':';
    q = put_dec(w + o_nm, nm);
    q[0] = '\t', q[1] = 'M', q[2] = 'D', q[3] = ':', q[4] = 'Z', q[5] = ':';
    w[o_md + md_len] = '\n';
  }
  // ---- the long fields, record by record, all lanes ----
  const uint32_t at_lo = (uint32_t)at, at_hi = (uint32_t)(at >> 32), no_lo = (uint32_t)no, no_hi = (uint32_t)(no >> 32);
  const uint32_t ro_lo = (uint32_t)ro, ro_hi = (uint32_t)(ro >> 32);
  const uint32_t L_seq = seq ? L : 0u;  // (no SEQ / QUAL bytes on a read's further records)
  struct Rec {
    uint8_t *w;
    const uint8_t *name, *rname, *md, *bases, *quals;
    uint32_t name_len, rname_len, md_len, L, o_rname, o_md, o_seq;
  };
  auto rl = [](uint32_t v, uint32_t i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)i); };
  auto record = [&](uint32_t i) -> Rec {
    Rec x;
    x.w = p.text + ((uint64_t)rl(at_lo, i) | (uint64_t)rl(at_hi, i) << 32);
    x.name = p.names + ((uint64_t)rl(no_lo, i) | (uint64_t)rl(no_hi, i) << 32);
    const uint64_t ro_i = (uint64_t)rl(ro_lo, i) | (uint64_t)rl(ro_hi, i) << 32;
    x.bases = p.bases + ro_i, x.quals = p.quals ? p.quals + ro_i : nullptr;
    x.rname = p.ref_names + rl(rn0, i), x.md = p.md + rl(m0, i);
    x.name_len = rl(name_len, i), x.rname_len = rl(rname_len, i), x.md_len = rl(md_len, i), x.L = rl(L_seq, i);
    x.o_rname = rl(o_rname, i), x.o_md = rl(o_md, i), x.o_seq = rl(o_seq, i);
    return x;
  };
  struct Bytes {
    uint8_t n, rn, m, sA, sB, qA, qB;
  };
  const uint32_t k1 = ln + 64u;
  auto load = [&](const Rec &x) -> Bytes {
    Bytes b{};
    if (ln < x.name_len) b.n = x.name[ln];
    if (ln < x.rname_len) b.rn = x.rname[ln];
    if (ln < x.md_len) b.m = x.md[ln];
    if (ln < x.L) b.sA = x.bases[ln];
    if (k1 < x.L) b.sB = x.bases[k1];
    if (x.quals) {
      if (ln < x.L) b.qA = x.quals[ln];
      if (k1 < x.L) b.qB = x.quals[k1];
    }
    return b;
  };
  auto store = [&](const Rec &x, const Bytes &b) {
    uint8_t *w_seq = x.w + x.o_seq, *w_rname = x.w + x.o_rname, *w_md = x.w + x.o_md;
    if (ln < x.name_len) x.w[ln] = b.n;
    if (ln < x.rname_len) w_rname[ln] = b.rn;
    if (ln < x.md_len) w_md[ln] = b.m;
    if (ln < x.L) w_seq[ln] = lut[b.sA];
    if (k1 < x.L) w_seq[k1] = lut[b.sB];
    if (x.quals) {
      if (ln < x.L) w_seq[x.L + 1u + ln] = b.qA;
      if (k1 < x.L) w_seq[x.L + 1u + k1] = b.qB;
    }
    // fields beyond what a lane holds (wave-uniform conditions): names and reference names over 64 characters, MD strings
    // over 64, reads over 128
    for (uint32_t k = k1; k < x.name_len; k += 64u) x.w[k] = x.name[k];
    for (uint32_t k = k1; k < x.rname_len; k += 64u) w_rname[k] = x.rname[k];
    for (uint32_t k = k1; k < x.md_len; k += 64u) w_md[k] = x.md[k];
    for (uint32_t k = ln + 128u; k < x.L; k += 64u) {
      w_seq[k] = lut[x.bases[k]];
      if (x.quals) w_seq[x.L + 1u + k] = x.quals[k];
    }
  };
  uint32_t i = 0;
  for (; i + 1u < n_here; i += 2u) {
    const Rec xa = record(i), xb = record(i + 1u);
    const Bytes ba = load(xa), bb = load(xb);
    store(xa, ba);
    store(xb, bb);
  }
  if (i < n_here) {
    const Rec xa = record(i);
    store(xa, load(xa));
  }
}

}  // namespace

struct Tail::Impl {
  DevBuf rec_begin, queue, ctl, u_cand, u_misc, s_cand, s_misc, s_read, t_ops, t_md, o_ops, o_md, ovf, rec_list, src_slot, n_ops, n_md,
      flag, tid, pos0, nm, cigar_off, md_off, cigar, md, scan_tmp;
  DevBuf line_len, line_off, text, qual_at;
  PinBuf h_ctl, h_rec_begin, h_flag, h_tid, h_pos0, h_nm, h_cigar_off, h_md_off, h_cigar, h_md, h_text, h_qual_at;
  uint32_t last_n = 0, last_nr = 0;  // what the last run() left on the device
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_text = nullptr;  // the SAM text has arrived in h_text
  ~Impl() {
    if (ev_text) (void)hipEventDestroy(ev_text);
    for (DevBuf *b : {&rec_begin, &queue, &ctl, &u_cand, &u_misc, &s_cand, &s_misc, &s_read, &t_ops, &t_md, &o_ops, &o_md, &ovf, &rec_list,
                      &src_slot, &n_ops, &n_md, &flag, &tid, &pos0, &nm, &cigar_off, &md_off, &cigar, &md, &scan_tmp, &line_len, &line_off, &text, &qual_at})
      b->release();
    for (PinBuf *b : {&h_ctl, &h_rec_begin, &h_flag, &h_tid, &h_pos0, &h_nm, &h_cigar_off, &h_md_off, &h_cigar, &h_md, &h_text, &h_qual_at})
      b->release();
    for (hipEvent_t e : ev)
      if (e) (void)hipEventDestroy(e);
  }
};

Tail::~Tail() { delete impl_; }

#define TAIL_TRY(expr)                                                             \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e_);           \
      return e_ == hipErrorOutOfMemory ? FEM_ERR_NOMEM : FEM_ERR_HIP;              \
    }                                                                              \
  } while (0)

// Everything run() and sam() allocate for a batch of n reads with nr records (device arrays sized by the records, the scans'
// scratch).  run() calls it with the batch's own numbers; a caller that knows what is coming (fem_dev_reserve_batch) calls it
// during its setup: thirty allocations per slot otherwise fall into the first batches' way home.
int Tail::reserve(uint32_t n, uint32_t nr, uint32_t max_len_in, int e, bool tiny, std::string *err) {
  if (!impl_) impl_ = new (std::nothrow) Impl();
  if (!impl_) return FEM_ERR_NOMEM;
  Impl &m = *impl_;
  const uint32_t fast_ops = std::min<uint32_t>(kOpsCap, 2u * (uint32_t)e + 2u);
  const uint32_t ops_cap = tiny ? 1u : std::max<uint32_t>(8u, fast_ops), md_cap = tiny ? 2u : kMdCap;
  (void)max_len_in;
  TAIL_TRY(m.rec_begin.need(((size_t)n + 1) * 4));
  TAIL_TRY(m.queue.need(std::max<size_t>(n, 1) * 4));
  TAIL_TRY(m.ctl.need(16));
  TAIL_TRY(m.h_ctl.need(32));
  const size_t r1 = (size_t)nr + 1;
  TAIL_TRY(m.u_cand.need(r1 * 8));
  TAIL_TRY(m.u_misc.need(r1 * 4));
  TAIL_TRY(m.s_cand.need(r1 * 8));
  TAIL_TRY(m.s_misc.need(r1 * 4));
  TAIL_TRY(m.s_read.need(r1 * 4));
  TAIL_TRY(m.t_ops.need(r1 * ops_cap * 4));
  TAIL_TRY(m.t_md.need(r1 * std::max<uint32_t>(md_cap, 12)));  // doubles as the ordering scratch (8 + 4 bytes per hit)
  TAIL_TRY(m.ovf.need(r1 * 4));
  TAIL_TRY(m.rec_list.need(r1 * 4));
  TAIL_TRY(m.src_slot.need(r1 * 4));
  TAIL_TRY(m.n_ops.need(r1 * 4));
  TAIL_TRY(m.n_md.need(r1 * 4));
  TAIL_TRY(m.flag.need(r1 * 2));
  TAIL_TRY(m.tid.need(r1 * 4));
  TAIL_TRY(m.pos0.need(r1 * 4));
  TAIL_TRY(m.nm.need(r1));
  TAIL_TRY(m.cigar_off.need(r1 * 4));
  TAIL_TRY(m.md_off.need(r1 * 4));
  TAIL_TRY(m.cigar.need(std::max<size_t>((size_t)nr * ops_cap, 1) * 4));
  TAIL_TRY(m.md.need(std::max<size_t>((size_t)nr * md_cap, 1)));
  TAIL_TRY(m.line_len.need(r1 * 8));
  TAIL_TRY(m.line_off.need(r1 * 8));
  TAIL_TRY(m.qual_at.need(((size_t)n + 1) * 8));
  TAIL_TRY(m.h_qual_at.need(((size_t)n + 1) * 8));
  size_t tmp_a = 0, tmp_b = 0, tmp_c = 0;
  TAIL_TRY(rocprim::exclusive_scan(nullptr, tmp_a, (const uint32_t *)nullptr, m.rec_begin.as<uint32_t>(), 0u, (size_t)n,
                                   rocprim::plus<uint32_t>(), (hipStream_t) nullptr));
  {
    auto lens = rocprim::make_zip_iterator(rocprim::make_tuple(m.n_ops.as<uint32_t>(), m.n_md.as<uint32_t>()));
    auto offs = rocprim::make_zip_iterator(rocprim::make_tuple(m.cigar_off.as<uint32_t>(), m.md_off.as<uint32_t>()));
    TAIL_TRY(rocprim::exclusive_scan(nullptr, tmp_b, lens, offs, rocprim::make_tuple(0u, 0u), r1, PairPlus(), (hipStream_t) nullptr));
  }
  TAIL_TRY(rocprim::exclusive_scan(nullptr, tmp_c, m.line_len.as<unsigned long long>(), m.line_off.as<unsigned long long>(), 0ull, r1,
                                   rocprim::plus<unsigned long long>(), (hipStream_t) nullptr));
  TAIL_TRY(m.scan_tmp.need(std::max<size_t>(std::max(tmp_a, std::max(tmp_b, tmp_c)), 16)));
  for (hipEvent_t &ev : m.ev)
    if (!ev) TAIL_TRY(hipEventCreate(&ev));
  if (!m.ev_text) TAIL_TRY(hipEventCreateWithFlags(&m.ev_text, hipEventDisableTiming));
  return FEM_OK;
}

// The first launch of a kernel loads its code object (this file's: 12-17 ms of host time in front of the first batch's
// ordering kernels, 7 more in front of its text) and the first use of a stream creates its queue: both belong to the setup.
// Asks for every kernel's attributes (which loads the code object) and runs the three scans over one element on `stream`.
int Tail::warm(hipStream_t stream, std::string *err) {
  if (!impl_) return FEM_ERR_STATE;
  Impl &m = *impl_;
  hipFuncAttributes a;
  const void *kernels[] = {(const void *)gather_kernel, (const void *)sort_kernel, (const void *)trace_ident_kernel,
                           (const void *)trace_fast_kernel<uint8_t, NoPlane>, (const void *)trace_fast_kernel<uint16_t, NoPlane>,
                           (const void *)trace_fast_kernel<uint16_t, uint8_t>, (const void *)trace_fast_kernel<uint32_t, NoPlane>,
                           (const void *)trace_fast_kernel<uint32_t, uint8_t>, (const void *)trace_fast_kernel<uint32_t, uint16_t>,
                           (const void *)trace_kernel, (const void *)compact_kernel, (const void *)sam_len_kernel,
                           (const void *)sam_write_kernel};
  for (const void *k : kernels) TAIL_TRY(hipFuncGetAttributes(&a, k));
  if (!m.scan_tmp.p || !m.rec_begin.p || !m.n_ops.p || !m.line_len.p) return FEM_OK;  // (nothing reserved: the scans load with the first batch)
  size_t tmp = m.scan_tmp.cap;
  TAIL_TRY(hipMemsetAsync(m.n_ops.p, 0, 4, stream));
  TAIL_TRY(hipMemsetAsync(m.n_md.p, 0, 4, stream));
  TAIL_TRY(hipMemsetAsync(m.line_len.p, 0, 8, stream));
  TAIL_TRY(rocprim::exclusive_scan(m.scan_tmp.p, tmp, (const uint32_t *)m.n_ops.as<uint32_t>(), m.rec_begin.as<uint32_t>(), 0u, (size_t)1,
                                   rocprim::plus<uint32_t>(), stream));
  {
    auto lens = rocprim::make_zip_iterator(rocprim::make_tuple(m.n_ops.as<uint32_t>(), m.n_md.as<uint32_t>()));
    auto offs = rocprim::make_zip_iterator(rocprim::make_tuple(m.cigar_off.as<uint32_t>(), m.md_off.as<uint32_t>()));
    tmp = m.scan_tmp.cap;
    TAIL_TRY(rocprim::exclusive_scan(m.scan_tmp.p, tmp, lens, offs, rocprim::make_tuple(0u, 0u), (size_t)1, PairPlus(), stream));
  }
  tmp = m.scan_tmp.cap;
  TAIL_TRY(rocprim::exclusive_scan(m.scan_tmp.p, tmp, m.line_len.as<unsigned long long>(), m.line_off.as<unsigned long long>(), 0ull, (size_t)1,
                                   rocprim::plus<unsigned long long>(), stream));
  if (m.text.p && m.h_text.p && m.text.cap >= (1u << 20) && m.h_text.cap >= (1u << 20))
    TAIL_TRY(hipMemcpyAsync(m.h_text.p, m.text.p, 1u << 20, hipMemcpyDeviceToHost, stream));
  TAIL_TRY(hipMemcpyAsync(m.h_ctl.p, m.ctl.p, 16, hipMemcpyDeviceToHost, stream));
  TAIL_TRY(hipStreamSynchronize(stream));
  return FEM_OK;
}

int Tail::run(const TailInput &in, hipStream_t stream, int n_cu, bool tiny, TailOutput *out, std::string *err, double *ms,
              bool copy_records) {
  if (!impl_) impl_ = new (std::nothrow) Impl();
  if (!impl_) return FEM_ERR_NOMEM;
  Impl &m = *impl_;
  if (in.n_records > 0xFFFFFFF0ull) {
    if (err) *err = "more than 2^32 mappings in one batch; split the batch";
    return FEM_ERR_UNSUPPORTED;
  }
  const uint32_t n = in.n_reads, nr = (uint32_t)in.n_records;
  for (hipEvent_t &e : m.ev)
    if (!e) TAIL_TRY(hipEventCreate(&e));

  // ---- LDS plan of the traceback kernel: lanes = records one 64-thread block walks at a time ----
  const uint32_t max_len = std::max<uint32_t>(in.max_len, 1);
  const uint32_t text_words = (max_len + 3) / 4 + 4, pat_words = (max_len + 2 * (uint32_t)in.e + 3) / 4 + 4;
  const uint32_t words_per_lane = text_words + pat_words + 2 * max_len;
  const uint32_t lanes = std::min<uint32_t>(64, (64u * 1024u / 4u) / words_per_lane);
  if (lanes == 0) {
    if (err) *err = "read too long for the device traceback";
    return FEM_ERR_UNSUPPORTED;
  }
  const uint32_t lds_bytes = lanes * words_per_lane * 4u;
  // first-pass kernel: one packed word per column (two fields of 2e+1 bits) + the run list
  const uint32_t hist_bytes = (2u * (2u * (uint32_t)in.e + 1u) + 7u) / 8u;  // 1, 1, 2, 2, 3, 3, 4, 4 for e = 0..7
  const uint32_t fast_ops = std::min<uint32_t>(kOpsCap, 2u * (uint32_t)in.e + 2u);  // a sane walk opens <= 2 ed + 1 runs
  const uint32_t fast_per_lane = max_len * hist_bytes + fast_ops * 2u;
  const uint32_t fast_lanes = std::min<uint32_t>(64, (64u * 1024u - 4u) / fast_per_lane);
  const uint32_t fast_lds = ((max_len * hist_bytes * fast_lanes + 3u) & ~3u) + fast_ops * 2u * fast_lanes;
  // (the staging's rows are read and written one record per lane: the shorter the row, the fewer lines a wave touches)
  const uint32_t ops_cap = tiny ? 1u : std::max<uint32_t>(8u, fast_ops), md_cap = tiny ? 2u : kMdCap;
  // longest possible walk: every step opens a run; the MD of a run never exceeds two characters per column
  const uint32_t o_ops_cap = 2 * max_len + 2 * (uint32_t)in.e + 8, o_md_cap = 8 * max_len + 128;

  static const bool trace_host = getenv("FEM_TESTING") && getenv("FEM_FETCH_TIMES");
  const auto t_in = std::chrono::steady_clock::now();
  auto since_in = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count(); };
  {
    const int rrc = reserve(n, nr, max_len, in.e, tiny, err);
    if (rrc) return rrc;
  }
  const double ms_reserved = since_in();
  const size_t r1 = (size_t)nr + 1;
  size_t tmp_bytes = m.scan_tmp.cap;

  Params p{};
  p.bases = in.bases, p.read_off = in.read_off, p.n_reads = n;
  p.ref_raw = in.ref_raw, p.ref_bytes = in.ref_bytes, p.seq_off = in.seq_off;
  p.planes = in.planes;
  p.packed = in.packed, p.packed_bpr = in.packed_bpr, p.exc_bits = in.exc_bits;
  p.cand = in.cand, p.ed = in.ed, p.end = in.end, p.cand_begin = in.cand_begin, p.cand_count = in.cand_count;
  p.e = in.e, p.n_records = nr;
  p.rec_begin = m.rec_begin.as<uint32_t>();
  p.u_cand = m.u_cand.as<uint64_t>(), p.u_misc = m.u_misc.as<uint32_t>();
  p.s_cand = m.s_cand.as<uint64_t>(), p.s_misc = m.s_misc.as<uint32_t>(), p.s_read = m.s_read.as<uint32_t>();
  p.queue = m.queue.as<uint32_t>(), p.ctl = m.ctl.as<uint32_t>();
  p.g_keys = m.t_md.as<uint64_t>();
  p.g_idx = (uint32_t *)(m.t_md.as<uint8_t>() + r1 * 8);
  p.lanes = lanes, p.text_words = text_words, p.pat_words = pat_words, p.max_len = max_len, p.fast_lanes = fast_lanes, p.fast_ops = fast_ops;
  p.t_ops = m.t_ops.as<uint32_t>(), p.t_md = m.t_md.as<uint8_t>(), p.ops_cap = ops_cap, p.md_cap = md_cap;
  p.ovf_queue = nullptr, p.ovf_out = m.ovf.as<uint32_t>(), p.src_slot = m.src_slot.as<uint32_t>();
  p.rec_list = m.rec_list.as<uint32_t>();
  p.n_ops = m.n_ops.as<uint32_t>(), p.n_md = m.n_md.as<uint32_t>();
  p.flag = m.flag.as<uint16_t>(), p.tid = m.tid.as<uint32_t>(), p.pos0 = m.pos0.as<uint32_t>(), p.nm = m.nm.as<uint8_t>();

  uint32_t *h_ctl = m.h_ctl.as<uint32_t>();
  TAIL_TRY(hipEventRecord(m.ev[0], stream));
  TAIL_TRY(hipMemsetAsync(m.ctl.p, 0, 16, stream));
  if (n) {
    TAIL_TRY(rocprim::exclusive_scan(m.scan_tmp.p, tmp_bytes, (const uint32_t *)in.n_map, m.rec_begin.as<uint32_t>(), 0u,
                                     (size_t)n, rocprim::plus<uint32_t>(), stream));
  }
  TAIL_TRY(hipMemsetD32Async((hipDeviceptr_t)(m.rec_begin.as<uint32_t>() + n), (int)nr, 1, stream));
  uint32_t n_overflow = 0;
  if (nr) {
    hipLaunchKernelGGL(gather_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, p);
    TAIL_TRY(hipGetLastError());
    hipLaunchKernelGGL(sort_kernel, dim3((uint32_t)n_cu * 4u), dim3(64), 0, stream, p);
    TAIL_TRY(hipGetLastError());
    TAIL_TRY(hipEventRecord(m.ev[1], stream));
    // zero-edit records first (no recurrence); what they leave goes through the walking kernel
    hipLaunchKernelGGL(trace_ident_kernel, dim3(std::min<uint32_t>((nr + kIdentChunk - 1u) / kIdentChunk, (uint32_t)n_cu * 8u)), dim3(256), 0, stream, p);
    TAIL_TRY(hipGetLastError());
    const uint32_t blocks = std::min<uint32_t>((nr + fast_lanes - 1) / fast_lanes, (uint32_t)n_cu * 32u);
    switch (hist_bytes) {
      case 1: hipLaunchKernelGGL((trace_fast_kernel<uint8_t, NoPlane>), dim3(blocks), dim3(64), fast_lds, stream, p); break;
      case 2: hipLaunchKernelGGL((trace_fast_kernel<uint16_t, NoPlane>), dim3(blocks), dim3(64), fast_lds, stream, p); break;
      case 3: hipLaunchKernelGGL((trace_fast_kernel<uint16_t, uint8_t>), dim3(blocks), dim3(64), fast_lds, stream, p); break;
      case 4: hipLaunchKernelGGL((trace_fast_kernel<uint32_t, NoPlane>), dim3(blocks), dim3(64), fast_lds, stream, p); break;
      case 5: hipLaunchKernelGGL((trace_fast_kernel<uint32_t, uint8_t>), dim3(blocks), dim3(64), fast_lds, stream, p); break;
      default: hipLaunchKernelGGL((trace_fast_kernel<uint32_t, uint16_t>), dim3(blocks), dim3(64), fast_lds, stream, p); break;
    }
    TAIL_TRY(hipGetLastError());
    TAIL_TRY(hipMemcpyAsync(h_ctl, m.ctl.p, 16, hipMemcpyDeviceToHost, stream));
    const double ms_queued = since_in();
    TAIL_TRY(hipStreamSynchronize(stream));
    if (trace_host) fprintf(stderr, "[tail] allocations %.2f ms, kernels queued %.2f, walked %.2f\n", ms_reserved, ms_queued, since_in());
    n_overflow = h_ctl[1];
    if (getenv("FEM_TESTING") && getenv("FEM_TAIL_DEBUG"))
      fprintf(stderr, "[tail] records %u, queued for ordering %u, walked %u, overflow pass %u, lanes %u/%u\n", nr, h_ctl[0], h_ctl[3], n_overflow, fast_lanes, lanes);
    if (n_overflow) {  // records whose CIGAR or MD outgrew the first staging: once more, with room for any walk
      TAIL_TRY(m.o_ops.need((size_t)n_overflow * o_ops_cap * 4));
      TAIL_TRY(m.o_md.need((size_t)n_overflow * o_md_cap));
      Params q = p;
      q.ovf_queue = m.ovf.as<uint32_t>();
      q.t_ops = m.o_ops.as<uint32_t>(), q.t_md = m.o_md.as<uint8_t>(), q.ops_cap = o_ops_cap, q.md_cap = o_md_cap;
      const uint32_t b2 = std::min<uint32_t>((n_overflow + lanes - 1) / lanes, (uint32_t)n_cu * 16u);
      hipLaunchKernelGGL(trace_kernel, dim3(b2), dim3(64), lds_bytes, stream, q);
      TAIL_TRY(hipGetLastError());
    }
  } else {
    TAIL_TRY(hipEventRecord(m.ev[1], stream));
  }
  TAIL_TRY(hipEventRecord(m.ev[2], stream));
  // ---- compaction: offsets by exclusive scans over n_records + 1 lengths (the last one zero) ----
  TAIL_TRY(hipMemsetAsync(m.n_ops.as<uint32_t>() + nr, 0, 4, stream));
  TAIL_TRY(hipMemsetAsync(m.n_md.as<uint32_t>() + nr, 0, 4, stream));
  {  // both offsets in one pass: a scan over (runs, MD characters) pairs
    auto lens = rocprim::make_zip_iterator(rocprim::make_tuple(m.n_ops.as<uint32_t>(), m.n_md.as<uint32_t>()));
    auto offs = rocprim::make_zip_iterator(rocprim::make_tuple(m.cigar_off.as<uint32_t>(), m.md_off.as<uint32_t>()));
    TAIL_TRY(rocprim::exclusive_scan(m.scan_tmp.p, tmp_bytes, lens, offs, rocprim::make_tuple(0u, 0u), r1, PairPlus(), stream));
  }
  // (the compacted arrays are sized by what the stagings can hold, so that no round trip to the host sits between the scan
  // and the kernel that uses it; the totals come back with everything else)
  TAIL_TRY(m.cigar.need(std::max<size_t>((size_t)nr * ops_cap + (size_t)n_overflow * o_ops_cap, 1) * 4));
  TAIL_TRY(m.md.need(std::max<size_t>((size_t)nr * md_cap + (size_t)n_overflow * o_md_cap, 1)));
  if (nr) {
    CompactParams c{};
    c.n_records = nr, c.src_slot = p.src_slot, c.n_ops = p.n_ops, c.n_md = p.n_md;
    c.cigar_off = m.cigar_off.as<uint32_t>(), c.md_off = m.md_off.as<uint32_t>();
    c.t_ops = p.t_ops, c.t_md = p.t_md, c.o_ops = m.o_ops.as<uint32_t>(), c.o_md = m.o_md.as<uint8_t>();
    c.ops_cap = ops_cap, c.md_cap = md_cap, c.o_ops_cap = o_ops_cap, c.o_md_cap = o_md_cap;
    c.cigar = m.cigar.as<uint32_t>(), c.md = m.md.as<uint8_t>();
    hipLaunchKernelGGL(compact_kernel, dim3((nr + 255u) / 256u), dim3(256), 0, stream, c);
    TAIL_TRY(hipGetLastError());
  }
  TAIL_TRY(hipEventRecord(m.ev[3], stream));
  TAIL_TRY(hipMemcpyAsync(h_ctl + 4, m.cigar_off.as<uint32_t>() + nr, 4, hipMemcpyDeviceToHost, stream));
  TAIL_TRY(hipMemcpyAsync(h_ctl + 5, m.md_off.as<uint32_t>() + nr, 4, hipMemcpyDeviceToHost, stream));
  m.last_n = n, m.last_nr = nr;
  if (!copy_records) {  // the caller renders them on the device (sam())
    TAIL_TRY(hipMemcpyAsync(h_ctl, m.ctl.p, 16, hipMemcpyDeviceToHost, stream));
    TAIL_TRY(hipStreamSynchronize(stream));
    if (h_ctl[2] != 0) {
      if (err) *err = "device traceback: a record outgrew the overflow staging (internal error)";
      return FEM_ERR_HIP;
    }
    if (ms) {
      for (int i = 0; i < 3; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, m.ev[i], m.ev[i + 1]) == hipSuccess) ms[i] += t;
      }
    }
    memset(out, 0, sizeof *out);
    out->n_reads = n, out->n_records = nr;
    return FEM_OK;
  }
  // ---- copy back ----
  TAIL_TRY(hipStreamSynchronize(stream));
  const uint32_t n_cigar = h_ctl[4], n_md = h_ctl[5];
  TAIL_TRY(m.h_rec_begin.need(((size_t)n + 1) * 4));
  TAIL_TRY(m.h_flag.need(r1 * 2));
  TAIL_TRY(m.h_tid.need(r1 * 4));
  TAIL_TRY(m.h_pos0.need(r1 * 4));
  TAIL_TRY(m.h_nm.need(r1));
  TAIL_TRY(m.h_cigar_off.need(r1 * 4));
  TAIL_TRY(m.h_md_off.need(r1 * 4));
  TAIL_TRY(m.h_cigar.need(std::max<size_t>(n_cigar, 1) * 4));
  TAIL_TRY(m.h_md.need(std::max<size_t>(n_md, 1)));
  TAIL_TRY(hipMemcpyAsync(m.h_rec_begin.p, m.rec_begin.p, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, stream));
  TAIL_TRY(hipMemcpyAsync(m.h_cigar_off.p, m.cigar_off.p, r1 * 4, hipMemcpyDeviceToHost, stream));
  TAIL_TRY(hipMemcpyAsync(m.h_md_off.p, m.md_off.p, r1 * 4, hipMemcpyDeviceToHost, stream));
  if (nr) {
    TAIL_TRY(hipMemcpyAsync(m.h_flag.p, m.flag.p, (size_t)nr * 2, hipMemcpyDeviceToHost, stream));
    TAIL_TRY(hipMemcpyAsync(m.h_tid.p, m.tid.p, (size_t)nr * 4, hipMemcpyDeviceToHost, stream));
    TAIL_TRY(hipMemcpyAsync(m.h_pos0.p, m.pos0.p, (size_t)nr * 4, hipMemcpyDeviceToHost, stream));
    TAIL_TRY(hipMemcpyAsync(m.h_nm.p, m.nm.p, (size_t)nr, hipMemcpyDeviceToHost, stream));
  }
  if (n_cigar) TAIL_TRY(hipMemcpyAsync(m.h_cigar.p, m.cigar.p, (size_t)n_cigar * 4, hipMemcpyDeviceToHost, stream));
  if (n_md) TAIL_TRY(hipMemcpyAsync(m.h_md.p, m.md.p, (size_t)n_md, hipMemcpyDeviceToHost, stream));
  TAIL_TRY(hipMemcpyAsync(h_ctl, m.ctl.p, 16, hipMemcpyDeviceToHost, stream));
  TAIL_TRY(hipStreamSynchronize(stream));
  if (h_ctl[2] != 0) {
    if (err) *err = "device traceback: a record outgrew the overflow staging (internal error)";
    return FEM_ERR_HIP;
  }
  if (ms) {
    for (int i = 0; i < 3; ++i) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, m.ev[i], m.ev[i + 1]) == hipSuccess) ms[i] += t;
    }
  }
  out->n_reads = n, out->n_records = nr;
  out->rec_begin = m.h_rec_begin.as<uint32_t>();
  out->flag = m.h_flag.as<uint16_t>(), out->tid = m.h_tid.as<uint32_t>(), out->pos0 = m.h_pos0.as<uint32_t>();
  out->nm = m.h_nm.as<uint8_t>();
  out->cigar_off = m.h_cigar_off.as<uint32_t>(), out->cigar = m.h_cigar.as<uint32_t>();
  out->md_off = m.h_md_off.as<uint32_t>(), out->md = m.h_md.as<char>();
  return FEM_OK;
}


int Tail::reserve_text(uint64_t bytes, std::string *err) {
  if (!impl_) impl_ = new (std::nothrow) Impl();
  if (!impl_) return FEM_ERR_NOMEM;
  TAIL_TRY(impl_->text.need((size_t)bytes));
  TAIL_TRY(impl_->h_text.need((size_t)bytes));
  return FEM_OK;
}

int Tail::wait_text() {
  if (!impl_ || !impl_->ev_text) return FEM_ERR_STATE;
  return hipEventSynchronize(impl_->ev_text) == hipSuccess ? FEM_OK : FEM_ERR_HIP;
}

int Tail::sam(const TailInput &in, const SamInput &names, hipStream_t stream, int n_cu, SamOutput *out, std::string *err, double *ms,
              bool wait, TextGate *gate) {
  if (!impl_ || !out) return FEM_ERR_STATE;
  Impl &m = *impl_;
  const uint32_t nr = m.last_nr;
  const size_t r1 = (size_t)nr + 1;
  for (hipEvent_t &e : m.ev)
    if (!e) TAIL_TRY(hipEventCreate(&e));
  TAIL_TRY(m.line_len.need(r1 * 8));
  TAIL_TRY(m.line_off.need(r1 * 8));
  TAIL_TRY(m.h_ctl.need(32));
  size_t tmp = 0;
  TAIL_TRY(rocprim::exclusive_scan(nullptr, tmp, m.line_len.as<unsigned long long>(), m.line_off.as<unsigned long long>(), 0ull, r1,
                                   rocprim::plus<unsigned long long>(), stream));
  TAIL_TRY(m.scan_tmp.need(std::max<size_t>(tmp, 16)));
  SamParams p{};
  p.n_records = nr, p.rec_begin = m.rec_begin.as<uint32_t>(), p.s_read = m.s_read.as<uint32_t>();
  p.flag = m.flag.as<uint16_t>(), p.tid = m.tid.as<uint32_t>(), p.pos0 = m.pos0.as<uint32_t>(), p.nm = m.nm.as<uint8_t>();
  p.cigar_off = m.cigar_off.as<uint32_t>(), p.cigar = m.cigar.as<uint32_t>(), p.md_off = m.md_off.as<uint32_t>(), p.md = m.md.as<uint8_t>();
  p.bases = in.bases, p.read_off = in.read_off;
  p.quals = names.quals, p.names = names.names, p.name_off = names.name_off, p.ref_names = names.ref_names, p.ref_name_off = names.ref_name_off;
  const bool hole = names.qual_hole && !names.quals;
  const size_t n_reads1 = (size_t)m.last_n + 1;
  if (hole) {  // where each read's QUAL field starts (all ones: the read has no record)
    TAIL_TRY(m.qual_at.need(n_reads1 * 8));
    TAIL_TRY(m.h_qual_at.need(n_reads1 * 8));
    TAIL_TRY(hipMemsetAsync(m.qual_at.p, 0xFF, n_reads1 * 8, stream));
    p.qual_at = m.qual_at.as<unsigned long long>(), p.qual_hole = 1u;
  }
  p.line_len = m.line_len.as<unsigned long long>(), p.line_off = m.line_off.as<unsigned long long>();
  p.asserted = m.ctl.as<uint32_t>() + 2;  // (ctl[2] is zero after a successful run())
  unsigned long long *h_total = (unsigned long long *)(m.h_ctl.as<uint32_t>() + 6);
  TAIL_TRY(hipEventRecord(m.ev[0], stream));
  hipLaunchKernelGGL(sam_len_kernel, dim3(std::max<uint32_t>(1u, std::min<uint32_t>((nr + 256u) / 256u, (uint32_t)n_cu * 16u))), dim3(256), 0, stream, p);
  TAIL_TRY(hipGetLastError());
  TAIL_TRY(rocprim::exclusive_scan(m.scan_tmp.p, m.scan_tmp.cap, m.line_len.as<unsigned long long>(), m.line_off.as<unsigned long long>(), 0ull,
                                   r1, rocprim::plus<unsigned long long>(), stream));
  TAIL_TRY(hipMemcpyAsync(h_total, m.line_off.as<unsigned long long>() + nr, 8, hipMemcpyDeviceToHost, stream));
  TAIL_TRY(hipMemcpyAsync(m.h_ctl.as<uint32_t>() + 2, m.ctl.as<uint32_t>() + 2, 4, hipMemcpyDeviceToHost, stream));
  TAIL_TRY(hipStreamSynchronize(stream));
  const uint64_t total = *h_total;
  TAIL_TRY(m.text.need(std::max<size_t>((size_t)total, 16)));
  TAIL_TRY(m.h_text.need(std::max<size_t>((size_t)total + total / 8, 1u << 20)));
  if (nr) {
    p.text = m.text.as<uint8_t>();
    const uint32_t blocks = (nr + 255u) / 256u;  // a wave per 64 records
    hipLaunchKernelGGL(sam_write_kernel, dim3(blocks), dim3(256), 0, stream, p);
    TAIL_TRY(hipGetLastError());
  }
  TAIL_TRY(hipEventRecord(m.ev[1], stream));
  if (!m.ev_text) TAIL_TRY(hipEventCreateWithFlags(&m.ev_text, hipEventDisableTiming));
  static const bool no_gate = getenv("FEM_TESTING") && getenv("FEM_TEXT_NO_GATE");  // (A/B)
  {
    std::unique_lock<std::mutex> turn;
    if (gate && !no_gate) {
      turn = std::unique_lock<std::mutex>(gate->mu);
      if (gate->last && gate->last != m.ev_text) TAIL_TRY(hipEventSynchronize(gate->last));  // (this slot's own last text is home: its stream is in order)
    }
    // (by the copy engine.  The shader cores' stores into the pinned buffer — no engine to queue in — bring a text home in 7-9.5
    //  ms where the engine takes 5.4, and FEM map from 130 to 117 Mreads/s.)
    if (total) TAIL_TRY(hipMemcpyAsync(m.h_text.p, m.text.p, (size_t)total, hipMemcpyDeviceToHost, stream));
    if (hole) TAIL_TRY(hipMemcpyAsync(m.h_qual_at.p, m.qual_at.p, n_reads1 * 8, hipMemcpyDeviceToHost, stream));
    TAIL_TRY(hipEventRecord(m.ev_text, stream));
    if (gate) gate->last = m.ev_text;
  }
  if (wait) TAIL_TRY(hipStreamSynchronize(stream));
  if (ms && wait) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, m.ev[0], m.ev[1]) == hipSuccess) *ms += t;
  }
  out->text = m.h_text.as<char>(), out->len = total, out->n_asserted = m.h_ctl.as<uint32_t>()[2];
  out->qual_at = hole ? m.h_qual_at.as<uint64_t>() : nullptr;
  return FEM_OK;
}

}  // namespace femt
