/*
 * fem_oracle.c — CPU restatement of FEM's per-read mapping hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see fem_oracle.h).
 *
 * Every function names the reference lines it restates.  Where the reference
 * has undefined behaviour the choice made here is marked "UB in reference".
 */
#include "fem_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* small growable arrays (the reference uses klib kvec; only push/clear/swap  */
/* semantics matter)                                                          */
/* ------------------------------------------------------------------------- */
#define VEC_DECL(name, type)                                                   \
  typedef struct {                                                             \
    type *a;                                                                   \
    size_t n, m;                                                               \
  } name;                                                                      \
  static inline void name##_push(name *v, type x) {                            \
    if (v->n == v->m) {                                                        \
      v->m = v->m ? v->m * 2 : 16;                                             \
      v->a = (type *)realloc(v->a, v->m * sizeof(type));                       \
    }                                                                          \
    v->a[v->n++] = x;                                                          \
  }                                                                            \
  static inline void name##_append(name *v, const type *x, size_t k) {         \
    for (size_t i_ = 0; i_ < k; ++i_) name##_push(v, x[i_]);                   \
  }                                                                            \
  static inline void name##_swap(name *x, name *y) {                           \
    name t = *x;                                                               \
    *x = *y;                                                                   \
    *y = t;                                                                    \
  }

VEC_DECL(v64, uint64_t)
VEC_DECL(v32, uint32_t)
VEC_DECL(v16, uint16_t)
VEC_DECL(vi16, int16_t)
VEC_DECL(v8, uint8_t)
VEC_DECL(vch, char)

/* ------------------------------------------------------------------------- */
/* encodings (src/utils.h:72-81)                                              */
/* ------------------------------------------------------------------------- */
static inline uint8_t base_code(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}
static const char code_char[8] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N'};

/* hash_seed_in_sequence (src/utils.h:83-99) */
uint32_t fo_hash_seed(uint64_t pos, int k, const char *seq, uint64_t len) {
  uint32_t mask = (((uint32_t)1) << (2 * k)) - 1;
  uint32_t h = 0;
  for (int i = 0; i < k; ++i) {
    if (pos + (uint64_t)i < len) {
      uint8_t b = base_code(seq[pos + i]);
      if (b < 4)
        h = ((h << 2) | b) & mask;
      else
        h = (h << 2) & mask; /* N -> A */
    } else {
      h = (h << 2) & mask; /* past the end: pad A */
    }
  }
  return h;
}

/* hash_all_seeds_in_sequence (src/utils.h:101-117), start 0.  The ambiguous
 * count looks only at the base entering the window, i.e. read offsets >= k. */
static void hash_all_seeds(int n_seeds, int k, const char *seq, uint64_t len, int *n_ambiguous, uint32_t *out) {
  uint32_t mask = (((uint32_t)1) << (2 * k)) - 1;
  uint32_t h = fo_hash_seed(0, k, seq, len);
  *n_ambiguous = 0;
  out[0] = h;
  for (int i = 1; i < n_seeds; ++i) {
    uint8_t b = base_code(seq[i + k - 1]);
    if (b < 4) {
      h = ((h << 2) | b) & mask;
    } else {
      h = (h << 2) & mask;
      ++*n_ambiguous;
    }
    out[i] = h;
  }
}

/* prepare_negative_sequence_at (src/sequence_batch.h:90-98) */
void fo_revcomp(const char *seq, uint32_t len, char *out) {
  for (uint32_t i = 0; i < len; ++i) out[i] = code_char[((uint8_t)3) ^ base_code(seq[len - i - 1])];
}

/* ------------------------------------------------------------------------- */
/* index (src/index.c:57-98, 100-168)                                         */
/* ------------------------------------------------------------------------- */
uint64_t fo_index_count(const fo_ref *ref, int k, int step) {
  uint64_t n = 0;
  for (uint32_t s = 0; s < ref->n_seq; ++s)
    for (uint32_t pos = 0; (uint64_t)pos + k - 1 < ref->len[s]; pos += step) ++n; /* src/index.c:65 */
  return n;
}

/* construct_index: the reference sorts (hash, location) pairs by hash with an
 * unstable radix sort and then re-sorts every bucket by location
 * (src/index.c:74,89-94), so the result is "each bucket ascending by
 * location".  A counting sort over positions visited in ascending location
 * order gives exactly that. */
int fo_index_build(const fo_ref *ref, int k, int step, uint32_t *lookup, uint64_t *occ) {
  size_t n_lookup = ((size_t)1 << (2 * k)) + 1;
  memset(lookup, 0, n_lookup * sizeof(uint32_t));
  for (uint32_t s = 0; s < ref->n_seq; ++s) {
    const char *seq = ref->text + ref->off[s];
    for (uint32_t pos = 0; (uint64_t)pos + k - 1 < ref->len[s]; pos += step)
      lookup[fo_hash_seed(pos, k, seq, ref->len[s]) + 1]++;
  }
  uint32_t run = 0; /* src/index.c:88-92: exclusive prefix sum, uint32 */
  for (size_t i = 1; i < n_lookup; ++i) {
    run += lookup[i];
    lookup[i] = run;
  }
  uint32_t *cursor = (uint32_t *)malloc((n_lookup - 1) * sizeof(uint32_t));
  if (!cursor) return -1;
  memcpy(cursor, lookup, (n_lookup - 1) * sizeof(uint32_t));
  for (uint32_t s = 0; s < ref->n_seq; ++s) {
    const char *seq = ref->text + ref->off[s];
    for (uint32_t pos = 0; (uint64_t)pos + k - 1 < ref->len[s]; pos += step) {
      uint32_t h = fo_hash_seed(pos, k, seq, ref->len[s]);
      occ[cursor[h]++] = ((uint64_t)s << 32) | pos; /* src/index.c:67 */
    }
  }
  free(cursor);
  return 0;
}

/* The same arrays built by n_threads threads (checker speed only: a 3 Gbp reference takes the loop above minutes).
 * Thread t owns the buckets [4^k t / T, 4^k (t+1) / T): every thread walks the whole reference in ascending
 * location order, hashes every indexed position and keeps the ones in its own range, so each bucket is filled in
 * the same order as by fo_index_build (tests/test_oracle_models.py compares the two). */
typedef struct {
  const fo_ref *ref;
  int k, step, t, n_threads, phase;
  uint32_t *lookup, *cursor;
  uint64_t *occ;
} fo_index_job;

static void *fo_index_worker(void *arg) {
  fo_index_job *j = (fo_index_job *)arg;
  const uint64_t n_buckets = (uint64_t)1 << (2 * j->k);
  const uint32_t lo = (uint32_t)(n_buckets * (uint64_t)j->t / (uint64_t)j->n_threads);
  const uint32_t hi = (uint32_t)(n_buckets * (uint64_t)(j->t + 1) / (uint64_t)j->n_threads);
  for (uint32_t s = 0; s < j->ref->n_seq; ++s) {
    const char *seq = j->ref->text + j->ref->off[s];
    const uint32_t len = j->ref->len[s];
    for (uint32_t pos = 0; (uint64_t)pos + j->k - 1 < len; pos += j->step) {
      const uint32_t h = fo_hash_seed(pos, j->k, seq, len);
      if (h < lo || h >= hi) continue;
      if (j->phase == 0) j->lookup[h + 1]++;
      else j->occ[j->cursor[h]++] = ((uint64_t)s << 32) | pos;
    }
  }
  return NULL;
}

int fo_index_build_mt(const fo_ref *ref, int k, int step, uint32_t *lookup, uint64_t *occ, int n_threads) {
  if (n_threads <= 1) return fo_index_build(ref, k, step, lookup, occ);
  if (n_threads > 64) n_threads = 64;
  size_t n_lookup = ((size_t)1 << (2 * k)) + 1;
  memset(lookup, 0, n_lookup * sizeof(uint32_t));
  uint32_t *cursor = (uint32_t *)malloc((n_lookup - 1) * sizeof(uint32_t));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  fo_index_job *job = (fo_index_job *)calloc((size_t)n_threads, sizeof(fo_index_job));
  if (!cursor || !th || !job) {
    free(cursor), free(th), free(job);
    return -1;
  }
  for (int phase = 0; phase < 2; ++phase) {
    for (int t = 0; t < n_threads; ++t) {
      job[t] = (fo_index_job){ref, k, step, t, n_threads, phase, lookup, cursor, occ};
      pthread_create(&th[t], NULL, fo_index_worker, &job[t]);
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    if (phase == 0) {
      uint32_t run = 0; /* src/index.c:88-92: exclusive prefix sum, uint32 */
      for (size_t i = 1; i < n_lookup; ++i) {
        run += lookup[i];
        lookup[i] = run;
      }
      memcpy(cursor, lookup, (n_lookup - 1) * sizeof(uint32_t));
    }
  }
  free(cursor), free(th), free(job);
  return 0;
}

/* save_index (src/index.c:133-168): int k | int step | uint32 lookup[4^k+1] |
 * size_t n | uint64 occ[n], native endian, no magic. */
int fo_index_save(const char *path, int k, int step, const uint32_t *lookup, uint64_t n_occ, const uint64_t *occ) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  int32_t hdr[2] = {k, step};
  size_t n_lookup = ((size_t)1 << (2 * k)) + 1;
  size_t n = (size_t)n_occ;
  int ok = fwrite(hdr, sizeof(int32_t), 2, f) == 2 && fwrite(lookup, sizeof(uint32_t), n_lookup, f) == n_lookup &&
           fwrite(&n, sizeof(size_t), 1, f) == 1 && fwrite(occ, sizeof(uint64_t), n, f) == n;
  fclose(f);
  return ok ? 0 : -1;
}

/* load_index (src/index.c:100-131) */
int fo_index_load(const char *path, int *k, int *step, uint32_t *lookup, uint64_t *n_occ, uint64_t *occ,
                  uint64_t occ_capacity) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  int32_t hdr[2];
  if (fread(hdr, sizeof(int32_t), 2, f) != 2) {
    fclose(f);
    return -2;
  }
  *k = hdr[0];
  *step = hdr[1];
  size_t n_lookup = ((size_t)1 << (2 * hdr[0])) + 1;
  int rc = 0;
  if (lookup) {
    if (fread(lookup, sizeof(uint32_t), n_lookup, f) != n_lookup) rc = -2;
  } else if (fseek(f, (long)(n_lookup * sizeof(uint32_t)), SEEK_CUR) != 0) {
    rc = -2;
  }
  size_t n = 0;
  if (rc == 0 && fread(&n, sizeof(size_t), 1, f) != 1) rc = -2;
  *n_occ = n;
  if (rc == 0 && occ) {
    if (n > occ_capacity)
      rc = -3;
    else if (fread(occ, sizeof(uint64_t), n, f) != n)
      rc = -2;
  }
  fclose(f);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* seeding + candidate filter (src/filter.c)                                  */
/* ------------------------------------------------------------------------- */
typedef struct { /* Seed (src/utils.h:119-124) */
  uint32_t hash, start, end, freq;
} seed_t;

/* generate_optimal_prefix_qgram_for_group_seeding (src/filter.c:3-43).
 * n_group = seeds in this phase group, lg = ceil(k/step).  Returns
 * M[R][C-1]; picked[] receives the chosen seeds in traceback (right-to-left)
 * order. */
static uint32_t pick_group_seeds(const fo_params *p, uint64_t n_occ, int lg, int n_group, const seed_t *seeds,
                                 seed_t *picked) {
  uint32_t n_rows = (uint32_t)(p->e + p->a + 1 + 1);
  uint32_t n_cols = (uint32_t)(n_group - (p->e + p->a + 1) * lg + 1 + 1);
  uint32_t *M = (uint32_t *)malloc((size_t)n_rows * n_cols * sizeof(uint32_t));
  uint8_t *D = (uint8_t *)malloc((size_t)n_rows * n_cols); /* 3 stop, 2 vertical, 1 horizontal */
#define MM(r, c) M[(size_t)(r) * n_cols + (c)]
#define DD(r, c) D[(size_t)(r) * n_cols + (c)]
  for (uint32_t i = 1; i < n_rows; ++i) {
    MM(i, 0) = (uint32_t)n_occ; /* src/filter.c:9, size_t truncated to uint32 */
    DD(i, 0) = 3;
  }
  for (uint32_t i = 1; i < n_cols; ++i) {
    MM(0, i) = 0;
    DD(0, i) = 3;
  }
  for (uint32_t row = 1; row < n_rows; ++row) {
    for (uint32_t col = 1; col < n_cols; ++col) {
      uint32_t pos = col + (row - 1) * (uint32_t)lg - 1;
      uint32_t with_new = MM(row - 1, col) + seeds[pos].freq; /* uint32 wrap as in the reference */
      if (with_new < MM(row, col - 1)) { /* strict: ties go horizontal (src/filter.c:20) */
        MM(row, col) = with_new;
        DD(row, col) = 2;
      } else {
        MM(row, col) = MM(row, col - 1);
        DD(row, col) = 1;
      }
    }
  }
  /* UB in reference: if the walk reaches column 0 before taking R seeds the
   * tail of optimal_seeds is uninitialised stack (src/filter.c:33-41).  Here
   * the tail is all-zero seeds with freq 0, i.e. they contribute nothing. */
  memset(picked, 0, (size_t)(n_rows - 1) * sizeof(seed_t));
  uint32_t r = n_rows - 1, c = n_cols - 1;
  int n_picked = 0;
  while (DD(r, c) != 3) {
    if (DD(r, c) == 2) {
      picked[n_picked++] = seeds[c + (r - 1) * (uint32_t)lg - 1];
      --r;
    } else {
      --c;
    }
  }
  uint32_t best = MM(n_rows - 1, n_cols - 1);
#undef MM
#undef DD
  free(M);
  free(D);
  return best;
}

/* qsort(compare_seed) (src/filter.c:204, src/utils.h:126-136).  glibc's qsort
 * is a merge sort for arrays this small, hence stable; an insertion sort that
 * only moves on strict '<' is the same permutation. */
static void sort_seeds_by_freq_stable(seed_t *s, int n) {
  for (int i = 1; i < n; ++i) {
    seed_t t = s[i];
    int j = i;
    while (j > 0 && t.freq < s[j - 1].freq) {
      s[j] = s[j - 1];
      --j;
    }
    s[j] = t;
  }
}

/* merge_candidate_locations (src/filter.c:80-116): fold the seeds' shifted
 * occurrence lists into buf1.  The last seed is only merged while buf1 still
 * has elements (src/filter.c:85). */
static void merge_seed_lists(const fo_index *idx, const seed_t *seeds, size_t n_seeds, v64 *buf1, v64 *buf2) {
  for (size_t si = 0; si < n_seeds; ++si) {
    size_t i1 = 0, io = 0;
    const uint64_t *list = idx->occ + idx->lookup[seeds[si].hash]; /* get_seed_occurrences, src/index.h:26 */
    uint32_t freq = seeds[si].freq, start = seeds[si].start;
    while (i1 < buf1->n || (si != n_seeds - 1 && io < freq)) {
      if (i1 < buf1->n) {
        uint64_t b = buf1->a[i1];
        if (io < freq) {
          if ((uint32_t)list[io] < start) {
            ++io;
          } else {
            uint64_t s = list[io] - start;
            if (s <= b) {
              v64_push(buf2, s);
              ++io;
            } else {
              v64_push(buf2, b);
              ++i1;
            }
          }
        } else {
          v64_push(buf2, b);
          ++i1;
        }
      } else {
        if ((uint32_t)list[io] >= start) v64_push(buf2, list[io] - start);
        ++io;
      }
    }
    v64_swap(buf1, buf2);
    buf2->n = 0;
  }
}

/* additional_qgram_filter (src/filter.c:118-131) */
static void qgram_window_filter(const fo_params *p, const v64 *in, v64 *out) {
  for (size_t ci = 0; ci < in->n; ++ci) {
    size_t in_range = 1;
    while (ci + in_range < in->n && in->a[ci + in_range] <= in->a[ci] + (uint64_t)p->e) {
      ++in_range;
      if (in_range > (size_t)p->a) break;
    }
    if (in_range > (size_t)p->a) v64_push(out, in->a[ci]);
  }
}

/* merge_kvec_t_uint64_t (src/filter.c:45-78): two-way merge, element of the
 * second list first on ties, keeping x only if x > last kept + e. */
static void merge_greedy_dedup(const fo_params *p, const v64 *b1, const v64 *b2, v64 *out) {
  size_t i1 = 0, i2 = 0;
  while (i1 < b1->n || i2 < b2->n) {
    uint64_t x;
    if (i1 < b1->n && (i2 >= b2->n || b1->a[i1] < b2->a[i2]))
      x = b1->a[i1++];
    else
      x = b2->a[i2++];
    if (out->n == 0 || x > out->a[out->n - 1] + (uint64_t)p->e) v64_push(out, x);
  }
}

/* remove_out_ranged_candidates (src/filter.c:133-144) */
static void clip_to_reference(const fo_params *p, uint32_t read_len, const fo_ref *ref, const v64 *in, v64 *out) {
  for (size_t i = 0; i < in->n; ++i) {
    uint64_t c = in->a[i];
    uint32_t seq = (uint32_t)(c >> 32), pos = (uint32_t)c;
    uint32_t slen = ref->len[seq];
    if (pos >= (uint32_t)p->e && pos + read_len + (uint32_t)p->e < slen) v64_push(out, c - (uint64_t)p->e);
  }
}

typedef struct {
  v64 buf1, buf2, cands;
} seed_scratch;

/* generate_group_seeding_candidates (src/filter.c:146-223) */
static uint32_t seed_candidates(const fo_params *p, const char *seq, uint32_t len, const fo_ref *ref,
                                const fo_index *idx, seed_scratch *s, uint32_t *pre_filter) {
  v64 *buf1 = &s->buf1, *buf2 = &s->buf2, *cands = &s->cands;
  buf1->n = buf2->n = cands->n = 0;
  *pre_filter = 0;
  int R = p->e + 1 + p->a;
  int lg = p->k / p->step + (p->k % p->step > 0 ? 1 : 0);
  int n_seeds = (int)len - p->k + 1;
  if (n_seeds <= 0) return 0; /* reference: assert(num_seeds_in_read > 0), src/filter.c:167 */
  if (R > n_seeds / p->step) return 0; /* src/filter.c:168-172 */
  /* UB in reference: with G < R*lg - 1 the uint32 column count underflows
   * (src/filter.c:5-7) and with G == R*lg - 1 the DP has no column to pick
   * from; the reference crashes or reads garbage.  Such (len, e, a) yield no
   * candidates here.  The smallest phase group decides. */
  {
    int g_min = (n_seeds - (p->step - 1)) / p->step;
    if (g_min - R * lg + 2 < 2) return 0;
  }
  uint32_t *hashes = (uint32_t *)malloc((size_t)n_seeds * sizeof(uint32_t));
  int n_amb = 0;
  hash_all_seeds(n_seeds, p->k, seq, len, &n_amb, hashes);
  if (n_amb > p->e) { /* src/filter.c:180-182 */
    free(hashes);
    return 0;
  }
  seed_t *group = (seed_t *)malloc((size_t)(n_seeds / p->step + 1) * sizeof(seed_t));
  seed_t *picked = (seed_t *)malloc((size_t)R * sizeof(seed_t));
  for (int si = 0; si < p->step; ++si) { /* src/filter.c:190-213 */
    int G = ((int)len - p->k + 1 - si) / p->step;
    for (int j = 0; j < G; ++j) {
      int at = si + j * p->step;
      group[j].hash = hashes[at];
      group[j].start = (uint32_t)at;
      group[j].end = (uint32_t)(at + p->k);
      group[j].freq = idx->lookup[hashes[at] + 1] - idx->lookup[hashes[at]]; /* get_seed_frequency */
    }
    *pre_filter += pick_group_seeds(p, idx->n_occ, lg, G, group, picked);
    sort_seeds_by_freq_stable(picked, R);
    buf1->n = buf2->n = 0;
    merge_seed_lists(idx, picked, (size_t)R, buf1, buf2);
    qgram_window_filter(p, buf1, buf2);
    v64_swap(buf1, cands); /* buf1 <- candidates so far */
    cands->n = 0;
    merge_greedy_dedup(p, buf1, buf2, cands);
  }
  v64_swap(buf1, cands);
  cands->n = 0;
  clip_to_reference(p, len, ref, buf1, cands);
  free(hashes);
  free(group);
  free(picked);
  return (uint32_t)cands->n;
}

uint32_t fo_seed_candidates(const fo_params *p, const char *seq, uint32_t len, const fo_ref *ref, const fo_index *idx,
                            uint64_t *cands, uint32_t cap, uint32_t *pre_filter) {
  seed_scratch s;
  memset(&s, 0, sizeof s);
  uint32_t n = seed_candidates(p, seq, len, ref, idx, &s, pre_filter);
  uint32_t rc = n;
  if (n > cap)
    rc = UINT32_MAX;
  else if (n)
    memcpy(cands, s.cands.a, (size_t)n * sizeof(uint64_t));
  free(s.buf1.a);
  free(s.buf2.a);
  free(s.cands.a);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* banded Myers verification (src/align.c:102-277)                            */
/* ------------------------------------------------------------------------- */
/* banded_edit_distance (src/align.c:102-147): 32-bit words, band 2e+1. */
int fo_banded_ed32(int e, const char *pattern, const char *text, int len, int *end) {
  uint32_t Peq[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * e; ++i) Peq[base_code(pattern[i])] |= (uint32_t)1 << i;
  uint32_t top = (uint32_t)1 << (2 * e);
  uint32_t VP = 0, VN = 0, X, D0, HN, HP;
  int score = 0;
  for (int i = 0; i < len; ++i) {
    Peq[base_code(pattern[i + 2 * e])] |= top;
    X = Peq[base_code(text[i])] | VN;
    D0 = ((VP + (X & VP)) ^ VP) | X;
    HN = VP & D0;
    HP = VN | ~(VP | D0);
    X = D0 >> 1;
    VN = X & HP;
    VP = HN | ~(X | HP);
    score += 1 - (int)(D0 & 1u);
    if (score > 3 * e) return e + 1;
    for (int c = 0; c < 5; ++c) Peq[c] >>= 1;
  }
  int best = score;
  *end = len - 1;
  for (int i = 0; i < 2 * e; ++i) {
    score += (int)((VP >> i) & 1u);
    score -= (int)((VN >> i) & 1u);
    if (score < best) { /* first strict minimum (src/align.c:141-144) */
      best = score;
      *end = len - 1 + 1 + i;
    }
  }
  return best;
}

/* vectorized_banded_edit_distance (src/align.c:149-277): the same recurrence
 * in 8 independent int16 lanes.  Differences kept on purpose: 16-bit wrap of
 * the add, logical 16-bit shifts, no per-lane early return (only when all 8
 * lanes exceed 3e), Peq primed by shifting the top bit down.  end[] must be
 * pre-set by the caller to len-1 (src/align.c:17-19). */
void fo_banded_ed16x8(int e, const char *const pattern[8], const char *text, int len, int16_t ed[8], int16_t end[8]) {
  uint16_t Peq[8][5];
  uint16_t VP[8], VN[8];
  int16_t score[8], best[8];
  uint16_t top = (uint16_t)(1u << (2 * e));
  memset(Peq, 0, sizeof Peq);
  for (int i = 0; i < 2 * e; ++i) {
    for (int l = 0; l < 8; ++l) Peq[l][base_code(pattern[l][i])] |= top;
    for (int l = 0; l < 8; ++l)
      for (int c = 0; c < 5; ++c) Peq[l][c] >>= 1;
  }
  for (int l = 0; l < 8; ++l) VP[l] = VN[l] = 0, score[l] = 0;
  for (int i = 0; i < len; ++i) {
    int all_over = 1;
    for (int l = 0; l < 8; ++l) {
      Peq[l][base_code(pattern[l][i + 2 * e])] |= top;
      uint16_t X = (uint16_t)(Peq[l][base_code(text[i])] | VN[l]);
      uint16_t D0 = (uint16_t)(X & VP[l]);
      D0 = (uint16_t)(D0 + VP[l]);
      D0 = (uint16_t)(D0 ^ VP[l]);
      D0 = (uint16_t)(D0 | X);
      uint16_t HN = (uint16_t)(VP[l] & D0);
      uint16_t HP = (uint16_t)(VP[l] | D0);
      HP = (uint16_t)(HP ^ 0xffffu);
      HP = (uint16_t)(HP | VN[l]);
      X = (uint16_t)(D0 >> 1);
      VN[l] = (uint16_t)(X & HP);
      VP[l] = (uint16_t)(X | HP);
      VP[l] = (uint16_t)(VP[l] ^ 0xffffu);
      VP[l] = (uint16_t)(VP[l] | HN);
      score[l] = (int16_t)(score[l] + (int16_t)((D0 & 1u) ^ 1u));
      if (!(score[l] > (int16_t)(3 * e))) all_over = 0;
    }
    if (all_over) { /* src/align.c:247-252 */
      for (int l = 0; l < 8; ++l) ed[l] = score[l];
      return;
    }
    for (int l = 0; l < 8; ++l)
      for (int c = 0; c < 5; ++c) Peq[l][c] >>= 1;
  }
  for (int l = 0; l < 8; ++l) best[l] = score[l];
  for (int i = 0; i < 2 * e; ++i) {
    for (int l = 0; l < 8; ++l) {
      score[l] = (int16_t)(score[l] + (int16_t)(VP[l] & 1u));
      score[l] = (int16_t)(score[l] - (int16_t)(VN[l] & 1u));
      if (score[l] < best[l]) {
        end[l] = (int16_t)(len - 1 + 1 + i);
        best[l] = score[l];
      }
      VP[l] >>= 1;
      VN[l] >>= 1;
    }
  }
  for (int l = 0; l < 8; ++l) ed[l] = best[l];
}

/* ------------------------------------------------------------------------- */
/* mapping order (src/align.c:53-57, src/ksort.h:101-151)                     */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint64_t key;
  uint32_t idx;
} keyed_t;

/* rs_insertsort (src/ksort.h:105-115) */
static void ks_insertion(keyed_t *beg, keyed_t *end) {
  for (keyed_t *i = beg + 1; i < end; ++i)
    if (i->key < (i - 1)->key) {
      keyed_t *j, tmp = *i;
      for (j = i; j > beg && tmp.key < (j - 1)->key; --j) *j = *(j - 1);
      *j = tmp;
    }
}

/* rs_sort (src/ksort.h:116-144): in-place MSD radix, 8 bits per level, cyclic
 * permutation inside each level, buckets of <=64 finished by insertion sort. */
static void ks_radix_level(keyed_t *beg, keyed_t *end, int n_bits, int s) {
  int size = 1 << n_bits, m = size - 1;
  struct bucket {
    keyed_t *b, *e;
  } b[256], *be = b + size, *k;
  for (k = b; k != be; ++k) k->b = k->e = beg;
  for (keyed_t *i = beg; i != end; ++i) ++b[i->key >> s & m].e;
  for (k = b + 1; k != be; ++k) k->e += (k - 1)->e - beg, k->b = (k - 1)->e;
  for (k = b; k != be;) {
    if (k->b != k->e) {
      struct bucket *l;
      if ((l = b + (k->b->key >> s & m)) != k) {
        keyed_t tmp = *k->b, swap;
        do {
          swap = tmp;
          tmp = *l->b;
          *l->b++ = swap;
          l = b + (tmp.key >> s & m);
        } while (l != k);
        *k->b++ = tmp;
      } else
        ++k->b;
    } else
      ++k;
  }
  for (b->b = beg, k = b + 1; k != be; ++k) k->b = (k - 1)->e;
  if (s) {
    s = s > n_bits ? s - n_bits : 0;
    for (k = b; k != be; ++k)
      if (k->e - k->b > 64)
        ks_radix_level(k->b, k->e, n_bits, s);
      else if (k->e - k->b > 1)
        ks_insertion(k->b, k->e);
  }
}

/* radix_sort_mapping (KRADIX_SORT_INIT(mapping, Mapping, MappingSortKey, 8),
 * src/align.c:54; entry point src/ksort.h:146-150) */
void fo_sort_mapping_keys(uint64_t *keys, uint32_t *perm, uint32_t n) {
  keyed_t *a = (keyed_t *)malloc((size_t)(n ? n : 1) * sizeof(keyed_t));
  for (uint32_t i = 0; i < n; ++i) a[i].key = keys[i], a[i].idx = i;
  if (n <= 64)
    ks_insertion(a, a + n);
  else
    ks_radix_level(a, a + n, 8, (8 - 1) * 8);
  for (uint32_t i = 0; i < n; ++i) keys[i] = a[i].key, perm[i] = a[i].idx;
  free(a);
}

/* ------------------------------------------------------------------------- */
/* traceback -> CIGAR, MD (src/align.c:279-544)                               */
/* ------------------------------------------------------------------------- */
/* generate_MD_tag (src/align.c:501-544) */
static int make_md(const char *pattern, const char *text, int start, const uint32_t *cigar, int n_cigar, char *md,
                   int md_cap) {
  const char *ref = pattern + start;
  int n_match = 0, rp = 0, tp = 0, o = 0;
#define MD_ROOM(k)            \
  if (o + (k) >= md_cap) return -1
  for (int ci = 0; ci < n_cigar; ++ci) {
    int op = (int)(cigar[ci] & 0xf), n = (int)(cigar[ci] >> 4);
    if (op == 0) {
      for (int i = 0; i < n; ++i) {
        if (ref[rp] == text[tp]) {
          ++n_match;
        } else {
          if (n_match != 0) {
            MD_ROOM(12);
            o += sprintf(md + o, "%d", n_match);
            n_match = 0;
          }
          MD_ROOM(1);
          md[o++] = ref[rp];
        }
        ++rp;
        ++tp;
      }
    } else if (op == 1) {
      tp += n;
    } else if (op == 2) {
      if (n_match != 0) {
        MD_ROOM(12);
        o += sprintf(md + o, "%d", n_match);
        n_match = 0;
      }
      MD_ROOM(1);
      md[o++] = '^';
      for (int i = 0; i < n; ++i) {
        MD_ROOM(1);
        md[o++] = ref[rp++];
      }
    }
  }
  if (n_match != 0) {
    MD_ROOM(12);
    o += sprintf(md + o, "%d", n_match);
  }
#undef MD_ROOM
  md[o] = 0;
  return o;
}

/* generate_alignment (src/align.c:279-499) */
int fo_align(int e, const char *pattern, const char *text, int len, int ed, int end, uint32_t *cigar, int cigar_cap,
             int *n_cigar, char *md, int md_cap) {
  int start = end - len + 1; /* src/align.c:285 */
  if (start < 0) return -1;  /* reference: assert */
  *n_cigar = 0;
  int n_err = 0;
  for (int i = 0; i < len; ++i)
    if (text[i] != pattern[start + i]) ++n_err;
  if (n_err == 0) { /* src/align.c:294-300 */
    if (cigar_cap < 1) return -2;
    cigar[0] = (uint32_t)len << 4;
    *n_cigar = 1;
    if (make_md(pattern, text, start, cigar, 1, md, md_cap) < 0) return -2;
    return start;
  }
  uint32_t *D0s = (uint32_t *)malloc((size_t)len * sizeof(uint32_t));
  uint32_t *HPs = (uint32_t *)malloc((size_t)len * sizeof(uint32_t));
  char *ops = (char *)malloc((size_t)len + 2);
  int *cnt = (int *)calloc((size_t)len + 2, sizeof(int));
  uint32_t Peq[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * e; ++i) Peq[base_code(pattern[i])] |= (uint32_t)1 << i;
  uint32_t top = (uint32_t)1 << (2 * e);
  uint32_t VP = 0, VN = 0, X, D0, HN, HP;
  for (int i = 0; i < len; ++i) { /* src/align.c:319-338 */
    Peq[base_code(pattern[i + 2 * e])] |= top;
    X = Peq[base_code(text[i])] | VN;
    D0 = ((VP + (X & VP)) ^ VP) | X;
    HN = VP & D0;
    HP = VN | ~(VP | D0);
    X = D0 >> 1;
    VN = X & HP;
    VP = HN | ~(X | HP);
    D0s[i] = D0;
    HPs[i] = HP;
    for (int c = 0; c < 5; ++c) Peq[c] >>= 1;
  }
  int rc = 0;
  int bit = end - len + 1; /* pattern_bit_position */
  int t = len - 1;         /* text_position */
  int pe = end;            /* mapping_end_position, walks left */
  char pre_op = 'S';
  int pre_n = 1;
  n_err = 0;
#define DBIT ((D0s[t] >> bit) & 1u)
#define HBIT ((HPs[t] >> bit) & 1u)
  /* first step (src/align.c:345-368) */
  if (DBIT && pattern[pe] == text[t]) {
    --t, --pe;
    pre_op = 'M', pre_n = 1;
  } else if (!DBIT) {
    --t, --pe, ++n_err;
    pre_op = 'S', pre_n = 1;
  } else if (DBIT && HBIT) {
    --t, ++bit, ++n_err;
    pre_op = 'S', pre_n = 1;
    ++start;
  } else {
    rc = -3; /* reference: assert(1 == 0) */
  }
  int oi = 0;
  while (rc == 0 && t >= 0) { /* src/align.c:373-440 */
    if (n_err == ed) break;
    if (bit < 0 || bit > 31 || pe < 0) {
      rc = -3; /* outside anything the reference could index safely */
      break;
    }
    if (DBIT && pattern[pe] == text[t]) { /* match */
      --t, --pe;
      if (pre_op != 'M') {
        ops[oi] = pre_op, cnt[oi] = pre_n, ++oi;
        pre_op = 'M', pre_n = 1;
      } else {
        ++pre_n;
      }
    } else if (!DBIT) { /* mismatch */
      --t, --pe, ++n_err;
      if (pre_op == 'S') {
        ++pre_n;
      } else if (pre_op != 'M') {
        ops[oi] = pre_op, cnt[oi] = pre_n, ++oi;
        pre_op = 'M', pre_n = 1;
      } else {
        ++pre_n;
      }
    } else if (DBIT && HBIT) { /* insertion */
      --t, ++bit, ++n_err;
      if (pre_op == 'S') {
        ++pre_n;
      } else if (pre_op != 'I') {
        ops[oi] = pre_op, cnt[oi] = pre_n, ++oi;
        pre_op = 'I', pre_n = 1;
      } else {
        ++pre_n;
      }
      ++start;
    } else { /* deletion */
      --bit, --pe, ++n_err;
      if (pre_op != 'D') {
        ops[oi] = pre_op, cnt[oi] = pre_n, ++oi;
        pre_op = 'D', pre_n = 1;
      } else {
        ++pre_n;
      }
      --start;
    }
  }
#undef DBIT
#undef HBIT
  if (rc == 0) {
    if (t >= 0) { /* src/align.c:445-459 */
      if (pre_op != 'M') {
        ops[oi] = pre_op, cnt[oi] = pre_n, ++oi;
        ops[oi] = 'M', cnt[oi] = t + 1;
      } else {
        ops[oi] = 'M', cnt[oi] = pre_n + t + 1;
      }
    } else {
      ops[oi] = pre_op, cnt[oi] = pre_n;
    }
    int lo = 0;
    if (ops[0] == 'S') { /* src/align.c:466-469 */
      cnt[1] += cnt[0];
      lo = 1;
    }
    for (int i = oi; i >= lo && rc == 0; --i) { /* src/align.c:470-496 */
      uint32_t c = (uint32_t)cnt[i] << 4;
      if (*n_cigar >= cigar_cap) {
        rc = -2;
      } else if (ops[i] == 'M') {
        cigar[(*n_cigar)++] = c | 0;
      } else if (ops[i] == 'I') {
        cigar[(*n_cigar)++] = c | 1;
      } else if (ops[i] == 'D') {
        cigar[(*n_cigar)++] = c | 2;
      } else {
        rc = -3; /* reference: assert(1 == 0) */
      }
    }
  }
  if (rc == 0 && make_md(pattern, text, start, cigar, *n_cigar, md, md_cap) < 0) rc = -2;
  free(D0s);
  free(HPs);
  free(ops);
  free(cnt);
  return rc == 0 ? start : rc;
}

/* ------------------------------------------------------------------------- */
/* batch driver                                                               */
/* ------------------------------------------------------------------------- */
struct fo_result {
  uint64_t n_reads;
  uint64_t stats[5];
  /* candidates */
  v64 cand_off; /* 2n+1 */
  v64 cands;
  v32 pre; /* 2n */
  v8 v_ed;
  vi16 v_end;
  /* mappings in verify order */
  v64 map_off; /* n+1 */
  v8 m_dir, m_ed;
  v64 m_cand;
  vi16 m_end;
  /* records */
  v64 rec_off; /* n+1 */
  v16 r_flag;
  v32 r_tid, r_pos;
  v8 r_nm;
  v64 cig_off;
  v32 cig;
  v64 md_off;
  vch md;
};

typedef struct {
  const fo_params *p;
  const fo_ref *ref;
  const fo_index *idx;
  const fo_reads *reads;
  uint64_t lo, hi;
  int stages;
  fo_result *out;
} worker_t;

/* verify_candidates (src/align.c:4-51) for one strand; appends to the
 * per-candidate verify arrays and to the mapping arrays. */
static uint32_t verify_strand(const fo_params *p, const char *text, int len, uint8_t dir, const fo_ref *ref,
                              const uint64_t *cands, uint32_t n, fo_result *o) {
  uint32_t n_map = 0;
  uint32_t n_groups = n / 8, n_rest = n % 8;
  for (uint32_t g = 0; g < n_groups; ++g) {
    const char *pat[8];
    int16_t ed[8], end[8];
    for (int l = 0; l < 8; ++l) {
      uint64_t c = cands[g * 8 + l];
      pat[l] = ref->text + ref->off[(uint32_t)(c >> 32)] + (uint32_t)c;
      end[l] = (int16_t)(len - 1);
    }
    fo_banded_ed16x8(p->e, pat, text, len, ed, end);
    for (int l = 0; l < 8; ++l) {
      if (ed[l] <= p->e) {
        v8_push(&o->v_ed, (uint8_t)ed[l]);
        vi16_push(&o->v_end, end[l]);
        v8_push(&o->m_dir, dir);
        v8_push(&o->m_ed, (uint8_t)ed[l]);
        v64_push(&o->m_cand, cands[g * 8 + l]);
        vi16_push(&o->m_end, end[l]);
        ++n_map;
      } else {
        v8_push(&o->v_ed, 0xFF);
        vi16_push(&o->v_end, 0);
      }
    }
  }
  for (uint32_t ci = 0; ci < n_rest; ++ci) {
    uint64_t c = cands[n_groups * 8 + ci];
    const char *pat = ref->text + ref->off[(uint32_t)(c >> 32)] + (uint32_t)c;
    int end = -len;
    int ed = fo_banded_ed32(p->e, pat, text, len, &end);
    if (ed <= p->e) {
      v8_push(&o->v_ed, (uint8_t)ed);
      vi16_push(&o->v_end, (int16_t)end);
      v8_push(&o->m_dir, dir);
      v8_push(&o->m_ed, (uint8_t)ed);
      v64_push(&o->m_cand, c);
      vi16_push(&o->m_end, (int16_t)end);
      ++n_map;
    } else {
      v8_push(&o->v_ed, 0xFF);
      vi16_push(&o->v_end, 0);
    }
  }
  return n_map;
}

/* process_mappings (src/align.c:56-92) for one read: sort, traceback, record
 * fields.  first = index of the read's first mapping in o->m_*. */
static void emit_records(const fo_params *p, const char *fwd, const char *rev, int len, const fo_ref *ref,
                         fo_result *o, size_t first, uint32_t n) {
  uint64_t *keys = (uint64_t *)malloc((size_t)n * sizeof(uint64_t));
  uint32_t *perm = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
  for (uint32_t i = 0; i < n; ++i) /* MappingSortKey (src/align.c:53) */
    keys[i] = ((uint64_t)o->m_ed.a[first + i] << 60) | ((uint64_t)o->m_dir.a[first + i] << 59) |
              (o->m_cand.a[first + i] + (uint64_t)(int64_t)o->m_end.a[first + i]);
  fo_sort_mapping_keys(keys, perm, n);
  int cig_cap = len + 2, md_cap = 16 * len + 64;
  uint32_t *cig = (uint32_t *)malloc((size_t)cig_cap * sizeof(uint32_t));
  char *md = (char *)malloc((size_t)md_cap);
  for (uint32_t mi = 0; mi < n; ++mi) {
    size_t m = first + perm[mi];
    uint8_t dir = o->m_dir.a[m];
    uint64_t cand = o->m_cand.a[m];
    const char *text = dir == 0 ? fwd : rev;
    const char *pat = ref->text + ref->off[(uint32_t)(cand >> 32)] + (uint32_t)cand;
    int n_cig = 0;
    int start = fo_align(p->e, pat, text, len, o->m_ed.a[m], o->m_end.a[m], cig, cig_cap, &n_cig, md, md_cap);
    uint16_t flag = dir == 0 ? 0 : 16; /* BAM_FREVERSE */
    if (mi > 0) flag |= 256;           /* BAM_FSECONDARY */
    if (start < 0) {
      flag |= 0x8000; /* marks "reference would have asserted"; never set on valid data */
      n_cig = 0;
      md[0] = 0;
      start = 0;
    }
    v16_push(&o->r_flag, flag);
    v32_push(&o->r_tid, (uint32_t)(cand >> 32));
    v32_push(&o->r_pos, (uint32_t)start + (uint32_t)cand); /* src/align.c:80 */
    v8_push(&o->r_nm, o->m_ed.a[m]);
    v32_append(&o->cig, cig, (size_t)n_cig);
    v64_push(&o->cig_off, o->cig.n);
    vch_append(&o->md, md, strlen(md));
    v64_push(&o->md_off, o->md.n);
  }
  free(keys);
  free(perm);
  free(cig);
  free(md);
}

/* single_end_read_mapping_thread (src/map.c:17-58) over reads [lo, hi) */
static void *map_range(void *arg) {
  worker_t *w = (worker_t *)arg;
  fo_result *o = w->out;
  const fo_params *p = w->p;
  seed_scratch scratch;
  memset(&scratch, 0, sizeof scratch);
  vch rev = {0, 0, 0};
  v64_push(&o->cand_off, 0);
  v64_push(&o->map_off, 0);
  v64_push(&o->rec_off, 0);
  v64_push(&o->cig_off, 0);
  v64_push(&o->md_off, 0);
  for (uint64_t ri = w->lo; ri < w->hi; ++ri) {
    const char *fwd = w->reads->bases + w->reads->off[ri];
    uint32_t len = (uint32_t)(w->reads->off[ri + 1] - w->reads->off[ri]);
    size_t first_map = o->m_cand.n;
    o->stats[0] += 1;
    for (int dir = 0; dir < 2; ++dir) {
      const char *text = fwd;
      if (dir == 1) { /* src/map.c:40 */
        rev.n = 0;
        for (uint32_t i = 0; i < len; ++i) vch_push(&rev, 0);
        fo_revcomp(fwd, len, rev.a);
        text = rev.a;
      }
      uint32_t pre = 0;
      uint32_t n = seed_candidates(p, text, len, w->ref, w->idx, &scratch, &pre);
      o->stats[2] += pre;
      o->stats[3] += n;
      v32_push(&o->pre, pre);
      v64_append(&o->cands, scratch.cands.a, n);
      v64_push(&o->cand_off, o->cands.n);
      if (n > 0 && (w->stages & FO_STAGE_VERIFY))
        o->stats[4] += verify_strand(p, text, (int)len, (uint8_t)dir, w->ref, scratch.cands.a, n, o);
    }
    uint32_t n_map = (uint32_t)(o->m_cand.n - first_map);
    v64_push(&o->map_off, o->m_cand.n);
    if (n_map > 0) {
      o->stats[1] += 1;
      if (w->stages & FO_STAGE_ALIGN) emit_records(p, fwd, rev.a, (int)len, w->ref, o, first_map, n_map);
    }
    v64_push(&o->rec_off, o->r_flag.n);
  }
  free(scratch.buf1.a);
  free(scratch.buf2.a);
  free(scratch.cands.a);
  free(rev.a);
  return NULL;
}

static void result_free_fields(fo_result *r) {
  free(r->cand_off.a), free(r->cands.a), free(r->pre.a), free(r->v_ed.a), free(r->v_end.a);
  free(r->map_off.a), free(r->m_dir.a), free(r->m_ed.a), free(r->m_cand.a), free(r->m_end.a);
  free(r->rec_off.a), free(r->r_flag.a), free(r->r_tid.a), free(r->r_pos.a), free(r->r_nm.a);
  free(r->cig_off.a), free(r->cig.a), free(r->md_off.a), free(r->md.a);
}

/* append a cumulative-offset vector (first entry 0) shifted by base */
static void append_offsets(v64 *dst, const v64 *src, uint64_t base) {
  for (size_t i = 1; i < src->n; ++i) v64_push(dst, src->a[i] + base);
}

fo_result *fo_map(const fo_params *p, const fo_ref *ref, const fo_index *idx, const fo_reads *reads, int n_threads,
                  int stages) {
  if (n_threads < 1) n_threads = 1;
  if ((uint64_t)n_threads > reads->n && reads->n > 0) n_threads = (int)reads->n;
  if (stages & FO_STAGE_ALIGN) stages |= FO_STAGE_VERIFY;
  worker_t *w = (worker_t *)calloc((size_t)n_threads, sizeof(worker_t));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  for (int t = 0; t < n_threads; ++t) {
    w[t].p = p, w[t].ref = ref, w[t].idx = idx, w[t].reads = reads, w[t].stages = stages;
    w[t].lo = reads->n * (uint64_t)t / (uint64_t)n_threads;
    w[t].hi = reads->n * (uint64_t)(t + 1) / (uint64_t)n_threads;
    w[t].out = (fo_result *)calloc(1, sizeof(fo_result));
  }
  if (n_threads == 1) {
    map_range(&w[0]);
  } else {
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, map_range, &w[t]);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  }
  fo_result *r = w[0].out; /* thread 0's buffers become the merged result */
  for (int t = 1; t < n_threads; ++t) { /* MappingStats reduction (src/FEM_map.c:200-212) */
    fo_result *o = w[t].out;
    for (int i = 0; i < 5; ++i) r->stats[i] += o->stats[i];
    append_offsets(&r->cand_off, &o->cand_off, r->cands.n);
    v64_append(&r->cands, o->cands.a, o->cands.n);
    v32_append(&r->pre, o->pre.a, o->pre.n);
    v8_append(&r->v_ed, o->v_ed.a, o->v_ed.n);
    vi16_append(&r->v_end, o->v_end.a, o->v_end.n);
    append_offsets(&r->map_off, &o->map_off, r->m_cand.n);
    v8_append(&r->m_dir, o->m_dir.a, o->m_dir.n);
    v8_append(&r->m_ed, o->m_ed.a, o->m_ed.n);
    v64_append(&r->m_cand, o->m_cand.a, o->m_cand.n);
    vi16_append(&r->m_end, o->m_end.a, o->m_end.n);
    append_offsets(&r->rec_off, &o->rec_off, r->r_flag.n);
    v16_append(&r->r_flag, o->r_flag.a, o->r_flag.n);
    v32_append(&r->r_tid, o->r_tid.a, o->r_tid.n);
    v32_append(&r->r_pos, o->r_pos.a, o->r_pos.n);
    v8_append(&r->r_nm, o->r_nm.a, o->r_nm.n);
    append_offsets(&r->cig_off, &o->cig_off, r->cig.n);
    v32_append(&r->cig, o->cig.a, o->cig.n);
    append_offsets(&r->md_off, &o->md_off, r->md.n);
    vch_append(&r->md, o->md.a, o->md.n);
    result_free_fields(o);
    free(o);
  }
  r->n_reads = reads->n;
  if (reads->n == 0 && r->cand_off.n == 0) { /* keep offset vectors well-formed */
    v64_push(&r->cand_off, 0), v64_push(&r->map_off, 0), v64_push(&r->rec_off, 0);
    v64_push(&r->cig_off, 0), v64_push(&r->md_off, 0);
  }
  free(w);
  free(th);
  return r;
}

void fo_result_free(fo_result *r) {
  if (!r) return;
  result_free_fields(r);
  free(r);
}

void fo_result_stats(const fo_result *r, uint64_t out[5]) { memcpy(out, r->stats, sizeof r->stats); }

uint64_t fo_result_candidates(const fo_result *r, const uint64_t **cand_off, const uint64_t **cands,
                              const uint32_t **pre) {
  *cand_off = r->cand_off.a, *cands = r->cands.a, *pre = r->pre.a;
  return r->cands.n;
}

void fo_result_verify(const fo_result *r, const uint8_t **ed, const int16_t **end) {
  *ed = r->v_ed.a, *end = r->v_end.a;
}

uint64_t fo_result_mappings(const fo_result *r, const uint64_t **map_off, const uint8_t **dir, const uint8_t **ed,
                            const uint64_t **cand, const int16_t **end) {
  *map_off = r->map_off.a, *dir = r->m_dir.a, *ed = r->m_ed.a, *cand = r->m_cand.a, *end = r->m_end.a;
  return r->m_cand.n;
}

uint64_t fo_result_records(const fo_result *r, const uint64_t **rec_off, const uint16_t **flag, const uint32_t **tid,
                           const uint32_t **pos0, const uint8_t **nm, const uint64_t **cigar_off,
                           const uint32_t **cigar, const uint64_t **md_off, const char **md) {
  *rec_off = r->rec_off.a, *flag = r->r_flag.a, *tid = r->r_tid.a, *pos0 = r->r_pos.a, *nm = r->r_nm.a;
  *cigar_off = r->cig_off.a, *cigar = r->cig.a, *md_off = r->md_off.a, *md = r->md.a;
  return r->r_flag.n;
}
