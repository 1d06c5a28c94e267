/* ref_klib.c — TEST INFRASTRUCTURE ONLY (see fem_oracle.h): a thin harness around the two klib headers of the reference
 * that compile on their own, built FROM WHERE THEY LIE under /root/reference (never copied): make -C oracle ref
 *
 *   kseq.h   the FASTA/FASTQ record reader, instantiated exactly as src/sequence_batch.h:13 does
 *            (KSEQ_INIT(gzFile, gzread)); opened and read as src/sequence_batch.c:30-37,47-66 do
 *   ksort.h  the radix sort behind radix_sort_mapping, instantiated as src/align.c:53-54 does
 *            (KRADIX_SORT_INIT(mapping, Mapping, MappingSortKey, 8)) on records that carry their key ready-made
 *
 * Everything else of the reference includes htslib (src/utils.h:18), which this image does not have, and is NOT built.
 * What this pins: the record rules of the read/reference parser (libfemhost's readers) and the order in which a read's
 * mappings are emitted, ties included (oracle, host tail, device ordering kernel).  Output: oracle/_ref/libfemref_klib.so.
 * Only tests/ and tests/golden/make_klib_golden.py load it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "kseq.h"
KSEQ_INIT(gzFile, gzread) /* src/sequence_batch.h:13 */

#include "ksort.h"
typedef struct {
  uint64_t key; /* MappingSortKey(m), src/align.c:53 */
  uint32_t tag; /* the record's index before sorting */
} ref_item;
#define ref_item_key(x) ((x).key)
KRADIX_SORT_INIT(refitem, ref_item, ref_item_key, 8) /* src/align.c:54 */

typedef struct {
  uint64_t n;
  int32_t last_rc; /* what kseq_read returned last: -1 end of file, -2 truncated quality, -3 stream error */
  uint64_t *seq_off, *name_off, *comment_off; /* n + 1 each */
  uint8_t *has_qual;                          /* n */
  char *seq, *qual, *name, *comment;          /* qual: same offsets as seq (zeros where a record has none) */
} ref_records;

static void push(char **buf, uint64_t *len, uint64_t *cap, const char *p, size_t n) {
  if (*len + n + 1 > *cap) {
    *cap = (*len + n + 1) * 2 + 64;
    *buf = (char *)realloc(*buf, *cap);
  }
  if (n) memcpy(*buf + *len, p, n);
  *len += n;
}

/* Every record kseq_read yields, zero-length ones included (the loader skips those, src/sequence_batch.c:50-52). */
int ref_kseq_read_all(const char *path, ref_records *out) {
  memset(out, 0, sizeof *out);
  gzFile fp = gzopen(path, "r"); /* src/sequence_batch.c:31 */
  if (!fp) return -1;
  kseq_t *ks = kseq_init(fp); /* :36 */
  uint64_t cap_n = 0, ls = 0, cs = 0, lq = 0, cq = 0, ln = 0, cn = 0, lc = 0, cc = 0;
  int l;
  while ((l = kseq_read(ks)) >= 0) {
    if (out->n + 2 > cap_n) {
      cap_n = cap_n * 2 + 1024;
      out->seq_off = (uint64_t *)realloc(out->seq_off, cap_n * sizeof(uint64_t));
      out->name_off = (uint64_t *)realloc(out->name_off, cap_n * sizeof(uint64_t));
      out->comment_off = (uint64_t *)realloc(out->comment_off, cap_n * sizeof(uint64_t));
      out->has_qual = (uint8_t *)realloc(out->has_qual, cap_n);
    }
    out->seq_off[out->n] = ls, out->name_off[out->n] = ln, out->comment_off[out->n] = lc;
    out->has_qual[out->n] = ks->qual.l != 0; /* "fastq file", src/sequence_batch.c:57 */
    push(&out->seq, &ls, &cs, ks->seq.s, ks->seq.l);
    if (ks->qual.l == ks->seq.l) {
      push(&out->qual, &lq, &cq, ks->qual.s, ks->qual.l);
    } else {
      static const char zeros[1] = {0};
      for (size_t i = 0; i < ks->seq.l; ++i) push(&out->qual, &lq, &cq, zeros, 1);
    }
    push(&out->name, &ln, &cn, ks->name.s, ks->name.l);
    push(&out->comment, &lc, &cc, ks->comment.s, ks->comment.l);
    ++out->n;
  }
  out->last_rc = l;
  if (!out->seq_off) {
    out->seq_off = (uint64_t *)calloc(1, sizeof(uint64_t));
    out->name_off = (uint64_t *)calloc(1, sizeof(uint64_t));
    out->comment_off = (uint64_t *)calloc(1, sizeof(uint64_t));
  }
  out->seq_off[out->n] = ls, out->name_off[out->n] = ln, out->comment_off[out->n] = lc;
  kseq_destroy(ks);
  gzclose(fp); /* :40-41 */
  return 0;
}

void ref_records_free(ref_records *r) {
  free(r->seq_off), free(r->name_off), free(r->comment_off), free(r->has_qual);
  free(r->seq), free(r->qual), free(r->name), free(r->comment);
  memset(r, 0, sizeof *r);
}

/* radix_sort_mapping on n keys: keys come back sorted, tags[i] = index the record at rank i had before. */
void ref_radix_sort(uint64_t *keys, uint32_t *tags, uint32_t n) {
  ref_item *a = (ref_item *)malloc((size_t)(n ? n : 1) * sizeof(ref_item));
  for (uint32_t i = 0; i < n; ++i) a[i].key = keys[i], a[i].tag = i;
  radix_sort_refitem(a, a + n); /* src/align.c:57 */
  for (uint32_t i = 0; i < n; ++i) keys[i] = a[i].key, tags[i] = a[i].tag;
  free(a);
}
