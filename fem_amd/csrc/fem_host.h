/*
 * fem_host.h — host side of the drop-in: everything around the device hot path that the
 * reference does on the CPU and that stays on the CPU here (C ABI, libfemhost.so).
 *
 *   sequence files   FASTA/FASTQ(.gz) loading           (reference src/sequence_batch.c:30-121, src/kseq.h:185-226)
 *   index files      byte-compatible save / load        (src/index.c:100-168)
 *   mapping tail     Mapping lists -> sorted records    (src/align.c:53-92, 279-544; src/ksort.h:101-151)
 *   SAM text         header + records                   (src/output_queue.c:93-116, src/align.c:546-632)
 *   synthetic data   seeded reference / read generator  (build-owned, not in the reference)
 *
 * No mapping arithmetic of the hot path (seeding, filtering, verification) lives here: that is libfemhip.so.
 */
#ifndef FEM_HOST_H_
#define FEM_HOST_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- sequence files ---------------- */
typedef struct fem_seqfile fem_seqfile; /* an open FASTA/FASTQ(.gz) stream */

/* A set of sequences held in memory: concatenated characters + n+1 offsets (+ names, + qualities). */
typedef struct {
  uint64_t n;
  char *bases;      /* concatenated sequence characters */
  uint64_t *off;    /* n+1 */
  char *quals;      /* concatenated quality strings (same offsets), or NULL for FASTA */
  char *names;      /* concatenated names */
  uint64_t *name_off; /* n+1 */
} fem_seqset;

fem_seqfile *fem_seqfile_open(const char *path);
void fem_seqfile_close(fem_seqfile *f);
/* Reads up to max_seqs records (0 = all), skipping zero-length ones as the reference does
 * (src/sequence_batch.c:50-52,88-89).  Returns 0, or <0 on a malformed / truncated file. */
int fem_seqfile_read(fem_seqfile *f, uint64_t max_seqs, fem_seqset *out);
/* Reads whole records covering about approx_bytes of input (0 = the rest of the file).  Plain 4-line FASTQ files
 * are memory-mapped and parsed by n_threads threads; gzip, FASTA and multi-line records go through the
 * sequential reader (same records either way). */
int fem_seqfile_read_bytes(fem_seqfile *f, uint64_t approx_bytes, int n_threads, fem_seqset *out);
void fem_seqset_free(fem_seqset *s);

/* The same batch in two phases, so that the fields land where the caller wants them without an intermediate copy
 * (FEM map points `bases` / `off` at the device library's PINNED staging buffers, include/fem_hip.h
 * fem_dev_acquire_stage; the reference's reader fills a reusable SequenceBatch, src/input_queue.c:53-79).
 *   plan: delimits whole records covering about approx_bytes (0 = the rest) and sizes the batch: plain 4-line FASTQ
 *         is only scanned (all threads); anything else is parsed by the sequential kseq-rule reader and held.
 *   fill: copies characters, offsets (off[0] = 0), qualities, names into the caller's buffers (all threads) and frees
 *         the plan.  quals may be NULL.  bases needs n_bases + 64 bytes (64 zero bytes follow the last read). */
typedef struct fem_batch_plan fem_batch_plan;
typedef struct {
  uint64_t n_reads, n_bases, n_name_bytes;
  uint32_t max_len;
  int32_t has_qual; /* every record carried qualities */
  uint32_t min_len; /* == max_len: reads of one length (fem_dev_commit_stage_uniform) */
} fem_batch_shape;
int fem_seqfile_plan(fem_seqfile *f, uint64_t approx_bytes, int n_threads, fem_batch_plan **plan, fem_batch_shape *shape);
int fem_seqfile_fill(fem_seqfile *f, fem_batch_plan *plan, int n_threads, char *bases, uint64_t *off, char *quals,
                     char *names, uint64_t *name_off);
/* fill with the bases written at TWO BITS PER BASE, the form fem_dev_commit_stage_packed (include/fem_hip.h) takes: `codes`
 * is the staging buffer; the characters outside "ACGT" are listed behind the codes as that call wants them (positions at
 * the 8-byte boundary behind n_reads * ceil(read_len / 4) code bytes, then the bytes; fem_dev_packed_layout gives exc_cap,
 * how many the buffer and the format take).  For plans whose shape has min_len == max_len == read_len.  quals, names and
 * name_off as in fem_seqfile_fill (qualities of read i at quals + i * read_len).  Returns 0 (plan freed, *n_exc set), 1 =
 * the batch holds more than exc_cap such characters (nothing is lost: the plan is still there, call fem_seqfile_fill for
 * the characters), < 0 on a bad argument. */
int fem_seqfile_fill_packed(fem_seqfile *f, fem_batch_plan *plan, int n_threads, uint32_t read_len, uint8_t *codes, uint64_t exc_cap,
                            uint64_t *n_exc, char *quals, char *names, uint64_t *name_off);
/* Reads that are not copied at all: where name, bases and qualities of read r lie in the input (arrays of n entries owned
 * by the caller; reads of one length).  fem_seqfile_fill_packed_refs fills them — and packs the bases as fem_seqfile_fill_packed
 * does — for batches whose records sit in the mapping of a plain, uncompressed 4-line FASTQ file (valid until
 * fem_seqfile_close); it returns 2 and leaves the plan alone for any other batch (gzip / BGZF windows are reused, the
 * sequential reader holds copies), 1 as fem_seqfile_fill_packed.  fem_records_sam_refs (below) renders records from them. */
typedef struct {
  uint64_t n;
  uint32_t read_len;
  const char **name;
  uint32_t *name_len;
  const char **seq;
  const char **qual;
} fem_read_refs;
int fem_seqfile_fill_packed_refs(fem_seqfile *f, fem_batch_plan *plan, int n_threads, uint32_t read_len, uint8_t *codes, uint64_t exc_cap,
                                 uint64_t *n_exc, fem_read_refs *refs);
void fem_batch_plan_free(fem_batch_plan *plan);
/* 1 when the NEXT batch may be planned (by another thread) while a plan of this file is still being filled: plain 4-line
 * FASTQ files read through a mapping (a plan's records then stay where they are).  gzip / BGZF windows and the sequential
 * reader reuse their buffers: 0, plan and fill alternate. */
int fem_seqfile_plan_ahead_ok(fem_seqfile *f);

/* ---------------- index files (src/index.c:100-168) ---------------- */
/* int32 k | int32 step | uint32 lookup[4^k+1] | size_t n | uint64 occ[n] */
int fem_index_save(const char *path, int32_t k, int32_t step, const uint32_t *lookup, uint64_t n_occ, const uint64_t *occ);
/* Allocates *lookup and *occ with malloc. */
int fem_index_load(const char *path, int32_t *k, int32_t *step, uint32_t **lookup, uint64_t *n_occ, uint64_t **occ);

/* ---------------- mapping tail ---------------- */
/* What the device hands back per batch (fem_batch_result of include/fem_hip.h). */
typedef struct {
  uint64_t n_reads;
  const uint32_t *cand_begin;
  const uint32_t *cand_count;
  const uint64_t *cand;
  const uint8_t *ed;
  const int16_t *end;
} fem_tail_input;

typedef struct {
  const char *text;      /* reference characters, concatenated (raw FASTA case) */
  const uint64_t *off;
  const uint32_t *len;
  uint32_t n_seq;
  const char *names;     /* concatenated names */
  const uint64_t *name_off;
} fem_tail_ref;

/* Records in output order, as arrays (tests compare these with the oracle's). */
typedef struct {
  uint64_t n_records;
  uint64_t *rec_off;   /* n_reads+1 */
  uint16_t *flag;
  uint32_t *tid;
  uint32_t *pos0;
  uint8_t *nm;
  uint64_t *cigar_off; /* n_records+1 */
  uint32_t *cigar;     /* BAM encoding len<<4|op */
  uint64_t *md_off;    /* n_records+1 */
  char *md;
} fem_records;

/* process_mappings (src/align.c:56-92) for every read of a batch: rebuild the Mapping list in
 * verify_candidates' order, radix_sort_mapping, traceback -> CIGAR/MD, record fields. */
int fem_tail_records(int32_t e, const fem_tail_ref *ref, const char *read_bases, const uint64_t *read_off,
                     const fem_tail_input *in, int n_threads, fem_records *out);
void fem_records_free(fem_records *r);

/* The same, rendered as SAM text (one line per record, reads in batch order, unmapped reads emit
 * nothing, src/map.c:50-55).  *text is malloc'd. */
int fem_tail_sam(int32_t e, const fem_tail_ref *ref, const fem_seqset *reads, const fem_tail_input *in, int n_threads,
                 char **text, uint64_t *text_len);
/* Records already computed (the device mapping tail, fem_dev_fetch_records of include/fem_hip.h), rendered as the
 * same SAM text as fem_tail_sam.  The fields mirror fem_batch_records. */
typedef struct {
  uint64_t n_reads;
  uint64_t n_records;
  const uint32_t *rec_begin; /* n_reads+1 */
  const uint16_t *flag;
  const uint32_t *tid;
  const uint32_t *pos0;
  const uint8_t *nm;
  const uint32_t *cigar_off; /* n_records+1 */
  const uint32_t *cigar;
  const uint32_t *md_off;    /* n_records+1 */
  const char *md;
} fem_record_view;
int fem_records_sam(const fem_tail_ref *ref, const fem_seqset *reads, const fem_record_view *rec, int n_threads,
                    char **text, uint64_t *text_len);
/* The same text without the concatenation: thread t formats its share of the reads into its own stretch of *buf (a
 * caller-owned buffer, reused from batch to batch and grown here with realloc when too small), parts[t] = {offset,
 * length} of that stretch (n_threads entries; the text is the parts in order).  *n_asserted counts records on which
 * the reference would have tripped an assertion (src/align.c:366-368): written with their 0x8000 marker bit cleared
 * from FLAG and CIGAR `*`. */
typedef struct {
  uint64_t offset, length;
} fem_text_part;
int fem_records_sam_parts(const fem_tail_ref *ref, const fem_seqset *reads, const fem_record_view *rec, int n_threads,
                          char **buf, uint64_t *cap, fem_text_part *parts, uint64_t *n_asserted);
/* fem_records_sam_parts for reads the parser did not copy (fem_seqfile_fill_packed_refs): QNAME, SEQ and QUAL of a line come
 * straight out of the input file's mapping. */
int fem_records_sam_refs(const fem_tail_ref *ref, const fem_read_refs *reads, const fem_record_view *rec, int n_threads, char **buf,
                         uint64_t *cap, fem_text_part *parts, uint64_t *n_asserted);
/* "@SQ\tSN:%s\tLN:%d\n" per sequence (src/output_queue.c:104-108). *text is malloc'd. */
int fem_sam_header(const fem_tail_ref *ref, char **text, uint64_t *text_len);
/* The other half of fem_dev_commit_names_stage / fem_dev_sam_quals (include/fem_hip.h): copies read r's quality string
 * (quals + off[r], off[r + 1] - off[r] characters; off == NULL: reads of one length, quals + r * read_len) to text + qual_at[r]
 * for every read with qual_at[r] != UINT64_MAX, on up to n_threads threads (they sleep between calls).  Returns 0, or -1 on a
 * null argument or a field that would end behind text_len. */
int fem_sam_fill_quals(char *text, uint64_t text_len, const uint64_t *qual_at, uint64_t n_reads, const char *quals, const uint64_t *off,
                       uint32_t read_len, int n_threads);

/* ---------------- synthetic data ---------------- */
/* iid uniform A/C/G/T; sequence i is a pure function of (seed, i). */
void fem_synth_reference(uint64_t seed, uint32_t n_seq, const uint64_t *seq_off, const uint32_t *seq_len, char *out,
                         int n_threads);
/* Read r (global index first_read + r) is a pure function of (seed, r): uniform start, 0..e edits
 * (60% substitution / 20% insertion / 20% deletion at interior offsets), truncated to L, reverse-complemented
 * with probability 1/2.  bases_out holds n_reads*L characters. */
void fem_synth_reads(uint64_t seed, const char *ref_text, const uint64_t *seq_off, const uint32_t *seq_len,
                     uint32_t n_seq, uint64_t first_read, uint64_t n_reads, uint32_t L, int32_t e, char *bases_out,
                     int n_threads);
/* the same, and the number of edits put into each read (n_err_out may be NULL) */
void fem_synth_reads_ex(uint64_t seed, const char *ref_text, const uint64_t *seq_off, const uint32_t *seq_len,
                        uint32_t n_seq, uint64_t first_read, uint64_t n_reads, uint32_t L, int32_t e, char *bases_out,
                        uint8_t *n_err_out, int n_threads);

/* the same reads at two bits per base, ceil(L / 4) bytes per read (the form fem_dev_commit_stage_packed takes; the
 * generator draws A C G T only: no exceptions), padded with zero bytes to the next multiple of 8 */
void fem_synth_reads_packed(uint64_t seed, const char *ref_text, const uint64_t *seq_off, const uint32_t *seq_len, uint32_t n_seq,
                            uint64_t first_read, uint64_t n_reads, uint32_t L, int32_t e, uint8_t *codes, int n_threads);

/* Writes reads as FASTQ ("@r<index>", constant quality 'I') or a reference as FASTA (60 columns); for the
 * end-to-end measurements and tests.  Returns 0 or <0. */
int fem_synth_write_fastq(const char *path, const char *bases, uint32_t L, uint64_t n_reads, uint64_t first_index);
int fem_synth_write_fasta(const char *path, const char *text, const uint64_t *seq_off, const uint32_t *seq_len,
                          uint32_t n_seq);

#ifdef __cplusplus
}
#endif
#endif /* FEM_HOST_H_ */
