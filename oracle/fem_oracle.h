/*
 * fem_oracle.h — CPU restatement of FEM's per-read mapping hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (libfemhip.so + libfemhost.so) never
 * links, loads or calls it.
 *
 * PARITY UNPINNED: the reference (haowenz/FEM v0.2) holds no tests, golden
 * vectors or fixtures, and cannot be compiled in this image (every hot-path
 * source includes the un-vendored htslib via src/utils.h:18).  This file is
 * therefore a line-by-line restatement of the reference algorithm, each
 * function citing the reference file:line it follows, cross-checked only by
 * independent brute-force models in tests/ (full-matrix edit distance,
 * closed-form candidate sets, CIGAR re-scoring).  One function is pinned
 * against reference code: fo_sort_mapping_keys equals the reference's own
 * ksort.h instantiation, built from /root/reference behind ref_klib.c
 * (tests/test_ref_klib.py).
 */
#ifndef FEM_ORACLE_H_
#define FEM_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FEMArgs (src/utils.h:63-70).  `map` always runs k=12, step=3
 * (src/FEM_map.c:67-68); 0<=e<=7, 0<=a<=2 (src/FEM_map.c:30,38). */
typedef struct {
  int32_t k;
  int32_t step;
  int32_t e;
  int32_t a;
} fo_params;

/* Reference text: all sequences concatenated, raw FASTA characters
 * (SequenceBatch of kseq_t, src/sequence_batch.h:45-55). */
typedef struct {
  const char *text;
  const uint64_t *off; /* n_seq start offsets into text */
  const uint32_t *len; /* n_seq lengths */
  uint32_t n_seq;
} fo_ref;

/* Index (src/index.h:7-14): CSR lookup + occurrence table. */
typedef struct {
  int32_t k;
  int32_t step;
  const uint32_t *lookup; /* 4^k + 1 prefix sums */
  uint64_t n_occ;
  const uint64_t *occ; /* seq<<32 | pos, ascending inside each bucket */
} fo_index;

/* One batch of reads: concatenated raw characters + n+1 offsets. */
typedef struct {
  const char *bases;
  const uint64_t *off;
  uint64_t n;
} fo_reads;

/* ---- index (src/index.c:57-98) ---- */
/* Number of index entries construct_index would emit. */
uint64_t fo_index_count(const fo_ref *ref, int k, int step);
/* Fills lookup[4^k+1] and occ[fo_index_count()]. Returns 0 on success. */
int fo_index_build(const fo_ref *ref, int k, int step, uint32_t *lookup, uint64_t *occ);
/* The same arrays, built by n_threads threads (each owns a range of buckets): only to make the checker usable on
 * BASELINE-sized references; verified against fo_index_build. */
int fo_index_build_mt(const fo_ref *ref, int k, int step, uint32_t *lookup, uint64_t *occ, int n_threads);
/* File format of save_index/load_index (src/index.c:100-168). */
int fo_index_save(const char *path, int k, int step, const uint32_t *lookup, uint64_t n_occ, const uint64_t *occ);
/* Reads header + sizes; call twice (first with lookup/occ NULL to get n_occ). */
int fo_index_load(const char *path, int *k, int *step, uint32_t *lookup, uint64_t *n_occ, uint64_t *occ, uint64_t occ_capacity);

/* ---- single-function entry points (unit-test granularity) ---- */
uint32_t fo_hash_seed(uint64_t pos, int k, const char *seq, uint64_t len);
/* prepare_negative_sequence_at (src/sequence_batch.h:90-98) */
void fo_revcomp(const char *seq, uint32_t len, char *out);
/* generate_group_seeding_candidates (src/filter.c:146-223) for one strand's
 * character sequence.  cands must hold cap entries; returns the number of
 * candidates, or UINT32_MAX if cap was too small (call again with more). */
uint32_t fo_seed_candidates(const fo_params *p, const char *seq, uint32_t len, const fo_ref *ref, const fo_index *idx,
                            uint64_t *cands, uint32_t cap, uint32_t *pre_filter);
/* banded_edit_distance (src/align.c:102-147) */
int fo_banded_ed32(int e, const char *pattern, const char *text, int len, int *end);
/* vectorized_banded_edit_distance (src/align.c:149-277), 8 lanes of int16 */
void fo_banded_ed16x8(int e, const char *const pattern[8], const char *text, int len, int16_t ed[8], int16_t end[8]);
/* generate_alignment + generate_MD_tag (src/align.c:279-544).  cigar gets BAM
 * encoded ops (len<<4|op), md a NUL-terminated string.  Returns the start
 * offset inside pattern, or <0 if the reference would have hit an assert. */
int fo_align(int e, const char *pattern, const char *text, int len, int ed, int end, uint32_t *cigar, int cigar_cap,
             int *n_cigar, char *md, int md_cap);
/* radix_sort_mapping (src/align.c:53-57, src/ksort.h:101-151) on parallel
 * key array; perm[i] = original index of the record that ends at rank i. */
void fo_sort_mapping_keys(uint64_t *keys, uint32_t *perm, uint32_t n);

/* ---- batch driver: single_end_read_mapping_thread (src/map.c:17-58) ---- */
typedef struct fo_result fo_result;

enum { FO_STAGE_SEED = 1, FO_STAGE_VERIFY = 2, FO_STAGE_ALIGN = 4 };

fo_result *fo_map(const fo_params *p, const fo_ref *ref, const fo_index *idx, const fo_reads *reads, int n_threads,
                  int stages);
void fo_result_free(fo_result *r);

/* MappingStats (src/utils.h:55-61): reads, mapped reads, candidates before
 * the additional q-gram filter, candidates, mappings. */
void fo_result_stats(const fo_result *r, uint64_t out[5]);

/* Candidates per (read, strand): cand_off has 2*n+1 entries, slot 2*i is the
 * + strand of read i, slot 2*i+1 the - strand. pre has 2*n entries. */
uint64_t fo_result_candidates(const fo_result *r, const uint64_t **cand_off, const uint64_t **cands,
                              const uint32_t **pre);
/* Verification outcome aligned with cands: ed (0xFF = rejected), end offset. */
void fo_result_verify(const fo_result *r, const uint8_t **ed, const int16_t **end);
/* Mappings in the order verify_candidates appends them (src/align.c:21-49). */
uint64_t fo_result_mappings(const fo_result *r, const uint64_t **map_off /* n+1 */, const uint8_t **dir,
                            const uint8_t **ed, const uint64_t **cand, const int16_t **end);
/* Records after process_mappings (src/align.c:56-92), in output order; record
 * j of read i is rec_off[i]+j.  cigar_off / md_off have n_rec+1 entries. */
uint64_t fo_result_records(const fo_result *r, const uint64_t **rec_off /* n+1 */, const uint16_t **flag,
                           const uint32_t **tid, const uint32_t **pos0, const uint8_t **nm,
                           const uint64_t **cigar_off, const uint32_t **cigar, const uint64_t **md_off,
                           const char **md);

#ifdef __cplusplus
}
#endif
#endif /* FEM_ORACLE_H_ */
