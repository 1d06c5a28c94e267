/*
 * fem_hip.h — C ABI of libfemhip.so: FEM's per-read mapping hot path on MI355X (gfx950).
 *
 * The reference (haowenz/FEM v0.2) has no plugin/FFI layer; the seam this
 * library replaces is the per-read call sequence inside
 * single_end_read_mapping_thread (src/map.c:27-55):
 *
 *     generate_group_seeding_candidates()   src/filter.h:9   (src/filter.c:146-223)
 *     verify_candidates()                   src/align.h:13   (src/align.c:4-51)
 *     MappingStats accumulation             src/map.c:25,32-33,37,43-44,48,51
 *
 * called once per read and strand.  Here the same work is done one BATCH at a
 * time: the index (src/index.h:7-14) and the reference text are uploaded once
 * and stay resident in HBM, reads are handed over as one contiguous byte
 * array + offsets, and the per-candidate verification outcome comes back as
 * flat arrays from which the host rebuilds the reference's Mapping lists
 * (src/utils.h:44-49) in the reference's order.
 *
 * Plain C types only; no exits or aborts: every call returns 0 (FEM_OK) or a
 * negative fem_status, and fem_dev_last_error() holds a message.
 * A handle is not thread-safe; distinct handles are independent.
 */
#ifndef FEM_HIP_H_
#define FEM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum fem_status {
  FEM_OK = 0,
  FEM_ERR_INVALID = -1,     /* bad argument (NULL, out-of-range parameter, bad slot) */
  FEM_ERR_HIP = -2,         /* a HIP runtime call failed; see fem_dev_last_error() */
  FEM_ERR_NOMEM = -3,       /* host or device allocation failed */
  FEM_ERR_STATE = -4,       /* call order violated (e.g. map before index upload) */
  FEM_ERR_UNSUPPORTED = -5, /* input outside what the device path handles (see fem_dev_limits) */
  FEM_ERR_RCCL = -6         /* an RCCL call failed */
};

/* FEMArgs (src/utils.h:63-70) minus threads/seeding_method.  `FEM map` always
 * runs k=12, step=3 (src/FEM_map.c:67-68); 0<=e<=7, 0<=a<=2 (src/FEM_map.c:30,38). */
typedef struct {
  int32_t k;
  int32_t step;
  int32_t e; /* error_threshold */
  int32_t a; /* num_additional_qgrams */
} fem_params;

/* One batch of reads: what a SequenceBatch (src/sequence_batch.h:15-24) holds
 * for the hot path — the raw read characters, concatenated, and n_reads+1 offsets. */
typedef struct {
  const char *bases;
  const uint64_t *offsets;
  uint64_t n_reads;
} fem_read_batch;

/* Outcome of one batch.  Pointers refer to pinned host memory owned by the
 * handle and stay valid until the slot is staged or mapped again.
 * Slot 2*i+d describes strand d (0 = POSITIVE_DIRECTION, 1 = NEGATIVE_DIRECTION,
 * src/utils.h:21-22) of read i: its candidates (src/filter.c:221-222, ascending,
 * already shifted by -e) are cand[cand_begin[s] .. cand_begin[s]+cand_count[s]),
 * and ed[]/end[] hold banded_edit_distance's result for each (src/align.c:102-147;
 * ed == 0xFF when the candidate is rejected, i.e. the reference would not push
 * a Mapping, src/align.c:22,40). */
typedef struct {
  uint64_t n_reads;
  uint64_t n_candidates; /* slots in cand/ed/end; includes padding slots that no (read, strand) refers to */
  const uint32_t *cand_begin; /* 2*n_reads */
  const uint32_t *cand_count; /* 2*n_reads */
  const uint64_t *cand;       /* n_candidates: seq<<32 | (pos - e) */
  const uint8_t *ed;          /* n_candidates */
  const int16_t *end;         /* n_candidates: Mapping.end_position_offset */
  /* MappingStats (src/utils.h:55-61): reads, mapped reads, candidates before the
   * additional q-gram filter, candidates, mappings */
  uint64_t stats[5];
} fem_batch_result;

/* The same outcome in the form that crosses the link (round 5): 11.4 instead of 30.8 bytes per read at BASELINE config 2.
 * The candidate slots of fem_batch_result are handed out in chunks per wavefront (padding between them, 16 bytes of
 * cand_begin / cand_count per read); here the candidates of strands [256 b, 256 b + 256) (strand s = 2 * read + direction,
 * as above) lie in strand order from seg_begin[b] on, without padding, and a strand has count[s] of them — count[s] == 255
 * means "255 or more": its real count is listed in big[] (n_big pairs: strand, count).  A caller walking the batch in read
 * order — what process_mappings does, src/align.c:56-92 — keeps one running index per segment and never needs an offset
 * per read.  Same lifetime as fem_batch_result. */
typedef struct {
  uint64_t n_reads;
  uint64_t n_candidates;     /* entries of cand/ed/end: every one belongs to a strand */
  const uint8_t *count;      /* 2*n_reads */
  const uint32_t *seg_begin; /* ceil(2*n_reads / 256) */
  const uint64_t *cand;      /* seq<<32 | (pos - e) */
  const uint8_t *ed;         /* 0xFF = rejected */
  const int16_t *end;
  const uint32_t *big;       /* 2*n_big: (strand, count) */
  uint32_t n_big;
  uint64_t stats[5];
} fem_batch_packed;

/* The records of one batch as process_mappings would hand them to the writer
 * (src/align.c:56-92): each mapped read's Mappings ordered by radix_sort_mapping
 * (src/align.c:53,66; klib semantics, src/ksort.h:101-151), then per Mapping
 * generate_alignment + generate_MD_tag (src/align.c:279-544) and the record
 * fields of generate_bam1_t (src/align.c:546-632).  Pointers refer to pinned
 * host memory owned by the handle, valid until the slot's next fetch_records. */
typedef struct {
  uint64_t n_reads;
  uint64_t n_records;
  const uint32_t *rec_begin; /* n_reads+1: records of read i are [rec_begin[i], rec_begin[i+1]); the first is the primary */
  const uint16_t *flag;      /* 16 = BAM_FREVERSE, 256 = BAM_FSECONDARY; 0x8000 = the reference would have asserted (empty CIGAR/MD) */
  const uint32_t *tid;       /* reference sequence index */
  const uint32_t *pos0;      /* 0-based leftmost position (src/align.c:80) */
  const uint8_t *nm;         /* edit distance */
  const uint32_t *cigar_off; /* n_records+1 */
  const uint32_t *cigar;     /* BAM encoding: len<<4 | op, M=0 I=1 D=2 */
  const uint32_t *md_off;    /* n_records+1 */
  const char *md;            /* MD tag characters, not NUL-terminated */
  uint64_t stats[5];
} fem_batch_records;

/* The batch's output as SAM text, rendered on the device (replaces process_mappings + generate_bam1_t + the SAM writer,
 * src/align.c:56-92,546-632, src/output_queue.c:93-116, for the batch): the lines of every mapped read, reads in batch
 * order, a read's records in the reference's order.  Pinned host memory owned by the library, valid until the slot is
 * fetched this way again. */
typedef struct {
  const char *text;
  uint64_t len;
  uint64_t n_reads, n_records;
  uint64_t n_asserted; /* records on which the reference would have tripped an assertion: written with CIGAR * */
  uint64_t stats[5];
} fem_batch_sam;

typedef struct fem_dev fem_dev;

/* Process-wide, before fem_dev_open: handles opened afterwards make their threads SLEEP while they wait for the device
 * (hipDeviceScheduleBlockingSync) instead of spinning.  For callers that keep several threads waiting on one handle while
 * other threads need the cores (FEM map: a thread per batch in flight beside the FASTQ parser); a caller with one thread
 * that waits rarely (bench.py's pipeline) is better off with the default. */
int fem_set_blocking_waits(int on);

/* ---- lifetime ---- */
int fem_dev_open(int device, fem_dev **out);
int fem_dev_close(fem_dev *h);
const char *fem_strerror(int rc);
const char *fem_dev_last_error(const fem_dev *h);
/* Limits of the device path: max read length, number of batch slots. */
int fem_dev_limits(const fem_dev *h, uint32_t *max_read_len, int32_t *n_slots);

/* ---- resident data (replaces load_index / the reference SequenceBatch;
 *      src/index.c:100-131, src/FEM_map.c:135-143) ---- */
/* lookup: 4^k+1 prefix sums; occ: seq<<32|pos, ascending in each bucket — byte
 * for byte the arrays of the index file (src/index.c:133-168). */
int fem_dev_upload_index(fem_dev *h, int32_t k, int32_t step, const uint32_t *lookup, uint64_t n_lookup,
                         const uint64_t *occ, uint64_t n_occ);
/* seq[i] points at seq_len[i] raw FASTA characters (any case; non-ACGT = N). */
int fem_dev_upload_reference(fem_dev *h, uint32_t n_seq, const char *const *seq, const uint32_t *seq_len);
/* Build the index on the device from the uploaded reference (construct_index,
 * src/index.c:57-98) and keep it resident.  If lookup_out/occ_out are non-NULL
 * the arrays are also copied back (occ_cap entries available); *n_occ_out is
 * always set.  The result is byte-identical to the reference's index arrays. */
int fem_dev_build_index(fem_dev *h, int32_t k, int32_t step, uint32_t *lookup_out, uint64_t *occ_out,
                        uint64_t occ_cap, uint64_t *n_occ_out);
/* Copies the resident index arrays (uploaded or built) to the host: what save_index
 * writes (src/index.c:133-168).  lookup_out holds 4^k+1 entries, occ_out occ_cap >= n_occ;
 * either may be NULL.  `FEM index` builds once, sizes its buffers from *n_occ_out, then fetches. */
int fem_dev_fetch_index(fem_dev *h, uint32_t *lookup_out, uint64_t *occ_out, uint64_t occ_cap);

/* ---- mapping one batch (replaces the loop body src/map.c:27-49) ---- */
/* submit = stage + map + start of the copy back; wait = finish + result.
 * Two (or more) slots let the transfer of one batch overlap the kernels of another. */
int fem_dev_map_batch_submit(fem_dev *h, int slot, const fem_params *p, const fem_read_batch *reads);
int fem_dev_map_batch_wait(fem_dev *h, int slot, fem_batch_result *out);

/* The same in separate phases. */
/* Copies the caller's batch into the slot's pinned staging buffers (several host threads), checks it
 * (offsets ascending, reads within fem_dev_limits) and starts the asynchronous copy to HBM.  Returns without
 * waiting for the copy; the caller's buffers are free again on return. */
int fem_dev_stage_reads(fem_dev *h, int slot, const fem_read_batch *reads);
/* A batch of equal-length reads crosses the link as two bits per base for the characters A C G T plus, for every other
 * byte (lower case, N, anything), its position and the byte, and is rebuilt byte for byte on the device.  Batches of
 * mixed lengths, or with more than one such byte in sixteen, go as characters + offsets.  FEM_NO_PACK=1 turns the
 * packing off.  fem_dev_stage_info: bytes the slot's last staging (either form) sent to the device, and whether they
 * were packed. */
int fem_dev_stage_info(fem_dev *h, int slot, uint64_t *h2d_bytes, int32_t *packed);
/* Zero-copy form (north_star: "reads streamed in pinned batches"; the reusable SequenceBatch ring of
 * src/input_queue.c:34-51): the library lends the slot's PINNED staging buffers, the FASTQ parser writes the
 * batch straight into them, commit starts the asynchronous H2D copy and returns at once.
 *   acquire: waits until the slot is idle, grows the buffers to n_reads_cap + 1 offsets and n_bases_cap
 *            characters (+ 64 bytes of slack the caller may overrun), returns them; valid until the next acquire.
 *            Once the slot's batch has been fetched (fem_dev_fetch*, fem_dev_sync) the same buffers may be
 *            refilled and committed again without another acquire.
 *   commit:  the batch is complete: n_reads reads, offsets[0] == 0, offsets ascending, offsets[n_reads]
 *            characters, longest read max_len (the parser knows it; no per-read host work happens here).
 *            A wrong max_len / offset table is the caller's bug: the kernels trust them. */
int fem_dev_acquire_stage(fem_dev *h, int slot, uint64_t n_reads_cap, uint64_t n_bases_cap, char **bases,
                          uint64_t **offsets);
int fem_dev_commit_stage(fem_dev *h, int slot, uint64_t n_reads, uint32_t max_len);
/* The same for a batch in which every read has exactly read_len characters (what a sequencer run usually is):
 * read i sits at bases + i * read_len, and the offset table is neither read from the staging buffer nor copied —
 * it is generated on the device (8 of the 108 bytes per 100 bp read that cross the host link otherwise). */
int fem_dev_commit_stage_uniform(fem_dev *h, int slot, uint64_t n_reads, uint32_t read_len);
/* The same for a batch the caller (a FASTQ parser, a sequencer's base caller) already holds at TWO BITS PER BASE — what the
 * device consumes; fem_dev_stage_reads produces this form from characters on the library's host threads, this entry point
 * takes it as it is: no host work per base, a quarter of the bytes over the link.  The reusable batch of
 * src/input_queue.c:34-79 in the form the GPU wants.  Layout inside the `bases` buffer lent by fem_dev_acquire_stage
 * (acquired for n_reads * read_len characters as before):
 *   codes    read i takes bytes [i * bpr, (i + 1) * bpr), bpr = ceil(read_len / 4); base j of a read sits in bits
 *            2 (j & 3) of its byte j / 4; A C G T = 0 1 2 3 (src/utils.h:72); unused bits zero;
 *   at byte exc_offset = n_reads * bpr rounded up to 8:  uint32 pos[n_exceptions], then uint8 chr[n_exceptions] —
 *            every character of the batch that is not one of the upper-case letters A C G T (N, lower case, anything):
 *            its index in the batch (i * read_len + j; its code bits are 0) and the byte itself.  The device rebuilds the
 *            batch byte for byte (the traceback compares characters, src/align.c:289-300).
 * fem_dev_packed_layout: bpr, exc_offset and the most exceptions such a batch may carry (one byte in sixteen, and what
 * the buffer has room for); a batch with more goes through fem_dev_commit_stage_uniform as characters.  Needs no handle.
 * commit_stage_packed checks the exception positions (a wrong one would write outside the batch) and nothing else. */
int fem_dev_packed_layout(uint64_t n_reads, uint32_t read_len, uint32_t *bytes_per_read, uint64_t *exc_offset, uint64_t *exc_cap);
int fem_dev_commit_stage_packed(fem_dev *h, int slot, uint64_t n_reads, uint32_t read_len, uint64_t n_exceptions);
int fem_dev_map_staged(fem_dev *h, int slot, const fem_params *p);          /* kernels only, asynchronous */
int fem_dev_sync(fem_dev *h, int slot);                                     /* wait; re-runs on scratch overflow */
int fem_dev_fetch_stats(fem_dev *h, int slot, uint64_t stats[5]);           /* sync + the five counters */
/* sync + full result to the host.  The arrays are the slot's pinned result buffers: valid until the slot is mapped
 * again (fem_dev_map_staged / fem_dev_map_batch_submit) — behind fem_dev_stage_reads the next batch's arrays are sent
 * home right behind its kernels, so that this call usually finds them there. */
int fem_dev_fetch(fem_dev *h, int slot, fem_batch_result *out);
/* The same outcome as fem_batch_packed (above): a third of the bytes over the link.  Once a slot has been fetched this way
 * its next batches are packed and sent home behind their kernels (as fem_dev_fetch's arrays are behind fem_dev_stage_reads);
 * fem_dev_fetch on the same batch still works (and switches the slot back).  FEM_ERR_UNSUPPORTED if more than 4 096 strands
 * of the batch have 255 candidates or more (-a 0 on a large reference): fem_dev_fetch takes any batch. */
int fem_dev_fetch_packed(fem_dev *h, int slot, fem_batch_packed *out);
/* sync + the mapping tail on the device (replaces process_mappings, src/map.c:50-54 -> src/align.c:56-92, up to
 * the point where the reference packs a bam1_t): sorted records with CIGAR and MD.  Independent of fem_dev_fetch. */
int fem_dev_fetch_records(fem_dev *h, int slot, fem_batch_records *out);

/* ---- SAM text on the device ----
 * upload_reference_names: the @SQ names (first token of every FASTA header), once, next to the reference.
 * acquire_text_stage / commit_text_stage: pinned staging for the batch's quality strings (same offsets as the bases)
 *   and read names (name i = names[name_off[i] .. name_off[i+1])), filled by the parser like the bases; commit after
 *   fem_dev_commit_stage* of the same batch (before or after its fem_dev_map_staged: the copies run on a stream of
 *   the slot's own, beside the batch's kernels, and only the SAM text waits for them), asynchronous.
 * fetch_sam: sync + mapping tail + text, all on the device; one D2H copy of the finished lines. */
int fem_dev_upload_reference_names(fem_dev *h, uint32_t n_seq, const char *names, const uint64_t *name_off);
int fem_dev_acquire_text_stage(fem_dev *h, int slot, uint64_t n_reads_cap, uint64_t n_bases_cap, uint64_t n_name_bytes_cap,
                               char **quals, char **names, uint64_t **name_off);
int fem_dev_commit_text_stage(fem_dev *h, int slot, uint64_t n_reads, uint64_t n_name_bytes);
/* The same without the qualities (the staging's `quals` need not be filled): they stay with the caller.  With the text from
 * the device 228 of the 473 bytes per 100-bp read on the link are the qualities going to the device and coming back unchanged;
 * this form sends none up and the SAM text comes back with the QUAL field of every read's first record sized but NOT WRITTEN.
 * fem_dev_sam_quals (once the text is home: after fem_dev_fetch_sam / fem_dev_sam_wait) gives qual_at[n_reads]: where in the
 * text read r's field starts, UINT64_MAX for a read without a record; the caller copies each read's quality string there
 * (libfemhost's fem_sam_fill_quals does it on several threads) before it uses the text.  Pinned memory of the slot, valid until
 * the slot's next SAM text. */
int fem_dev_commit_names_stage(fem_dev *h, int slot, uint64_t n_reads, uint64_t n_name_bytes);
int fem_dev_sam_quals(fem_dev *h, int slot, const uint64_t **qual_at, uint64_t *n_reads);
/* Optional: device and pinned buffers of the slot for batches of this shape and `text_bytes` of SAM text, allocated now
 * (pinning host memory costs ~0.25 ms per MB; otherwise the first batch of every slot pays for it). */
int fem_dev_reserve_text(fem_dev *h, int slot, uint64_t n_reads, uint64_t n_bases, uint64_t n_name_bytes, uint64_t text_bytes);
/* Optional, after the index is resident: everything else a batch of up to n_reads reads of up to max_len characters with up
 * to n_records mappings allocates in this slot on its way through fem_dev_map_staged and fem_dev_fetch_records /
 * fem_dev_fetch_sam (per-read and per-candidate arrays, the selection's hand-over, the mapping tail's thirty arrays) —
 * otherwise the first batch of every slot makes these allocations between its kernels (FEM map, 16 M reads: 20 of the job's
 * 145 ms).  Larger batches still grow what they need. */
int fem_dev_reserve_batch(fem_dev *h, int slot, uint64_t n_reads, uint64_t n_records, uint32_t max_len, const fem_params *p);
int fem_dev_fetch_sam(fem_dev *h, int slot, fem_batch_sam *out);
/* The same, returning as soon as the copy of the text to the host is queued: out->text must not be read before
 * fem_dev_sam_wait(h, slot) has returned — the one call that may be made from another thread than the one driving
 * the handle (a writer thread waits there while the GPU's thread starts on the next batch). */
int fem_dev_fetch_sam_nowait(fem_dev *h, int slot, fem_batch_sam *out);
int fem_dev_sam_wait(fem_dev *h, int slot);

/* Name of the seed + filter kernel fem_dev_map_staged would launch first for these parameters on the resident
 * index ("seed_join_kernel" — behind its "seed_select_kernel"; "seed_join_banked_kernel" where the reference's sequences
 * need more than one 32-bit coordinate space —, "seed_fast_kernel<hash>", "seed_fast_kernel<lean>" or
 * "seed_filter_kernel"); the generic seed_filter_kernel always follows for whatever those queue.  Static string. */
const char *fem_dev_seed_kernel(const fem_dev *h, const fem_params *p);
/* Which derived tables the resident index has (one line of text, NUL-terminated, cut at cap): "dense: 32-bit occurrence
 * table, strided with pads (8192 MiB), 1 bank, ..." / "... compact ..." / "sparse: bucket summaries" — so that a
 * measurement can say what it ran on (the strided table is taken from 16 entries per bucket on, and may be declined when
 * memory is short). */
int fem_dev_index_info(const fem_dev *h, char *buf, uint64_t cap);

/* ---- measurement ---- */
/* With timing on, every kernel launch is bracketed by HIP events on the stream
 * it is launched on; fem_dev_kernel_time reports their sum and count since
 * the last reset.  kernel: 0 = seed/filter kernel (fast form, k=12 step=3; on a dense
 * index the join kernel, whose selection kernel is 8),
 * 1 = verify kernel, 2 = seed/filter kernel (generic form: the reads the fast
 * form queued, or every read when the fast form does not apply); of
 * fem_dev_fetch_records: 3 = ordering of the mappings, 4 = traceback + MD,
 * 5 = compaction (one entry per call, several kernels each);
 * 6 = unused, always 0 (the count kernel of rounds 1-2; the counters come out of kernel 1 now); of fem_dev_fetch_sam:
 * 7 = the SAM text kernels;
 * 8 = seed selection kernel of the dense-index path (it runs beside the previous batch's kernel 0: its event time is
 * what it takes there, not what it would take alone). */
int fem_dev_set_timing(fem_dev *h, int on);
int fem_dev_reset_timing(fem_dev *h);
int fem_dev_kernel_time(fem_dev *h, int kernel, double *ms_total, uint64_t *launches);
/* Achieved device-to-device copy bandwidth in GB/s over `bytes` (roofline cross-check). */
int fem_dev_copy_bandwidth(fem_dev *h, uint64_t bytes, int iters, double *gb_per_s);
/* Achieved pinned-host-to-device bandwidth in GB/s (what bounds the read stream). */
int fem_dev_h2d_bandwidth(fem_dev *h, uint64_t bytes, int iters, double *gb_per_s);

/* ---- host placement (new; the reference leaves its threads to the scheduler, src/FEM_map.c:172-198) ---- */
/* NUMA node of the host memory GPU `device` is attached to (-1 if the system does not say) and that node's CPUs
 * as a Linux cpulist ("64-127,192-255").  Needs no handle. */
int fem_device_numa(int device, int32_t *node, char *cpulist, uint64_t cap);
/* Restricts the CALLING thread (and the threads it creates afterwards) to those CPUs, so that the pinned staging
 * buffers and the threads that fill them sit next to the GPU.  0 = bound; 1 = nothing done (unknown topology, none of
 * the node's CPUs allowed to this process, or FEM_NUMA_BIND=0).  Call it before the first fem_dev_open on a thread. */
int fem_bind_thread_near_device(int device);

/* ---- multi-GPU (replaces the thread reduction src/FEM_map.c:200-212) ---- */
/* Sums the five MappingStats counters of n handles (one per GPU of this
 * process) with one RCCL all-reduce; stats is n x 5, reduced in place.  The
 * communicator is created on the first call and kept until one of its
 * handles is closed. */
int fem_dev_allreduce_stats(fem_dev *const *h, int n, uint64_t *stats);

#ifdef __cplusplus
}
#endif
#endif /* FEM_HIP_H_ */
