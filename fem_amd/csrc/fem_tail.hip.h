// fem_tail.hip.h — the mapping tail on the device: what process_mappings does for every mapped read
// (reference src/align.c:53-92) — order the read's Mappings with radix_sort_mapping, then for each one
// generate_alignment (src/align.c:279-499: ungapped shortcut, or the Myers recurrence re-run with D0/HP kept per
// column and the traceback with its 'S' pseudo-run) and generate_MD_tag (src/align.c:501-544).
// Input: the per-candidate verification outcome the mapping kernels left in HBM.  Output: records in the reference's
// order (FLAG, reference id, POS, NM, BAM-encoded CIGAR, MD), compacted on the device, copied to pinned host memory.
#pragma once
#include <mutex>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace femt {

struct TailInput {  // device pointers unless noted
  const uint8_t *bases;        // raw read characters, concatenated
  const uint64_t *read_off;    // n_reads + 1
  uint32_t n_reads;
  uint32_t max_len;            // host: longest read of the batch
  const uint8_t *ref_raw;      // raw reference characters (case kept: the traceback compares characters, src/align.c:355)
  uint64_t ref_bytes;          // host: bytes in ref_raw, its slack included
  // the batch as it arrived packed (fem_dev_stage_reads on equal-length reads): two bits per base, four bases per byte,
  // packed_bpr bytes per read; bit r of exc_bits: read r has a character that is not one of ACGT (kept in `bases` only).
  // nullptr: the batch came as characters.
  const uint8_t *packed;
  uint32_t packed_bpr;
  const uint32_t *exc_bits;
  const uint8_t *planes;       // bit planes over the reference (femk::plane_window): code bits 0..2, and "character is none of ACGTN"
  const uint64_t *seq_off;
  const uint64_t *cand;        // per candidate slot
  const uint8_t *ed;           // 0xFF = rejected
  const int16_t *end;
  const uint32_t *cand_begin;  // 2 * n_reads
  const uint32_t *cand_count;
  const uint32_t *n_map;       // n_reads: accepted candidates of each read
  int32_t e;
  uint64_t n_records;          // host: total accepted candidates (the "number of mapping" counter)
};

struct TailOutput {  // pinned host memory owned by the Tail object, valid until its next run
  uint64_t n_reads, n_records;
  const uint32_t *rec_begin;  // n_reads + 1: records of read i are [rec_begin[i], rec_begin[i+1]), primary first
  const uint16_t *flag;       // 16 = reverse strand, 256 = secondary; 0x8000 = the reference would have asserted
  const uint32_t *tid;
  const uint32_t *pos0;       // 0-based leftmost reference position
  const uint8_t *nm;
  const uint32_t *cigar_off;  // n_records + 1
  const uint32_t *cigar;      // BAM encoding len << 4 | op (M 0, I 1, D 2)
  const uint32_t *md_off;     // n_records + 1
  const char *md;
};

// ---- SAM text on the device: the records of the last run() rendered as the reference's output lines
//      (generate_bam1_t + htslib's SAM writer, src/align.c:546-632, src/output_queue.c:93-116) ----
struct SamInput {  // device pointers
  const uint8_t *quals;          // quality characters, same offsets as the bases
  const uint8_t *names;          // read names, concatenated
  const uint64_t *name_off;      // n_reads + 1
  const uint8_t *ref_names;      // reference sequence names, concatenated
  const uint32_t *ref_name_off;  // n_seq + 1
  bool qual_hole;                // quals == nullptr and the QUAL field of a primary record is LEFT UNWRITTEN (the caller fills it: SamOutput::qual_at)
};
struct SamOutput {  // pinned host memory owned by the Tail object, valid until its next sam()
  const char *text;
  uint64_t len;
  uint64_t n_asserted;  // records on which the reference would have tripped an assertion (written with CIGAR *)
  const uint64_t *qual_at;  // qual_hole: per READ, where in `text` its QUAL field starts (~0: the read has no record); else nullptr
};

// One text on its way home at a time (per GPU).  Two device-to-host copies queued in the copy engines take both of them, and
// the next batch's reads wait for their copy in until every queued text is home (scratch/sdma_probe.hip: a 30 MB copy in
// behind ONE 300 MB copy out is done after 0.6 ms, behind four after all four, 22 ms); FEM map's device then alternated
// between four batches' kernels and four batches' texts going home.  The thread that queues a text's copy waits here for
// the previous text to have arrived.
struct TextGate {
  std::mutex mu;
  hipEvent_t last = nullptr;  // the latest text's arrival (an event of some slot's Tail; never destroyed before the handle's tails are)
};

class Tail {
 public:
  Tail() = default;
  ~Tail();
  Tail(const Tail &) = delete;
  Tail &operator=(const Tail &) = delete;
  // Runs on `stream` and waits for it.  ms[0..2] (optional) receive the device time of: ordering, traceback, compaction.
  // tiny = test hook: per-record CIGAR/MD staging starts far too small so that the overflow pass runs.
  // copy_records = false: the records stay on the device (for sam()); *out then only carries the counts.
  int run(const TailInput &in, hipStream_t stream, int n_cu, bool tiny, TailOutput *out, std::string *err, double *ms,
          bool copy_records = true);
  // Makes room for `bytes` of SAM text ahead of time (pinning host memory costs ~0.25 ms per MB: better spent before
  // the first batch than inside it).
  int reserve_text(uint64_t bytes, std::string *err);
  // ... and for everything else run() and sam() allocate for a batch of n_reads reads with n_records records.
  int reserve(uint32_t n_reads, uint32_t n_records, uint32_t max_len, int e, bool tiny, std::string *err);
  // ... and loads the kernels (a code object is loaded by its first launch otherwise) and wakes `stream`.
  int warm(hipStream_t stream, std::string *err);
  // The records of the last run() as SAM lines, in record order.  ms (optional) receives the device time.
  // wait = false: returns once the copy of the text to the host has been queued; wait_text() (which may be called from
  // another thread) returns when it has arrived.
  int sam(const TailInput &in, const SamInput &names, hipStream_t stream, int n_cu, SamOutput *out, std::string *err, double *ms,
          bool wait = true, TextGate *gate = nullptr);
  int wait_text();

 private:
  struct Impl;
  Impl *impl_ = nullptr;
};

}  // namespace femt
