// fem_hip.hip — host side of libfemhip.so: the C ABI declared in include/fem_hip.h.
// Owns device memory, streams, pinned result buffers and the launch logic of the
// kernels in fem_kernels.hip.h.  No CPU fallback: every entry point needs a GPU.
#include "../../include/fem_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fem_index_build.hip.h"
#include "fem_pack.h"
#include "fem_kernels.hip.h"
#include "fem_seed_fast.hip.h"
#include "fem_seed_join.hip.h"
#include "fem_tail.hip.h"

namespace {

#ifndef FEM_SLOTS
#define FEM_SLOTS 4
#endif
constexpr int kSlots = FEM_SLOTS;
constexpr uint32_t kMaxReadLen = 1024;
constexpr size_t kFrontPad = 16;  // kernels fetch a reverse-strand chunk from up to 15 bytes in front of a read
constexpr uint32_t kXcapSmall = 512, kFcap = 128, kCcap = 128;

constexpr int kTimedKernels = 10;  // fem_dev_kernel_time ids: 0 seed (join), 1 verify, 2 generic seed, 3-5 tail, 6 pack_results_kernel (round 5; the count kernel of rounds 1-2 before), 7 SAM text, 8 seed selection
struct TimedLaunch {
  int kernel;
  hipEvent_t start, stop;
  bool counts = true;  // false: a further part of a launch that is already counted (a batch mapped in parts, launch_batch)
};
constexpr int kMaxParts = 4;

struct Slot {
  hipStream_t stream = nullptr;
  // inputs
  uint8_t *d_bases_alloc = nullptr;  // kFrontPad bytes of padding, then the batch's characters, then 64 bytes of slack
  uint8_t *bases() const { return d_bases_alloc + 16; }
  size_t bases_cap = 0;
  uint64_t *d_off = nullptr;
  size_t off_cap = 0;
  uint64_t n_reads = 0;
  uint64_t n_bases = 0;
  uint32_t max_len = 0;
  bool staged = false;
  // pinned host staging lent to the parser (fem_dev_acquire_stage)
  char *h_bases = nullptr;
  size_t h_bases_cap = 0;
  uint64_t *h_off = nullptr;
  size_t h_off_cap = 0;
  uint64_t acq_reads = 0, acq_bases = 0;  // what the last fem_dev_acquire_stage asked for
  // qualities and names for the device SAM text (fem_dev_acquire_text_stage / fem_dev_fetch_sam): pinned staging + copies in HBM
  char *h_quals = nullptr, *h_names = nullptr;
  uint64_t *h_name_off = nullptr;
  size_t h_quals_cap = 0, h_names_cap = 0, h_name_off_cap = 0;
  uint8_t *d_quals = nullptr, *d_names = nullptr;
  uint64_t *d_name_off = nullptr;
  size_t d_quals_cap = 0, d_names_cap = 0, d_name_off_cap = 0;
  bool text_staged = false;
  bool host_quals = false;             // the batch's qualities stayed on the host (fem_dev_commit_names_stage): the text leaves their field open
  const uint64_t *qual_at = nullptr;   // ... and where each read's field starts in the slot's last text (pinned, the tail's)
  hipStream_t text_stream = nullptr;   // qualities and names go to the device beside the batch's kernels, not in front of them
  hipStream_t out_stream = nullptr;    // the batch's way out (mapping tail, SAM text, its copy home): HIGH priority, see out_stream_of
  hipEvent_t ev_text_staged = nullptr; // ... and have arrived (the SAM text's kernels wait for it)
  hipEvent_t ev_text_order = nullptr;  // the slot's last SAM text has been rendered (its kernels read the same arrays)
  bool have_text_order = false;
  uint8_t *d_packed = nullptr;            // packed transfer (fem_dev_stage_reads): 2-bit codes + positions of other characters
  size_t packed_cap = 0;
  uint32_t *d_exc_bits = nullptr;         // ... bit r: read r has such a character (the device tail reads the others' bases from d_packed)
  size_t exc_bits_cap = 0;
  uint32_t packed_bpr = 0;
  // fem_dev_fetch callers get the result arrays sent home behind the kernels, without the host waiting for the batch first:
  // the per-read arrays whole, the per-candidate arrays up to what the slot's previous batch needed (the rest at fetch)
  bool prefetch_results = false, staged_by_copy = false;
  uint64_t prefetched_reads2 = 0, prefetched_cand = 0, last_n_cand = 0;
  uint64_t h2d_bytes = 0;                 // what the last staging sent over the link
  bool sent_packed = false;
  // A batch committed to an IDLE device is sent and mapped in `parts` pieces (reads [part_begin[q], part_begin[q + 1])), so that
  // the join of its first piece starts after a quarter of the copy and a quarter of the selection (launch_batch): the fill of
  // the pipeline.  ev_part[q]: piece q's characters are in HBM; ev_sel[q]: its selection is done.
  int parts = 1;
  uint32_t part_begin[kMaxParts + 1] = {0, 0, 0, 0, 0};
  hipEvent_t ev_part[kMaxParts] = {nullptr, nullptr, nullptr, nullptr}, ev_sel[kMaxParts] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_zeroed = nullptr;
  hipEvent_t ev_results = nullptr, ev_home = nullptr;  // the batch's results are final on the device / have arrived on the host
  // outputs on the device
  uint64_t *d_cand = nullptr;
  uint32_t *d_meta = nullptr;
  uint8_t *d_ed = nullptr;
  int16_t *d_end = nullptr;
  uint32_t cand_cap = 0;
  uint32_t *d_begin = nullptr, *d_count = nullptr, *d_nmap = nullptr;
  size_t per_read_cap = 0;
  // counters: ctr[4] (u32) | arena_ctr[2] (u64) | stats[4] (u64)
  uint8_t *d_ctl = nullptr;
  uint8_t *h_ctl = nullptr;  // pinned mirror
  uint64_t *d_arena = nullptr;
  uint64_t arena_cap = 0;
  uint32_t *d_slow = nullptr;  // reads the fast seed kernel leaves to the generic one
  uint32_t slow_cap = 0;
  // dense indexes: what seed_select_kernel hands seed_join_kernel (6 R selected seeds + one header per read)
  uint2 *d_sel = nullptr, *d_sel_hdr = nullptr;
  size_t sel_cap = 0, sel_hdr_cap = 0;
  // pinned host results
  uint32_t *h_begin = nullptr, *h_count = nullptr;
  size_t h_per_read_cap = 0;
  uint64_t *h_cand = nullptr;
  uint8_t *h_ed = nullptr;
  int16_t *h_end = nullptr;
  size_t h_cand_cap = 0;
  // the outcome in the form that crosses the link (fem_dev_fetch_packed; pack_results_kernel): on the device, pinned on the host
  uint8_t *d_count8 = nullptr, *d_ped = nullptr;
  uint32_t *d_seg = nullptr;
  uint64_t *d_pcand = nullptr;
  int16_t *d_pend = nullptr;
  uint2 *d_big = nullptr;
  size_t count8_cap = 0, seg_cap = 0, pcand_cap = 0;
  uint8_t *h_count8 = nullptr, *h_ped = nullptr;
  uint32_t *h_seg = nullptr;
  uint64_t *h_pcand = nullptr;
  int16_t *h_pend = nullptr;
  uint2 *h_big = nullptr;
  size_t h_count8_cap = 0, h_seg_cap = 0, h_pcand_cap = 0;
  bool want_packed = false;    // the slot's last fetch was fem_dev_fetch_packed: launch_batch packs and sends home behind the kernels
  bool packed_enqueued = false;  // pack_results_kernel ran (or is queued) for the slot's current batch
  uint64_t packed_home = 0, last_n_packed = 0;  // packed candidates copied home behind the kernels / of the slot's previous batch
  bool packed_per_read_home = false;
  uint32_t n_packed = 0, n_big = 0;
  // state of the last launch
  fem_params params{};
  bool mapped = false, synced = false;
  uint64_t stats[5] = {0, 0, 0, 0, 0};
  uint32_t n_cand = 0;
  std::vector<TimedLaunch> pending;
  femt::Tail *tail = nullptr;  // device mapping tail (fem_dev_fetch_records), created on first use
};

// ctr[4] | arena_ctr[2] | stats[4] | pack cursor[2]
constexpr size_t kCtlPackCursor = 4 * sizeof(uint32_t) + 2 * sizeof(uint64_t) + 4 * sizeof(uint64_t);  // pack_results_kernel: packed candidates, big[] entries
constexpr size_t kCtlBytes = kCtlPackCursor + 2 * sizeof(uint32_t);
constexpr uint32_t kBigCap = 4096;  // strands with 255 candidates and more that a packed result lists (beyond: fetch the plain form)
// ... and, in a cache line of its own behind them, the work cursor of seed_fast_kernel
constexpr size_t kCtlWorkCursor = 128, kCtlWorkCursor2 = 192, kCtlPartStride = 128, kCtlAlloc = 1024;  // (seed_select_kernel / seed_join_kernel; one pair per part)
static_assert(kCtlWorkCursor2 + (kMaxParts - 1) * kCtlPartStride + 64 <= kCtlAlloc, "control block layout");
static_assert(kCtlBytes <= kCtlWorkCursor, "control block layout");

}  // namespace

namespace {
// The host threads of fem_dev_stage_reads: created once per handle and parked between batches (a batch every few ms:
// starting and joining a dozen threads per batch cost more than the packing gained, and under a cgroup CPU quota the
// short-lived threads, scattered over the machine's CPUs, had the whole process throttled now and then).
class StagePool {
 public:
  ~StagePool() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
  }
  // work(t) for t in [0, n): t = 0 on the calling thread, the rest on the pool
  void run(unsigned n, const std::function<void(unsigned)> &work) {
    if (n <= 1) return work(0u);
    {
      std::lock_guard<std::mutex> l(m_);
      while (threads_.size() + 1 < n) {
        const unsigned id = (unsigned)threads_.size() + 1;
        threads_.emplace_back([this, id] { loop(id); });
      }
      work_ = &work, n_ = n, pending_ = n - 1, ++epoch_;
    }
    cv_.notify_all();
    work(0u);
    std::unique_lock<std::mutex> l(m_);
    done_.wait(l, [&] { return pending_ == 0; });
    work_ = nullptr;
  }

 private:
  void loop(unsigned id) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(unsigned)> *w = nullptr;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || (epoch_ != seen && id < n_); });
        if (stop_) return;
        seen = epoch_, w = work_;
      }
      (*w)(id);
      std::lock_guard<std::mutex> l(m_);
      if (--pending_ == 0) done_.notify_one();
    }
  }
  std::mutex m_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> threads_;
  const std::function<void(unsigned)> *work_ = nullptr;
  unsigned n_ = 0, pending_ = 0;
  uint64_t epoch_ = 0;
  bool stop_ = false;
};
}  // namespace

struct fem_dev {
  // Calls on DIFFERENT slots of one handle may come from different threads (round 5: `FEM map` retires its batches in flight on
  // a thread each, so that one batch's host round trips — records counted, text sized — do not hold up the next one's): what
  // the slots share (the kernels' chaining events, the event pool, the timing sums, the error string) is touched under this
  // lock; a thread waiting for the device does not hold it.
  femt::TextGate text_gate;  // one SAM text on its way to the host at a time (fem_tail.hip.h)
  std::recursive_mutex mu;
  int device = 0;
  StagePool *stage_pool = nullptr;  // host threads of fem_dev_stage_reads
  uint8_t *d_ref_names = nullptr;   // reference sequence names for the device SAM text
  uint32_t *d_ref_name_off = nullptr;
  int n_cu = 256;
  std::string err;
  // index
  uint32_t *d_lookup = nullptr;
  uint64_t n_lookup = 0;
  uint64_t *d_occ = nullptr;
  uint64_t n_occ = 0;
  int32_t k = 0, step = 0;
  uint64_t fast_occ_key = ~0ull;  // seed_fast_kernel residency, cached per (R, form, LDS bytes)
  int fast_occ_blocks = 0;
  uint32_t *d_summary = nullptr;  // bucket summaries (femk::SeedParams::summary), built for sparse indexes only
  // dense indexes: occurrence table in 32-bit global coordinates + its sequence tables (fem_seed_dense.hip.h)
  uint32_t *d_occ32 = nullptr, *d_goff = nullptr, *d_blkseq = nullptr;
  uint32_t list_shift = 0;        // != 0: d_occ32 is the strided table (bucket h at h << list_shift, fem_seed_dense.hip.h); 0: compact
  bool no_strided = false;        // FEM_NO_STRIDED=1: keep the compact 32-bit table (test hook / A-B)
  uint32_t *d_freq11 = nullptr;  // saturated byte frequencies per 11-mer (fem_seed_select.hip.h), 64 MiB
  // banks of sequences, each with 32-bit coordinates of its own (fem_seed_dense.hip.h); 1 = the whole reference in one
  uint32_t n_banks = 1, bank_first[5] = {0, 0, 0, 0, 0};
  uint32_t *d_bank_lo = nullptr;  // [n_banks - 1][n_buckets]: where each further bank's part of a bucket's list starts
  uint64_t bank_limit = 0;        // FEM_TEST_BANK_BASES: coordinates per bank (tests: banks on small references); 0 = kDenseLimit
  uint32_t bank_seqs = 0;         // FEM_TEST_BANK_SEQS: sequences per bank (tests); 0 = kDenseMaxSeq
  int select_occ_blocks = 0, join_occ_blocks = 0;
  uint64_t select_occ_key = ~0ull, join_occ_key = ~0ull;
  // reference
  // bit q of the codes, one bit per base (verify_kernel's windows); [3]: the uploaded character is not one of "ACGTN"
  uint8_t *d_planes = nullptr;  // femk::plane_window
  uint8_t *d_ref_raw = nullptr;  // the characters as uploaded (the traceback and MD compare and print them)
  uint64_t ref_bytes = 0;
  uint64_t *d_seq_off = nullptr;
  uint32_t *d_seq_len = nullptr;
  uint32_t n_seq = 0;
  std::vector<uint64_t> seq_off;
  std::vector<uint32_t> seq_len;
  Slot slot[kSlots];
  bool timing = false;
  int verify_blocks_per_cu = 0;  // resident 256-thread blocks of verify_kernel per CU (queried once)
  double t_ms[kTimedKernels] = {};
  uint64_t t_n[kTimedKernels] = {};
  bool force_generic = false;  // FEM_FORCE_GENERIC=1: skip the fast seed kernel (test hook)
  bool force_hash = false;     // FEM_FORCE_HASH=1: always use the hash-join form of the fast kernel (test hook)
  bool force_dense = false;    // FEM_FORCE_DENSE=1: build the 32-bit tables and run seed_select_kernel + seed_join_kernel whatever the index density (test hook)
  bool no_dense = false;       // FEM_NO_DENSE=1: never take the dense-index path (measurement / A-B hook)
  bool tiny_buffers = false;   // FEM_TEST_TINY_BUFFERS=1: start every scratch buffer tiny so the grow + re-run paths run (test hook)
  std::vector<hipEvent_t> event_pool;
  // The slots' streams overlap copies with kernels, but the kernels of different batches run one after the other
  // (each batch's first kernel waits for the previous batch's last): two batches' seed kernels side by side only
  // evict each other's index lines, and per-kernel event times stay those of a kernel that has the chip to itself.
  hipEvent_t ev_kernels_done = nullptr;
  bool have_kernels_done = false;
  // Dense indexes: seed_select_kernel of batch i + 1 runs BESIDE batch i's seed_join_kernel (it is bound by the rate of
  // table sectors the fabric delivers and needs four waves per CU for that; the join is bound by instruction issue):
  // selections are chained among themselves, and a batch's join waits for its own selection and the previous batch's
  // kernels.  FEM_NO_OVERLAP=1: one after the other (measurement hook).
  hipEvent_t ev_select_done = nullptr;
  bool have_select_done = false;
  bool no_overlap = false;
  // The batches' H2D copies go one after the other (each waits for the previous one's): four batches committed at once —
  // the start of every job — used to share the link, and the first one's kernels started when all four had arrived.
  hipEvent_t ev_h2d_done = nullptr;
  bool have_h2d_done = false;
  hipStream_t side_stream = nullptr;  // the selections of a batch mapped in parts (beside the joins on the slot's stream)
  bool no_parts = false;              // FEM_NO_PARTS=1 (measurement hook)
  int max_parts = 2;                  // parts of a batch that meets an idle device (FEM_PARTS, measurement hook): two halves —
                                      // every further launch of the join costs ~0.35 ms of ramp and tail, what a finer first part saves
  // FEM_TIMELINE=1 (with FEM_TESTING=1 and timing on): every timed launch and copy is printed with its start and end in ms since
  // the last fem_dev_reset_timing — the pipeline's fill and drain made visible (ids >= kTimedKernels: 20 H2D + unpack, 21 D2H)
  bool timeline = false, have_epoch = false;
  hipEvent_t ev_epoch = nullptr;
};

namespace {

// Test / measurement switches of the library are read only when FEM_TESTING=1 (tests/conftest.py sets it): a production
// process does not steer kernels through its environment.
bool testing_switch(const char *name) {
  const char *t = getenv("FEM_TESTING");  // (read every time: a process may open a plain handle first and a test handle later)
  if (!(t && t[0] == '1')) return false;
  const char *v = getenv(name);
  return v && v[0] == '1';
}

int fail(fem_dev *h, int rc, const std::string &msg) {
  if (h) {
    std::lock_guard<std::recursive_mutex> lock(h->mu);
    h->err = msg;
  }
  return rc;
}
#define FEM_LOCK(h) std::lock_guard<std::recursive_mutex> fem_lock_(h->mu)

#define HIP_TRY(h, expr)                                                                          \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(h, e_ == hipErrorOutOfMemory ? FEM_ERR_NOMEM : FEM_ERR_HIP,                     \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                             \
  } while (0)

template <typename T>
int dev_realloc(fem_dev *h, T **p, size_t *cap, size_t want, bool keep = false) {
  if (want <= *cap && *p) return FEM_OK;
  size_t n = std::max(want, *cap + *cap / 2);
  T *q = nullptr;
  HIP_TRY(h, hipMalloc((void **)&q, std::max<size_t>(n, 1) * sizeof(T)));
  if (keep && *p && *cap) HIP_TRY(h, hipMemcpy(q, *p, *cap * sizeof(T), hipMemcpyDeviceToDevice));
  if (*p) (void)hipFree(*p);
  *p = q;
  *cap = n;
  return FEM_OK;
}

template <typename T>
int pinned_realloc(fem_dev *h, T **p, size_t *cap, size_t want) {
  if (want <= *cap && *p) return FEM_OK;
  size_t n = std::max(want, *cap + *cap / 2);
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  HIP_TRY(h, hipHostMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault));
  *cap = n;
  return FEM_OK;
}

bool params_ok(const fem_params *p) {
  return p && p->k >= 1 && p->k <= 16 && p->step >= 1 && p->step <= 16 && p->e >= 0 && p->e <= 7 && p->a >= 0 &&
         p->a <= 2;
}

// LDS carve-up of one wave for reads up to max_len (see femk::SeedLayout)
femk::SeedLayout make_layout(const fem_params &p, uint32_t max_len) {
  femk::SeedLayout l{};
  const uint32_t R = (uint32_t)(p.e + 1 + p.a);
  const uint32_t lg = (uint32_t)(p.k / p.step + (p.k % p.step ? 1 : 0));
  const uint32_t n_groups = 2u * (uint32_t)p.step;
  uint32_t smax = max_len >= (uint32_t)p.k ? max_len - (uint32_t)p.k + 1u : 1u;
  int c = (int)(smax / (uint32_t)p.step) - (int)(R * lg) + 2;
  uint32_t cmax = (uint32_t)std::max(c, 2);
  l.smax = smax;
  l.cmax = cmax;
  l.cw = (cmax + 31u) / 32u;
  l.n_words = (max_len + 15u) / 16u + 2u;
  l.xcap = kXcapSmall, l.fcap = kFcap, l.ccap = kCcap;
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) {
    uint32_t at = o;
    o += (bytes + 15u) & ~15u;
    return at;
  };
  l.pkw = take(l.n_words * 4u);
  l.nkw = take(l.n_words * 4u);
  l.sf = take(2u * smax * 8u);
  l.dp_rows = take(n_groups * 2u * cmax * 4u);
  l.dp_bits = take(std::max(n_groups * R * l.cw * 4u, n_groups * R * 8u));  // serial form: bit rows; DPP form: one ballot per row
  l.picked = take(n_groups * R * 16u);
  l.rb = take(2u * (R + 1u) * 4u);
  l.X = take(l.xcap * 8u);
  l.F = take(l.fcap * 8u);
  l.A = take(l.ccap * 8u);
  l.B = take(l.ccap * 8u);
  l.wave_bytes = o;
  return l;
}

// LDS of one wave of seed_fast_kernel: packed bases, (lookup, frequency) per seed, DP take bits, selected seeds, scatter
femk::SeedLayout make_layout_fast(const fem_params &p, uint32_t max_len, bool hash) {
  femk::SeedLayout l{};
  const uint32_t R = (uint32_t)(p.e + 1 + p.a);
  const uint32_t n_groups = 2u * (uint32_t)p.step;
  l.smax = max_len >= (uint32_t)p.k ? max_len - (uint32_t)p.k + 1u : 1u;
  l.n_words = (max_len + 15u) / 16u + 2u;
  l.xcap = 64;
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) {
    uint32_t at = o;
    o += (bytes + 15u) & ~15u;
    return at;
  };
  if (hash) {  // (the lean form encodes a block of reads at once: strm)
    l.pkw = take(l.n_words * 4u);
    l.nkw = take(l.n_words * 4u);
  }
  // (hash, frequency) of every seed of both strands, reused for the strands' candidates once the seeds are selected
  // — hash-join form only: the lean form looks its seeds up straight into the group queue
  if (hash) l.sf = take(std::max(2u * l.smax * 8u, 2u * 64u * 8u));
  l.dp_bits = take(std::max(n_groups, femk::kGroupQueue) * R * 8u);  // one ballot per (pass, row); up to kGroupQueue passes
  if (hash) {
    l.X = take(64u * 8u);                 // scatter
    l.A = take(femk::kListScratchBytes);  // lists_in_lanes
  }  // (lean form: no per-read list phase; flush_small lays its scratch over the group queue, dead by then)
  l.B = take(2u * femk::kReadBlock * 8u);  // the block's begin/count entries
  if (!hash) {
    // One block of reads (kReadBlock consecutive ones, contiguous in the batch) is staged and encoded at once; blocks
    // longer than this many characters (reads over 256 bases among them) go to the generic kernel.
    const uint32_t blk_chars = femk::kReadBlock * std::min(max_len, 256u);
    l.strm_words = blk_chars / 16u + 3u;
    l.strm = take(3u * l.strm_words * 4u);
    const uint32_t q_at = o;
    l.F = take(femk::kQueueBytes);  // queue of small reads' seeds
    // queue of live phase groups: a read may add all six of its groups, each with G - Lg + 1 words
    const uint32_t g0 = l.smax / (uint32_t)femk::kStep;
    const uint32_t n_used = g0 > (uint32_t)femk::kLg ? g0 - (uint32_t)femk::kLg + 1u : 1u;
    l.gq_cap = std::max(384u, n_groups * n_used);
    l.gq = take(std::max(l.gq_cap * 4u + femk::kGroupQueue * 16u + 2u * 32u * 4u, femk::kFlushScratchBytes));
    // the block's raw characters (+ slack for the 16-byte copies) are dead once the streams exist, and both queues
    // are empty at that point: they share the space
    l.blk_bytes = blk_chars + 32u;
    l.blk = q_at;
    if (o < q_at + l.blk_bytes) take(q_at + l.blk_bytes - o);
  }
  if (hash) {              // hash-join form: open-addressing table; xcap = most occurrences one group may select
    l.xcap = (uint32_t)femk::bloom_chunks((int)R) * 64u;
    l.F = take(femk::bloom_slots((int)R) / 16u * 4u);  // bitmap: two bits per key slot
  }
  l.wave_bytes = o;
  return l;
}

// LDS of one wave of seed_select_kernel (fem_seed_select.hip.h): the block's read offsets, the sub-block's two 2-bit
// streams, one frequency byte per seed, strand and phase group, the per-read words.  The reads of a block are worked
// on `nb` at a time, as many as a budget of 3 KB of 16-bit frequencies holds (a block of four waves then stays within the
// 16 KB that six blocks of the join leave of a CU's LDS) (the kernel runs beside seed_join_kernel, whose
// bitmaps want the LDS).
femk::SeedLayout make_layout_select(const fem_params &p, uint32_t max_len) {
  femk::SeedLayout l{};
  const uint32_t R = (uint32_t)(p.e + 1 + p.a);
  l.smax = max_len >= (uint32_t)p.k ? max_len - (uint32_t)p.k + 1u : 1u;
  // a phase group's DP takes at most kSelMaxCols columns: groups of more than kSelMaxCols - 1 + 4 R seeds go to the generic kernel
  const uint32_t g_max = std::min<uint32_t>((l.smax + 2u) / 3u, femk::kSelMaxCols - 1u + 4u * R);
  // one array of 16-bit frequencies per strand, in seed order (group g's seed c at 3 c + g): even, and an odd number of
  // 32-bit words, so that the rows of the DP's lanes spread over the LDS banks
  l.gstride = 3u * g_max + 2u;
  l.gstride += l.gstride & 1u;
  if (!((l.gstride >> 1) & 1u)) l.gstride += 2u;
  l.nb = std::max<uint32_t>(1u, std::min<uint32_t>(femk::kReadBlock, 3100u / (2u * 2u * l.gstride)));
  l.strm_words = (l.nb * max_len + 15u) / 16u + 2u;
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) {
    uint32_t at = o;
    o += (bytes + 15u) & ~15u;
    return at;
  };
  l.rb = take((femk::kReadBlock + 2u) * 8u);
  l.rinfo = take(4u * femk::kReadBlock * 4u);
  l.strm = take(2u * l.strm_words * 4u);
  l.fq = take(l.nb * 2u * l.gstride * 2u + 256u);  // (+ what an idle lane of the last strand reads past its array)
  l.wave_bytes = o;
  return l;
}

// LDS of one wave of seed_join_kernel: the strands' candidates, flagged values per phase group, scatter, the block's
// begin/count entries, the sequence table, the join's bitmap
femk::SeedLayout make_layout_join(const fem_params &p, bool banked, bool padded = false) {
  femk::SeedLayout l{};
  const uint32_t R = (uint32_t)(p.e + 1 + p.a);
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) {
    uint32_t at = o;
    o += (bytes + 15u) & ~15u;
    return at;
  };
  l.sf = take(2u * 64u * 4u);
  l.X = take(64u * 4u);
  l.A = take(3u * (femk::dense_flag_cap((int)R) + 1u) * 4u);
  l.B = take(2u * femk::kReadBlock * 8u);
  l.F = take(femk::join_bitmap_words((int)R, padded) * 4u);
  if (banked) l.gq = take(2u * 64u * 8u);  // a strand's candidates of bank after bank (seed_join_body<R, true>)
  l.wave_bytes = o;
  l.picked = 0;  // the block's sequence table: set by the launcher (behind the waves' regions)
  return l;
}

// (`banked`: the reference's sequences lie in more than one coordinate space, fem_seed_dense.hip.h)
template <int R>
void launch_select_r(bool banked, dim3 grid, dim3 block, uint32_t lds, hipStream_t st, const femk::SeedParams &sp) {
  if (banked)
    hipLaunchKernelGGL((femk::seed_select_kernel<R, true>), grid, block, lds, st, sp);
  else
    hipLaunchKernelGGL((femk::seed_select_kernel<R>), grid, block, lds, st, sp);
}
template <int R>
int select_blocks_per_cu_r(bool banked, int block, uint32_t lds) {
  int nb = 0;
  const hipError_t e = banked ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, femk::seed_select_kernel<R, true>, block, lds)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, femk::seed_select_kernel<R>, block, lds);
  return e == hipSuccess ? nb : 0;
}
typedef void (*JoinKernel)(femk::SeedParams);
// which = 0: the compact table, 1: references in banks (compact, cut at bank_lo), 2: the strided table with its pads
JoinKernel join_kernel(int R, int which) {
  static const JoinKernel k[3][femk::kMaxR] = {
      {femk::seed_join_kernel_r1, femk::seed_join_kernel_r2, femk::seed_join_kernel_r3, femk::seed_join_kernel_r4, femk::seed_join_kernel_r5,
       femk::seed_join_kernel_r6, femk::seed_join_kernel_r7, femk::seed_join_kernel_r8, femk::seed_join_kernel_r9, femk::seed_join_kernel_r10},
      {femk::seed_join_banked_kernel_r1, femk::seed_join_banked_kernel_r2, femk::seed_join_banked_kernel_r3, femk::seed_join_banked_kernel_r4,
       femk::seed_join_banked_kernel_r5, femk::seed_join_banked_kernel_r6, femk::seed_join_banked_kernel_r7, femk::seed_join_banked_kernel_r8,
       femk::seed_join_banked_kernel_r9, femk::seed_join_banked_kernel_r10},
      {femk::seed_join_kernel_padded_r1, femk::seed_join_kernel_padded_r2, femk::seed_join_kernel_padded_r3, femk::seed_join_kernel_padded_r4,
       femk::seed_join_kernel_padded_r5, femk::seed_join_kernel_padded_r6, femk::seed_join_kernel_padded_r7, femk::seed_join_kernel_padded_r8,
       femk::seed_join_kernel_padded_r9, femk::seed_join_kernel_padded_r10}};
  return k[which][std::min(std::max(R, 1), femk::kMaxR) - 1];
}

#define FEM_DENSE_SWITCH(R, CALL)      \
  switch (R) {                         \
    case 1: CALL(1); break;            \
    case 2: CALL(2); break;            \
    case 3: CALL(3); break;            \
    case 4: CALL(4); break;            \
    case 5: CALL(5); break;            \
    case 6: CALL(6); break;            \
    case 7: CALL(7); break;            \
    case 8: CALL(8); break;            \
    case 9: CALL(9); break;            \
    default: CALL(10); break;          \
  }
template <int R>
uint32_t kernel_regs_r(bool join, int which) {
  hipFuncAttributes a{};
  const bool banked = which == 1;
  const void *f = join ? (const void *)join_kernel(R, which)
                       : banked ? (const void *)femk::seed_select_kernel<R, true> : (const void *)femk::seed_select_kernel<R>;
  return hipFuncGetAttributes(&a, f) == hipSuccess && a.numRegs > 0 ? (uint32_t)a.numRegs : 128u;
}
// vector registers per lane of seed_join_kernel<R> / seed_select_kernel<R>
// (handles of several GPUs launch from their own threads: the cache is atomic; every thread would store the same value)
uint32_t kernel_regs(int R, bool join, int which = 0) {
  static std::atomic<uint32_t> cache[3][2][femk::kMaxR + 1] = {};
  std::atomic<uint32_t> &slot = cache[which][join ? 1 : 0][std::min(std::max(R, 1), femk::kMaxR)];
  uint32_t c = slot.load(std::memory_order_relaxed);
  if (c) return c;
#define FEM_CALL(r) c = kernel_regs_r<r>(join, which)
  FEM_DENSE_SWITCH(R, FEM_CALL)
#undef FEM_CALL
  slot.store(c, std::memory_order_relaxed);
  return c;
}
int select_blocks_per_cu(int R, bool banked, int block, uint32_t lds) {
  int nb = 0;
#define FEM_CALL(r) nb = select_blocks_per_cu_r<r>(banked, block, lds)
  FEM_DENSE_SWITCH(R, FEM_CALL)
#undef FEM_CALL
  return nb;
}
void launch_select(int R, bool banked, dim3 grid, dim3 block, uint32_t lds, hipStream_t st, const femk::SeedParams &sp) {
#define FEM_CALL(r) launch_select_r<r>(banked, grid, block, lds, st, sp)
  FEM_DENSE_SWITCH(R, FEM_CALL)
#undef FEM_CALL
}
int join_blocks_per_cu(int R, int which, int block, uint32_t lds) {
  int nb = 0;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, join_kernel(R, which), block, lds) == hipSuccess ? nb : 0;
}
void launch_join(int R, int which, dim3 grid, dim3 block, uint32_t lds, hipStream_t st, const femk::SeedParams &sp) {
  hipLaunchKernelGGL(join_kernel(R, which), grid, block, lds, st, sp);
}

template <int R>
void launch_fast(bool hash, dim3 grid, dim3 block, uint32_t lds, hipStream_t st, const femk::SeedParams &sp) {
  if (hash)
    hipLaunchKernelGGL((femk::seed_fast_kernel<R, true>), grid, block, lds, st, sp);
  else
    hipLaunchKernelGGL((femk::seed_fast_kernel<R, false>), grid, block, lds, st, sp);
}

template <int R>
int fast_blocks_per_cu_r(bool hash, int block, uint32_t lds) {
  int nb = 0;
  hipError_t err = hash ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, femk::seed_fast_kernel<R, true>, block, lds)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, femk::seed_fast_kernel<R, false>, block, lds);
  return err == hipSuccess ? nb : 0;
}
// blocks of seed_fast_kernel<R, hash> one CU holds at a time (registers and LDS both count); 0 = unknown
int fast_blocks_per_cu(int R, bool hash, int block, uint32_t lds) {
  switch (R) {
    case 1: return fast_blocks_per_cu_r<1>(hash, block, lds);
    case 2: return fast_blocks_per_cu_r<2>(hash, block, lds);
    case 3: return fast_blocks_per_cu_r<3>(hash, block, lds);
    case 4: return fast_blocks_per_cu_r<4>(hash, block, lds);
    case 5: return fast_blocks_per_cu_r<5>(hash, block, lds);
    case 6: return fast_blocks_per_cu_r<6>(hash, block, lds);
    case 7: return fast_blocks_per_cu_r<7>(hash, block, lds);
    case 8: return fast_blocks_per_cu_r<8>(hash, block, lds);
    case 9: return fast_blocks_per_cu_r<9>(hash, block, lds);
    default: return fast_blocks_per_cu_r<10>(hash, block, lds);
  }
}

hipEvent_t get_event(fem_dev *h) {
  if (!h->event_pool.empty()) {
    hipEvent_t e = h->event_pool.back();
    h->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

// (timeline only) events around a copy on `st`
struct Span {
  fem_dev *h;
  Slot &s;
  hipStream_t st;
  TimedLaunch t;
  bool on;
  Span(fem_dev *h_, Slot &s_, int id, hipStream_t st_) : h(h_), s(s_), st(st_), t{id, nullptr, nullptr, false}, on(h_->timing && h_->timeline) {
    if (on) {
      t.start = get_event(h), t.stop = get_event(h);
      (void)hipEventRecord(t.start, st);
    }
  }
  ~Span() {
    if (on) {
      (void)hipEventRecord(t.stop, st);
      s.pending.push_back(t);
    }
  }
};

void drain_timing(fem_dev *h, Slot &s) {
  for (auto &t : s.pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess && t.kernel < kTimedKernels) {
      h->t_ms[t.kernel] += ms;
      h->t_n[t.kernel] += t.counts ? 1 : 0;
    }
    if (h->timeline && h->have_epoch) {
      float a = 0.f, b = 0.f;
      if (hipEventElapsedTime(&a, h->ev_epoch, t.start) == hipSuccess && hipEventElapsedTime(&b, h->ev_epoch, t.stop) == hipSuccess)
        fprintf(stderr, "TL slot %d id %2d  %9.3f %9.3f  (%.3f)\n", (int)(&s - h->slot), t.kernel, a, b, b - a);
    }
    h->event_pool.push_back(t.start);
    h->event_pool.push_back(t.stop);
  }
  s.pending.clear();
}

int ensure_outputs(fem_dev *h, Slot &s) {
  size_t want_reads = (size_t)s.n_reads;
  if (want_reads > s.per_read_cap || !s.d_begin) {
    size_t cap = std::max<size_t>(want_reads, 1);
    if (s.d_begin) (void)hipFree(s.d_begin);
    if (s.d_count) (void)hipFree(s.d_count);
    if (s.d_nmap) (void)hipFree(s.d_nmap);
    s.d_begin = s.d_count = s.d_nmap = nullptr;
    HIP_TRY(h, hipMalloc((void **)&s.d_begin, cap * 2 * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc((void **)&s.d_count, cap * 2 * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc((void **)&s.d_nmap, cap * sizeof(uint32_t)));
    s.per_read_cap = cap;
  }
  if (!s.d_cand || !s.d_meta || !s.d_ed || !s.d_end || s.cand_cap == 0) {
    for (void *p : {(void *)s.d_cand, (void *)s.d_meta, (void *)s.d_ed, (void *)s.d_end})
      if (p) (void)hipFree(p);
    s.d_cand = nullptr, s.d_meta = nullptr, s.d_ed = nullptr, s.d_end = nullptr;
    uint64_t want = 2 * s.n_reads + (5u << 20);  // ~1 candidate per strand on typical data + chunk padding
    if (h->tiny_buffers) want = 512;
    want = std::min<uint64_t>(want, 0xFFFFFFF0ull);
    s.cand_cap = 0;
    size_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int rc;
    if ((rc = dev_realloc(h, &s.d_cand, &c0, want))) return rc;
    if ((rc = dev_realloc(h, &s.d_meta, &c1, want))) return rc;
    if ((rc = dev_realloc(h, &s.d_ed, &c2, want))) return rc;
    if ((rc = dev_realloc(h, &s.d_end, &c3, want))) return rc;
    s.cand_cap = (uint32_t)want;
  }
  if (!s.d_ctl) {
    HIP_TRY(h, hipMalloc((void **)&s.d_ctl, kCtlAlloc));
    HIP_TRY(h, hipHostMalloc((void **)&s.h_ctl, kCtlBytes, hipHostMallocDefault));
  }
  if (!s.d_arena) {
    s.arena_cap = h->tiny_buffers ? 256u : (4u << 20);  // entries (32 MiB); grown on demand
    HIP_TRY(h, hipMalloc((void **)&s.d_arena, s.arena_cap * sizeof(uint64_t)));
  }
  {
    // With long occurrence lists (large references) nearly every read overflows the lanes of the fast kernel
    // and is queued; with short ones almost none is.  Size the queue for the likely case, grow + re-run otherwise.
    const double avg_bucket = (double)h->n_occ / (double)(h->n_lookup ? h->n_lookup : 1);
    uint64_t want = avg_bucket > 1.0 ? s.n_reads + (1u << 20) : (1u << 18);
    if (h->tiny_buffers) want = s.d_slow ? s.slow_cap : 32;
    if (want > s.slow_cap || !s.d_slow) {
      if (s.d_slow) (void)hipFree(s.d_slow);
      s.d_slow = nullptr;
      HIP_TRY(h, hipMalloc((void **)&s.d_slow, want * sizeof(uint32_t)));
      s.slow_cap = (uint32_t)want;
    }
  }
  return FEM_OK;
}

int grow_candidates(fem_dev *h, Slot &s, uint64_t want) {
  if (want > 0xFFFFFFF0ull) return fail(h, FEM_ERR_UNSUPPORTED, "more than 2^32 candidates in one batch; split the batch");
  s.cand_cap = 0;  // published again only once all four arrays exist (ensure_outputs re-allocates otherwise)
  for (void *p : {(void *)s.d_cand, (void *)s.d_meta, (void *)s.d_ed, (void *)s.d_end})
    if (p) (void)hipFree(p);
  s.d_cand = nullptr, s.d_meta = nullptr, s.d_ed = nullptr, s.d_end = nullptr;
  HIP_TRY(h, hipMalloc((void **)&s.d_cand, want * sizeof(uint64_t)));
  HIP_TRY(h, hipMalloc((void **)&s.d_meta, want * sizeof(uint32_t)));
  HIP_TRY(h, hipMalloc((void **)&s.d_ed, want * sizeof(uint8_t)));
  HIP_TRY(h, hipMalloc((void **)&s.d_end, want * sizeof(int16_t)));
  s.cand_cap = (uint32_t)want;
  return FEM_OK;
}

// The link is handed from one batch's copy to the next: call before and after a slot's H2D copies.
int h2d_begin(fem_dev *h, Slot &s) {
  if (h->have_h2d_done) HIP_TRY(h, hipStreamWaitEvent(s.stream, h->ev_h2d_done, 0));
  return FEM_OK;
}
int h2d_end(fem_dev *h, Slot &s) {
  HIP_TRY(h, hipEventRecord(h->ev_h2d_done, s.stream));
  h->have_h2d_done = true;
  return FEM_OK;
}
// No batch's kernels are pending on the device: the batch being committed starts a pipeline (it is sent and mapped in parts)
bool device_idle(fem_dev *h) {
  return (!h->have_kernels_done || hipEventQuery(h->ev_kernels_done) == hipSuccess) &&
         (!h->have_select_done || hipEventQuery(h->ev_select_done) == hipSuccess);
}
constexpr uint64_t kPartMinReads = 1u << 16;  // a part is at least this many reads

// The stream a slot's results go home on: its own.  (Tried: one stream for every slot's results behind an event of the slot's
// stream — consistent 0.63 ms per 32 MB where a slot's own stream takes 0.6 to 1.5 beside the next batch's H2D copy, but the
// calls that enqueue the copies then kept the host until the batch's kernels were through: 326 -> 284 Mreads/s.  Ten streams
// timed one by one all copy at 54 GB/s: the slow copies are the link shared with the other direction, not a slow engine.)
hipStream_t d2h_begin(fem_dev *h, Slot &s) {
  (void)h;
  return s.stream;
}
// ... and the slot's stream (what fem_dev_sync waits for) continues behind them.
int d2h_end(fem_dev *h, Slot &s, hipStream_t st) {
  if (st == s.stream) return FEM_OK;
  HIP_TRY(h, hipEventRecord(s.ev_home, st));
  HIP_TRY(h, hipStreamWaitEvent(s.stream, s.ev_home, 0));
  return FEM_OK;
}

// pack_results_kernel behind the slot's verification, and (send_home) the packed arrays' copies behind it: the per-strand
// bytes and the segment table whole, the per-candidate arrays up to what the slot's previous batch needed (the rest at fetch).
int enqueue_pack(fem_dev *h, Slot &s, bool kernel, bool send_home) {
  const size_t n2 = (size_t)s.n_reads * 2, n_seg = (n2 + 255) / 256;
  int rc;
  if ((rc = dev_realloc(h, &s.d_count8, &s.count8_cap, n2 + 16))) return rc;
  if ((rc = dev_realloc(h, &s.d_seg, &s.seg_cap, n_seg + 4))) return rc;
  if (s.pcand_cap < s.cand_cap || !s.d_pcand) {
    for (void *q : {(void *)s.d_pcand, (void *)s.d_ped, (void *)s.d_pend})
      if (q) (void)hipFree(q);
    s.d_pcand = nullptr, s.d_ped = nullptr, s.d_pend = nullptr, s.pcand_cap = 0;
    HIP_TRY(h, hipMalloc((void **)&s.d_pcand, (size_t)s.cand_cap * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc((void **)&s.d_ped, (size_t)s.cand_cap * sizeof(uint8_t)));
    HIP_TRY(h, hipMalloc((void **)&s.d_pend, (size_t)s.cand_cap * sizeof(int16_t)));
    s.pcand_cap = s.cand_cap;
  }
  if (!s.d_big) HIP_TRY(h, hipMalloc((void **)&s.d_big, kBigCap * sizeof(uint2)));
  if (n2 + 16 > s.h_count8_cap || !s.h_count8) {
    if (s.h_count8) (void)hipHostFree(s.h_count8);
    s.h_count8 = nullptr, s.h_count8_cap = 0;
    if ((rc = pinned_realloc(h, &s.h_count8, &s.h_count8_cap, n2 + 16))) return rc;
  }
  if (n_seg + 4 > s.h_seg_cap || !s.h_seg) {
    if (s.h_seg) (void)hipHostFree(s.h_seg);
    s.h_seg = nullptr, s.h_seg_cap = 0;
    if ((rc = pinned_realloc(h, &s.h_seg, &s.h_seg_cap, n_seg + 4))) return rc;
  }
  if (!s.h_big) {
    size_t c = 0;
    if ((rc = pinned_realloc(h, &s.h_big, &c, (size_t)kBigCap))) return rc;
  }
  if (kernel) {
  femk::PackParams pp{};
  pp.cand_begin = s.d_begin, pp.cand_count = s.d_count, pp.cand = s.d_cand, pp.ed = s.d_ed, pp.end = s.d_end;
  pp.ctr = (const uint32_t *)s.d_ctl, pp.n_strands = (uint32_t)n2;
  pp.count8 = s.d_count8, pp.seg_begin = s.d_seg, pp.pcand = s.d_pcand, pp.ped = s.d_ped, pp.pend = s.d_pend, pp.pcap = (uint32_t)s.pcand_cap;
  pp.cursor = (uint32_t *)(s.d_ctl + kCtlPackCursor), pp.big = s.d_big, pp.big_cap = kBigCap;
  const uint32_t grid = (uint32_t)std::max<size_t>(1, (n_seg + femk::kPackSegs - 1) / femk::kPackSegs);  // (a block per sixteen segments)
  {
    TimedLaunch t{6, nullptr, nullptr, true};  // (kernel id 6: pack_results_kernel)
    if (h->timing) {
      t.start = get_event(h), t.stop = get_event(h);
      HIP_TRY(h, hipEventRecord(t.start, s.stream));
    }
    hipLaunchKernelGGL(femk::pack_results_kernel, dim3(grid), dim3(256), 0, s.stream, pp);
    HIP_TRY(h, hipGetLastError());
    if (h->timing) {
      HIP_TRY(h, hipEventRecord(t.stop, s.stream));
      s.pending.push_back(t);
    }
  }
  s.packed_enqueued = true;
  }
  if (send_home) {
    hipStream_t st = d2h_begin(h, s);
    {
      Span span(h, s, 21, st);
      HIP_TRY(h, hipMemcpyAsync(s.h_count8, s.d_count8, n2, hipMemcpyDeviceToHost, st));
      HIP_TRY(h, hipMemcpyAsync(s.h_seg, s.d_seg, n_seg * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      s.packed_per_read_home = true;
      const size_t guess = std::min<size_t>({(size_t)(s.last_n_packed + s.last_n_packed / 32 + 1024), s.h_pcand_cap, s.pcand_cap});
      if (s.last_n_packed && s.h_pcand && guess) {
        HIP_TRY(h, hipMemcpyAsync(s.h_pcand, s.d_pcand, guess * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipMemcpyAsync(s.h_ped, s.d_ped, guess * sizeof(uint8_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipMemcpyAsync(s.h_pend, s.d_pend, guess * sizeof(int16_t), hipMemcpyDeviceToHost, st));
        s.packed_home = guess;
      }
    }
    if ((rc = d2h_end(h, s, st))) return rc;
  }
  return FEM_OK;
}

// Enqueue the two kernels of one batch on the slot's stream (asynchronous).
int launch_batch(fem_dev *h, Slot &s) {
  const fem_params &p = s.params;
  uint32_t *d_ctr = (uint32_t *)s.d_ctl;
  unsigned long long *d_arena_ctr = (unsigned long long *)(s.d_ctl + 4 * sizeof(uint32_t));
  unsigned long long *d_stats = (unsigned long long *)(s.d_ctl + 4 * sizeof(uint32_t) + 2 * sizeof(uint64_t));
  // A batch that meets an IDLE device (the start of a job) is mapped in parts on a dense index — see below; its counters are
  // then zeroed on the side stream, ahead of the slot's stream, which may still be busy with the rest of the batch's copy.
  int parts = 1;
  uint32_t part_begin[kMaxParts + 1] = {0, 0, 0, 0, 0};
  const bool copied_in_parts = s.parts > 1;
  {
    const int R_ = p.e + 1 + p.a;
    const bool dense_overlap = !h->force_generic && p.k == femk::kK && p.step == femk::kStep && R_ >= 1 && R_ <= femk::kMaxR && h->d_occ32 &&
                               h->d_freq11 && !h->no_overlap;
    if (dense_overlap && !h->no_parts && s.n_reads >= 2 * kPartMinReads && (copied_in_parts || device_idle(h))) {
      parts = copied_in_parts ? s.parts : (int)std::min<uint64_t>((uint64_t)h->max_parts, s.n_reads / kPartMinReads);
      for (int q = 0; q <= parts; ++q)
        part_begin[q] = copied_in_parts ? s.part_begin[q] : q == parts ? (uint32_t)s.n_reads : (uint32_t)((s.n_reads * q / parts) & ~63ull);
    }
    s.parts = 1;  // (consumed: the slot's batch mapped again starts from the device's state then)
  }
  if (parts > 1) {
    if (h->have_select_done) HIP_TRY(h, hipStreamWaitEvent(h->side_stream, h->ev_select_done, 0));
    HIP_TRY(h, hipMemsetAsync(s.d_ctl, 0, kCtlAlloc, h->side_stream));
  } else {
    HIP_TRY(h, hipMemsetAsync(s.d_ctl, 0, kCtlAlloc, s.stream));
  }

  femk::SeedParams sp{};
  sp.bases = s.bases();
  sp.read_off = s.d_off;
  sp.n_reads = (uint32_t)s.n_reads;
  sp.read_begin = 0;
  sp.lookup = h->d_lookup;
  sp.occ = h->d_occ;
  sp.inf32 = (uint32_t)h->n_occ;
  sp.seq_len = h->d_seq_len;
  sp.e = p.e, sp.a = p.a, sp.R = p.e + 1 + p.a, sp.k = p.k, sp.step = p.step;
  sp.lg = p.k / p.step + (p.k % p.step ? 1 : 0);
  sp.cand = s.d_cand, sp.cand_meta = s.d_meta, sp.cand_cap = s.cand_cap;
  sp.cand_begin = s.d_begin, sp.cand_count = s.d_count;
  sp.ctr = d_ctr;
  sp.work_cursor = (uint32_t *)(s.d_ctl + kCtlWorkCursor);
  sp.stats = d_stats;
  sp.arena = s.d_arena, sp.arena_cap = s.arena_cap, sp.arena_ctr = d_arena_ctr;
  sp.lay = make_layout(p, std::max<uint32_t>(s.max_len, (uint32_t)p.k));

  sp.n_seq = h->n_seq;
  sp.summary = h->d_summary;
  sp.slow_queue = s.d_slow, sp.slow_cap = s.slow_cap;
  sp.work_queue = nullptr;
  const int R = p.e + 1 + p.a;
  const bool use_fast = !h->force_generic && p.k == femk::kK && p.step == femk::kStep && R >= 1 && R <= femk::kMaxR;

  auto shape = [&](const femk::SeedLayout &lay, uint32_t *wpb_out, uint32_t *lds_out, uint32_t *grid_out) {
    uint32_t wpb = std::min<uint32_t>(4u, std::max<uint32_t>(1u, (64u * 1024u) / lay.wave_bytes));
    uint32_t lds_bytes = wpb * lay.wave_bytes;
    uint32_t waves_per_cu = std::min<uint32_t>(32u, (160u * 1024u / lds_bytes) * wpb);
    uint64_t blocks_wanted = (s.n_reads + wpb - 1) / wpb;
    uint64_t blocks_resident = (uint64_t)h->n_cu * std::max<uint32_t>(1u, waves_per_cu / wpb);
    *wpb_out = wpb, *lds_out = lds_bytes;
    *grid_out = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(blocks_wanted, blocks_resident * 4));
  };
  if (sp.lay.wave_bytes > 64u * 1024u) return fail(h, FEM_ERR_UNSUPPORTED, "read too long for the device path");

  auto timed = [&](int id, hipStream_t st, auto &&launch, bool counts = true) -> int {
    TimedLaunch t{id, nullptr, nullptr, counts};
    if (h->timing) {
      t.start = get_event(h), t.stop = get_event(h);
      HIP_TRY(h, hipEventRecord(t.start, st));
    }
    launch();
    HIP_TRY(h, hipGetLastError());
    if (h->timing) {
      HIP_TRY(h, hipEventRecord(t.stop, st));
      s.pending.push_back(t);
    }
    return FEM_OK;
  };

  if (s.n_reads) {
    int rc;
    const bool split_dense = use_fast && h->d_occ32 && h->d_freq11;
    const bool overlap = split_dense && !h->no_overlap;
    if (!overlap && h->have_kernels_done) HIP_TRY(h, hipStreamWaitEvent(s.stream, h->ev_kernels_done, 0));
    femk::VerifyParams vp{};
    vp.bases = s.bases(), vp.read_off = s.d_off;
    vp.planes = h->d_planes, vp.seq_off = h->d_seq_off;
    vp.cand = s.d_cand, vp.cand_meta = s.d_meta;
    vp.ctr = d_ctr, vp.cand_cap = s.cand_cap, vp.e = p.e;
    vp.ed = s.d_ed, vp.end = s.d_end;
    vp.n_map = s.d_nmap, vp.stats = d_stats;
    HIP_TRY(h, hipMemsetAsync(s.d_nmap, 0, (size_t)s.n_reads * sizeof(uint32_t), s.stream));
    // grid-stride kernel: exactly the blocks that are resident together, or the ones that start late set the makespan
    if (h->verify_blocks_per_cu == 0) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, femk::verify_kernel, 256, 0) != hipSuccess || nb <= 0) nb = 4;
      h->verify_blocks_per_cu = nb;
    }
    const uint32_t vgrid = (uint32_t)h->n_cu * (uint32_t)h->verify_blocks_per_cu;
    if (split_dense) {
      // dense index: seed selection for blocks of reads (fem_seed_select.hip.h), then the bitmap join on 32-bit
      // coordinates, a wave per read (fem_seed_dense.hip.h)
      const uint32_t max_len = std::max<uint32_t>(s.max_len, (uint32_t)p.k);
      const bool banked = h->n_banks > 1;
      const size_t want_sel = (size_t)s.n_reads * 6u * (size_t)R * h->n_banks;
      if ((rc = dev_realloc(h, &s.d_sel, &s.sel_cap, want_sel))) return rc;
      if ((rc = dev_realloc(h, &s.d_sel_hdr, &s.sel_hdr_cap, (size_t)s.n_reads))) return rc;
      femk::SeedParams fp = sp;
      fp.occ32 = h->d_occ32, fp.list_shift = h->list_shift, fp.goff = h->d_goff, fp.blkseq = h->d_blkseq;
      fp.freq11 = h->d_freq11, fp.sel = s.d_sel, fp.sel_hdr = s.d_sel_hdr;
      fp.n_banks = h->n_banks, fp.bank_lo = h->d_bank_lo, fp.n_buckets = (uint32_t)(h->n_lookup - 1);
      fp.blk_stride = (femk::kDenseRemap >> femk::kDenseBlkShift) + 1u;
      for (uint32_t b = 0; b <= femk::kDenseMaxBanks; ++b) fp.bank_first[b] = h->bank_first[b];
      // A batch committed to an idle device (enqueue_packed: s.parts > 1) is mapped in parts: part q's selection on the side
      // stream as soon as its characters are in HBM, its join on the slot's stream behind its selection — beside the selection
      // of part q + 1.  Any other batch is one part, selection and join on the slot's stream as before.
      uint32_t select_lds = 0, select_threads = 256;
      fp.lay = make_layout_select(p, max_len);
      if (fp.lay.wave_bytes > 64u * 1024u) return fail(h, FEM_ERR_UNSUPPORTED, "read too long for the device path");
      const femk::SeedLayout lay_select = fp.lay;
      femk::SeedLayout lay_join = make_layout_join(p, banked, !banked && h->list_shift != 0);
      const uint32_t wpb_s = std::min<uint32_t>(4u, std::max<uint32_t>(1u, (64u * 1024u) / lay_select.wave_bytes));
      const uint32_t wpb_j = std::min<uint32_t>(4u, std::max<uint32_t>(1u, (60u * 1024u) / lay_join.wave_bytes));
      lay_join.picked = wpb_j * lay_join.wave_bytes;
      const uint32_t lds_j = wpb_j * lay_join.wave_bytes + 64u * 8u;
      const int which = banked ? 1 : h->list_shift ? 2 : 0;  // (join_kernel)
      uint64_t per_cu_s, per_cu_j;
      {
        const uint32_t lds_bytes = wpb_s * lay_select.wave_bytes;
        const uint64_t key = ((uint64_t)banked << 48) | ((uint64_t)R << 40) | lds_bytes;
        if (h->select_occ_key != key) h->select_occ_key = key, h->select_occ_blocks = select_blocks_per_cu(R, banked, (int)(64u * wpb_s), lds_bytes);
        per_cu_s = h->select_occ_blocks > 0 ? (uint64_t)h->select_occ_blocks : std::max<uint64_t>(1, 160u * 1024u / lds_bytes);
        if (overlap) per_cu_s = 1;  // (3.3 ms per 2.5 M reads of C3 with one block per CU as with five: sectors per second, not waves)
        select_lds = lds_bytes, select_threads = 64u * wpb_s;
      }
      {
        const uint64_t key = ((uint64_t)which << 48) | ((uint64_t)R << 40) | lds_j;
        if (h->join_occ_key != key) h->join_occ_key = key, h->join_occ_blocks = join_blocks_per_cu(R, which, (int)(64u * wpb_j), lds_j);
        per_cu_j = h->join_occ_blocks > 0 ? (uint64_t)h->join_occ_blocks : std::max<uint64_t>(1, 160u * 1024u / lds_j);
        if (overlap) {
          // leave one block of the next batch's seed_select_kernel room on every CU: registers (512 per lane and SIMD,
          // handed out in eights), LDS (160 KB) and wave slots (8 per SIMD) of both kernels together
          const uint32_t vj = (kernel_regs(R, true, which) + 7u) & ~7u, vs = (kernel_regs(R, false, banked ? 1 : 0) + 7u) & ~7u;
          const uint32_t wj = 64u * wpb_j / 256u ? 64u * wpb_j / 256u : 1u, ws = select_threads / 256u ? select_threads / 256u : 1u;  // waves per SIMD and block
          const uint32_t lj = (lds_j + 511u) & ~511u, ls = (select_lds + 511u) & ~511u;  // (LDS is handed out in pieces of 512 bytes)
          while (per_cu_j > 1 && (per_cu_j * wj * vj + ws * vs > 512u || per_cu_j * lj + ls > 160u * 1024u || per_cu_j * wj + ws > 8u)) --per_cu_j;
        }
      }
      if (parts > 1 && !copied_in_parts) {  // (the copy went whole, on the slot's stream: the side stream starts behind it)
        HIP_TRY(h, hipEventRecord(s.ev_zeroed, s.stream));
        HIP_TRY(h, hipStreamWaitEvent(h->side_stream, s.ev_zeroed, 0));
      }
      for (int q = 0; q < parts; ++q) {
        const uint32_t r_lo = parts > 1 ? part_begin[q] : 0u, r_hi = parts > 1 ? part_begin[q + 1] : (uint32_t)s.n_reads;
        const uint64_t blocks_of_reads = ((uint64_t)(r_hi - r_lo) + femk::kReadBlock - 1) / femk::kReadBlock;
        fp.read_begin = r_lo, fp.n_reads = r_hi;
        hipStream_t st_sel = parts > 1 ? h->side_stream : s.stream;
        {
          fp.lay = lay_select;
          if (parts > 1) {
            if (copied_in_parts) HIP_TRY(h, hipStreamWaitEvent(st_sel, s.ev_part[q], 0));
          } else if (overlap && h->have_select_done) {
            HIP_TRY(h, hipStreamWaitEvent(s.stream, h->ev_select_done, 0));
          }
          const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((blocks_of_reads + wpb_s - 1) / wpb_s, (uint64_t)h->n_cu * per_cu_s));
          fp.work_cursor = (uint32_t *)(s.d_ctl + kCtlWorkCursor + (size_t)q * kCtlPartStride);
          rc = timed(8, st_sel, [&] { launch_select(R, banked, dim3(grid), dim3(64u * wpb_s), select_lds, st_sel, fp); }, q == 0);
          if (rc) return rc;
          if (parts > 1) {
            HIP_TRY(h, hipEventRecord(s.ev_sel[q], st_sel));
            HIP_TRY(h, hipStreamWaitEvent(s.stream, s.ev_sel[q], 0));
          }
          if (overlap && q + 1 == parts) {
            HIP_TRY(h, hipEventRecord(h->ev_select_done, st_sel));
            h->have_select_done = true;
          }
          if (overlap && q == 0 && h->have_kernels_done) HIP_TRY(h, hipStreamWaitEvent(s.stream, h->ev_kernels_done, 0));
        }
        {
          fp.lay = lay_join;
          const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((blocks_of_reads + wpb_j - 1) / wpb_j, (uint64_t)h->n_cu * per_cu_j));
          fp.work_cursor = (uint32_t *)(s.d_ctl + kCtlWorkCursor2 + (size_t)q * kCtlPartStride);
          rc = timed(0, s.stream, [&] { launch_join(R, which, dim3(grid), dim3(64u * wpb_j), lds_j, s.stream, fp); }, q == 0);
          if (rc) return rc;
        }
      }
      sp.work_queue = s.d_slow;  // the generic kernel finishes what the two queued
    } else if (use_fast) {
      femk::SeedParams fp = sp;
      // long occurrence lists (dense index): the hash-join form of the kernel; short ones: lists in lanes only
      const bool hash = h->force_hash || (double)h->n_occ > (double)h->n_lookup;
      fp.lay = make_layout_fast(p, std::max<uint32_t>(s.max_len, (uint32_t)p.k), hash);
      uint32_t wpb, lds_bytes, grid;
      shape(fp.lay, &wpb, &lds_bytes, &grid);
      {
        // The kernel's waves pull blocks of reads from a cursor: the grid is exactly what is resident at a time
        // (registers included).  (With a fixed stride per wave the same grid took 6.9 ms against 6.2 at six times as
        // many waves: the CUs do not all run at the same pace.  Every wave pads its last chunk of candidate slots, so
        // extra waves cost the verify kernel lanes.)  FEM_GRID_MULT overrides the multiple (measurement only).
        const uint64_t key = ((uint64_t)R << 40) | ((uint64_t)hash << 32) | lds_bytes;
        if (h->fast_occ_key != key) h->fast_occ_key = key, h->fast_occ_blocks = fast_blocks_per_cu((int)R, hash, (int)(64u * wpb), lds_bytes);
        static const uint64_t mult = testing_switch("FEM_TESTING") && getenv("FEM_GRID_MULT") ? (uint64_t)std::max(1, atoi(getenv("FEM_GRID_MULT"))) : 1;
        const uint64_t per_cu = h->fast_occ_blocks > 0 ? (uint64_t)h->fast_occ_blocks : std::max<uint64_t>(1, 160u * 1024u / lds_bytes);
        const uint64_t wanted = (s.n_reads + (uint64_t)femk::kReadBlock * wpb - 1) / ((uint64_t)femk::kReadBlock * wpb);
        grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(wanted, (uint64_t)h->n_cu * per_cu * mult));
      }
      auto launch_range = [&](uint32_t lo, uint32_t hi) {
        femk::SeedParams q = fp;
        q.read_begin = lo, q.n_reads = hi;
        dim3 g(grid), b(64u * wpb);
        switch (R) {
          case 1: launch_fast<1>(hash, g, b, lds_bytes, s.stream, q); break;
          case 2: launch_fast<2>(hash, g, b, lds_bytes, s.stream, q); break;
          case 3: launch_fast<3>(hash, g, b, lds_bytes, s.stream, q); break;
          case 4: launch_fast<4>(hash, g, b, lds_bytes, s.stream, q); break;
          case 5: launch_fast<5>(hash, g, b, lds_bytes, s.stream, q); break;
          case 6: launch_fast<6>(hash, g, b, lds_bytes, s.stream, q); break;
          case 7: launch_fast<7>(hash, g, b, lds_bytes, s.stream, q); break;
          case 8: launch_fast<8>(hash, g, b, lds_bytes, s.stream, q); break;
          case 9: launch_fast<9>(hash, g, b, lds_bytes, s.stream, q); break;
          default: launch_fast<10>(hash, g, b, lds_bytes, s.stream, q); break;
        }
      };
      rc = timed(0, s.stream, [&] { launch_range(0, (uint32_t)s.n_reads); });
      if (rc) return rc;
      sp.work_queue = s.d_slow;  // the generic kernel finishes what the fast one queued
    }
    {
      uint32_t wpb, lds_bytes, grid;
      shape(sp.lay, &wpb, &lds_bytes, &grid);
      rc = timed(2, s.stream, [&] { hipLaunchKernelGGL(femk::seed_filter_kernel, dim3(grid), dim3(64u * wpb), lds_bytes, s.stream, sp); });
      if (rc) return rc;
    }
    rc = timed(1, s.stream, [&] { hipLaunchKernelGGL(femk::verify_kernel, dim3(vgrid), dim3(256), 0, s.stream, vp); });
    if (rc) return rc;
    s.packed_enqueued = false, s.packed_home = 0, s.packed_per_read_home = false;
    // (the packing inside the chain of the batches' kernels, 0.07 ms: beside the next batch's join — three kernels starting at
    //  once — it cost that join 0.6 ms; on a sparse index, where nothing runs beside the seed kernel, it goes behind the chain)
    if (s.want_packed && split_dense && (rc = enqueue_pack(h, s, true, false))) return rc;
    HIP_TRY(h, hipEventRecord(h->ev_kernels_done, s.stream));
    h->have_kernels_done = true;
    if (s.want_packed && (rc = enqueue_pack(h, s, !split_dense, true))) return rc;
  } else {
    s.packed_enqueued = false, s.packed_home = 0, s.packed_per_read_home = false;
  }
  HIP_TRY(h, hipMemcpyAsync(s.h_ctl, s.d_ctl, kCtlBytes, hipMemcpyDeviceToHost, s.stream));
  s.prefetched_reads2 = 0, s.prefetched_cand = 0;
  static const bool no_prefetch = testing_switch("FEM_NO_PREFETCH");
  // (only behind fem_dev_stage_reads, whose caller packs the next batch in the meantime; a caller of the zero-copy form is
  // idle until it fetches, and the extra traffic next to its four-times-larger H2D cost 5 % there)
  if (s.prefetch_results && !s.want_packed && s.staged_by_copy && !no_prefetch) {
    hipStream_t st = d2h_begin(h, s);
    {
      Span span(h, s, 21, st);
      const size_t n2 = (size_t)s.n_reads * 2;
      if (n2 && s.h_begin && n2 <= s.h_per_read_cap) {
        HIP_TRY(h, hipMemcpyAsync(s.h_begin, s.d_begin, n2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipMemcpyAsync(s.h_count, s.d_count, n2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        s.prefetched_reads2 = n2;
      }
      const size_t guess = std::min<size_t>({(size_t)(s.last_n_cand + s.last_n_cand / 32 + 1024), s.h_cand_cap, (size_t)s.cand_cap});
      if (s.last_n_cand && s.h_cand && guess) {
        HIP_TRY(h, hipMemcpyAsync(s.h_cand, s.d_cand, guess * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipMemcpyAsync(s.h_ed, s.d_ed, guess * sizeof(uint8_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipMemcpyAsync(s.h_end, s.d_end, guess * sizeof(int16_t), hipMemcpyDeviceToHost, st));
        s.prefetched_cand = guess;
      }
    }
    int rc2 = d2h_end(h, s, st);
    if (rc2) return rc2;
  }
  s.mapped = true;
  s.synced = false;
  return FEM_OK;
}

// After the index is resident: for sparse indexes build the bucket summaries the fast seed kernel tests first.
int refresh_summary(fem_dev *h) {
  if (h->d_summary) (void)hipFree(h->d_summary);
  h->d_summary = nullptr;
  const uint64_t n_buckets = h->n_lookup - 1;
  // dense index: nearly every bucket is non-empty, the tests would not pay.  (Fewer than 2^31 entries also keeps bit
  // 31 of a lookup value free: the fast seed kernel tags deferred lookups with it.)
  if (h->n_occ >= n_buckets || h->n_occ >= 0x80000000ull || (n_buckets >> 3) >= (1ull << 22)) return FEM_OK;
  const uint64_t words = n_buckets / femk::kSummaryBuckets + 2;
  HIP_TRY(h, hipMalloc((void **)&h->d_summary, words * sizeof(uint32_t)));
  HIP_TRY(h, hipMemset(h->d_summary, 0, words * sizeof(uint32_t)));
  hipLaunchKernelGGL(femk::bucket_summary_kernel, dim3((uint32_t)h->n_cu * 8u), dim3(256), 0, 0, h->d_lookup, n_buckets, h->d_summary);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipDeviceSynchronize());
  return FEM_OK;
}

// After index AND reference are resident: for dense indexes (long occurrence lists) derive the occurrence table in
// 32-bit global coordinates that seed_join_kernel joins on (fem_seed_dense.hip.h) and the byte frequencies seed_select_kernel reads.  Skipped (the 64-bit hash-join
// form of seed_fast_kernel runs instead) when the coordinates do not fit 32 bits.
constexpr double kDenseMinAvgBucket = 4.0;
int refresh_dense(fem_dev *h) {
  for (void *p : {(void *)h->d_occ32, (void *)h->d_goff, (void *)h->d_blkseq, (void *)h->d_freq11, (void *)h->d_bank_lo})
    if (p) (void)hipFree(p);
  h->d_occ32 = nullptr, h->d_goff = nullptr, h->d_blkseq = nullptr, h->d_freq11 = nullptr, h->d_bank_lo = nullptr;
  h->n_banks = 1, h->list_shift = 0;
  if (!h->d_occ || !h->d_ref_raw || h->no_dense || h->k != femk::kK || h->step != femk::kStep || h->n_occ == 0) return FEM_OK;
  const uint64_t n_buckets = h->n_lookup - 1;
  if (!h->force_dense && (double)h->n_occ < kDenseMinAvgBucket * (double)n_buckets) return FEM_OK;
  // Coordinates: goff[seq] + pos, a gap between sequences; where the next sequence would pass the 32-bit limit a new BANK
  // starts with coordinates of its own (fem_seed_dense.hip.h) — up to kDenseMaxBanks of them, else the 64-bit join
  const uint64_t limit = h->bank_limit ? std::min<uint64_t>(h->bank_limit, femk::kDenseLimit) : femk::kDenseLimit;
  const uint32_t seq_limit = h->bank_seqs ? std::min<uint32_t>(h->bank_seqs, femk::kDenseMaxSeq) : femk::kDenseMaxSeq;
  std::vector<uint32_t> goff(h->n_seq + 1);
  uint32_t n_banks = 1, bank_first[5] = {0, 0, 0, 0, 0};
  // Where more than one bank is needed they are cut about EQUAL, not the first ones filled to the brim: the join runs once
  // per bank on that bank's part of every list, and a pass over lists of up to 64 entries costs about half of one over
  // longer lists (their second chunks).  Filling each bank up to the mean + one sequence never needs more banks.
  uint64_t cap = limit;
  {
    uint64_t total_c = femk::kDenseGap, longest = 0;
    for (uint32_t i = 0; i < h->n_seq; ++i) total_c += (uint64_t)h->seq_len[i] + femk::kDenseGap, longest = std::max<uint64_t>(longest, h->seq_len[i]);
    const uint64_t want = (total_c + limit - 1) / limit;
    if (want > 1) cap = std::min<uint64_t>(limit, (total_c + want - 1) / want + longest + femk::kDenseGap);
  }
  uint64_t at = femk::kDenseGap;
  for (uint32_t i = 0; i < h->n_seq; ++i) {
    // (a bank also ends at kDenseMaxSeq sequences: the remapped near-start entries carry the index within the bank)
    if ((at + (uint64_t)h->seq_len[i] + femk::kDenseGap > cap && at != femk::kDenseGap) || i - bank_first[n_banks - 1] >= seq_limit) {
      if (n_banks == femk::kDenseMaxBanks) return FEM_OK;
      bank_first[n_banks++] = i;
      at = femk::kDenseGap;
    }
    goff[i] = (uint32_t)at;
    at += (uint64_t)h->seq_len[i] + femk::kDenseGap;
    if (at > femk::kDenseLimit) return FEM_OK;  // one sequence beyond 32 bits: 64-bit join
  }
  goff[h->n_seq] = (uint32_t)at;
  for (uint32_t b = n_banks; b <= femk::kDenseMaxBanks; ++b) bank_first[b] = h->n_seq;
  const uint32_t n_blk = (femk::kDenseRemap >> femk::kDenseBlkShift) + 1u;
  std::vector<uint32_t> blkseq((size_t)n_blk * n_banks, 0);  // per bank: the last of its sequences starting at or before each 2^20 block
  for (uint32_t bank = 0; bank < n_banks; ++bank)
    for (uint32_t b = 0, sq = bank_first[bank]; b < n_blk; ++b) {
      const uint64_t first = (uint64_t)b << femk::kDenseBlkShift;
      while (sq + 1u < bank_first[bank + 1] && goff[sq + 1u] <= first) ++sq;
      blkseq[(size_t)bank * n_blk + b] = sq;
    }
  // These tables are optional (4 bytes per occurrence: 4 GB at 3 Gbp): a failed allocation declines the dense form —
  // the 64-bit hash-join form of seed_fast_kernel runs without them — instead of failing the upload.
  uint32_t *d_bad = nullptr;
  auto decline = [&]() {
    for (void *q : {(void *)h->d_occ32, (void *)h->d_goff, (void *)h->d_blkseq, (void *)h->d_freq11, (void *)h->d_bank_lo, (void *)d_bad})
      if (q) (void)hipFree(q);
    h->d_occ32 = nullptr, h->d_goff = nullptr, h->d_blkseq = nullptr, h->d_freq11 = nullptr, h->d_bank_lo = nullptr;
    h->n_banks = 1, h->list_shift = 0;
    (void)hipGetLastError();  // (clears the out-of-memory error)
    return FEM_OK;
  };
  // One coordinate space: the STRIDED 32-bit table (fem_seed_dense.hip.h: 512 bytes per bucket, lists on line boundaries, found
  // by the hash alone).  Where it does not fit, and for references in banks (whose lists are cut at bank_lo), the compact one.
  // It is 8.6 GB whatever the reference, so it is taken where the compact table is at least an eighth of that (16 entries per
  // bucket and more: references from ~0.8 Gbp; at 3 Gbp it is 2.1 x the compact table and bought 7 % of the step) — and
  // wherever FEM_FORCE_DENSE asks for the dense kernels on a small reference (the parity tests of the padded join).
  uint32_t list_shift = 0;
  const bool strided_pays = h->force_dense || h->n_occ >= (n_buckets << 4);
  if (n_banks == 1 && !h->no_strided && strided_pays) {
    const size_t words = ((size_t)n_buckets + femk::kDensePadBuckets) << femk::kDenseListShift;
    bool ok = hipMalloc((void **)&h->d_occ32, words * sizeof(uint32_t)) == hipSuccess;
    if (ok) {  // every slot reads "pad" (fem_seed_dense.hip.h) until dense_occ32_strided_kernel writes a bucket's entries over its first ones
      hipLaunchKernelGGL(femk::dense_pad_kernel, dim3((uint32_t)h->n_cu * 16u), dim3(256), 0, 0, (uint4 *)h->d_occ32, (uint64_t)(words / 4));
      ok = hipGetLastError() == hipSuccess;
    }
    if (ok) {
      list_shift = femk::kDenseListShift;
    } else {
      if (h->d_occ32) (void)hipFree(h->d_occ32);
      h->d_occ32 = nullptr;
      (void)hipGetLastError();
    }
  }
  if (hipMalloc((void **)&h->d_goff, goff.size() * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc((void **)&h->d_blkseq, blkseq.size() * sizeof(uint32_t)) != hipSuccess ||
      (!list_shift && hipMalloc((void **)&h->d_occ32, (h->n_occ + 256) * sizeof(uint32_t)) != hipSuccess) ||
      hipMalloc((void **)&h->d_freq11, (size_t)femk::kX11 * 4u * sizeof(uint32_t)) != hipSuccess ||
      (n_banks > 1 && hipMalloc((void **)&h->d_bank_lo, (size_t)(n_banks - 1) * n_buckets * sizeof(uint32_t)) != hipSuccess) ||
      hipMalloc((void **)&d_bad, sizeof(uint32_t)) != hipSuccess)
    return decline();
  if (hipMemset(d_bad, 0, sizeof(uint32_t)) != hipSuccess ||
      hipMemcpy(h->d_goff, goff.data(), goff.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->d_blkseq, blkseq.data(), blkseq.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
    (void)decline();
    return fail(h, FEM_ERR_HIP, "dense tables: copy to the device failed");
  }
  femk::BankFirst bf{};
  bf.n = n_banks;
  for (uint32_t b = 0; b <= femk::kDenseMaxBanks; ++b) bf.first[b] = bank_first[b];
  if (list_shift)
    hipLaunchKernelGGL(femk::dense_occ32_strided_kernel, dim3((uint32_t)h->n_cu * 8u), dim3(256), 0, 0, h->d_occ, h->d_lookup, (uint32_t)n_buckets, h->d_goff,
                       h->n_seq, h->d_occ32, d_bad);
  else
    hipLaunchKernelGGL(femk::dense_occ32_kernel, dim3((uint32_t)h->n_cu * 8u), dim3(256), 0, 0, h->d_occ, h->n_occ, h->d_goff, h->n_seq, bf,
                       h->d_occ32, d_bad);
  uint32_t bad = 0;
  if (hipGetLastError() != hipSuccess || hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess) {
    (void)decline();
    return fail(h, FEM_ERR_HIP, "dense tables: building the 32-bit occurrence table failed");
  }
  if (bad) return decline();  // the index names sequences the reference does not have: leave that to the 64-bit path's checks
  (void)hipFree(d_bad);
  d_bad = nullptr;
  for (uint32_t b = 1; b < n_banks; ++b)  // where bank b's part of every list starts
    hipLaunchKernelGGL(femk::bank_split_kernel, dim3((uint32_t)h->n_cu * 8u), dim3(256), 0, 0, h->d_occ, h->d_lookup, (uint32_t)n_buckets, bank_first[b],
                       h->d_bank_lo + (size_t)(b - 1) * n_buckets);
  h->n_banks = n_banks, h->list_shift = list_shift;
  for (uint32_t b = 0; b <= femk::kDenseMaxBanks; ++b) h->bank_first[b] = bank_first[b];
  // byte frequencies per 11-mer for seed_select_kernel (fem_seed_select.hip.h)
  hipLaunchKernelGGL(femk::freq11_kernel, dim3((uint32_t)h->n_cu * 8u), dim3(256), 0, 0, h->d_lookup, h->d_freq11);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipDeviceSynchronize());
  return FEM_OK;
}

// The RCCL communicator of fem_dev_allreduce_stats (one process, one handle per GPU): kept across calls.
struct CommSet {
  std::vector<int> devs;
  std::vector<ncclComm_t> comms;
  std::vector<uint64_t *> bufs;
};
std::mutex g_comm_mu;
CommSet *g_comm = nullptr;
void destroy_comm_set() {  // caller holds g_comm_mu
  if (!g_comm) return;
  for (size_t i = 0; i < g_comm->devs.size(); ++i) {
    (void)hipSetDevice(g_comm->devs[i]);
    if (g_comm->bufs[i]) (void)hipFree(g_comm->bufs[i]);
    if (g_comm->comms[i]) ncclCommDestroy(g_comm->comms[i]);
  }
  delete g_comm;
  g_comm = nullptr;
}

int check_slot(fem_dev *h, int slot) {
  if (!h) return FEM_ERR_INVALID;
  if (slot < 0 || slot >= kSlots) return fail(h, FEM_ERR_INVALID, "slot out of range");
  return FEM_OK;
}


// ---- packed read transfer: host side in fem_pack.h (the device side is unpack_reads_kernel) ----
using fempack::pack_bases;

// Device side of a packed batch that sits in the slot's pinned staging (codes | uint32 positions | bytes, fem_pack.h):
// one copy, then the characters are rebuilt in HBM byte for byte and the offset table is generated there.
int enqueue_packed(fem_dev *h, Slot &s, uint64_t n, uint32_t len, uint64_t n_exc) {
  const uint32_t bpr = fempack::bytes_per_read(len);
  const uint64_t code_bytes = fempack::code_bytes(n, len), n_bases = n * (uint64_t)len;
  const uint64_t total = code_bytes + n_exc * 5u;
  int rc;
  if ((rc = dev_realloc(h, &s.d_packed, &s.packed_cap, (size_t)total + 64))) return rc;
  if ((rc = dev_realloc(h, &s.d_bases_alloc, &s.bases_cap, kFrontPad + (size_t)n_bases + 64))) return rc;
  if ((rc = dev_realloc(h, &s.d_off, &s.off_cap, (size_t)n + 1))) return rc;
  if ((rc = dev_realloc(h, &s.d_exc_bits, &s.exc_bits_cap, (size_t)n / 32 + 2))) return rc;
  // parts: only where nothing is pending on the device (the start of a job), for batches without exceptions (their scatter
  // follows the whole batch) and of some size
  s.parts = 1;
  if (!h->no_parts && !h->no_overlap && n_exc == 0 && n >= 2 * kPartMinReads && device_idle(h))
    s.parts = (int)std::min<uint64_t>((uint64_t)h->max_parts, n / kPartMinReads);
  for (int q = 0; q <= s.parts; ++q) s.part_begin[q] = q == s.parts ? (uint32_t)n : (uint32_t)((n * q / s.parts) & ~63ull);
  HIP_TRY(h, hipMemsetAsync(s.d_exc_bits, 0, ((size_t)n / 32 + 1) * sizeof(uint32_t), s.stream));
  hipLaunchKernelGGL(femk::uniform_offsets_kernel, dim3((uint32_t)std::min<uint64_t>((n + 256) / 256, (uint64_t)h->n_cu * 8u)), dim3(256), 0,
                     s.stream, s.d_off, n, len);
  if ((rc = h2d_begin(h, s))) return rc;
  for (int q = 0; q < s.parts; ++q) {
    const uint64_t r0 = s.part_begin[q], r1 = s.part_begin[q + 1];
    // (the last part takes the codes' padding and the exceptions along)
    const uint64_t b0 = r0 * bpr, b1 = q + 1 == s.parts ? total : r1 * bpr;
    Span span(h, s, 20, s.stream);
    if (b1 > b0) HIP_TRY(h, hipMemcpyAsync(s.d_packed + b0, s.h_bases + b0, b1 - b0, hipMemcpyHostToDevice, s.stream));
    // (the link goes to the next batch's copy as soon as this batch's last byte is over — not behind the expansion kernel,
    //  which waits for a free wave slot beside the running kernels: that chained C2's batches at 2.0 ms apiece)
    if (q + 1 == s.parts && (rc = h2d_end(h, s))) return rc;
    if (r1 > r0) {
      const uint32_t grid = (uint32_t)std::min<uint64_t>(((r1 - r0) * bpr + 255) / 256, (uint64_t)h->n_cu * 16u);
      hipLaunchKernelGGL(femk::unpack_reads_kernel, dim3(grid), dim3(256), 0, s.stream, (const uint8_t *)s.d_packed + b0, r1 - r0, len, bpr,
                         s.bases() + r0 * len);
    }
    if (s.parts > 1) HIP_TRY(h, hipEventRecord(s.ev_part[q], s.stream));
  }
  if (n_exc) {
    const dim3 g((uint32_t)std::min<uint64_t>((n_exc + 255) / 256, (uint64_t)h->n_cu * 4u));
    hipLaunchKernelGGL(femk::scatter_chars_kernel, g, dim3(256), 0, s.stream, (const uint32_t *)(s.d_packed + code_bytes),
                       (const uint8_t *)(s.d_packed + code_bytes + 4u * n_exc), n_exc, s.bases());
    hipLaunchKernelGGL(femk::mark_exception_reads_kernel, g, dim3(256), 0, s.stream, (const uint32_t *)(s.d_packed + code_bytes), n_exc, len,
                       s.d_exc_bits);
  }
  s.packed_bpr = bpr;
  HIP_TRY(h, hipGetLastError());
  s.n_reads = n, s.n_bases = n_bases, s.max_len = len;
  s.staged = true, s.mapped = false, s.synced = false;
  s.h2d_bytes = total, s.sent_packed = true;
  return FEM_OK;
}

unsigned stage_threads(uint64_t n_reads) {
  unsigned want = 12;  // (measured on the GPU box's 16-core share: 8 -> 12 threads still gains, 16 does not)
  if (const char *e = getenv("FEM_STAGE_THREADS")) want = (unsigned)std::max(1, atoi(e));
  const unsigned hw = std::thread::hardware_concurrency();
  return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)(hw ? hw : 1), (uint64_t)want, n_reads / 65536 + 1}));
}

}  // namespace

extern "C" {

const char *fem_strerror(int rc) {
  switch (rc) {
    case FEM_OK: return "ok";
    case FEM_ERR_INVALID: return "invalid argument";
    case FEM_ERR_HIP: return "HIP runtime error";
    case FEM_ERR_NOMEM: return "out of memory";
    case FEM_ERR_STATE: return "call order violated";
    case FEM_ERR_UNSUPPORTED: return "input not supported by the device path";
    case FEM_ERR_RCCL: return "RCCL error";
    default: return "unknown error";
  }
}

const char *fem_dev_last_error(const fem_dev *h) { return h ? h->err.c_str() : "null handle"; }

static std::atomic<int> g_blocking_waits{0};
int fem_set_blocking_waits(int on) {
  g_blocking_waits.store(on ? 1 : 0);
  return FEM_OK;
}

int fem_dev_open(int device, fem_dev **out) {
  if (!out) return FEM_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FEM_ERR_HIP;  // no GPU: fail loudly, no fallback
  if (device < 0 || device >= n) return FEM_ERR_INVALID;
  fem_dev *h = new (std::nothrow) fem_dev();
  if (!h) return FEM_ERR_NOMEM;
  h->device = device;
  if (hipSetDevice(device) != hipSuccess) {
    delete h;
    return FEM_ERR_HIP;
  }
  // Threads that wait for the device sleep instead of spinning (fem_set_blocking_waits): FEM map has a thread per batch in
  // flight waiting most of the time, on hosts where the parser wants every core (16 M reads: 15.6 -> 13.9 cores busy).
  if (g_blocking_waits.load() || testing_switch("FEM_BLOCKING_SYNC")) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->n_cu = prop.multiProcessorCount;
  for (int i = 0; i < kSlots; ++i) {
    if (hipStreamCreateWithFlags(&h->slot[i].stream, hipStreamNonBlocking) != hipSuccess) {
      delete h;
      return FEM_ERR_HIP;
    }
  }
  bool ok_ev = hipEventCreateWithFlags(&h->ev_kernels_done, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&h->ev_select_done, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&h->ev_h2d_done, hipEventDisableTiming) == hipSuccess &&
               hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; ok_ev && i < kSlots; ++i) {
    Slot &sl = h->slot[i];
    ok_ev = hipEventCreateWithFlags(&sl.ev_zeroed, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&sl.ev_results, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&sl.ev_home, hipEventDisableTiming) == hipSuccess;
    for (int q = 0; ok_ev && q < kMaxParts; ++q)
      ok_ev = hipEventCreateWithFlags(&sl.ev_part[q], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&sl.ev_sel[q], hipEventDisableTiming) == hipSuccess;
  }
  if (!ok_ev) {
    (void)fem_dev_close(h);
    return FEM_ERR_HIP;
  }
  // Kernel-choice and buffer-size overrides exist for the parity tests and for A/B measurements only: a production process
  // does not look at its environment for them unless FEM_TESTING=1 says so.
  if (testing_switch("FEM_TESTING")) {
    h->no_overlap = testing_switch("FEM_NO_OVERLAP");
    h->no_parts = testing_switch("FEM_NO_PARTS");
    if (const char *mp = getenv("FEM_PARTS")) h->max_parts = std::min(kMaxParts, std::max(1, atoi(mp)));
    h->timeline = testing_switch("FEM_TIMELINE");
    if (h->timeline && hipEventCreate(&h->ev_epoch) == hipSuccess && hipEventRecord(h->ev_epoch, h->side_stream) == hipSuccess)
      h->timing = true, h->have_epoch = true;  // (a process that never asks for timing, FEM map, is timed from its handle's opening)
    h->force_generic = testing_switch("FEM_FORCE_GENERIC");
    h->force_hash = testing_switch("FEM_FORCE_HASH");
    h->force_dense = testing_switch("FEM_FORCE_DENSE");
    h->no_dense = testing_switch("FEM_NO_DENSE");
    h->no_strided = testing_switch("FEM_NO_STRIDED");
    h->tiny_buffers = testing_switch("FEM_TEST_TINY_BUFFERS");
    if (const char *bl = getenv("FEM_TEST_BANK_BASES")) h->bank_limit = strtoull(bl, nullptr, 10);
    if (const char *bs = getenv("FEM_TEST_BANK_SEQS")) h->bank_seqs = (uint32_t)strtoul(bs, nullptr, 10);
  }
  *out = h;
  return FEM_OK;
}

int fem_dev_close(fem_dev *h) {
  if (!h) return FEM_ERR_INVALID;
  {
    std::lock_guard<std::mutex> lock(g_comm_mu);
    if (g_comm && std::find(g_comm->devs.begin(), g_comm->devs.end(), h->device) != g_comm->devs.end()) destroy_comm_set();
  }
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  delete h->stage_pool;
  h->stage_pool = nullptr;
  for (auto &s : h->slot) {
    drain_timing(h, s);
    for (void *p : {(void *)s.d_bases_alloc, (void *)s.d_off, (void *)s.d_cand, (void *)s.d_meta, (void *)s.d_ed,
                    (void *)s.d_end, (void *)s.d_begin, (void *)s.d_count, (void *)s.d_nmap, (void *)s.d_ctl,
                    (void *)s.d_arena, (void *)s.d_slow, (void *)s.d_sel, (void *)s.d_sel_hdr, (void *)s.d_packed, (void *)s.d_exc_bits, (void *)s.d_quals, (void *)s.d_names,
                    (void *)s.d_name_off, (void *)s.d_count8, (void *)s.d_seg, (void *)s.d_pcand, (void *)s.d_ped, (void *)s.d_pend, (void *)s.d_big})
      if (p) (void)hipFree(p);
    for (void *p : {(void *)s.h_ctl, (void *)s.h_begin, (void *)s.h_count, (void *)s.h_cand, (void *)s.h_ed,
                    (void *)s.h_end, (void *)s.h_bases, (void *)s.h_off, (void *)s.h_quals, (void *)s.h_names, (void *)s.h_name_off,
                    (void *)s.h_count8, (void *)s.h_seg, (void *)s.h_pcand, (void *)s.h_ped, (void *)s.h_pend, (void *)s.h_big})
      if (p) (void)hipHostFree(p);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.text_stream) (void)hipStreamDestroy(s.text_stream);
    if (s.out_stream) (void)hipStreamDestroy(s.out_stream);
    if (s.ev_text_staged) (void)hipEventDestroy(s.ev_text_staged);
    if (s.ev_text_order) (void)hipEventDestroy(s.ev_text_order);
    delete s.tail;
    s.tail = nullptr;
  }
  if (h->d_ref_names) (void)hipFree(h->d_ref_names);
  if (h->d_ref_name_off) (void)hipFree(h->d_ref_name_off);
  for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
  if (h->ev_kernels_done) (void)hipEventDestroy(h->ev_kernels_done);
  if (h->ev_select_done) (void)hipEventDestroy(h->ev_select_done);
  if (h->ev_h2d_done) (void)hipEventDestroy(h->ev_h2d_done);
  if (h->ev_epoch) (void)hipEventDestroy(h->ev_epoch);
  if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
  for (int i = 0; i < kSlots; ++i) {
    Slot &sl = h->slot[i];
    if (sl.ev_zeroed) (void)hipEventDestroy(sl.ev_zeroed);
    if (sl.ev_results) (void)hipEventDestroy(sl.ev_results);
    if (sl.ev_home) (void)hipEventDestroy(sl.ev_home);
    for (int q = 0; q < kMaxParts; ++q) {
      if (sl.ev_part[q]) (void)hipEventDestroy(sl.ev_part[q]);
      if (sl.ev_sel[q]) (void)hipEventDestroy(sl.ev_sel[q]);
    }
  }
  for (void *p : {(void *)h->d_lookup, (void *)h->d_occ, (void *)h->d_ref_raw, (void *)h->d_seq_off,
                  (void *)h->d_seq_len, (void *)h->d_summary, (void *)h->d_planes, (void *)h->d_occ32, (void *)h->d_goff, (void *)h->d_blkseq,
                  (void *)h->d_freq11, (void *)h->d_bank_lo})
    if (p) (void)hipFree(p);
  delete h;
  return FEM_OK;
}

int fem_dev_limits(const fem_dev *h, uint32_t *max_read_len, int32_t *n_slots) {
  if (!h) return FEM_ERR_INVALID;
  if (max_read_len) *max_read_len = kMaxReadLen;
  if (n_slots) *n_slots = kSlots;
  return FEM_OK;
}

int fem_dev_upload_index(fem_dev *h, int32_t k, int32_t step, const uint32_t *lookup, uint64_t n_lookup,
                         const uint64_t *occ, uint64_t n_occ) {
  if (!h || !lookup || (!occ && n_occ)) return FEM_ERR_INVALID;
  if (k < 1 || k > 16 || step < 1) return fail(h, FEM_ERR_INVALID, "k must be 1..16 and step >= 1");
  if (n_lookup != (1ull << (2 * k)) + 1) return fail(h, FEM_ERR_INVALID, "lookup table must have 4^k + 1 entries");
  if (n_occ > 0xFFFFFFFFull) return fail(h, FEM_ERR_INVALID, "occurrence table larger than its uint32 prefix sums");
  HIP_TRY(h, hipSetDevice(h->device));
  if (h->d_lookup) (void)hipFree(h->d_lookup);
  if (h->d_occ) (void)hipFree(h->d_occ);
  h->d_lookup = nullptr, h->d_occ = nullptr;
  HIP_TRY(h, hipMalloc((void **)&h->d_lookup, n_lookup * sizeof(uint32_t)));
  HIP_TRY(h, hipMalloc((void **)&h->d_occ, std::max<uint64_t>(n_occ, 1) * sizeof(uint64_t)));
  HIP_TRY(h, hipMemcpy(h->d_lookup, lookup, n_lookup * sizeof(uint32_t), hipMemcpyHostToDevice));
  if (n_occ) HIP_TRY(h, hipMemcpy(h->d_occ, occ, n_occ * sizeof(uint64_t), hipMemcpyHostToDevice));
  h->n_lookup = n_lookup, h->n_occ = n_occ, h->k = k, h->step = step;
  int rc = refresh_summary(h);
  return rc ? rc : refresh_dense(h);
}

namespace {
// The reference as base codes 0..4 (what the bit planes and the index build are made from) in a buffer of its own, from the
// resident characters.  A temporary: nothing on the mapping path reads it, so it is not kept (3 GB at 3 Gbp).
int make_codes(fem_dev *h, uint8_t **out) {
  *out = nullptr;
  uint8_t *codes = nullptr;
  const uint64_t total = h->ref_bytes;
  HIP_TRY(h, hipMalloc((void **)&codes, total + 128));
  hipError_t e = hipMemcpy(codes, h->d_ref_raw, total, hipMemcpyDeviceToDevice);
  if (e == hipSuccess) e = hipMemset(codes + total, 4, 128);
  if (e == hipSuccess && total) {
    hipLaunchKernelGGL(femk::ref_encode_kernel, dim3((uint32_t)h->n_cu * 8u), dim3(256), 0, 0, codes, total);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  if (e != hipSuccess) {
    (void)hipFree(codes);
    HIP_TRY(h, e);
  }
  *out = codes;
  return FEM_OK;
}
}  // namespace

int fem_dev_upload_reference(fem_dev *h, uint32_t n_seq, const char *const *seq, const uint32_t *seq_len) {
  if (!h || !seq || !seq_len || n_seq == 0) return FEM_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  h->seq_off.assign(n_seq, 0);
  h->seq_len.assign(seq_len, seq_len + n_seq);
  uint64_t total = 0;
  for (uint32_t i = 0; i < n_seq; ++i) {
    h->seq_off[i] = total;
    total += seq_len[i];
  }
  for (void *p : {(void *)h->d_ref_raw, (void *)h->d_seq_off, (void *)h->d_seq_len, (void *)h->d_planes})
    if (p) (void)hipFree(p);
  h->d_ref_raw = nullptr, h->d_seq_off = nullptr, h->d_seq_len = nullptr;
  h->d_planes = nullptr;
  h->ref_bytes = 0, h->n_seq = 0;
  // 64 bytes of slack so that 4-byte window reads at the very end stay inside the allocation
  HIP_TRY(h, hipMalloc((void **)&h->d_ref_raw, total + 128));
  HIP_TRY(h, hipMalloc((void **)&h->d_seq_off, n_seq * sizeof(uint64_t)));
  HIP_TRY(h, hipMalloc((void **)&h->d_seq_len, n_seq * sizeof(uint32_t)));
  for (uint32_t i = 0; i < n_seq; ++i)
    if (seq_len[i]) HIP_TRY(h, hipMemcpy(h->d_ref_raw + h->seq_off[i], seq[i], seq_len[i], hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(h->d_seq_off, h->seq_off.data(), n_seq * sizeof(uint64_t), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(h->d_seq_len, h->seq_len.data(), n_seq * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemset(h->d_ref_raw + total, 'N', 128));
  h->ref_bytes = total, h->n_seq = n_seq;
  {  // bit planes of the codes, 64 bases of slack (code 4) included; the codes themselves are not kept
    // (a failure from here on must not leave a reference without its planes behind: fem_dev_map_staged and
    //  fem_dev_build_index test d_ref_raw)
    auto drop_reference = [&]() {
      for (void *q : {(void *)h->d_ref_raw, (void *)h->d_seq_off, (void *)h->d_seq_len, (void *)h->d_planes})
        if (q) (void)hipFree(q);
      h->d_ref_raw = nullptr, h->d_seq_off = nullptr, h->d_seq_len = nullptr, h->d_planes = nullptr;
      h->ref_bytes = 0, h->n_seq = 0;
    };
    uint8_t *codes = nullptr;
    int rc = make_codes(h, &codes);
    if (rc) {
      drop_reference();
      return rc;
    }
    const uint64_t n_pb = (total + 64 + 7) / 8;
    hipError_t e = hipMalloc((void **)&h->d_planes, femk::plane_bytes(n_pb));
    if (e == hipSuccess) e = hipMemset(h->d_planes, 0, femk::plane_bytes(n_pb));
    if (e == hipSuccess) {
      hipLaunchKernelGGL(femk::ref_planes_kernel, dim3((uint32_t)h->n_cu * 8u), dim3(256), 0, 0, codes, h->d_ref_raw, n_pb, h->d_planes);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    (void)hipFree(codes);
    if (e != hipSuccess) drop_reference();
    HIP_TRY(h, e);
  }
  return refresh_dense(h);
}

int fem_dev_build_index(fem_dev *h, int32_t k, int32_t step, uint32_t *lookup_out, uint64_t *occ_out, uint64_t occ_cap,
                        uint64_t *n_occ_out) {
  if (!h) return FEM_ERR_INVALID;
  if (!h->d_ref_raw) return fail(h, FEM_ERR_STATE, "upload the reference before building the index");
  if (k < 1 || k > 16 || step < 1) return fail(h, FEM_ERR_INVALID, "k must be 1..16 and step >= 1");
  HIP_TRY(h, hipSetDevice(h->device));
  if (h->d_lookup) (void)hipFree(h->d_lookup);
  if (h->d_occ) (void)hipFree(h->d_occ);
  h->d_lookup = nullptr, h->d_occ = nullptr;
  uint64_t n_occ = 0;
  std::string err;
  uint8_t *codes = nullptr;  // (base codes of the reference: made for the build, not kept)
  int rc = make_codes(h, &codes);
  if (rc) return rc;
  rc = femix::build_index(codes, h->seq_off, h->seq_len, k, step, h->n_cu, &h->d_lookup, &h->d_occ, &n_occ, &err);
  (void)hipFree(codes);
  if (rc != FEM_OK) return fail(h, rc, err);
  h->n_lookup = (1ull << (2 * k)) + 1, h->n_occ = n_occ, h->k = k, h->step = step;
  if ((rc = refresh_summary(h))) return rc;
  if ((rc = refresh_dense(h))) return rc;
  if (n_occ_out) *n_occ_out = n_occ;
  if (lookup_out)
    HIP_TRY(h, hipMemcpy(lookup_out, h->d_lookup, h->n_lookup * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (occ_out) {
    if (occ_cap < n_occ) return fail(h, FEM_ERR_INVALID, "occ_out too small");
    if (n_occ) HIP_TRY(h, hipMemcpy(occ_out, h->d_occ, n_occ * sizeof(uint64_t), hipMemcpyDeviceToHost));
  }
  return FEM_OK;
}

int fem_dev_fetch_index(fem_dev *h, uint32_t *lookup_out, uint64_t *occ_out, uint64_t occ_cap) {
  if (!h) return FEM_ERR_INVALID;
  if (!h->d_lookup) return fail(h, FEM_ERR_STATE, "no index is resident");
  HIP_TRY(h, hipSetDevice(h->device));
  if (lookup_out) HIP_TRY(h, hipMemcpy(lookup_out, h->d_lookup, h->n_lookup * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (occ_out) {
    if (occ_cap < h->n_occ) return fail(h, FEM_ERR_INVALID, "occ_out too small");
    if (h->n_occ) HIP_TRY(h, hipMemcpy(occ_out, h->d_occ, h->n_occ * sizeof(uint64_t), hipMemcpyDeviceToHost));
  }
  return FEM_OK;
}

int fem_dev_acquire_stage(fem_dev *h, int slot, uint64_t n_reads_cap, uint64_t n_bases_cap, char **bases, uint64_t **offsets) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  FEM_LOCK(h);
  if (!bases || !offsets) return fail(h, FEM_ERR_INVALID, "null output pointer");
  if (n_reads_cap > 0x3FFFFFF0ull) return fail(h, FEM_ERR_UNSUPPORTED, "more than 2^30 reads in one batch");
  Slot &s = h->slot[slot];
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(s.stream));  // the previous batch's copy out of these buffers is done
  if (s.out_stream) HIP_TRY(h, hipStreamSynchronize(s.out_stream));  // ... and its records and text (they read the slot's device arrays)
  drain_timing(h, s);
  if ((rc = pinned_realloc(h, &s.h_bases, &s.h_bases_cap, (size_t)n_bases_cap + 64))) return rc;
  if ((rc = pinned_realloc(h, &s.h_off, &s.h_off_cap, (size_t)n_reads_cap + 1))) return rc;
  s.staged = false, s.mapped = false, s.synced = false, s.text_staged = false;
  s.acq_reads = n_reads_cap, s.acq_bases = n_bases_cap;
  *bases = s.h_bases, *offsets = s.h_off;
  return FEM_OK;
}

int fem_dev_commit_stage(fem_dev *h, int slot, uint64_t n_reads, uint32_t max_len) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  FEM_LOCK(h);
  Slot &s = h->slot[slot];
  if (!s.h_bases || !s.h_off) return fail(h, FEM_ERR_STATE, "acquire the slot's staging buffers first");
  if (n_reads > s.acq_reads) return fail(h, FEM_ERR_INVALID, "more reads than the staging buffers were acquired for");
  if (max_len > kMaxReadLen)
    return fail(h, FEM_ERR_UNSUPPORTED, "read longer than the device path supports (" + std::to_string(kMaxReadLen) + ")");
  const uint64_t n_bases = n_reads ? s.h_off[n_reads] : 0;
  if (n_reads && (s.h_off[0] != 0 || n_bases > s.acq_bases)) return fail(h, FEM_ERR_INVALID, "staged offsets must start at 0 and end inside the buffer");
  HIP_TRY(h, hipSetDevice(h->device));
  if ((rc = dev_realloc(h, &s.d_bases_alloc, &s.bases_cap, kFrontPad + (size_t)n_bases + 64))) return rc;
  if ((rc = dev_realloc(h, &s.d_off, &s.off_cap, (size_t)n_reads + 1))) return rc;
  if ((rc = h2d_begin(h, s))) return rc;
  if (n_bases) HIP_TRY(h, hipMemcpyAsync(s.bases(), s.h_bases, n_bases, hipMemcpyHostToDevice, s.stream));
  if (n_reads) HIP_TRY(h, hipMemcpyAsync(s.d_off, s.h_off, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s.stream));
  if ((rc = h2d_end(h, s))) return rc;
  s.parts = 1;
  s.n_reads = n_reads, s.n_bases = n_bases, s.max_len = max_len;
  s.staged = true, s.mapped = false, s.synced = false;
  s.h2d_bytes = n_bases + (n_reads ? (n_reads + 1) * sizeof(uint64_t) : 0), s.sent_packed = false, s.staged_by_copy = false;
  return FEM_OK;
}

int fem_dev_commit_stage_uniform(fem_dev *h, int slot, uint64_t n_reads, uint32_t read_len) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  FEM_LOCK(h);
  Slot &s = h->slot[slot];
  if (!s.h_bases) return fail(h, FEM_ERR_STATE, "acquire the slot's staging buffers first");
  if (read_len > kMaxReadLen)
    return fail(h, FEM_ERR_UNSUPPORTED, "read longer than the device path supports (" + std::to_string(kMaxReadLen) + ")");
  const uint64_t n_bases = n_reads * (uint64_t)read_len;
  if (n_reads > s.acq_reads || n_bases > s.acq_bases) return fail(h, FEM_ERR_INVALID, "more reads than the staging buffers were acquired for");
  HIP_TRY(h, hipSetDevice(h->device));
  if ((rc = dev_realloc(h, &s.d_bases_alloc, &s.bases_cap, kFrontPad + (size_t)n_bases + 64))) return rc;
  if ((rc = dev_realloc(h, &s.d_off, &s.off_cap, (size_t)n_reads + 1))) return rc;
  if ((rc = h2d_begin(h, s))) return rc;
  if (n_bases) HIP_TRY(h, hipMemcpyAsync(s.bases(), s.h_bases, n_bases, hipMemcpyHostToDevice, s.stream));
  if ((rc = h2d_end(h, s))) return rc;
  s.parts = 1;
  hipLaunchKernelGGL(femk::uniform_offsets_kernel, dim3((uint32_t)std::min<uint64_t>((n_reads + 256) / 256, (uint64_t)h->n_cu * 8u)), dim3(256), 0,
                     s.stream, s.d_off, n_reads, read_len);
  HIP_TRY(h, hipGetLastError());
  s.n_reads = n_reads, s.n_bases = n_bases, s.max_len = read_len;
  s.staged = true, s.mapped = false, s.synced = false;
  s.h2d_bytes = n_bases, s.sent_packed = false, s.staged_by_copy = false;
  return FEM_OK;
}

int fem_dev_packed_layout(uint64_t n_reads, uint32_t read_len, uint32_t *bytes_per_read, uint64_t *exc_offset, uint64_t *exc_cap) {
  if (read_len == 0 || read_len > kMaxReadLen) return FEM_ERR_INVALID;
  const uint64_t n_bases = n_reads * (uint64_t)read_len, cb = fempack::code_bytes(n_reads, read_len);
  if (bytes_per_read) *bytes_per_read = fempack::bytes_per_read(read_len);
  if (exc_offset) *exc_offset = cb;
  // what a staging buffer acquired for n_reads * read_len characters has left behind the codes, and never more than one
  // byte in sixteen (beyond that the characters themselves are the smaller message: fem_dev_commit_stage_uniform)
  if (exc_cap) *exc_cap = std::min<uint64_t>((n_bases + 64 - std::min<uint64_t>(cb, n_bases + 64)) / 5u, n_bases / 16u);
  return FEM_OK;
}

int fem_dev_commit_stage_packed(fem_dev *h, int slot, uint64_t n_reads, uint32_t read_len, uint64_t n_exceptions) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  FEM_LOCK(h);
  Slot &s = h->slot[slot];
  if (!s.h_bases) return fail(h, FEM_ERR_STATE, "acquire the slot's staging buffers first");
  if (read_len == 0 && n_reads) return fail(h, FEM_ERR_INVALID, "packed batches hold reads of one non-zero length");
  if (read_len > kMaxReadLen)
    return fail(h, FEM_ERR_UNSUPPORTED, "read longer than the device path supports (" + std::to_string(kMaxReadLen) + ")");
  const uint64_t n_bases = n_reads * (uint64_t)read_len;
  if (n_reads > s.acq_reads || n_bases > s.acq_bases) return fail(h, FEM_ERR_INVALID, "more reads than the staging buffers were acquired for");
  if (n_bases >= 0xFFFFFFF0ull) return fail(h, FEM_ERR_UNSUPPORTED, "a packed batch holds fewer than 2^32 bases; split the batch");
  const uint64_t code_bytes = fempack::code_bytes(n_reads, read_len);
  if (code_bytes + n_exceptions * 5u > s.acq_bases + 64 || n_exceptions > n_bases / 16u)
    return fail(h, FEM_ERR_INVALID, "more exceptions than a packed batch may carry (fem_dev_packed_layout): commit the characters instead");
  // the positions are the one thing here a wrong value of which would make a kernel write outside its buffers
  const uint32_t *pos = (const uint32_t *)(s.h_bases + code_bytes);
  for (uint64_t i = 0; i < n_exceptions; ++i)
    if (pos[i] >= n_bases) return fail(h, FEM_ERR_INVALID, "exception position outside the batch");
  HIP_TRY(h, hipSetDevice(h->device));
  if ((rc = enqueue_packed(h, s, n_reads, read_len, n_exceptions))) return rc;
  s.staged_by_copy = true;  // (a quarter of the characters' bytes go out: the results may come home behind the kernels, launch_batch)
  return FEM_OK;
}

int fem_dev_stage_reads(fem_dev *h, int slot, const fem_read_batch *reads) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  if (!reads || (reads->n_reads && (!reads->bases || !reads->offsets))) return fail(h, FEM_ERR_INVALID, "null read batch");
  if (reads->n_reads > 0x3FFFFFF0ull) return fail(h, FEM_ERR_UNSUPPORTED, "more than 2^30 reads in one batch");
  const uint64_t n = reads->n_reads;
  const uint64_t base0 = n ? reads->offsets[0] : 0;
  if (n && reads->offsets[n] < base0) return fail(h, FEM_ERR_INVALID, "read offsets must be ascending");
  const uint64_t n_bases = n ? reads->offsets[n] - base0 : 0;
  char *hb = nullptr;
  uint64_t *ho = nullptr;
  if ((rc = fem_dev_acquire_stage(h, slot, n, n_bases, &hb, &ho))) return rc;
  Slot &s = h->slot[slot];
  const unsigned n_thr = stage_threads(n);
  if (!h->stage_pool) h->stage_pool = new (std::nothrow) StagePool();
  if (!h->stage_pool) return fail(h, FEM_ERR_NOMEM, "out of host memory");
  auto run_threads = [&](unsigned nt, const std::function<void(unsigned)> &work) { h->stage_pool->run(nt, work); };
  // ---- pass 1 (a few host threads): order, longest and shortest read ----
  std::vector<uint32_t> t_max(n_thr, 0), t_min(n_thr, 0xFFFFFFFFu);
  std::atomic<int> bad{0};
  run_threads(n_thr, [&](unsigned t) {
    const uint64_t r_lo = n * t / n_thr, r_hi = n * (t + 1) / n_thr;
    uint32_t mx = 0, mn = 0xFFFFFFFFu;
    for (uint64_t i = r_lo; i < r_hi; ++i) {
      const uint64_t o0 = reads->offsets[i], o1 = reads->offsets[i + 1];
      if (o1 < o0) {
        bad.store(1);
        return;
      }
      const uint64_t len = o1 - o0;
      if (len > kMaxReadLen) {
        bad.store(2);
        return;
      }
      mx = std::max<uint32_t>(mx, (uint32_t)len), mn = std::min<uint32_t>(mn, (uint32_t)len);
    }
    t_max[t] = mx, t_min[t] = mn;
  });
  if (bad.load() == 1) return fail(h, FEM_ERR_INVALID, "read offsets must be ascending");
  if (bad.load() == 2)
    return fail(h, FEM_ERR_UNSUPPORTED, "read longer than the device path supports (" + std::to_string(kMaxReadLen) + ")");
  uint32_t max_len = 0, min_len = 0xFFFFFFFFu;
  for (unsigned t = 0; t < n_thr; ++t) max_len = std::max(max_len, t_max[t]), min_len = std::min(min_len, t_min[t]);
  // ---- reads of one length: two bits per base cross the link instead of eight ----
  // (test hook, read per batch on purpose: tests/test_gpu_packed.py switches it between two batches of one handle; one
  // getenv per batch of >= 10^4 reads is not measurable)
  const bool no_pack = testing_switch("FEM_NO_PACK");
  if (n && min_len == max_len && max_len > 0 && n_bases < 0xFFFFFFF0ull && !no_pack) {
    const uint32_t len = max_len, bpr = (len + 3u) / 4u;
    const uint64_t code_bytes = (n * bpr + 7u) & ~7ull;
    const uint64_t exc_cap = (n_bases + 64 - std::min<uint64_t>(code_bytes, n_bases + 64)) / 5u;  // what is left of the staging buffer
    std::vector<std::vector<uint64_t>> exc(n_thr);
    const uint8_t *src = (const uint8_t *)reads->bases + base0;
    // (pieces of 32 Ki reads handed out as the threads ask for them: one thread held up by the host's other tenants
    // holds up its piece, not a twelfth of the batch)
    constexpr uint64_t kPiece = 32768;
    std::atomic<uint64_t> next_piece{0};
    run_threads(n_thr, [&](unsigned t) {
      for (;;) {
        const uint64_t r_lo = next_piece.fetch_add(1, std::memory_order_relaxed) * kPiece;
        if (r_lo >= n) break;
        const uint64_t r_hi = std::min(n, r_lo + kPiece);
        if ((len & 3u) == 0u) {  // no padding: the piece's reads are one stream
          pack_bases(src + r_lo * len, (r_hi - r_lo) * len, (uint8_t *)hb + r_lo * bpr, r_lo * len, exc[t]);
        } else {
          for (uint64_t i = r_lo; i < r_hi; ++i) pack_bases(src + i * len, len, (uint8_t *)hb + i * bpr, i * len, exc[t]);
        }
      }
    });
    uint64_t n_exc = 0;
    for (const auto &v : exc) n_exc += v.size();
    if (n_exc <= exc_cap && n_exc <= n_bases / 16u) {  // (more than that and the characters themselves are the smaller message)
      // codes | positions (uint32 each) | the bytes that belong there
      uint32_t *he = (uint32_t *)(hb + code_bytes);
      uint8_t *hc = (uint8_t *)(he + n_exc);
      for (const auto &v : exc)
        for (uint64_t x : v) *he++ = (uint32_t)(x >> 8), *hc++ = (uint8_t)x;
      {
        FEM_LOCK(h);
        if ((rc = enqueue_packed(h, s, n, len, n_exc))) return rc;
      }
      s.staged_by_copy = true;
      return FEM_OK;
    }
  }
  // ---- any other batch: the characters and the offsets as they are ----
  run_threads(n_thr, [&](unsigned t) {
    const uint64_t r_lo = n * t / n_thr, r_hi = n * (t + 1) / n_thr;
    for (uint64_t i = r_lo; i < r_hi; ++i) ho[i] = reads->offsets[i] - base0;
    if (r_hi == n) ho[n] = reads->offsets[n] - base0;
    if (r_hi > r_lo) memcpy(hb + (reads->offsets[r_lo] - base0), reads->bases + reads->offsets[r_lo], reads->offsets[r_hi] - reads->offsets[r_lo]);
  });
  if (n == 0) ho[0] = 0;
  rc = fem_dev_commit_stage(h, slot, n, max_len);
  s.staged_by_copy = true;
  return rc;
}

int fem_dev_stage_info(fem_dev *h, int slot, uint64_t *h2d_bytes, int32_t *packed) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  if (h2d_bytes) *h2d_bytes = h->slot[slot].h2d_bytes;
  if (packed) *packed = h->slot[slot].sent_packed ? 1 : 0;
  return FEM_OK;
}

int fem_dev_map_staged(fem_dev *h, int slot, const fem_params *p) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  FEM_LOCK(h);
  if (!params_ok(p)) return fail(h, FEM_ERR_INVALID, "parameters out of range (k 1..16, step 1..16, e 0..7, a 0..2)");
  if (!h->d_lookup || !h->d_ref_raw || !h->d_planes) return fail(h, FEM_ERR_STATE, "index and reference must be uploaded first");
  if (p->k != h->k) return fail(h, FEM_ERR_INVALID, "k differs from the uploaded index");
  Slot &s = h->slot[slot];
  if (!s.staged) return fail(h, FEM_ERR_STATE, "no reads staged in this slot");
  HIP_TRY(h, hipSetDevice(h->device));
  s.params = *p;
  if ((rc = ensure_outputs(h, s))) return rc;
  return launch_batch(h, s);
}

int fem_dev_sync(fem_dev *h, int slot) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  Slot &s = h->slot[slot];
  if (!s.mapped) return fail(h, FEM_ERR_STATE, "nothing was mapped in this slot");
  HIP_TRY(h, hipSetDevice(h->device));  // the callers (fetch, fetch_records) allocate and launch on this device
  if (s.synced) return FEM_OK;
  for (int attempt = 0; attempt < 8; ++attempt) {
    HIP_TRY(h, hipStreamSynchronize(s.stream));  // (without the handle's lock: other slots' calls go on meanwhile)
    FEM_LOCK(h);
    drain_timing(h, s);
    const uint32_t *ctr = (const uint32_t *)s.h_ctl;
    const uint64_t *arena_ctr = (const uint64_t *)(s.h_ctl + 4 * sizeof(uint32_t));
    const uint64_t *st = (const uint64_t *)(s.h_ctl + 4 * sizeof(uint32_t) + 2 * sizeof(uint64_t));
    uint32_t flags = ctr[1];
    if (flags & femk::kFlagTooLarge)
      return fail(h, FEM_ERR_UNSUPPORTED, "a read selects more than 2^31 occurrence entries in one seed group");
    if (flags == 0) {
      s.n_cand = ctr[0];
      s.n_packed = ((const uint32_t *)(s.h_ctl + kCtlPackCursor))[0], s.n_big = ((const uint32_t *)(s.h_ctl + kCtlPackCursor))[1];
      s.stats[0] = s.n_reads;
      s.stats[1] = st[3];
      s.stats[2] = st[0];
      s.stats[3] = st[1];
      s.stats[4] = st[2];
      s.synced = true;
      return FEM_OK;
    }
    // scratch too small: grow exactly what was asked for and run the batch again
    if (flags & femk::kFlagCandOverflow) {
      uint64_t want = (uint64_t)ctr[0] + ctr[0] / 8 + 1024;
      if ((rc = grow_candidates(h, s, std::min<uint64_t>(want, 0xFFFFFFF0ull)))) return rc;
    }
    if (flags & femk::kFlagQueueOverflow) {
      uint64_t want = (uint64_t)ctr[2] + ctr[2] / 8 + 4096;
      (void)hipFree(s.d_slow);
      s.d_slow = nullptr;
      s.slow_cap = 0;
      hipError_t e = hipMalloc((void **)&s.d_slow, want * sizeof(uint32_t));
      if (e != hipSuccess) return fail(h, FEM_ERR_NOMEM, "slow-read queue does not fit in device memory");
      s.slow_cap = (uint32_t)std::min<uint64_t>(want, 0xFFFFFFF0ull);
    }
    if (flags & femk::kFlagArenaOverflow) {
      uint64_t want = arena_ctr[1] + arena_ctr[1] / 8 + 1024;
      (void)hipFree(s.d_arena);
      s.d_arena = nullptr;
      hipError_t e = hipMalloc((void **)&s.d_arena, want * sizeof(uint64_t));
      if (e != hipSuccess) {
        s.arena_cap = 0;
        return fail(h, FEM_ERR_NOMEM, "scratch arena for oversized seed lists does not fit in device memory; use smaller batches");
      }
      s.arena_cap = want;
    }
    if ((rc = launch_batch(h, s))) return rc;
  }
  return fail(h, FEM_ERR_HIP, "batch kept overflowing its scratch buffers");
}

int fem_dev_fetch_stats(fem_dev *h, int slot, uint64_t stats[5]) {
  int rc = fem_dev_sync(h, slot);
  if (rc) return rc;
  if (stats) memcpy(stats, h->slot[slot].stats, sizeof(uint64_t) * 5);
  h->slot[slot].prefetch_results = false;  // this caller takes the counters only
  return FEM_OK;
}

int fem_dev_fetch(fem_dev *h, int slot, fem_batch_result *out) {
  int rc = fem_dev_sync(h, slot);
  if (rc) return rc;
  if (!out) return fail(h, FEM_ERR_INVALID, "null result");
  Slot &s = h->slot[slot];
  const size_t n2 = (size_t)s.n_reads * 2, nc = s.n_cand;
  bool regrown = false;
  if (n2 > s.h_per_read_cap || !s.h_begin) {
    regrown = true;
    size_t c0 = 0, c1 = 0;
    if (s.h_begin) (void)hipHostFree(s.h_begin);
    if (s.h_count) (void)hipHostFree(s.h_count);
    s.h_begin = s.h_count = nullptr;
    if ((rc = pinned_realloc(h, &s.h_begin, &c0, std::max<size_t>(n2, 2)))) return rc;
    if ((rc = pinned_realloc(h, &s.h_count, &c1, std::max<size_t>(n2, 2)))) return rc;
    s.h_per_read_cap = c0;
  }
  if (nc > s.h_cand_cap || !s.h_cand) {
    regrown = true;
    size_t c0 = 0, c1 = 0, c2 = 0;
    for (void *p : {(void *)s.h_cand, (void *)s.h_ed, (void *)s.h_end})
      if (p) (void)hipHostFree(p);
    s.h_cand = nullptr, s.h_ed = nullptr, s.h_end = nullptr;
    size_t want = std::max<size_t>(nc + nc / 4, 1024);
    if ((rc = pinned_realloc(h, &s.h_cand, &c0, want))) return rc;
    if ((rc = pinned_realloc(h, &s.h_ed, &c1, want))) return rc;
    if ((rc = pinned_realloc(h, &s.h_end, &c2, want))) return rc;
    s.h_cand_cap = c0;
  }
  // what did not come home behind the kernels already (launch_batch; nothing did if a buffer was regrown just now)
  const bool per_read_home = !regrown && s.prefetched_reads2 == n2;
  const size_t c_from = regrown ? 0 : std::min<size_t>(s.prefetched_cand, nc);
  if (n2 && !per_read_home) {
    HIP_TRY(h, hipMemcpyAsync(s.h_begin, s.d_begin, n2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(h, hipMemcpyAsync(s.h_count, s.d_count, n2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
  }
  if (nc > c_from) {
    HIP_TRY(h, hipMemcpyAsync(s.h_cand + c_from, s.d_cand + c_from, (nc - c_from) * sizeof(uint64_t), hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(h, hipMemcpyAsync(s.h_ed + c_from, s.d_ed + c_from, (nc - c_from) * sizeof(uint8_t), hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(h, hipMemcpyAsync(s.h_end + c_from, s.d_end + c_from, (nc - c_from) * sizeof(int16_t), hipMemcpyDeviceToHost, s.stream));
  }
  if ((n2 && !per_read_home) || nc > c_from) HIP_TRY(h, hipStreamSynchronize(s.stream));
  s.prefetch_results = true, s.last_n_cand = nc, s.want_packed = false;
  out->n_reads = s.n_reads;
  out->n_candidates = nc;
  out->cand_begin = s.h_begin, out->cand_count = s.h_count;
  out->cand = s.h_cand, out->ed = s.h_ed, out->end = s.h_end;
  memcpy(out->stats, s.stats, sizeof s.stats);
  return FEM_OK;
}

int fem_dev_fetch_packed(fem_dev *h, int slot, fem_batch_packed *out) {
  int rc = fem_dev_sync(h, slot);
  if (rc) return rc;
  if (!out) return fail(h, FEM_ERR_INVALID, "null result");
  Slot &s = h->slot[slot];
  const size_t n2 = (size_t)s.n_reads * 2, n_seg = (n2 + 255) / 256;
  if (!s.packed_enqueued && s.n_reads) {  // the slot's first packed fetch: pack now; from the next batch on it happens behind the kernels
    if ((rc = enqueue_pack(h, s, true, false))) return rc;
    HIP_TRY(h, hipMemcpyAsync(s.h_ctl, s.d_ctl, kCtlBytes, hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(h, hipStreamSynchronize(s.stream));
    s.n_packed = ((const uint32_t *)(s.h_ctl + kCtlPackCursor))[0], s.n_big = ((const uint32_t *)(s.h_ctl + kCtlPackCursor))[1];
  }
  if (s.n_big > kBigCap) return fail(h, FEM_ERR_UNSUPPORTED, "more strands with 255 candidates and more than a packed result lists: use fem_dev_fetch");
  const size_t np = s.n_reads ? s.n_packed : 0;
  bool regrown = false;
  if (np > s.h_pcand_cap || !s.h_pcand) {
    regrown = true;
    for (void *q : {(void *)s.h_pcand, (void *)s.h_ped, (void *)s.h_pend})
      if (q) (void)hipHostFree(q);
    s.h_pcand = nullptr, s.h_ped = nullptr, s.h_pend = nullptr;
    size_t c0 = 0, c1 = 0, c2 = 0;
    const size_t want = std::max<size_t>(np + np / 4, 1024);
    if ((rc = pinned_realloc(h, &s.h_pcand, &c0, want))) return rc;
    if ((rc = pinned_realloc(h, &s.h_ped, &c1, want))) return rc;
    if ((rc = pinned_realloc(h, &s.h_pend, &c2, want))) return rc;
    s.h_pcand_cap = c0;
  }
  bool copied = false;
  if (n2 && !s.packed_per_read_home) {
    HIP_TRY(h, hipMemcpyAsync(s.h_count8, s.d_count8, n2, hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(h, hipMemcpyAsync(s.h_seg, s.d_seg, n_seg * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
    s.packed_per_read_home = true, copied = true;
  }
  const size_t c_from = regrown ? 0 : std::min<size_t>(s.packed_home, np);
  if (np > c_from) {
    HIP_TRY(h, hipMemcpyAsync(s.h_pcand + c_from, s.d_pcand + c_from, (np - c_from) * sizeof(uint64_t), hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(h, hipMemcpyAsync(s.h_ped + c_from, s.d_ped + c_from, (np - c_from) * sizeof(uint8_t), hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(h, hipMemcpyAsync(s.h_pend + c_from, s.d_pend + c_from, (np - c_from) * sizeof(int16_t), hipMemcpyDeviceToHost, s.stream));
    s.packed_home = np, copied = true;
  }
  if (s.n_big) {
    HIP_TRY(h, hipMemcpyAsync(s.h_big, s.d_big, (size_t)s.n_big * sizeof(uint2), hipMemcpyDeviceToHost, s.stream));
    copied = true;
  }
  if (copied) HIP_TRY(h, hipStreamSynchronize(s.stream));
  s.want_packed = true, s.last_n_packed = np;
  out->n_reads = s.n_reads;
  out->n_candidates = np;
  out->count = s.h_count8, out->seg_begin = s.h_seg;
  out->cand = s.h_pcand, out->ed = s.h_ped, out->end = s.h_pend;
  out->big = (const uint32_t *)s.h_big, out->n_big = s.n_big;
  memcpy(out->stats, s.stats, sizeof s.stats);
  return FEM_OK;
}

int fem_dev_fetch_records(fem_dev *h, int slot, fem_batch_records *out) {
  int rc = fem_dev_sync(h, slot);
  if (rc) return rc;
  if (!out) return fail(h, FEM_ERR_INVALID, "null result");
  Slot &s = h->slot[slot];
  s.prefetch_results = false;  // this caller takes records, not the per-candidate arrays
  if (!s.tail) s.tail = new (std::nothrow) femt::Tail();
  if (!s.tail) return fail(h, FEM_ERR_NOMEM, "out of host memory");
  femt::TailInput in{};
  in.bases = s.bases(), in.read_off = s.d_off, in.n_reads = (uint32_t)s.n_reads, in.max_len = s.max_len;
  in.ref_raw = h->d_ref_raw, in.ref_bytes = h->ref_bytes + 64, in.seq_off = h->d_seq_off;
  in.planes = h->d_planes;
  if (s.sent_packed) in.packed = s.d_packed, in.packed_bpr = s.packed_bpr, in.exc_bits = s.d_exc_bits;
  in.cand = s.d_cand, in.ed = s.d_ed, in.end = s.d_end, in.cand_begin = s.d_begin, in.cand_count = s.d_count;
  in.n_map = s.d_nmap, in.e = s.params.e, in.n_records = s.stats[4];
  femt::TailOutput t{};
  double ms[3] = {0, 0, 0};
  std::string err;
  if (s.out_stream) HIP_TRY(h, hipStreamSynchronize(s.out_stream));  // (a SAM text of this slot still being made reads the tail's arrays)
  rc = s.tail->run(in, s.stream, h->n_cu, h->tiny_buffers, &t, &err, h->timing ? ms : nullptr);
  if (rc) return fail(h, rc, err);
  if (h->timing)
    for (int i = 0; i < 3; ++i) h->t_ms[3 + i] += ms[i], h->t_n[3 + i] += 1;
  out->n_reads = t.n_reads, out->n_records = t.n_records;
  out->rec_begin = t.rec_begin, out->flag = t.flag, out->tid = t.tid, out->pos0 = t.pos0, out->nm = t.nm;
  out->cigar_off = t.cigar_off, out->cigar = t.cigar, out->md_off = t.md_off, out->md = t.md;
  memcpy(out->stats, s.stats, sizeof s.stats);
  return FEM_OK;
}

int fem_dev_upload_reference_names(fem_dev *h, uint32_t n_seq, const char *names, const uint64_t *name_off) {
  if (!h || !names || !name_off || n_seq == 0) return FEM_ERR_INVALID;
  if (h->n_seq && n_seq != h->n_seq) return fail(h, FEM_ERR_INVALID, "as many names as reference sequences, please");
  if (name_off[n_seq] < name_off[0] || name_off[n_seq] - name_off[0] > 0xFFFFFFF0ull) return fail(h, FEM_ERR_INVALID, "name offsets out of range");
  HIP_TRY(h, hipSetDevice(h->device));
  std::vector<uint32_t> off(n_seq + 1);
  for (uint32_t i = 0; i <= n_seq; ++i) {
    if (name_off[i] < name_off[0] || (i && name_off[i] < name_off[i - 1])) return fail(h, FEM_ERR_INVALID, "name offsets must be ascending");
    off[i] = (uint32_t)(name_off[i] - name_off[0]);
  }
  if (h->d_ref_names) (void)hipFree(h->d_ref_names);
  if (h->d_ref_name_off) (void)hipFree(h->d_ref_name_off);
  h->d_ref_names = nullptr, h->d_ref_name_off = nullptr;
  HIP_TRY(h, hipMalloc((void **)&h->d_ref_names, std::max<size_t>(off[n_seq], 1)));
  HIP_TRY(h, hipMalloc((void **)&h->d_ref_name_off, (n_seq + 1) * sizeof(uint32_t)));
  if (off[n_seq]) HIP_TRY(h, hipMemcpy(h->d_ref_names, names + name_off[0], off[n_seq], hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(h->d_ref_name_off, off.data(), (n_seq + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
  return FEM_OK;
}

int fem_dev_acquire_text_stage(fem_dev *h, int slot, uint64_t n_reads_cap, uint64_t n_bases_cap, uint64_t n_name_bytes_cap, char **quals,
                               char **names, uint64_t **name_off) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  FEM_LOCK(h);
  if (!quals || !names || !name_off) return fail(h, FEM_ERR_INVALID, "null output pointer");
  Slot &s = h->slot[slot];
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(s.stream));  // the previous batch's copies out of these buffers are done
  if (s.text_stream) HIP_TRY(h, hipStreamSynchronize(s.text_stream));
  if (s.out_stream) HIP_TRY(h, hipStreamSynchronize(s.out_stream));
  if ((rc = pinned_realloc(h, &s.h_quals, &s.h_quals_cap, (size_t)n_bases_cap + 64))) return rc;
  if ((rc = pinned_realloc(h, &s.h_names, &s.h_names_cap, (size_t)n_name_bytes_cap + 64))) return rc;
  if ((rc = pinned_realloc(h, &s.h_name_off, &s.h_name_off_cap, (size_t)n_reads_cap + 1))) return rc;
  s.text_staged = false;
  *quals = s.h_quals, *names = s.h_names, *name_off = s.h_name_off;
  return FEM_OK;
}

int fem_dev_reserve_text(fem_dev *h, int slot, uint64_t n_reads, uint64_t n_bases, uint64_t n_name_bytes, uint64_t text_bytes) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  FEM_LOCK(h);
  Slot &s = h->slot[slot];
  HIP_TRY(h, hipSetDevice(h->device));
  if ((rc = dev_realloc(h, &s.d_bases_alloc, &s.bases_cap, kFrontPad + (size_t)n_bases + 64))) return rc;
  if ((rc = dev_realloc(h, &s.d_off, &s.off_cap, (size_t)n_reads + 1))) return rc;
  if ((rc = dev_realloc(h, &s.d_quals, &s.d_quals_cap, (size_t)n_bases + 64))) return rc;
  if ((rc = dev_realloc(h, &s.d_names, &s.d_names_cap, (size_t)n_name_bytes + 64))) return rc;
  if ((rc = dev_realloc(h, &s.d_name_off, &s.d_name_off_cap, (size_t)n_reads + 1))) return rc;
  if (!s.tail) s.tail = new (std::nothrow) femt::Tail();
  if (!s.tail) return fail(h, FEM_ERR_NOMEM, "out of host memory");
  std::string err;
  if ((rc = s.tail->reserve_text(text_bytes, &err))) return fail(h, rc, err);
  return FEM_OK;
}

// The stream a batch's records and text are made on.  The mapping has been waited for by then (fem_dev_sync), so nothing ties
// this work to the slot's own stream — and on that stream, at the same priority as every other slot's, the small kernels of a
// batch on its way OUT queue behind the joins of the batches coming IN (a join's persistent blocks fill the chip): with more
// batches in flight each took longer, finished out of order and kept its slot (FEM map, 8 slots: a batch's records 56 ms after
// its submission).  At high priority the oldest batch's kernels take the first resources that come free.
static hipStream_t out_stream_of(fem_dev *h, Slot &s) {
  static const bool plain = testing_switch("FEM_NO_OUT_PRIORITY");  // (A/B)
  if (plain) return s.stream;
  if (!s.out_stream) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
        hipStreamCreateWithPriority(&s.out_stream, hipStreamNonBlocking, greatest) != hipSuccess)
      s.out_stream = nullptr;
  }
  (void)h;
  return s.out_stream ? s.out_stream : s.stream;
}

int fem_dev_reserve_batch(fem_dev *h, int slot, uint64_t n_reads, uint64_t n_records, uint32_t max_len, const fem_params *p) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  FEM_LOCK(h);
  if (!params_ok(p)) return fail(h, FEM_ERR_INVALID, "parameters out of range (k 1..16, step 1..16, e 0..7, a 0..2)");
  if (!h->d_lookup) return fail(h, FEM_ERR_STATE, "the index must be uploaded first (what a batch needs depends on it)");
  if (n_reads == 0 || n_reads > 0x7FFFFFF0ull || n_records > 0xFFFFFFF0ull || max_len == 0 || max_len > kMaxReadLen)
    return fail(h, FEM_ERR_INVALID, "batch shape out of range");
  Slot &s = h->slot[slot];
  if (s.mapped && !s.synced) return fail(h, FEM_ERR_STATE, "a batch is in flight in this slot");
  HIP_TRY(h, hipSetDevice(h->device));
  // the mapping's own arrays: per read, per candidate (ensure_outputs sizes them by the slot's read count), the packed form's
  const uint64_t n_was = s.n_reads;
  s.n_reads = n_reads;
  rc = ensure_outputs(h, s);
  s.n_reads = n_was;
  if (rc) return rc;
  const uint64_t n_bases = n_reads * (uint64_t)max_len;
  if ((rc = dev_realloc(h, &s.d_packed, &s.packed_cap, (size_t)fempack::code_bytes(n_reads, max_len) + 64))) return rc;
  if ((rc = dev_realloc(h, &s.d_bases_alloc, &s.bases_cap, kFrontPad + (size_t)n_bases + 64))) return rc;
  if ((rc = dev_realloc(h, &s.d_off, &s.off_cap, (size_t)n_reads + 1))) return rc;
  if ((rc = dev_realloc(h, &s.d_exc_bits, &s.exc_bits_cap, (size_t)n_reads / 32 + 2))) return rc;
  if (h->d_occ32 && h->d_freq11 && p->k == 12 && p->step == 3) {  // dense index: the selection's hand-over to the join
    const size_t R = (size_t)(p->e + 1 + p->a);
    if ((rc = dev_realloc(h, &s.d_sel, &s.sel_cap, (size_t)n_reads * 6u * R * h->n_banks))) return rc;
    if ((rc = dev_realloc(h, &s.d_sel_hdr, &s.sel_hdr_cap, (size_t)n_reads))) return rc;
  }
  // the mapping tail and the SAM text's bookkeeping
  if (!s.tail) s.tail = new (std::nothrow) femt::Tail();
  if (!s.tail) return fail(h, FEM_ERR_NOMEM, "out of host memory");
  std::string err;
  if ((rc = s.tail->reserve((uint32_t)n_reads, (uint32_t)n_records, max_len, p->e, h->tiny_buffers, &err))) return fail(h, rc, err);
  // first uses: the slot's streams (a stream's queue is made by its first command) and the tail's code object
  if (!s.text_stream) HIP_TRY(h, hipStreamCreateWithFlags(&s.text_stream, hipStreamNonBlocking));
  HIP_TRY(h, hipMemsetAsync(s.d_sel_hdr ? (void *)s.d_sel_hdr : (void *)s.d_off, 0, 8, s.text_stream));
  HIP_TRY(h, hipStreamSynchronize(s.text_stream));
  HIP_TRY(h, hipMemsetAsync(s.d_off, 0, 8, h->side_stream));
  HIP_TRY(h, hipStreamSynchronize(h->side_stream));
  // ... and the copy paths a batch takes, in both directions, from and to the buffers it will use
  const size_t probe = 1u << 20;
  if (s.h_bases && s.h_bases_cap >= probe && s.bases_cap >= kFrontPad + probe)
    HIP_TRY(h, hipMemcpyAsync(s.d_bases_alloc + kFrontPad, s.h_bases, probe, hipMemcpyHostToDevice, s.stream));
  if (s.h_quals && s.d_quals && s.h_quals_cap >= probe && s.d_quals_cap >= probe)
    HIP_TRY(h, hipMemcpyAsync(s.d_quals, s.h_quals, probe, hipMemcpyHostToDevice, s.text_stream));
  HIP_TRY(h, hipMemcpyAsync(s.h_ctl, s.d_ctl, kCtlBytes, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(h, hipStreamSynchronize(s.text_stream));
  HIP_TRY(h, hipStreamSynchronize(s.stream));
  if ((rc = s.tail->warm(out_stream_of(h, s), &err))) return fail(h, rc, err);
  return FEM_OK;
}

static int commit_text(fem_dev *h, int slot, uint64_t n_reads, uint64_t n_name_bytes, bool with_quals) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  FEM_LOCK(h);
  Slot &s = h->slot[slot];
  if (!s.staged) return fail(h, FEM_ERR_STATE, "commit the reads of the batch first");
  if ((with_quals && !s.h_quals) || !s.h_names || !s.h_name_off) return fail(h, FEM_ERR_STATE, "acquire the slot's text staging buffers first");
  if (n_reads != s.n_reads) return fail(h, FEM_ERR_INVALID, "as many names as reads, please");
  if ((with_quals && s.n_bases + 64 > s.h_quals_cap) || n_name_bytes + 64 > s.h_names_cap || n_reads + 1 > s.h_name_off_cap)
    return fail(h, FEM_ERR_INVALID, "more qualities or names than the text staging buffers were acquired for");
  if (n_reads && (s.h_name_off[0] != 0 || s.h_name_off[n_reads] != n_name_bytes))
    return fail(h, FEM_ERR_INVALID, "name offsets must start at 0 and end at the number of name bytes");
  HIP_TRY(h, hipSetDevice(h->device));
  if (with_quals && (rc = dev_realloc(h, &s.d_quals, &s.d_quals_cap, (size_t)s.n_bases + 64))) return rc;
  if ((rc = dev_realloc(h, &s.d_names, &s.d_names_cap, (size_t)n_name_bytes + 64))) return rc;
  if ((rc = dev_realloc(h, &s.d_name_off, &s.d_name_off_cap, (size_t)n_reads + 1))) return rc;
  // On a stream of their own: nothing before the SAM text reads them, and on the slot's stream these copies (1.25 times the
  // characters of the reads: 2.5 ms per million 100-bp reads) stood between the batch's reads and its first kernel.
  if (!s.text_stream) HIP_TRY(h, hipStreamCreateWithFlags(&s.text_stream, hipStreamNonBlocking));
  if (!s.ev_text_staged) HIP_TRY(h, hipEventCreateWithFlags(&s.ev_text_staged, hipEventDisableTiming));
  // (the slot's previous batch may still be rendering its text out of the same arrays)
  if (s.have_text_order) HIP_TRY(h, hipStreamWaitEvent(s.text_stream, s.ev_text_order, 0));
  if (with_quals && s.n_bases) HIP_TRY(h, hipMemcpyAsync(s.d_quals, s.h_quals, s.n_bases, hipMemcpyHostToDevice, s.text_stream));
  if (n_name_bytes) HIP_TRY(h, hipMemcpyAsync(s.d_names, s.h_names, n_name_bytes, hipMemcpyHostToDevice, s.text_stream));
  HIP_TRY(h, hipMemcpyAsync(s.d_name_off, s.h_name_off, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s.text_stream));
  HIP_TRY(h, hipEventRecord(s.ev_text_staged, s.text_stream));
  s.text_staged = true, s.host_quals = !with_quals, s.qual_at = nullptr;
  return FEM_OK;
}

int fem_dev_commit_text_stage(fem_dev *h, int slot, uint64_t n_reads, uint64_t n_name_bytes) { return commit_text(h, slot, n_reads, n_name_bytes, true); }
// Names only: the qualities stay where the caller has them (228 of the 473 bytes per read that cross the link with the device's
// SAM text are the qualities going up and coming back unchanged).  The text then leaves the QUAL field of every read's first
// record unwritten, and fem_dev_sam_quals says where each read's field is.
int fem_dev_commit_names_stage(fem_dev *h, int slot, uint64_t n_reads, uint64_t n_name_bytes) { return commit_text(h, slot, n_reads, n_name_bytes, false); }

int fem_dev_sam_quals(fem_dev *h, int slot, const uint64_t **qual_at, uint64_t *n_reads) {
  int rc = check_slot(h, slot);
  if (rc) return rc;
  FEM_LOCK(h);
  Slot &s = h->slot[slot];
  if (!qual_at) return fail(h, FEM_ERR_INVALID, "null output pointer");
  if (!s.host_quals || !s.qual_at) return fail(h, FEM_ERR_STATE, "the slot's last SAM text was rendered with its qualities (or there is none)");
  *qual_at = s.qual_at;
  if (n_reads) *n_reads = s.n_reads;
  return FEM_OK;
}

static int fetch_sam(fem_dev *h, int slot, fem_batch_sam *out, bool wait);
int fem_dev_fetch_sam(fem_dev *h, int slot, fem_batch_sam *out) { return fetch_sam(h, slot, out, true); }
int fem_dev_fetch_sam_nowait(fem_dev *h, int slot, fem_batch_sam *out) { return fetch_sam(h, slot, out, false); }
int fem_dev_sam_wait(fem_dev *h, int slot) {
  if (!h || slot < 0 || slot >= kSlots) return FEM_ERR_INVALID;
  Slot &s = h->slot[slot];
  if (!s.tail) return FEM_ERR_STATE;
  if (hipSetDevice(h->device) != hipSuccess) return FEM_ERR_HIP;
  return s.tail->wait_text();  // (touches nothing but the slot's event: safe next to the thread that drives the handle)
}

static int fetch_sam(fem_dev *h, int slot, fem_batch_sam *out, bool wait) {
  static const bool trace_host = testing_switch("FEM_FETCH_TIMES");  // host time of the call's three stretches, on stderr
  const auto t_in = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  int rc = fem_dev_sync(h, slot);
  if (rc) return rc;
  const double ms_sync = since(t_in);
  if (!out) return fail(h, FEM_ERR_INVALID, "null result");
  Slot &s = h->slot[slot];
  if (!s.text_staged) return fail(h, FEM_ERR_STATE, "qualities and names of this batch were not committed (fem_dev_commit_text_stage)");
  if (!h->d_ref_names) return fail(h, FEM_ERR_STATE, "reference names must be uploaded first (fem_dev_upload_reference_names)");
  s.prefetch_results = false;
  if (!s.tail) s.tail = new (std::nothrow) femt::Tail();
  if (!s.tail) return fail(h, FEM_ERR_NOMEM, "out of host memory");
  femt::TailInput in{};
  in.bases = s.bases(), in.read_off = s.d_off, in.n_reads = (uint32_t)s.n_reads, in.max_len = s.max_len;
  in.ref_raw = h->d_ref_raw, in.ref_bytes = h->ref_bytes + 64, in.seq_off = h->d_seq_off;
  in.planes = h->d_planes;
  if (s.sent_packed) in.packed = s.d_packed, in.packed_bpr = s.packed_bpr, in.exc_bits = s.d_exc_bits;
  in.cand = s.d_cand, in.ed = s.d_ed, in.end = s.d_end, in.cand_begin = s.d_begin, in.cand_count = s.d_count;
  in.n_map = s.d_nmap, in.e = s.params.e, in.n_records = s.stats[4];
  femt::TailOutput t{};
  double ms[3] = {0, 0, 0}, ms_text = 0;
  std::string err;
  hipStream_t os = out_stream_of(h, s);
  rc = s.tail->run(in, os, h->n_cu, h->tiny_buffers, &t, &err, h->timing ? ms : nullptr, false);
  if (rc) return fail(h, rc, err);
  const double ms_run = since(t_in);
  femt::SamInput names{};
  names.quals = s.host_quals ? nullptr : s.d_quals, names.names = s.d_names, names.name_off = s.d_name_off;
  names.qual_hole = s.host_quals;
  names.ref_names = h->d_ref_names, names.ref_name_off = h->d_ref_name_off;
  femt::SamOutput text{};
  HIP_TRY(h, hipStreamWaitEvent(os, s.ev_text_staged, 0));  // qualities and names came on the slot's text stream
  rc = s.tail->sam(in, names, os, h->n_cu, &text, &err, h->timing ? &ms_text : nullptr, wait, &h->text_gate);
  if (rc) return fail(h, rc, err);
  if (!s.ev_text_order) HIP_TRY(h, hipEventCreateWithFlags(&s.ev_text_order, hipEventDisableTiming));
  HIP_TRY(h, hipEventRecord(s.ev_text_order, os));
  s.have_text_order = true;
  if (trace_host) fprintf(stderr, "[fetch_sam] slot %d: mapping synced after %.2f ms, records %.2f, text sized and queued %.2f\n", slot, ms_sync, ms_run, since(t_in));
  if (h->timing) {
    FEM_LOCK(h);
    for (int i = 0; i < 3; ++i) h->t_ms[3 + i] += ms[i], h->t_n[3 + i] += 1;
    if (wait) h->t_ms[7] += ms_text, h->t_n[7] += 1;  // (without the wait no elapsed time is read: nothing to count)
  }
  s.qual_at = text.qual_at;
  out->text = text.text, out->len = text.len, out->n_asserted = text.n_asserted;
  out->n_reads = t.n_reads, out->n_records = t.n_records;
  memcpy(out->stats, s.stats, sizeof s.stats);
  return FEM_OK;
}

int fem_dev_map_batch_submit(fem_dev *h, int slot, const fem_params *p, const fem_read_batch *reads) {
  int rc = fem_dev_stage_reads(h, slot, reads);
  if (rc) return rc;
  return fem_dev_map_staged(h, slot, p);
}

int fem_dev_map_batch_wait(fem_dev *h, int slot, fem_batch_result *out) { return fem_dev_fetch(h, slot, out); }

int fem_dev_index_info(const fem_dev *h, char *buf, uint64_t cap) {
  if (!h || !buf || cap == 0) return FEM_ERR_INVALID;
  const uint64_t n_buckets = h->n_lookup ? h->n_lookup - 1 : 0;
  std::string s;
  if (!h->d_lookup) {
    s = "no index";
  } else if (h->d_occ32 && h->d_freq11) {
    s = "dense: 32-bit occurrence table, ";
    s += h->list_shift ? "strided with pads (" + std::to_string(((n_buckets + femk::kDensePadBuckets) << h->list_shift) * 4 >> 20) + " MiB)"
                       : "compact (" + std::to_string(h->n_occ * 4 >> 20) + " MiB)";
    s += ", " + std::to_string(h->n_banks) + (h->n_banks == 1 ? " bank" : " banks") + ", freq11 64 MiB";
  } else if (h->d_summary) {
    s = "sparse: bucket summaries";
  } else {
    s = "64-bit occurrence table only";
  }
  s += "; " + std::to_string(h->n_occ) + " entries in " + std::to_string(n_buckets) + " buckets";
  snprintf(buf, (size_t)cap, "%s", s.c_str());
  return FEM_OK;
}

const char *fem_dev_seed_kernel(const fem_dev *h, const fem_params *p) {
  if (!h || !params_ok(p)) return "";
  const int R = p->e + 1 + p->a;
  const bool use_fast = !h->force_generic && p->k == femk::kK && p->step == femk::kStep && R >= 1 && R <= femk::kMaxR;
  if (!use_fast) return "seed_filter_kernel";
  if (h->d_occ32 && h->d_freq11) return h->n_banks > 1 ? "seed_join_banked_kernel" : "seed_join_kernel";
  return (h->force_hash || (double)h->n_occ > (double)h->n_lookup) ? "seed_fast_kernel<hash>" : "seed_fast_kernel<lean>";
}

int fem_dev_set_timing(fem_dev *h, int on) {
  if (!h) return FEM_ERR_INVALID;
  h->timing = on != 0;
  if (h->timing && h->event_pool.size() < 64) {
    // the events of a few batches in flight, made now: created one by one inside the first timed launches they cost those
    // launches a millisecond each (the pipeline of bench.py was visibly slower over its first steps)
    HIP_TRY(h, hipSetDevice(h->device));
    while (h->event_pool.size() < 64) {
      hipEvent_t e = nullptr;
      HIP_TRY(h, hipEventCreate(&e));
      h->event_pool.push_back(e);
    }
  }
  return FEM_OK;
}

int fem_dev_reset_timing(fem_dev *h) {
  if (!h) return FEM_ERR_INVALID;
  for (int i = 0; i < kTimedKernels; ++i) h->t_ms[i] = 0, h->t_n[i] = 0;
  if (h->timeline) {
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->ev_epoch) HIP_TRY(h, hipEventCreate(&h->ev_epoch));
    HIP_TRY(h, hipEventRecord(h->ev_epoch, h->side_stream));
    h->have_epoch = true;
  }
  return FEM_OK;
}

int fem_dev_kernel_time(fem_dev *h, int kernel, double *ms_total, uint64_t *launches) {
  if (!h || kernel < 0 || kernel >= kTimedKernels) return FEM_ERR_INVALID;
  if (ms_total) *ms_total = h->t_ms[kernel];
  if (launches) *launches = h->t_n[kernel];
  return FEM_OK;
}

int fem_dev_copy_bandwidth(fem_dev *h, uint64_t bytes, int iters, double *gb_per_s) {
  if (!h || !gb_per_s || bytes == 0 || iters <= 0) return FEM_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  void *a = nullptr, *b = nullptr;
  HIP_TRY(h, hipMalloc(&a, bytes));
  if (hipMalloc(&b, bytes) != hipSuccess) {
    (void)hipFree(a);
    return fail(h, FEM_ERR_NOMEM, "copy bandwidth probe: out of memory");
  }
  hipStream_t st = h->slot[0].stream;
  hipEvent_t e0 = get_event(h), e1 = get_event(h);
  (void)hipMemsetAsync(a, 1, bytes, st);
  (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, st);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, st);
  (void)hipEventRecord(e1, st);
  hipError_t e = hipStreamSynchronize(st);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  h->event_pool.push_back(e0);
  h->event_pool.push_back(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  if (e != hipSuccess) return fail(h, FEM_ERR_HIP, hipGetErrorString(e));
  *gb_per_s = ms > 0 ? (2.0 * (double)bytes * iters) / (ms * 1e6) : 0.0;  // read + write
  return FEM_OK;
}

int fem_dev_h2d_bandwidth(fem_dev *h, uint64_t bytes, int iters, double *gb_per_s) {
  if (!h || !gb_per_s || bytes == 0 || iters <= 0) return FEM_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  void *a = nullptr, *b = nullptr;
  HIP_TRY(h, hipHostMalloc(&a, bytes, hipHostMallocDefault));
  if (hipMalloc(&b, bytes) != hipSuccess) {
    (void)hipHostFree(a);
    return fail(h, FEM_ERR_NOMEM, "h2d bandwidth probe: out of memory");
  }
  memset(a, 1, bytes);
  hipStream_t st = h->slot[0].stream;
  hipEvent_t e0 = get_event(h), e1 = get_event(h);
  (void)hipMemcpyAsync(b, a, bytes, hipMemcpyHostToDevice, st);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) (void)hipMemcpyAsync(b, a, bytes, hipMemcpyHostToDevice, st);
  (void)hipEventRecord(e1, st);
  hipError_t e = hipStreamSynchronize(st);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  h->event_pool.push_back(e0);
  h->event_pool.push_back(e1);
  (void)hipHostFree(a);
  (void)hipFree(b);
  if (e != hipSuccess) return fail(h, FEM_ERR_HIP, hipGetErrorString(e));
  *gb_per_s = ms > 0 ? ((double)bytes * iters) / (ms * 1e6) : 0.0;
  return FEM_OK;
}

#ifdef FEM_STAMPS
// diagnostic build only: read and clear the seed kernel's per-phase cycle totals
int fem_dbg_stamps(uint64_t *out, int n) {
  unsigned long long tmp[femk::kNumStamps];
  if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(femk::g_stamp_cycles), sizeof tmp) != hipSuccess) return -1;
  for (int i = 0; i < n && i < femk::kNumStamps; ++i) out[i] = tmp[i];
  memset(tmp, 0, sizeof tmp);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(femk::g_stamp_cycles), tmp, sizeof tmp);
  return 0;
}
#endif

// ---- host placement ----
// The pinned staging buffers are what the GPU's copy engines read: they should sit in the memory of the socket the GPU
// hangs off, and so should the threads that fill them (measured on a two-socket MI355X host, FASTQ -> SAM on 16 M reads:
// 0.30 s with the process on the GPU's node, 0.35-0.45 s unbound, 0.39-0.42 s on the other node).
int fem_device_numa(int device, int32_t *node, char *cpulist, uint64_t cap) {
  if (node) *node = -1;
  if (cpulist && cap) cpulist[0] = 0;
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return FEM_ERR_HIP;
  for (char *c = bdf; *c; ++c) *c = (char)tolower((unsigned char)*c);  // sysfs spells the address in lower case
  char path[160];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE *f = fopen(path, "r");
  if (!f) return FEM_OK;  // (no sysfs: unknown, not an error)
  int n = -1;
  if (fscanf(f, "%d", &n) != 1) n = -1;
  fclose(f);
  if (node) *node = n;
  if (n >= 0 && cpulist && cap) {
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", n);
    if ((f = fopen(path, "r"))) {
      if (!fgets(cpulist, (int)std::min<uint64_t>(cap, 1u << 20), f)) cpulist[0] = 0;
      fclose(f);
      for (char *c = cpulist; *c; ++c)
        if (*c == '\n') *c = 0;
    }
  }
  return FEM_OK;
}

int fem_bind_thread_near_device(int device) {
  const char *env = getenv("FEM_NUMA_BIND");
  if (env && env[0] == '0') return 1;
  int32_t node = -1;
  std::vector<char> list(4096, 0);
  if (fem_device_numa(device, &node, list.data(), list.size()) != FEM_OK || node < 0 || !list[0]) return 1;
  cpu_set_t now, want;
  CPU_ZERO(&now);
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof now, &now) != 0) return 1;
  int n_set = 0;
  for (const char *c = list.data(); *c;) {  // "a-b,c,d-e"
    char *end = nullptr;
    const long lo = strtol(c, &end, 10);
    if (end == c) break;
    long hi = lo;
    if (*end == '-') hi = strtol(end + 1, &end, 10);
    for (long i = lo; i <= hi && i < CPU_SETSIZE; ++i)
      if (i >= 0 && CPU_ISSET((int)i, &now)) {
        CPU_SET((int)i, &want);
        ++n_set;
      }
    if (*end != ',') break;
    c = end + 1;
  }
  if (n_set == 0) return 1;  // the node's CPUs are not ours to run on: leave the thread where it is
  return sched_setaffinity(0, sizeof want, &want) == 0 ? 0 : 1;
}

int fem_dev_allreduce_stats(fem_dev *const *hs, int n, uint64_t *stats) {
  if (!hs || n <= 0 || !stats) return FEM_ERR_INVALID;
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) {
    if (!hs[i]) return FEM_ERR_INVALID;
    devs[i] = hs[i]->device;
  }
  fem_dev *h0 = hs[0];
  std::lock_guard<std::mutex> lock(g_comm_mu);
  // one communicator per job: created on the first call, reused while the same devices take part
  if (!g_comm || g_comm->devs != devs) {
    destroy_comm_set();
    CommSet *cs = new (std::nothrow) CommSet();
    if (!cs) return fail(h0, FEM_ERR_NOMEM, "out of host memory");
    cs->devs = devs;
    cs->comms.assign(n, nullptr);
    cs->bufs.assign(n, nullptr);
    if (ncclCommInitAll(cs->comms.data(), n, devs.data()) != ncclSuccess) {
      delete cs;
      return fail(h0, FEM_ERR_RCCL, "ncclCommInitAll failed");
    }
    g_comm = cs;
    for (int i = 0; i < n; ++i)
      if (hipSetDevice(devs[i]) != hipSuccess || hipMalloc((void **)&cs->bufs[i], 5 * sizeof(uint64_t)) != hipSuccess) {
        destroy_comm_set();
        return fail(h0, FEM_ERR_HIP, "allreduce: allocating the counter buffers failed");
      }
  }
  CommSet &cs = *g_comm;
  int rc = FEM_OK;
  for (int i = 0; i < n && rc == FEM_OK; ++i) {
    if (hipSetDevice(devs[i]) != hipSuccess ||
        hipMemcpyAsync(cs.bufs[i], stats + 5 * i, 5 * sizeof(uint64_t), hipMemcpyHostToDevice, hs[i]->slot[0].stream) != hipSuccess)
      rc = fail(h0, FEM_ERR_HIP, "allreduce: staging the counters failed");
  }
  if (rc == FEM_OK) {
    ncclGroupStart();
    for (int i = 0; i < n; ++i) {
      (void)hipSetDevice(devs[i]);
      if (ncclAllReduce(cs.bufs[i], cs.bufs[i], 5, ncclUint64, ncclSum, cs.comms[i], hs[i]->slot[0].stream) != ncclSuccess)
        rc = fail(h0, FEM_ERR_RCCL, "ncclAllReduce failed");
    }
    if (ncclGroupEnd() != ncclSuccess) rc = fail(h0, FEM_ERR_RCCL, "ncclGroupEnd failed");
  }
  for (int i = 0; i < n && rc == FEM_OK; ++i) {
    (void)hipSetDevice(devs[i]);
    if (hipStreamSynchronize(hs[i]->slot[0].stream) != hipSuccess ||
        hipMemcpy(stats + 5 * i, cs.bufs[i], 5 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess)
      rc = fail(h0, FEM_ERR_HIP, "allreduce: reading the counters back failed");
  }
  return rc;
}

}  // extern "C"
