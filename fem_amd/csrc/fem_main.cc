// fem_main.cc — the drop-in command line: `FEM index` and `FEM map` (reference src/FEM.c, src/FEM_index.c,
// src/FEM_map.c).  Same verbs, flags, index file format and SAM output; the per-read hot path runs on the GPU
// through libfemhip.so (include/fem_hip.h).  There is no CPU mapping path in this binary: without a GPU it fails.
//
// New, optional: `--gpus N` (map) spreads read batches over N GPUs of this node (whichever GPU has a free slot takes the
// next batch; record order in the SAM file follows completion, parity is modulo record order); `--batch N` sets reads
// per batch.
#include <errno.h>
#include <fcntl.h>
#include <getopt.h>
#include <unistd.h>
#include <malloc.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fem_hip.h"
#include "fem_host.h"

#define FEM_VERSION "0.2"

namespace {

double real_time() {
  struct timeval tp;
  gettimeofday(&tp, nullptr);
  return tp.tv_sec + tp.tv_usec * 1e-6;
}
double cpu_time() {
  struct rusage r;
  getrusage(RUSAGE_SELF, &r);
  return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}

void usage_main() {
  fprintf(stderr, "\nProgram: FEM (Fast and Efficient short read Mapper), MI355X build\n");
  fprintf(stderr, "Version: %s\n\n", FEM_VERSION);
  fprintf(stderr, "Usage:   FEM <command> [options]\n\n");
  fprintf(stderr, "Command: index   build index for reference\n");
  fprintf(stderr, "         map     map reads\n\n");
  fprintf(stderr, "Note: To use FEM, you need to first index the genome with `FEM index'.\n\n");
}
void usage_index() { fprintf(stderr, "Usage: FEM index <window_size> <step_size> <reference> <output>\n"); }
void usage_map() {
  fprintf(stderr, "\nUsage:  FEM map [options] \n\nOptions:\n");
  fprintf(stderr, "        -e       INT  error threshold \n");
  fprintf(stderr, "        -t       INT  number of threads \n");
  fprintf(stderr, "        -f       STR  seeding algorithm: \"g\" for group seeding and \"v\" for variable-length seeding \n");
  fprintf(stderr, "        -a       INT  # additional q-grams (only for test)\n");
  fprintf(stderr, "        --gpus   INT  number of GPUs to shard read batches over [1]\n");
  fprintf(stderr, "        --batch  INT  reads per device batch [250000]\n\n");
  fprintf(stderr, "Input/output: \n");
  fprintf(stderr, "        --ref    STR  Input reference file\n");
  fprintf(stderr, "        --index  STR  Input index file\n");
  fprintf(stderr, "        --read1  STR  Input read1 file\n");
  fprintf(stderr, "        -o       STR  Output SAM file \n\n");
}

struct Reference {
  fem_seqset set{};
  std::vector<uint32_t> len;
  fem_tail_ref view{};
  bool load(const char *path) {
    double t0 = real_time();
    fem_seqfile *f = fem_seqfile_open(path);
    if (!f) {
      fprintf(stderr, "Cannot find sequence file!");
      return false;
    }
    int rc = fem_seqfile_read(f, 0, &set);
    fem_seqfile_close(f);
    if (rc != 0) {
      fprintf(stderr, "Didn't reach the end of sequence file, which might be corrupted!");
      return false;
    }
    len.resize(set.n);
    for (uint64_t i = 0; i < set.n; ++i) {
      uint64_t l = set.off[i + 1] - set.off[i];
      if (l > 0xFFFFFFFFull) {
        fprintf(stderr, "Reference sequence longer than 2^32 bases.\n");
        return false;
      }
      len[i] = (uint32_t)l;
    }
    view.text = set.bases, view.off = set.off, view.len = len.data(), view.n_seq = (uint32_t)set.n;
    view.names = set.names, view.name_off = set.name_off;
    fprintf(stderr, "Number of sequences: %lu\n", (unsigned long)set.n);
    fprintf(stderr, "Number of bases: %lu\n", (unsigned long)set.off[set.n]);
    fprintf(stderr, "Loaded all sequences successfully in %fs\n", real_time() - t0);
    return set.n > 0;
  }
  int upload(fem_dev *h) const {
    std::vector<const char *> ptr(set.n);
    for (uint64_t i = 0; i < set.n; ++i) ptr[i] = set.bases + set.off[i];
    return fem_dev_upload_reference(h, (uint32_t)set.n, ptr.data(), len.data());
  }
  ~Reference() { fem_seqset_free(&set); }
};

int dev_fail(fem_dev *h, const char *what, int rc) {
  fprintf(stderr, "[FEM] %s failed: %s (%s)\n", what, fem_strerror(rc), h ? fem_dev_last_error(h) : "");
  return EXIT_FAILURE;
}

// ---------------------------------------------------------------- FEM index (src/FEM_index.c:11-39)
int index_main(int argc, char **argv) {
  if (argc < 5) {
    fprintf(stderr, "%s\n", "Too few args!");
    usage_index();
    exit(EXIT_FAILURE);
  }
  int k = atoi(argv[1]), step = atoi(argv[2]);
  const char *ref_path = argv[3], *out_path = argv[4];
  fprintf(stderr, "k: %d, step size: %d, reference: %s, output: %s\n", k, step, ref_path, out_path);
  if (k < 1 || k > 16 || step < 1) {
    fprintf(stderr, "window_size must be 1..16 and step_size >= 1.\n");
    usage_index();
    exit(EXIT_FAILURE);
  }
  (void)fem_bind_thread_near_device(0);  // host threads and buffers next to the GPU (FEM_NUMA_BIND=0: leave them alone)
  Reference ref;
  if (!ref.load(ref_path)) exit(EXIT_FAILURE);
  fem_dev *h = nullptr;
  int rc = fem_dev_open(0, &h);
  if (rc) return dev_fail(nullptr, "fem_dev_open (the index is built on the GPU; no CPU path)", rc);
  double t0 = real_time();
  if ((rc = ref.upload(h))) return dev_fail(h, "reference upload", rc);
  uint64_t n_occ = 0;
  if ((rc = fem_dev_build_index(h, k, step, nullptr, nullptr, 0, &n_occ))) return dev_fail(h, "index build", rc);
  std::vector<uint32_t> lookup(((size_t)1 << (2 * k)) + 1);
  std::vector<uint64_t> occ(n_occ ? n_occ : 1);
  if ((rc = fem_dev_fetch_index(h, lookup.data(), occ.data(), occ.size()))) return dev_fail(h, "index fetch", rc);
  fprintf(stderr, "Collected %lu seeds.\n", (unsigned long)n_occ);
  fprintf(stderr, "Lookup table size: %lu, occurrence table size: %lu.\n", (unsigned long)lookup.size(), (unsigned long)n_occ);
  fprintf(stderr, "Built index in %fs.\n", real_time() - t0);
  fem_dev_close(h);
  if (fem_index_save(out_path, k, step, lookup.data(), n_occ, occ.data()) != 0) {
    fprintf(stderr, "Write error while initializing hash table.\n");
    exit(EXIT_FAILURE);
  }
  return 0;
}

// ---------------------------------------------------------------- FEM map (src/FEM_map.c:57-227)
//
// The reference runs 1 reader + T mapping threads + 1 writer over two queues (src/FEM_map.c:172-198,
// src/input_queue.c, src/output_queue.c).  Here the T mapping threads are the GPUs:
//
//   reader     parses the next FASTQ window (all -t threads; bases, qualities, names) STRAIGHT INTO the pinned staging buffers of a free
//              (GPU, slot) pair — whichever GPU has one free, so the GPUs balance by themselves
//   worker[g]  one thread per GPU, the only one that talks to that GPU's handle: starts H2D + kernels of a filled
//              slot (asynchronous), keeps two batches in flight, then fetches the finished batch's SAM text
//              (sort + traceback + CIGAR/MD + the text itself run on the device, fem_dev_fetch_sam)
//   writer     writes the text
//   formatter  only with FEM_HOST_FORMAT=1 / FEM_HOST_TAIL=1: renders records as SAM text on the host threads
// A slot goes  free -> filled -> in flight -> fetched -> written -> free  and there are four per GPU.
template <typename T>
class Channel {  // unbounded hand-off between pipeline stages (the number of slots bounds what is in flight)
 public:
  void push(T v) {
    {
      std::lock_guard<std::mutex> l(m_);
      q_.push_back(std::move(v));
    }
    cv_.notify_one();
  }
  T pop() {
    std::unique_lock<std::mutex> l(m_);
    cv_.wait(l, [&] { return !q_.empty(); });
    T v = std::move(q_.front());
    q_.pop_front();
    return v;
  }
  bool try_pop_for(T &v, double seconds) {  // waits up to `seconds` for an element
    std::unique_lock<std::mutex> l(m_);
    if (!cv_.wait_for(l, std::chrono::duration<double>(seconds), [&] { return !q_.empty(); })) return false;
    v = std::move(q_.front());
    q_.pop_front();
    return true;
  }
  bool try_pop(T &v) {
    std::lock_guard<std::mutex> l(m_);
    if (q_.empty()) return false;
    v = std::move(q_.front());
    q_.pop_front();
    return true;
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<T> q_;
};

struct Growable {  // a reusable host array: only ever grows
  char *p = nullptr;
  uint64_t cap = 0;
  bool reserve(uint64_t n) {
    if (n <= cap) return true;
    const uint64_t want = n + n / 8 + 4096;
    char *q = (char *)realloc(p, want);
    if (!q) return false;
    p = q, cap = want;
    return true;
  }
  ~Growable() { free(p); }
};

struct BatchBuf {  // everything about the batch that sits in one (GPU, slot) pair
  uint64_t seq = 0;  // submission number on its GPU: results are handed on in this order
  int gpu = 0, slot = 0;
  char *bases = nullptr;      // pinned staging lent by the device library (fem_dev_acquire_stage)
  uint64_t *off = nullptr;
  uint64_t reads_cap = 0, bases_cap = 0;
  Growable quals, names, name_off;  // host formatting (FEM_HOST_FORMAT=1 / FEM_HOST_TAIL=1): names and qualities stay on the host
  bool packed = false;                          // the parser wrote this batch at two bits per base (fem_seqfile_fill_packed)
  uint64_t n_exc = 0;                           // ... and this many characters outside "ACGT" behind the codes
  bool spliced = false;                         // ... and copied nothing else: names, bases, qualities stay in the input's mapping
  Growable r_name, r_name_len, r_seq, r_qual;   //     (where each read's fields lie: fem_seqfile_fill_packed_refs)
  fem_read_refs refs{};
  char *q_stage = nullptr, *n_stage = nullptr;  // device SAM text: pinned staging lent by the library
  uint64_t *no_stage = nullptr;
  uint64_t names_cap = 0, want_names = 0;
  fem_batch_sam sam{};
  fem_batch_shape shape{};
  uint64_t want_reads = 0, want_bases = 0;  // set by the reader when the staging buffers are too small
  fem_batch_records rec{};                  // device tail's records (default path)
  fem_batch_result res{};                   // per-candidate outcome (FEM_HOST_TAIL=1)
  double t_submit = 0;
  double t_slot = 0, t_filled = 0, t_submitted = 0, t_retired = 0, t_text = 0;  // FEM_STAGE_TIMES=2: the batch's way through the stages
};

struct TextOut {  // one batch of SAM text: parts[i] of buf, in order
  char *buf = nullptr;
  uint64_t cap = 0;
  std::vector<fem_text_part> parts;
  bool owned_elsewhere = false;  // buf was malloc'd by fem_tail_sam: free it after writing
};

enum MsgKind { kFilled, kRecycle, kRegrow, kStop };
struct Msg {
  MsgKind kind = kStop;
  BatchBuf *b = nullptr;
};

int map_main(int argc, char **argv) {
  // Batches come and go as a few 100 MB allocations: keep them in the heap instead of mapping and unmapping them, so
  // that a recycled buffer's pages are already there (page faults were a visible share of the host time).
  mallopt(M_MMAP_MAX, 0);
  mallopt(M_TRIM_THRESHOLD, -1);
  char *ref_path = nullptr, *index_path = nullptr, *read_path = nullptr, *out_path = nullptr;
  fem_params params{12, 3, 2, 1};  // src/FEM_map.c:67-70: k and step are fixed, whatever the index header says
  int n_threads = 1, n_gpus = 1;
  // reads per batch: 250 k fills the pipeline soonest on small inputs; a batch costs three host round trips on its way through
  // the tail, which 1 M-read batches amortise on large ones (16 M reads of C2 to /dev/null: 105 -> 116 Mreads/s, 8 M of C3:
  // 71 -> 81; round 4) — chosen by the size of the read file unless --batch says otherwise
  uint64_t batch_reads = 250000;
  bool batch_given = false;
  const char *short_opt = "ha:f:e:t:o:r:i:b:";
  static struct option long_opt[] = {{"help", no_argument, nullptr, 'h'},       {"ref", required_argument, nullptr, 'r'},
                                     {"index", required_argument, nullptr, 'i'}, {"read1", required_argument, nullptr, 'b'},
                                     {"gpus", required_argument, nullptr, 'G'},  {"batch", required_argument, nullptr, 'B'},
                                     {nullptr, 0, nullptr, 0}};
  int c, oi = 0;
  while ((c = getopt_long(argc, argv, short_opt, long_opt, &oi)) >= 0) {
    switch (c) {
      case 'r': ref_path = optarg; break;
      case 'i': index_path = optarg; break;
      case 'b': read_path = optarg; break;
      case 'e': params.e = atoi(optarg); break;
      case 't': n_threads = atoi(optarg); break;
      case 'a': params.a = atoi(optarg); break;
      case 'G': n_gpus = atoi(optarg); break;
      case 'B': batch_reads = strtoull(optarg, nullptr, 10), batch_given = true; break;
      case 'f':
        if (strcmp(optarg, "v") != 0 && strcmp(optarg, "g") != 0) {  // parsed and ignored (src/FEM_map.c:108-118)
          fprintf(stderr, "%s\n", "Wrong name of seeding algorithm!");
          usage_map();
          exit(EXIT_FAILURE);
        }
        break;
      case 'o': out_path = optarg; break;
      default:
        usage_map();
        exit(EXIT_SUCCESS);
    }
  }
  // check_args (src/FEM_map.c:29-55)
  const char *bad = nullptr;
  if (params.e < 0 || params.e > 7) bad = "Wrong error threshold.";
  else if (n_threads <= 0) bad = "Wrong number of threads.";
  else if (params.a < 0 || params.a > 2) bad = "Wrong number of additional q-grams.";
  else if (!ref_path) bad = "Reference file path is required.";
  else if (!index_path) bad = "Index file path is required.";
  else if (!read_path) bad = "Read file path is required.";
  else if (!out_path) bad = "Output file path is required.";
  else if (n_gpus < 1 || n_gpus > 64) bad = "Wrong number of GPUs.";
  else if (batch_reads < 1) bad = "Wrong batch size.";
  if (bad) {
    fprintf(stderr, "%s\n", bad);
    usage_map();
    exit(EXIT_FAILURE);
  }

  // Host placement: with one GPU the whole process (parser, formatter, staging buffers) moves next to it, before any
  // thread pool exists; with several, each GPU's worker thread does so for itself and the buffers it acquires.
  const char *sg = getenv("FEM_TEST_SHARE_GPU"), *tg = getenv("FEM_TESTING");  // (test hook: honoured under FEM_TESTING=1 only)
  const bool share_gpu = sg && sg[0] == '1' && tg && tg[0] == '1';
  if (n_gpus == 1 || share_gpu) (void)fem_bind_thread_near_device(0);
  Reference ref;
  if (!ref.load(ref_path)) exit(EXIT_FAILURE);
  double t_idx = real_time();
  int32_t ik = 0, istep = 0;
  uint32_t *lookup = nullptr;
  uint64_t *occ = nullptr, n_occ = 0;
  if (fem_index_load(index_path, &ik, &istep, &lookup, &n_occ, &occ) != 0) {
    fprintf(stderr, "Failed to open index file %s\n", index_path);
    exit(EXIT_FAILURE);
  }
  fprintf(stderr, "Loaded index in %fs!\n", real_time() - t_idx);
  if (ik != params.k) {  // the reference would silently index a 4^12 table with the wrong hashes; refuse instead
    fprintf(stderr, "Index was built with k=%d but map always uses k=%d.\n", ik, params.k);
    exit(EXIT_FAILURE);
  }

  const char *ht = getenv("FEM_HOST_TAIL");
  const bool host_tail = ht && ht[0] == '1';
  // Where the SAM text is made.  Default: on the device (fem_dev_fetch_sam: qualities and names go there with the bases, the
  // finished text comes back: ~365 bytes per read over the link).  FEM_HOST_FORMAT=1: the device maps, orders, traces back and
  // hands over RECORDS (FLAG, RNAME id, POS, CIGAR, NM, MD: ~30 bytes per read) and the host threads splice them between
  // QNAME, SEQ and QUAL, which never leave the host — for a plain FASTQ file they are not even copied: the formatter reads them
  // out of the input's mapping (fem_seqfile_fill_packed_refs; FEM_SPLICE=0: copied by the parser).  ~55 bytes per read cross
  // the link; the price is ~0.07 core-microseconds of formatting per read, which on a 16-core share of the host is more than
  // the link costs (measured, 16 M reads of C2 to /dev/null: 75-93 Mreads/s against 100-117), so it is the
  // option, not the default.  FEM_HOST_TAIL=1: ordering, traceback and text by the host threads from the per-candidate outcome.
  const char *hf = getenv("FEM_HOST_FORMAT"), *spl = getenv("FEM_SPLICE");
  const bool device_text = !host_tail && !(hf && hf[0] == '1');
  // ... with the qualities kept on the host (fem_dev_commit_names_stage: the device leaves their field open and the batch's
  // retiring thread fills it in): they are 228 of the 473 bytes per read on the link otherwise — and 0.027 core-µs per read on
  // the host then, next to the parser's 0.078.  With 16 cores per GPU the two forms are within +7 / -16 % of each other from box
  // to box (the run is bound by the link in one and by the cores in the other); from 24 threads on the qualities stay on the
  // host.  FEM_HOST_QUALS=1 / FEM_DEVICE_QUALS=1 decide it by hand.
  const char *dq = getenv("FEM_DEVICE_QUALS"), *hq = getenv("FEM_HOST_QUALS");
  const bool host_quals = device_text && !(dq && dq[0] == '1') && ((hq && hq[0] == '1') || n_threads >= 24);
  const bool splice = !host_tail && !device_text && !(spl && spl[0] == '0');
  // With the text on the device the link is what bounds the run: batches of equal-length reads then cross it at two bits per
  // base — the parser writes that form straight into the pinned staging (fem_seqfile_fill_packed ->
  // fem_dev_commit_stage_packed: no host work per base beyond the parse itself) (FEM_PACK_BASES=0: always as characters).
  const char *pk = getenv("FEM_PACK_BASES");
  const bool pack_bases = !host_tail && !(pk && pk[0] == '0');
  if (!batch_given) {
    struct stat st;
    if (stat(read_path, &st) == 0 && st.st_size >= (off_t)(1ll << 30)) batch_reads = 1000000;  // (plain FASTQ of >= ~4 M reads)
  }
  const uint64_t batch_bytes = batch_reads * 250ull;  // header + bases + '+' + qualities of a ~100 bp record
  // a FASTQ window of batch_bytes characters holds fewer than batch_bytes / 2 bases; records under 32 bytes are unusual
  // (the reader asks for larger buffers when a batch needs them)
  const uint64_t reads_cap0 = batch_bytes / 32 + 16, bases_cap0 = batch_bytes / 2 + 4096, names_cap0 = bases_cap0 / 2 + 4096;
  std::vector<fem_dev *> devs((size_t)n_gpus, nullptr);
  // FEM_TEST_SHARE_GPU=1 (test hook for one-GPU boxes): all `--gpus N` workers open GPU 0, each with its own handle, and
  // the counters are summed on the host (RCCL refuses two ranks on one device)
  // What a batch will look like, from the input's first records: the device library then makes a batch's allocations during
  // the setup (fem_dev_reserve_batch) instead of between the first batches' kernels.
  uint64_t res_reads = 0;
  uint32_t res_len = 0;
  if (device_text) {
    if (fem_seqfile *f0 = fem_seqfile_open(read_path)) {
      fem_batch_plan *pl = nullptr;
      fem_batch_shape sh{};
      if (fem_seqfile_plan(f0, 1u << 18, 1, &pl, &sh) == 0 && pl && sh.n_reads > 0) {
        const uint64_t per_record = (2 * sh.n_bases + sh.n_name_bytes) / sh.n_reads + 6;
        res_reads = batch_bytes / per_record + batch_bytes / per_record / 8 + 1024, res_len = sh.max_len;
      }
      fem_batch_plan_free(pl);
      fem_seqfile_close(f0);
    }
  }
  (void)fem_set_blocking_waits(1);  // (a thread per batch in flight waits for the device; the parser needs the cores)
  const double t_dev = real_time();
  {  // one thread per GPU uploads the replicated reference + index (src/FEM_map.c:135-143)
    std::vector<int> up_rc((size_t)n_gpus, 0);
    std::vector<std::thread> up;
    for (int g = 0; g < n_gpus; ++g)
      up.emplace_back([&, g] {
        if (n_gpus > 1 && !share_gpu) (void)fem_bind_thread_near_device(g);  // the handle's pinned result buffers
        int rc = fem_dev_open(share_gpu ? 0 : g, &devs[(size_t)g]);
        if (!rc) rc = ref.upload(devs[(size_t)g]);
        if (!rc) rc = fem_dev_upload_index(devs[(size_t)g], ik, istep, lookup, ((uint64_t)1 << (2 * ik)) + 1, occ, n_occ);
        // the slots' pinned staging (and, for the device's SAM text, its buffers): pinning host memory takes ~0.25 ms per
        // MB, 35 ms per slot here, and belongs to the setup rather than to the first batches
        int32_t ns = 4;
        if (!rc) (void)fem_dev_limits(devs[(size_t)g], nullptr, &ns);
        for (int sl = 0; !rc && sl < ns; ++sl) {
          char *pb = nullptr, *pq = nullptr, *pn = nullptr;
          uint64_t *po = nullptr, *pno = nullptr;
          rc = fem_dev_acquire_stage(devs[(size_t)g], sl, reads_cap0, bases_cap0, &pb, &po);
          if (!rc && device_text) rc = fem_dev_acquire_text_stage(devs[(size_t)g], sl, reads_cap0, bases_cap0, names_cap0, &pq, &pn, &pno);
          if (!rc && device_text) rc = fem_dev_reserve_text(devs[(size_t)g], sl, reads_cap0, bases_cap0, names_cap0, batch_bytes + batch_bytes / 4);
          if (!rc && device_text && res_reads) rc = fem_dev_reserve_batch(devs[(size_t)g], sl, res_reads, res_reads + res_reads / 8, res_len, &params);
        }
        up_rc[(size_t)g] = rc;
      });
    for (auto &t : up) t.join();
    for (int g = 0; g < n_gpus; ++g)
      if (up_rc[(size_t)g]) return dev_fail(devs[(size_t)g], "device setup (mapping runs on the GPU; no CPU path)", up_rc[(size_t)g]);
  }
  free(lookup), free(occ);
  fprintf(stderr, "Reference and index resident on %d GPU%s in %fs.\n", n_gpus, n_gpus > 1 ? "s" : "", real_time() - t_dev);

  const int out_fd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (out_fd < 0) {
    fprintf(stderr, "Cannot open output file %s\n", out_path);
    exit(EXIT_FAILURE);
  }
  std::atomic<int> exit_code{0};
  // The SAM text of a batch is a handful of stretches (one per formatter thread), written one after the other.  (Side
  // by side with pwrite from four threads was tried twice: round 3 on tmpfs, 2.8 GB/s against 5.5 from one thread; round 5 on
  // the GPU box's /tmp, where four writers into four FILES take 56-61 GB/s against 16.5 from one: into ONE file 10.4 GB/s with
  // one, four or eight writers — buffered writes to a file take its inode's lock.)
  auto write_all = [&](const char *p, uint64_t n) -> bool {
    while (n) {
      ssize_t w = write(out_fd, p, n);
      if (w < 0 && errno == EINTR) continue;
      if (w <= 0) return false;
      p += w, n -= (uint64_t)w;
    }
    return true;
  };
  {
    char *hdr = nullptr;
    uint64_t hl = 0;
    fem_sam_header(&ref.view, &hdr, &hl);
    if (!write_all(hdr, hl)) {
      fprintf(stderr, "[FEM] write error on %s\n", out_path);
      exit_code = EXIT_FAILURE;
    }
    free(hdr);
  }

  if (device_text)
    for (fem_dev *h : devs) {
      int rc = fem_dev_upload_reference_names(h, (uint32_t)ref.set.n, ref.set.names, ref.set.name_off);
      if (rc) return dev_fail(h, "reference names", rc);
    }

  double t_start = real_time();
  auto cpu_seconds = [](double *user, double *sys) {
    struct rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    *user = ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec, *sys = ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec;
  };
  double cpu_u0 = 0, cpu_s0 = 0;
  cpu_seconds(&cpu_u0, &cpu_s0);
  // -t threads are shared by the two host stages that run side by side (FEM_SPLIT_THREADS=0: each gets all of them)
  const char *sp_env = getenv("FEM_SPLIT_THREADS");
  const bool split = !(sp_env && sp_env[0] == '0') && n_threads >= 4;
  int rd_share = 50;  // per cent of the threads that parse (FEM_READER_SHARE)
  if (const char *rs = getenv("FEM_READER_SHARE")) rd_share = std::min(90, std::max(10, atoi(rs)));
  const int rd_threads = !device_text && split ? std::max(1, std::min(n_threads - 1, (n_threads * rd_share + 50) / 100)) : n_threads;
  const int fmt_threads = !device_text && split ? n_threads - rd_threads : n_threads;
  int32_t n_slots = 4;
  (void)fem_dev_limits(devs[0], nullptr, &n_slots);
  // FEM_STAGE_TIMES=1: busy seconds of each pipeline stage on stderr at the end
  const char *st_env = getenv("FEM_STAGE_TIMES");
  const bool stage_times = st_env && (st_env[0] == '1' || st_env[0] == '2');
  const bool batch_times = st_env && st_env[0] == '2';  // ... =2: and every batch's way through them
  double busy_read = 0, busy_text = 0, busy_write = 0, busy_plan = 0;
  std::mutex stat_mu;  // (busy_text: the batches' retiring threads add to it)
  double wait_slot = 0, wait_records = 0, wait_text_buf = 0;  // reader waiting for a free slot, formatter for records / a text buffer
  std::vector<double> busy_submit((size_t)n_gpus, 0.0), busy_recycle((size_t)n_gpus, 0.0);
  double t_first_slot = 0, t_first_filled = 0, t_reader_done = 0, t_workers_done = 0;
  std::vector<double> busy_wait((size_t)n_gpus, 0.0);
  std::atomic<uint64_t> n_asserted{0};

  std::vector<BatchBuf> bufs((size_t)n_gpus * (size_t)n_slots);
  Channel<BatchBuf *> free_q, text_q, regrown_q;
  std::vector<Channel<Msg>> work_q((size_t)n_gpus);
  struct WriteItem {
    TextOut *t = nullptr;   // host-rendered text, or
    BatchBuf *b = nullptr;  // a batch whose text the device rendered (b->sam)
  };
  Channel<TextOut *> text_free_q;
  Channel<WriteItem> write_q;
  std::vector<TextOut> texts(3);
  for (TextOut &t : texts) text_free_q.push(&t);
  std::vector<uint64_t> per_gpu((size_t)n_gpus * 5, 0);

  // ---- writer (src/output_queue.c:60-91) ----
  std::thread writer([&] {
    for (;;) {
      WriteItem it = write_q.pop();
      if (!it.t && !it.b) break;
      if (it.b) {  // text rendered on the device: one stretch of the library's pinned memory
        const int wrc = fem_dev_sam_wait(devs[(size_t)it.b->gpu], it.b->slot);  // the copy of the text to the host has arrived
        // (this thread must not read the handle's error string, which the GPU's worker thread may be writing: the code says enough)
        if (wrc && !exit_code.exchange(EXIT_FAILURE)) fprintf(stderr, "[FEM] SAM text failed: %s\n", fem_strerror(wrc));
        double t0 = real_time();
        it.b->t_text = t0;
        const double t_w = real_time();
        bool ok = wrc != 0 || it.b->sam.len == 0 || write_all(it.b->sam.text, it.b->sam.len);
        if (!ok && !exit_code.exchange(EXIT_FAILURE)) fprintf(stderr, "[FEM] write error on %s\n", out_path);
        busy_write += real_time() - t_w;
        fprintf(stderr, "Mapped read batch in %fs.\n", real_time() - it.b->t_submit);
        if (batch_times)
          fprintf(stderr, "[FEM] batch %lu gpu %d slot %d (ms): slot %.2f filled %.2f submit %.2f..%.2f retired %.2f text home %.2f written %.2f\n",
                  (unsigned long)it.b->seq, it.b->gpu, it.b->slot, 1e3 * (it.b->t_slot - t_start), 1e3 * (it.b->t_filled - t_start),
                  1e3 * (it.b->t_submit - t_start), 1e3 * (it.b->t_submitted - t_start), 1e3 * (it.b->t_retired - t_start),
                  1e3 * (it.b->t_text - t_start), 1e3 * (real_time() - t_start));
        free_q.push(it.b);  // written: the slot's buffers may be refilled
        continue;
      }
      TextOut *t = it.t;
      double t0 = real_time();
      bool ok = true;
      for (const fem_text_part &p : t->parts)
        if (ok && p.length) ok = write_all(t->buf + p.offset, p.length);
      if (!ok && !exit_code.exchange(EXIT_FAILURE)) fprintf(stderr, "[FEM] write error on %s\n", out_path);
      if (t->owned_elsewhere) {
        free(t->buf);
        t->buf = nullptr, t->cap = 0, t->owned_elsewhere = false;
      }
      busy_write += real_time() - t0;
      text_free_q.push(t);
    }
  });

  // ---- formatter: records -> SAM text (src/align.c:546-632, src/output_queue.c:93-116) ----
  std::thread formatter([&] {
    for (;;) {
      const double t_pop = real_time();
      BatchBuf *b = text_q.pop();
      if (!b) break;
      const double t_got = real_time();
      wait_records += t_got - t_pop;
      TextOut *t = text_free_q.pop();
      double t0 = real_time();
      wait_text_buf += t0 - t_got;
      fem_seqset reads{};
      reads.n = b->shape.n_reads, reads.bases = b->bases, reads.off = b->off, reads.quals = b->quals.p, reads.names = b->names.p;
      reads.name_off = (uint64_t *)b->name_off.p;
      int fmt;
      if (host_tail) {  // FEM_HOST_TAIL=1: ordering / traceback / MD by libfemhost from the per-candidate outcome
        fem_tail_input in{b->res.n_reads, b->res.cand_begin, b->res.cand_count, b->res.cand, b->res.ed, b->res.end};
        char *text = nullptr;
        uint64_t len = 0;
        fmt = fem_tail_sam(params.e, &ref.view, &reads, &in, fmt_threads, &text, &len);
        if (!fmt) {
          free(t->buf);
          t->buf = text, t->cap = len, t->owned_elsewhere = true;
          t->parts.assign(1, fem_text_part{0, len});
        }
      } else {  // default: the records come off the device, the host only renders text
        const fem_batch_records &rec = b->rec;
        fem_record_view rv{rec.n_reads, rec.n_records, rec.rec_begin, rec.flag, rec.tid, rec.pos0, rec.nm,
                           rec.cigar_off, rec.cigar, rec.md_off, rec.md};
        t->parts.assign((size_t)fmt_threads, fem_text_part{0, 0});
        uint64_t na = 0;
        fmt = b->spliced ? fem_records_sam_refs(&ref.view, &b->refs, &rv, fmt_threads, &t->buf, &t->cap, t->parts.data(), &na)
                         : fem_records_sam_parts(&ref.view, &reads, &rv, fmt_threads, &t->buf, &t->cap, t->parts.data(), &na);
        n_asserted += na;
      }
      busy_text += real_time() - t0;
      if (fmt != 0) {
        if (!exit_code.exchange(EXIT_FAILURE)) fprintf(stderr, "[FEM] out of memory while formatting SAM records\n");
        t->parts.clear();
      }
      write_q.push(WriteItem{t, nullptr});
      fprintf(stderr, "Mapped read batch in %fs.\n", real_time() - b->t_submit);
      free_q.push(b);  // fetched and rendered: the staging buffers may be refilled (they stay acquired, include/fem_hip.h)
    }
  });

  // ---- one worker per GPU (the mapping threads of src/FEM_map.c:182-185) ----
  std::vector<std::thread> workers;
  for (int g = 0; g < n_gpus; ++g)
    workers.emplace_back([&, g] {
      fem_dev *h = devs[(size_t)g];
      if (n_gpus > 1 && !share_gpu) (void)fem_bind_thread_near_device(g);
      auto acquire = [&](BatchBuf *b, uint64_t reads_cap, uint64_t bases_cap) -> bool {
        char *pb = nullptr;
        uint64_t *po = nullptr;
        int rc = 0;
        rc = fem_dev_acquire_stage(h, b->slot, reads_cap, bases_cap, &pb, &po);
        if (rc) {
          if (!exit_code.exchange(EXIT_FAILURE)) dev_fail(h, "staging buffers", rc);
          return false;
        }
        b->bases = pb, b->off = po, b->reads_cap = reads_cap, b->bases_cap = bases_cap;
        if (device_text) {  // qualities and names go to the GPU as well: pinned staging for them
          const uint64_t names_cap = std::max<uint64_t>(b->want_names, std::max<uint64_t>(b->names_cap, names_cap0));
          rc = fem_dev_acquire_text_stage(h, b->slot, reads_cap, bases_cap, names_cap, &b->q_stage, &b->n_stage, &b->no_stage);
          if (rc) {
            if (!exit_code.exchange(EXIT_FAILURE)) dev_fail(h, "text staging buffers", rc);
            return false;
          }
          b->names_cap = names_cap;
        }
        return true;
      };
      for (int s = 0; s < n_slots; ++s) {
        BatchBuf *b = &bufs[(size_t)g * (size_t)n_slots + (size_t)s];
        b->gpu = g, b->slot = s;
        if (!acquire(b, reads_cap0, bases_cap0)) b->bases = nullptr;  // (allocated during the setup: this only hands them out)
        free_q.push(b);  // (without buffers when the acquisition failed: the reader stops on it instead of waiting for ever)
      }
      // Round 5: the thread that submits does not retire.  A batch's way home has host round trips in it (the mapping's
      // counters, the records counted, the text sized: fem_dev_fetch_sam), each of which can wait behind another batch's
      // kernels on the device; with one thread per GPU doing both, those waits came one after the other and the GPU idled
      // (16 M reads of C3 to /dev/null: 91 Mreads/s with 9.4 ms of "device wait" per 1 M-read batch, of which 5 were
      // kernels).  Now every batch in flight is retired by a thread of its own (the library takes calls on different
      // slots of a handle from different threads), and a sequencer hands the results on in submission order.
      std::mutex seq_mu;
      uint64_t seq_next = 0, seq_submit = 0;
      std::map<uint64_t, std::function<void()>> seq_held;
      auto deliver = [&](uint64_t seq, std::function<void()> fn) {
        std::lock_guard<std::mutex> l(seq_mu);
        seq_held.emplace(seq, std::move(fn));
        for (auto it = seq_held.find(seq_next); it != seq_held.end(); it = seq_held.find(seq_next)) {
          it->second();
          seq_held.erase(it);
          ++seq_next;
        }
      };
      Channel<BatchBuf *> retire_q;
      auto retire = [&](BatchBuf *b) {
        double t0 = real_time();
        int rc = host_tail     ? fem_dev_map_batch_wait(h, b->slot, &b->res)
                 : device_text ? fem_dev_fetch_sam_nowait(h, b->slot, &b->sam)  // (the writer waits for the text itself)
                               : fem_dev_fetch_records(h, b->slot, &b->rec);
        const double waited = real_time() - t0;
        b->t_retired = t0 + waited;
        double placing = 0;
        if (!rc && host_quals) {
          // the qualities never left the host: once the text is home, into the fields the device left open — on this thread
          // (a batch's own), so that the writer only writes
          rc = fem_dev_sam_wait(h, b->slot);
          if (!rc && b->sam.len) {
            const double t1 = real_time();
            const uint64_t *qual_at = nullptr;
            uint64_t n_q = 0;
            int qrc = fem_dev_sam_quals(h, b->slot, &qual_at, &n_q);
            if (!qrc)
              qrc = fem_sam_fill_quals(const_cast<char *>(b->sam.text), b->sam.len, qual_at, n_q, b->q_stage, b->packed ? nullptr : b->off, b->shape.max_len,
                                       std::max(1, n_threads / 2));
            if (qrc) {
              if (!exit_code.exchange(EXIT_FAILURE)) fprintf(stderr, "[FEM] qualities could not be placed in the SAM text (%d)\n", qrc);
              rc = FEM_ERR_STATE;
            }
            placing = real_time() - t1;
          }
        }
        deliver(b->seq, [&, b, rc, waited, placing] {
          busy_wait[(size_t)g] += waited;
          {
            std::lock_guard<std::mutex> l(stat_mu);
            busy_text += placing;
          }
          if (rc) {
            if (!exit_code.exchange(EXIT_FAILURE)) dev_fail(h, "mapping", rc);
            work_q[(size_t)g].push(Msg{kRecycle, b});
            return;
          }
          const uint64_t *st = host_tail ? b->res.stats : device_text ? b->sam.stats : b->rec.stats;
          for (int i = 0; i < 5; ++i) per_gpu[(size_t)g * 5 + (size_t)i] += st[i];
          if (device_text) {
            n_asserted += b->sam.n_asserted;
            write_q.push(WriteItem{nullptr, b});
          } else {
            text_q.push(b);
          }
        });
      };
      int n_retire = std::max(1, n_slots);  // (one per slot: a batch's thread also waits for its text and places the qualities)
      if (const char *fl = getenv("FEM_FLIGHT")) n_retire = std::max(1, std::min(n_slots, atoi(fl)));
      std::vector<std::thread> retirers;
      for (int r = 0; r < n_retire; ++r)
        retirers.emplace_back([&] {
          if (n_gpus > 1 && !share_gpu) (void)fem_bind_thread_near_device(g);
          for (;;) {
            BatchBuf *b = retire_q.pop();
            if (!b) break;
            retire(b);
          }
        });
      for (;;) {
        Msg m = work_q[(size_t)g].pop();
        if (m.kind == kStop) break;
        if (m.kind == kRecycle) {
          const double t0 = real_time();
          if (!acquire(m.b, m.b->reads_cap, m.b->bases_cap)) m.b->bases = nullptr;
          busy_recycle[(size_t)g] += real_time() - t0;
          free_q.push(m.b);
          continue;
        }
        if (m.kind == kRegrow) {
          if (!acquire(m.b, m.b->want_reads, m.b->want_bases)) m.b->bases = nullptr;
          regrown_q.push(m.b);
          continue;
        }
        BatchBuf *b = m.b;  // kFilled
        b->t_submit = real_time();
        int rc = exit_code ? FEM_ERR_STATE
                 : b->packed ? fem_dev_commit_stage_packed(h, b->slot, b->shape.n_reads, b->shape.max_len, b->n_exc)
                 : b->shape.min_len == b->shape.max_len  // reads of one length: the offsets need not cross the link
                     ? fem_dev_commit_stage_uniform(h, b->slot, b->shape.n_reads, b->shape.max_len)
                     : fem_dev_commit_stage(h, b->slot, b->shape.n_reads, b->shape.max_len);
        const double t_c1 = real_time();
        if (!rc) rc = fem_dev_map_staged(h, b->slot, &params);
        const double t_c2 = real_time();
        // (qualities and names behind the mapping's launches: the call that hands the link 140 MB can wait for room in the copy
        //  engine's queue, 8-10 ms now and then, and the kernels need none of it)
        if (!rc && device_text)
          rc = host_quals ? fem_dev_commit_names_stage(h, b->slot, b->shape.n_reads, b->shape.n_name_bytes)
                          : fem_dev_commit_text_stage(h, b->slot, b->shape.n_reads, b->shape.n_name_bytes);
        if (batch_times && real_time() - b->t_submit > 1e-3)
          fprintf(stderr, "[FEM] slow submit (ms): reads committed %.2f, mapping queued %.2f, text committed %.2f\n", 1e3 * (t_c1 - b->t_submit),
                  1e3 * (t_c2 - t_c1), 1e3 * (real_time() - t_c2));
        if (rc) {
          if (!exit_code.exchange(EXIT_FAILURE)) dev_fail(h, "batch submit", rc);
          work_q[(size_t)g].push(Msg{kRecycle, b});
          continue;
        }
        b->t_submitted = real_time();
        busy_submit[(size_t)g] += b->t_submitted - b->t_submit;
        b->seq = seq_submit++;
        retire_q.push(b);
      }
      for (int r = 0; r < n_retire; ++r) retire_q.push(nullptr);
      for (auto &t : retirers) t.join();
    });

  // ---- reader (src/input_queue.c:53-79): this thread ----
  fem_seqfile *read_file = nullptr;
  {
    fem_seqfile *f = read_file = fem_seqfile_open(read_path);
    if (!f) {
      fprintf(stderr, "Cannot find sequence file!");  // the reference exits here (src/sequence_batch.c:33-35)
      exit_code = EXIT_FAILURE;
    }
    // The planner: cuts the next batch out of the input (one pass over it: where the records end, how many, how long) on a
    // thread of its own, ahead of the batch being filled where the file allows it (plain FASTQ through a mapping: a plan's records
    // stay where they are) — a batch's plan needs no staging slot, so it is also made while the reader waits for one.  On boxes
    // whose cores are slow the reader's plan-then-fill in a row bounded the run (32 M reads of C3: 118-121 Mreads/s against
    // 130-140 elsewhere, the reader busy 0.19-0.24 of the run's 0.25-0.27 s).
    struct Planned {
      fem_batch_plan *plan = nullptr;
      fem_batch_shape shape{};
      int rc = 0;
    };
    Channel<Planned> planned_q;
    Channel<int> plan_tokens;  // a plan is made per token: one while nothing may run ahead, two where it may
    std::atomic<bool> plan_stop{false};
    const bool ahead = f && fem_seqfile_plan_ahead_ok(f) && !(getenv("FEM_PLAN_AHEAD") && getenv("FEM_PLAN_AHEAD")[0] == '0');
    plan_tokens.push(1);
    if (ahead) plan_tokens.push(1);
    std::thread planner([&] {
      while (f) {
        (void)plan_tokens.pop();
        if (plan_stop) break;
        Planned pl;
        const double t_p = real_time();
        pl.rc = fem_seqfile_plan(f, batch_bytes, rd_threads, &pl.plan, &pl.shape);
        busy_plan += real_time() - t_p;
        const bool last = !pl.plan || pl.rc != 0 || pl.shape.n_reads == 0;
        planned_q.push(pl);
        if (last) break;
      }
    });
    bool planner_done = false;  // the planner has handed over its last plan (end of input or failure)
    while (f && !exit_code) {
      const double t_pop = real_time();
      BatchBuf *b = free_q.pop();
      if (!b->bases) break;  // its GPU could not provide staging buffers (reported by the worker)
      double t0 = real_time();
      wait_slot += t0 - t_pop;
      b->t_slot = t0;
      if (t_first_slot == 0) t_first_slot = t0;
      Planned pd = planned_q.pop();
      t0 = real_time();  // (the reader is busy from here: the plan was made beside the previous batch)
      fem_batch_plan *plan = pd.plan;
      b->shape = pd.shape;
      int rc = pd.rc;
      if (!plan || rc != 0 || pd.shape.n_reads == 0) planner_done = true;
      bool ok = plan != nullptr;
      if (rc != 0) {  // the reference exits on a truncated file (src/sequence_batch.c:63-66): nothing of this batch is mapped
        fprintf(stderr, "Didn't reach the end of sequence file, which might be corrupted!");
        exit_code = EXIT_FAILURE;
        ok = false;
      }
      if (ok && b->shape.n_reads > 0 && !b->shape.has_qual) {
        fprintf(stderr, "Reads without qualities (FASTA) are not supported: the SAM records need QUAL.\n");
        exit_code = EXIT_FAILURE;
        ok = false;
      }
      if (ok && b->shape.n_reads > 0 &&
          (b->shape.n_reads > b->reads_cap || b->shape.n_bases + 64 > b->bases_cap || (device_text && b->shape.n_name_bytes + 64 > b->names_cap))) {
        b->want_reads = std::max(b->reads_cap, b->shape.n_reads + b->shape.n_reads / 8);
        b->want_bases = std::max(b->bases_cap, b->shape.n_bases + b->shape.n_bases / 8 + 4096);
        b->want_names = b->shape.n_name_bytes + b->shape.n_name_bytes / 8 + 4096;
        work_q[(size_t)b->gpu].push(Msg{kRegrow, b});
        BatchBuf *back = regrown_q.pop();  // (one request at a time: it is ours)
        ok = back->bases != nullptr;
      }
      if (ok && b->shape.n_reads > 0 && !device_text) {
        ok = b->quals.reserve(b->shape.n_bases + 1) && b->names.reserve(b->shape.n_name_bytes + 1) &&
             b->name_off.reserve((b->shape.n_reads + 1) * sizeof(uint64_t));
        if (!ok) {
          fprintf(stderr, "[FEM] out of memory while reading\n");
          exit_code = EXIT_FAILURE;
        }
      }
      if (!ok || b->shape.n_reads == 0) {
        fem_batch_plan_free(plan);
        busy_read += real_time() - t0;
        break;  // end of input (or failure)
      }
      b->packed = false, b->spliced = false, b->n_exc = 0;
      rc = 1;
      if (pack_bases && b->shape.min_len == b->shape.max_len && b->shape.n_bases < 0xFFFFFFF0ull) {
        // reads of one length: two bits per base straight into the staging; 1 = too many characters outside "ACGT" for that form
        uint64_t exc_cap = 0;
        const bool lay = fem_dev_packed_layout(b->shape.n_reads, b->shape.max_len, nullptr, nullptr, &exc_cap) == FEM_OK;
        if (lay && splice) {
          // ... and nothing else copied: where each read's name, bases and qualities lie in the file's mapping (2 = the records
          // do not sit in a mapping that stays: gzip / BGZF windows, the sequential reader)
          const uint64_t n = b->shape.n_reads;
          if (b->r_name.reserve(n * sizeof(char *)) && b->r_seq.reserve(n * sizeof(char *)) && b->r_qual.reserve(n * sizeof(char *)) &&
              b->r_name_len.reserve(n * sizeof(uint32_t))) {
            b->refs.name = (const char **)b->r_name.p, b->refs.seq = (const char **)b->r_seq.p, b->refs.qual = (const char **)b->r_qual.p;
            b->refs.name_len = (uint32_t *)b->r_name_len.p;
            rc = fem_seqfile_fill_packed_refs(f, plan, rd_threads, b->shape.max_len, (uint8_t *)b->bases, exc_cap, &b->n_exc, &b->refs);
            b->packed = b->spliced = rc == 0;
            if (rc == 2) rc = 1;
          }
        } else if (lay && device_text) {
          rc = fem_seqfile_fill_packed(f, plan, rd_threads, b->shape.max_len, (uint8_t *)b->bases, exc_cap, &b->n_exc, b->q_stage, b->n_stage,
                                       b->no_stage);
          b->packed = rc == 0;
        }
      }
      if (rc == 1)
        rc = device_text ? fem_seqfile_fill(f, plan, rd_threads, b->bases, b->off, b->q_stage, b->n_stage, b->no_stage)
                         : fem_seqfile_fill(f, plan, rd_threads, b->bases, b->off, b->quals.p, b->names.p, (uint64_t *)b->name_off.p);
      busy_read += real_time() - t0;
      plan_tokens.push(1);  // (this batch's plan is consumed: the planner may cut the next but one)
      if (rc) {
        fprintf(stderr, "[FEM] reading failed\n");
        exit_code = EXIT_FAILURE;
        break;
      }
      b->t_filled = real_time();
      if (t_first_filled == 0) t_first_filled = b->t_filled;
      work_q[(size_t)b->gpu].push(Msg{kFilled, b});
    }
    // the planner: told to stop if it is still waiting for a token, its unread plans freed
    plan_stop = true;
    plan_tokens.push(1), plan_tokens.push(1);
    planner.join();
    {
      Planned pd;
      while (!planner_done && planned_q.try_pop_for(pd, 0.0)) fem_batch_plan_free(pd.plan);
    }
    // (the file stays open until the mapping phase is over: unmapping 4 GB of faulted-in pages takes 0.08 s, which the GPUs
    // would otherwise spend waiting for their stop message)
    t_reader_done = real_time();
  }
  for (int g = 0; g < n_gpus; ++g) work_q[(size_t)g].push(Msg{kStop, nullptr});
  for (auto &w : workers) w.join();
  t_workers_done = real_time();
  text_q.push(nullptr);
  formatter.join();
  write_q.push(WriteItem{});
  writer.join();
  if (stage_times) {
    double bw = 0;
    for (double x : busy_wait) bw += x;
    double bs = 0, br = 0;
    for (double x : busy_submit) bs += x;
    for (double x : busy_recycle) br += x;
    fprintf(stderr, "[FEM] stage busy seconds: reader %.3f (+ planner %.3f), device wait %.3f, SAM text %.3f, writer %.3f\n", busy_read, busy_plan, bw,
            busy_text, busy_write);
    fprintf(stderr, "[FEM] waiting seconds: reader for a free slot %.3f, formatter for records %.3f and for a text buffer %.3f; "
                    "GPU threads: submit %.3f, slot recycling %.3f\n", wait_slot, wait_records, wait_text_buf, bs, br);
    double cpu_u1 = 0, cpu_s1 = 0;
    cpu_seconds(&cpu_u1, &cpu_s1);
    fprintf(stderr, "[FEM] host CPU in the mapping phase: %.3f s user + %.3f s system = %.1f cores on average\n", cpu_u1 - cpu_u0, cpu_s1 - cpu_s0,
            (cpu_u1 - cpu_u0 + cpu_s1 - cpu_s0) / std::max(1e-9, real_time() - t_start));
    fprintf(stderr, "[FEM] timeline (s after the mapping phase began): first staging slot %.3f, first batch parsed %.3f, input read %.3f, "
                    "devices done %.3f, output written %.3f\n", t_first_slot - t_start, t_first_filled - t_start, t_reader_done - t_start,
            t_workers_done - t_start, real_time() - t_start);
  }
  if (close(out_fd) != 0 && !exit_code.exchange(EXIT_FAILURE)) fprintf(stderr, "[FEM] write error on %s\n", out_path);
  if (n_asserted)
    fprintf(stderr, "[FEM] %lu records on which the reference would have tripped an assertion (src/align.c:366-368) were "
                    "written with CIGAR *\n", (unsigned long)n_asserted.load());

  // MappingStats reduction (src/FEM_map.c:200-212): across GPUs it is one RCCL all-reduce of 5 counters
  if (n_gpus > 1 && !exit_code && !share_gpu) {
    int rc = fem_dev_allreduce_stats(devs.data(), n_gpus, per_gpu.data());
    if (rc) exit_code = dev_fail(devs[0], "stats all-reduce", rc);
  } else if (n_gpus > 1) {
    for (int g = 1; g < n_gpus; ++g)
      for (int i = 0; i < 5; ++i) per_gpu[(size_t)i] += per_gpu[(size_t)g * 5 + (size_t)i];
  }
  uint64_t totals[5];
  for (int i = 0; i < 5; ++i) totals[i] = per_gpu[(size_t)i];
  const double t_mapping = real_time() - t_start;  // (the reference's timer stops after the counter reduction, src/FEM_map.c:219)
  for (TextOut &t : texts) free(t.buf);
  if (read_file) fem_seqfile_close(read_file);
  for (fem_dev *h : devs) fem_dev_close(h);
  if (exit_code) return exit_code;
  fprintf(stderr, "The number of read: %lu\n", (unsigned long)totals[0]);
  fprintf(stderr, "The number of mapped read: %lu\n", (unsigned long)totals[1]);
  fprintf(stderr, "The number of candidate before additional q-gram filter: %lu\n", (unsigned long)totals[2]);
  fprintf(stderr, "The number of candidate: %lu\n", (unsigned long)totals[3]);
  fprintf(stderr, "The number of mapping: %lu\n", (unsigned long)totals[4]);
  fprintf(stderr, "Time: %fs\n", t_mapping);
  return 0;
}

}  // namespace

int main(int argc, char *argv[]) {
  if (argc < 2) {
    fprintf(stderr, "%s\n", "Too few arguements.");
    usage_main();
    exit(EXIT_FAILURE);
  }
  int rv = 0;
  double t0 = real_time(), c0 = cpu_time();
  if (strcmp(argv[1], "index") == 0) {
    rv = index_main(argc - 1, argv + 1);
  } else if (strcmp(argv[1], "map") == 0) {
    rv = map_main(argc - 1, argv + 1);
  } else {
    fprintf(stderr, "[%s] unrecognized command '%s'\n", __func__, argv[1]);
    exit(EXIT_FAILURE);
  }
  if (rv == 0) {
    fprintf(stderr, "[%s] Version: %s\n", __func__, FEM_VERSION);
    fprintf(stderr, "[%s] CMD:", __func__);
    for (int i = 0; i < argc; ++i) fprintf(stderr, " %s", argv[i]);
    fprintf(stderr, "\n[%s] Real time: %.3f sec; CPU: %.3f sec\n", __func__, real_time() - t0, cpu_time() - c0);
  }
  return rv;
}
