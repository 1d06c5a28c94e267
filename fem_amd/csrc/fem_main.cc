// fem_main.cc — the drop-in command line: `FEM index` and `FEM map` (reference src/FEM.c, src/FEM_index.c,
// src/FEM_map.c).  Same verbs, flags, index file format and SAM output; the per-read hot path runs on the GPU
// through libfemhip.so (include/fem_hip.h).  There is no CPU mapping path in this binary: without a GPU it fails.
//
// New, optional: `--gpus N` (map) shards read batches over N GPUs of this node; `--batch N` sets reads per batch.
#include <getopt.h>
#include <malloc.h>
#include <sys/resource.h>
#include <sys/time.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
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
struct Batch {
  uint64_t id = 0;
  fem_seqset reads{};
  bool last = false;
};

template <typename T>
class Channel {  // bounded hand-off between pipeline stages
 public:
  explicit Channel(size_t cap) : cap_(cap) {}
  void push(T v) {
    std::unique_lock<std::mutex> l(m_);
    not_full_.wait(l, [&] { return q_.size() < cap_; });
    q_.push_back(std::move(v));
    not_empty_.notify_one();
  }
  T pop() {
    std::unique_lock<std::mutex> l(m_);
    not_empty_.wait(l, [&] { return !q_.empty(); });
    T v = std::move(q_.front());
    q_.pop_front();
    not_full_.notify_one();
    return v;
  }

 private:
  std::mutex m_;
  std::condition_variable not_empty_, not_full_;
  std::deque<T> q_;
  size_t cap_;
};

int map_main(int argc, char **argv) {
  // Batches come and go as a few 100 MB allocations: keep them in the heap instead of mapping and unmapping them, so
  // that a recycled buffer's pages are already there (page faults were a visible share of the host time).
  mallopt(M_MMAP_MAX, 0);
  mallopt(M_TRIM_THRESHOLD, -1);
  char *ref_path = nullptr, *index_path = nullptr, *read_path = nullptr, *out_path = nullptr;
  fem_params params{12, 3, 2, 1};  // src/FEM_map.c:67-70: k and step are fixed, whatever the index header says
  int n_threads = 1, n_gpus = 1;
  uint64_t batch_reads = 250000;  // (8 M reads, FASTQ -> SAM: 250 k per batch 17-22 Mreads/s, 1 M per batch 9-14: the pipeline fills sooner)
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
      case 'B': batch_reads = strtoull(optarg, nullptr, 10); break;
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

  std::vector<fem_dev *> devs((size_t)n_gpus, nullptr);
  for (int g = 0; g < n_gpus; ++g) {
    int rc = fem_dev_open(g, &devs[(size_t)g]);
    if (rc) return dev_fail(nullptr, "fem_dev_open (mapping runs on the GPU; no CPU path)", rc);
    if ((rc = ref.upload(devs[(size_t)g]))) return dev_fail(devs[(size_t)g], "reference upload", rc);
    if ((rc = fem_dev_upload_index(devs[(size_t)g], ik, istep, lookup, ((uint64_t)1 << (2 * ik)) + 1, occ, n_occ)))
      return dev_fail(devs[(size_t)g], "index upload", rc);
  }
  free(lookup), free(occ);

  FILE *out = fopen(out_path, "w");
  if (!out) {
    fprintf(stderr, "Cannot open output file %s\n", out_path);
    exit(EXIT_FAILURE);
  }
  {
    char *hdr = nullptr;
    uint64_t hl = 0;
    fem_sam_header(&ref.view, &hdr, &hl);
    fwrite(hdr, 1, hl, out);
    free(hdr);
  }

  double t_start = real_time();
  const uint64_t batch_bytes = batch_reads * 250ull;  // header + bases + '+' + qualities of a ~100 bp record
  // stage 4: one writer thread drains formatted SAM text (src/output_queue.c:60-91)
  struct Text {
    char *p;
    uint64_t n;
  };
  Channel<Text> to_write(4);
  // FEM_STAGE_TIMES=1: busy seconds of each pipeline stage on stderr at the end (reader, device wait, SAM text, writer)
  const char *st_env = getenv("FEM_STAGE_TIMES");
  const bool stage_times = st_env && st_env[0] == '1';
  double busy_read = 0, busy_wait = 0, busy_text = 0, busy_write = 0;
  std::thread writer([&] {
    for (;;) {
      Text t = to_write.pop();
      if (!t.p) break;
      double t0 = real_time();
      fwrite(t.p, 1, t.n, out);
      free(t.p);
      busy_write += real_time() - t0;
    }
  });
  // stage 1: one reader thread parses FASTQ into batches (src/input_queue.c:53-79)
  Channel<Batch *> parsed(4);
  std::thread reader([&] {
    fem_seqfile *f = fem_seqfile_open(read_path);
    uint64_t id = 0;
    bool ok = f != nullptr;
    if (!ok) fprintf(stderr, "Cannot find sequence file!");
    for (;;) {
      Batch *b = new Batch();
      b->id = id++;
      // batches are cut by bytes (~ batch_reads records of this file's shape); plain FASTQ is parsed by all threads
      double t0 = real_time();
      int rc = ok ? fem_seqfile_read_bytes(f, batch_bytes, n_threads, &b->reads) : -1;
      busy_read += real_time() - t0;
      if (rc != 0 && ok) fprintf(stderr, "Didn't reach the end of sequence file, which might be corrupted!");
      if (rc != 0 || b->reads.n == 0) {
        b->last = true;
        parsed.push(b);
        break;
      }
      parsed.push(b);
    }
    if (f) fem_seqfile_close(f);
  });

  // stage 2 (this thread): device submit / wait, batches dealt round-robin to GPUs, two slots per GPU;
  // stage 3: mapping tail + SAM text of the previous batch overlaps the kernels of the next one.
  struct InFlight {
    Batch *b;
    int gpu, slot;
  };
  std::deque<InFlight> flight;
  uint64_t totals[5] = {0, 0, 0, 0, 0};
  std::vector<uint64_t> per_gpu((size_t)n_gpus * 5, 0);
  int exit_code = 0;
  const char *ht = getenv("FEM_HOST_TAIL");
  const bool host_tail = ht && ht[0] == '1';
  auto retire = [&](InFlight f) {
    double t0 = real_time();
    uint64_t stats[5] = {0, 0, 0, 0, 0};
    char *text = nullptr;
    uint64_t len = 0;
    int rc, fmt = 0;
    if (host_tail) {  // FEM_HOST_TAIL=1: ordering / traceback / MD by libfemhost from the per-candidate outcome
      fem_batch_result res;
      rc = fem_dev_map_batch_wait(devs[(size_t)f.gpu], f.slot, &res);
      if (!rc) {
        fem_tail_input in{res.n_reads, res.cand_begin, res.cand_count, res.cand, res.ed, res.end};
        fmt = fem_tail_sam(params.e, &ref.view, &f.b->reads, &in, n_threads, &text, &len);
        memcpy(stats, res.stats, sizeof stats);
      }
    } else {  // default: the records come off the device, the host only renders text
      fem_batch_records rec;
      rc = fem_dev_fetch_records(devs[(size_t)f.gpu], f.slot, &rec);
      busy_wait += real_time() - t0;
      if (!rc) {
        double t1 = real_time();
        fem_record_view rv{rec.n_reads, rec.n_records, rec.rec_begin, rec.flag, rec.tid, rec.pos0, rec.nm,
                           rec.cigar_off, rec.cigar, rec.md_off, rec.md};
        fmt = fem_records_sam(&ref.view, &f.b->reads, &rv, n_threads, &text, &len);
        busy_text += real_time() - t1;
        memcpy(stats, rec.stats, sizeof stats);
      }
    }
    if (rc) {
      exit_code = dev_fail(devs[(size_t)f.gpu], "mapping", rc);
    } else {
      if (fmt != 0) {
        fprintf(stderr, "[FEM] out of memory while formatting SAM records\n");
        exit_code = EXIT_FAILURE;
      } else {
        to_write.push(Text{text, len});
      }
      for (int i = 0; i < 5; ++i) per_gpu[(size_t)f.gpu * 5 + (size_t)i] += stats[i];
      fprintf(stderr, "Mapped read batch in %fs.\n", real_time() - t0);
    }
    fem_seqset_free(&f.b->reads);
    delete f.b;
  };
  uint64_t n_submitted = 0;
  for (;;) {
    Batch *b = parsed.pop();
    if (b->last || exit_code) {
      fem_seqset_free(&b->reads);
      bool was_last = b->last;
      delete b;
      if (was_last) break;
      continue;
    }
    if (!b->reads.quals) {
      fprintf(stderr, "Reads without qualities (FASTA) are not supported: the SAM records need QUAL.\n");
      exit_code = EXIT_FAILURE;
      fem_seqset_free(&b->reads);
      delete b;
      continue;
    }
    int gpu = (int)(n_submitted % (uint64_t)n_gpus), slot = (int)((n_submitted / (uint64_t)n_gpus) % 2);
    while (flight.size() >= (size_t)n_gpus * 2 ||
           (!flight.empty() && flight.front().gpu == gpu && flight.front().slot == slot)) {
      retire(flight.front());
      flight.pop_front();
    }
    fem_read_batch rb{b->reads.bases, b->reads.off, b->reads.n};
    int rc = fem_dev_map_batch_submit(devs[(size_t)gpu], slot, &params, &rb);
    if (rc) {
      exit_code = dev_fail(devs[(size_t)gpu], "batch submit", rc);
      fem_seqset_free(&b->reads);
      delete b;
      continue;
    }
    flight.push_back({b, gpu, slot});
    ++n_submitted;
  }
  while (!flight.empty()) {
    retire(flight.front());
    flight.pop_front();
  }
  reader.join();
  to_write.push(Text{nullptr, 0});
  writer.join();
  if (stage_times)
    fprintf(stderr, "[FEM] stage busy seconds: reader %.3f, device wait %.3f, SAM text %.3f, writer %.3f\n", busy_read,
            busy_wait, busy_text, busy_write);
  fclose(out);

  // MappingStats reduction (src/FEM_map.c:200-212): across GPUs it is one RCCL all-reduce of 5 counters
  if (n_gpus > 1) {
    int rc = fem_dev_allreduce_stats(devs.data(), n_gpus, per_gpu.data());
    if (rc) exit_code = dev_fail(devs[0], "stats all-reduce", rc);
  }
  for (int i = 0; i < 5; ++i) totals[i] = per_gpu[(size_t)i];
  for (fem_dev *h : devs) fem_dev_close(h);
  if (exit_code) return exit_code;
  fprintf(stderr, "The number of read: %lu\n", (unsigned long)totals[0]);
  fprintf(stderr, "The number of mapped read: %lu\n", (unsigned long)totals[1]);
  fprintf(stderr, "The number of candidate before additional q-gram filter: %lu\n", (unsigned long)totals[2]);
  fprintf(stderr, "The number of candidate: %lu\n", (unsigned long)totals[3]);
  fprintf(stderr, "The number of mapping: %lu\n", (unsigned long)totals[4]);
  fprintf(stderr, "Time: %fs\n", real_time() - t_start);
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
