// fem_host.cc — host side of the drop-in (see fem_host.h).  C++17, OpenMP for the per-read loops.
#include "fem_host.h"
#include "fem_pack.h"

#include <fcntl.h>
#include <omp.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
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
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------------------
// The FASTQ parser's threads.  Not OpenMP's: libgomp's threads SPIN for milliseconds after every parallel region, and FEM map
// has two regions per batch every 8 ms — sixteen threads that never sleep (FEM map, 32 M reads of C3 on 16 cores: 14.9 cores
// busy, 9.7 with the spinning off, same Mreads/s; the wait policy is read when libgomp is loaded: a program cannot set it for
// itself).  These sleep on a condition variable between jobs; the caller takes indices like a worker.
// ------------------------------------------------------------------------------------------------
namespace {
class SleepingPool {
 public:
  ~SleepingPool() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
  }
  // fn(0) .. fn(n - 1), each once, on up to n threads
  void run(int n, const std::function<void(int)> &fn) {
    if (n <= 1) {
      for (int i = 0; i < n; ++i) fn(i);
      return;
    }
    std::lock_guard<std::mutex> one_job(call_mu_);
    {
      std::lock_guard<std::mutex> l(mu_);
      while ((int)threads_.size() < n - 1 && threads_.size() < 255) threads_.emplace_back([this] { worker(); });
      fn_ = &fn, n_ = n, next_ = 0, left_ = n, ++gen_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> l(mu_);
    done_cv_.wait(l, [&] { return left_ == 0; });
    fn_ = nullptr;
  }

 private:
  void work() {
    for (;;) {
      int i;
      const std::function<void(int)> *fn;
      {
        std::lock_guard<std::mutex> l(mu_);
        if (!fn_ || next_ >= n_) return;
        i = next_++, fn = fn_;
      }
      (*fn)(i);
      std::lock_guard<std::mutex> l(mu_);
      if (--left_ == 0) done_cv_.notify_all();
    }
  }
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return gen_ != seen || stop_; });
        if (stop_) return;
        seen = gen_;
      }
      work();
    }
  }
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> threads_;
  const std::function<void(int)> *fn_ = nullptr;
  int n_ = 0, next_ = 0, left_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};
SleepingPool &parser_pool() {
  static SleepingPool pool;
  return pool;
}
SleepingPool &planner_pool() {  // the plan's pass has threads of its own: the NEXT batch may be planned while this one is filled
  static SleepingPool pool;
  return pool;
}
}  // namespace

namespace {

// ------------------------------------------------------------------------------------------------
// encodings (reference src/utils.h:72-81)
// ------------------------------------------------------------------------------------------------
struct CodeTable {
  uint8_t t[256];
  constexpr CodeTable() : t() {
    for (int i = 0; i < 256; ++i) t[i] = 4;
    t['A'] = t['a'] = 0;
    t['C'] = t['c'] = 1;
    t['G'] = t['g'] = 2;
    t['T'] = t['t'] = 3;
  }
};
constexpr CodeTable kCode;
inline uint8_t code_of(char c) { return kCode.t[(uint8_t)c]; }
constexpr char kCodeChar[8] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N'};

template <typename T>
T *dup_vec(const std::vector<T> &v) {
  T *p = (T *)malloc(std::max<size_t>(v.size(), 1) * sizeof(T));
  if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

// ------------------------------------------------------------------------------------------------
// sequence files: a FASTA/FASTQ reader with kseq's record rules (reference src/kseq.h:185-226)
// ------------------------------------------------------------------------------------------------
class ByteStream {
 public:
  // compressed source: read(buf, cap) returns the bytes it produced, 0 at the end of the stream, -1 on an error
  explicit ByteStream(std::function<int(char *, unsigned)> read) : read_(std::move(read)), buf_(1u << 20) {}
  ByteStream(const char *mem, size_t len) : mem_(mem), mem_len_(len) {}
  // bytes handed out so far (memory source: the offset of the next byte in the map)
  size_t tell() const { return mem_ ? mem_pos_ - (end_ - pos_) : consumed_ - (end_ - pos_); }
  // gz source: `n` bytes that were read ahead (fem_seqfile's window) go back in front of the stream
  void prepend(const char *p, size_t n) {
    pre_.assign(p, p + n);
    pre_pos_ = 0;
  }
  void set_error() { eof_ = true, last_rc_ = -3; }
  void set_eof() { eof_ = true, last_rc_ = -1; }
  void seek_mem(size_t at) {  // memory source only
    mem_pos_ = at;
    pos_ = end_ = 0;
    eof_ = false;
  }
  int get() {  // next byte, -1 at end of file, -3 on a read error
    if (pos_ >= end_) {
      if (!refill()) return last_rc_;
    }
    return (unsigned char)cur_[pos_++];
  }
  // Append bytes up to (not including) the next delimiter; returns the delimiter, -1 at EOF (with or
  // without bytes appended; *got tells), -3 on error.  line=true: delimiter '\n'; else any isspace().
  int until(bool line, std::string *dst, bool *got) {
    *got = false;
    for (;;) {
      if (pos_ >= end_) {
        if (!refill()) return last_rc_;
      }
      size_t i = pos_;
      if (line) {
        const char *nl = (const char *)memchr(cur_ + pos_, '\n', end_ - pos_);
        i = nl ? (size_t)(nl - cur_) : end_;
      } else {
        while (i < end_ && !isspace((unsigned char)cur_[i])) ++i;
      }
      dst->append(cur_ + pos_, i - pos_);
      *got = true;
      pos_ = i;
      if (i < end_) {
        ++pos_;
        return (unsigned char)cur_[i];
      }
    }
  }

 private:
  bool refill() {
    if (pre_pos_ < pre_.size()) {  // bytes handed back by the window reader come first
      cur_ = pre_.data() + pre_pos_;
      const size_t n = pre_.size() - pre_pos_;
      pre_pos_ = pre_.size();
      consumed_ += n;
      pos_ = 0, end_ = n;
      return true;
    }
    if (eof_) {
      return false;  // (last_rc_ keeps -1 or -3)
    }
    if (mem_) {
      if (mem_pos_ >= mem_len_) {
        eof_ = true;
        last_rc_ = -1;
        return false;
      }
      size_t n = std::min<size_t>(mem_len_ - mem_pos_, 1u << 20);
      cur_ = mem_ + mem_pos_;
      mem_pos_ += n;
      pos_ = 0, end_ = n;
      return true;
    }
    int n = read_(buf_.data(), (unsigned)buf_.size());
    if (n <= 0) {
      eof_ = true;
      last_rc_ = n < 0 ? -3 : -1;
      return false;
    }
    cur_ = buf_.data();
    consumed_ += (size_t)n;
    pos_ = 0, end_ = (size_t)n;
    return true;
  }
  std::function<int(char *, unsigned)> read_;
  const char *mem_ = nullptr;
  size_t mem_len_ = 0, mem_pos_ = 0, consumed_ = 0;
  std::vector<char> buf_, pre_;
  size_t pre_pos_ = 0;
  const char *cur_ = nullptr;
  size_t pos_ = 0, end_ = 0;
  bool eof_ = false;
  int last_rc_ = -1;
};

}  // namespace

namespace {
// ---- BGZF blocks (the SAM specification, section 4.1: a gzip member with the extra subfield 'B' 'C' = block size - 1) ----
struct BgzfBlock {
  size_t at = 0, payload = 0, payload_len = 0;  // member offset, start and length of its deflate stream
  uint32_t isize = 0, crc = 0;                   // uncompressed length and CRC-32 from the trailer
  size_t next = 0;                               // offset of the next member
};
// false: no well-formed BGZF member at z[at] (or it runs past the end of the file)
bool bgzf_block_at(const unsigned char *z, size_t len, size_t at, BgzfBlock *b) {
  if (at + 18 > len) return false;
  const unsigned char *h = z + at;
  if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
  const size_t xlen = h[10] | (size_t)h[11] << 8;
  if (at + 12 + xlen > len) return false;
  size_t bsize = 0;
  for (size_t x = 0; x + 4 <= xlen;) {  // subfields: SI1 SI2 SLEN(2) data
    const unsigned char *sf = h + 12 + x;
    const size_t slen = sf[2] | (size_t)sf[3] << 8;
    if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (sf[4] | (size_t)sf[5] << 8) + 1;
    x += 4 + slen;
  }
  if (bsize < 12 + xlen + 8 || at + bsize > len) return false;
  b->at = at, b->payload = at + 12 + xlen, b->payload_len = bsize - (12 + xlen) - 8, b->next = at + bsize;
  const unsigned char *t = z + at + bsize - 8;
  b->crc = t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
  b->isize = t[4] | (uint32_t)t[5] << 8 | (uint32_t)t[6] << 16 | (uint32_t)t[7] << 24;
  return b->isize <= 65536u;
}
bool bgzf_inflate(const unsigned char *z, const BgzfBlock &b, char *out) {
  if (b.isize == 0) return true;
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (inflateInit2(&zs, -15) != Z_OK) return false;
  zs.next_in = const_cast<Bytef *>(z + b.payload), zs.avail_in = (uInt)b.payload_len;
  zs.next_out = (Bytef *)out, zs.avail_out = b.isize;
  const int rc = inflate(&zs, Z_FINISH);
  const bool ok = rc == Z_STREAM_END && zs.total_out == b.isize;
  inflateEnd(&zs);
  return ok && (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)out, b.isize) == b.crc;
}
}  // namespace

struct fem_seqfile {
  gzFile gz = nullptr;
  ByteStream *in = nullptr;
  const char *map = nullptr;  // plain (not gzip) regular files are memory-mapped
  size_t map_len = 0;
  // gzip input: inflated a window at a time (one thread: a gzip stream has no entry points), each window parsed by
  // all threads like a stretch of a mapped file
  std::vector<char> win;
  size_t win_lo = 0, win_len = 0;  // unparsed bytes of the window: [win_lo, win_len)
  bool gz_eof = false, gz_err = false;
  // BGZF (bgzip; gzip members of at most 64 KiB that carry their own size): the compressed file is mapped and the blocks
  // of a window are inflated by all threads
  const unsigned char *zmap = nullptr;
  size_t zlen = 0, zpos = 0;
  int threads = 1;  // host threads the caller of the current batch call allows
  bool fast_ok = true;        // 4-line FASTQ so far: the multi-threaded parser may be used
  bool read_again = false;    // the mapping's pages are read a second time (the spliced formatter): no longer advised as sequential
  int last_char = 0;  // header character already consumed by the previous record
  std::string name, comment, seq, qual;
  // one record; returns sequence length, -1 end of file, -2 truncated quality, -3 stream error
  long next() {
    int c;
    if (last_char == 0) {
      while ((c = in->get()) >= 0 && c != '>' && c != '@') {
      }
      if (c < 0) return c;
      last_char = c;
    }
    name.clear(), comment.clear(), seq.clear(), qual.clear();
    bool got;
    c = in->until(false, &name, &got);
    if (c < 0 && !got) return c;
    if (c >= 0 && c != '\n') {
      int d = in->until(true, &comment, &got);
      if (d == -3) return -3;
    }
    while ((c = in->get()) >= 0 && c != '>' && c != '+' && c != '@') {
      if (c == '\n') continue;
      seq.push_back((char)c);
      int d = in->until(true, &seq, &got);
      if (d == -3) return -3;
      if (seq.size() > 1 && seq.back() == '\r') seq.pop_back();  // kseq strips a trailing CR (src/kseq.h ks_getuntil2)
    }
    if (c == '>' || c == '@') last_char = c;
    if (c != '+') {
      if (c != '>' && c != '@') last_char = 0;
      return (long)seq.size();  // FASTA
    }
    while ((c = in->get()) >= 0 && c != '\n') {
    }
    if (c == -1) return -2;
    do {  // at least one line, even behind an empty sequence (src/kseq.h:222: the call sits in the loop's condition)
      int d = in->until(true, &qual, &got);
      if (d == -3) return -3;
      if (qual.size() > 1 && qual.back() == '\r') qual.pop_back();
      if (d < 0 && !got) break;  // end of file with nothing read
    } while (qual.size() < seq.size());
    last_char = 0;
    if (seq.size() != qual.size()) return -2;
    return (long)seq.size();
  }
};

extern "C" {

fem_seqfile *fem_seqfile_open(const char *path) {
  // plain regular file -> memory map (parallel parser possible); gzip or anything else -> zlib stream
  int fd = open(path, O_RDONLY);
  if (fd < 0) return nullptr;
  struct stat st;
  unsigned char magic[2] = {0, 0};
  bool plain = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && pread(fd, magic, 2, 0) == 2 &&
               !(magic[0] == 0x1f && magic[1] == 0x8b);
  fem_seqfile *f = new fem_seqfile();
  if (plain) {
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m != MAP_FAILED) {
      (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
      f->map = (const char *)m;
      f->map_len = (size_t)st.st_size;
      f->in = new ByteStream(f->map, f->map_len);
      close(fd);
      return f;
    }
  }
  // BGZF: regular file whose first member carries the 'BC' subfield -> map the compressed bytes, inflate block-wise
  if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 28) {
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    BgzfBlock b;
    if (m != MAP_FAILED && bgzf_block_at((const unsigned char *)m, (size_t)st.st_size, 0, &b)) {
      close(fd);
      f->zmap = (const unsigned char *)m, f->zlen = (size_t)st.st_size;
      f->in = new ByteStream([f](char *buf, unsigned cap) -> int {  // the sequential reader: one block at a time
        for (;;) {
          if (f->zpos >= f->zlen) return 0;
          BgzfBlock blk;
          if (!bgzf_block_at(f->zmap, f->zlen, f->zpos, &blk) || blk.isize > cap) return -1;
          f->zpos = blk.next;
          if (blk.isize == 0) continue;  // (the empty end-of-file block, or an empty member in between)
          return bgzf_inflate(f->zmap, blk, buf) ? (int)blk.isize : -1;
        }
      });
      return f;
    }
    if (m != MAP_FAILED) munmap(m, (size_t)st.st_size);
  }
  close(fd);
  gzFile gz = gzopen(path, "r");
  if (!gz) {
    delete f;
    return nullptr;
  }
  gzbuffer(gz, 1u << 20);
  f->gz = gz;
  f->in = new ByteStream([gz](char *buf, unsigned cap) -> int {
    const int n = gzread(gz, buf, cap);
    int zerr = Z_OK;
    if (n <= 0) (void)gzerror(gz, &zerr);  // a stream that stops short reads as 0 bytes + Z_BUF_ERROR, not as -1
    return n < 0 || zerr == Z_BUF_ERROR || zerr == Z_DATA_ERROR ? -1 : n;
  });
  return f;
}

void fem_seqfile_close(fem_seqfile *f) {
  if (!f) return;
  delete f->in;
  if (f->map) munmap((void *)f->map, f->map_len);
  if (f->zmap) munmap((void *)f->zmap, f->zlen);
  if (f->gz) gzclose(f->gz);
  delete f;
}

namespace {

struct ParsedChunk {
  std::string bases, quals, names;
  std::vector<uint32_t> len, name_len;
  bool any_qual = false, all_qual = true;
};

int finish_seqset(const std::vector<ParsedChunk> &parts, fem_seqset *out) {
  uint64_t n = 0, nb = 0, nn = 0;
  bool any_qual = false, all_qual = true;
  for (const ParsedChunk &c : parts) {
    n += c.len.size(), nb += c.bases.size(), nn += c.names.size();
    any_qual = any_qual || c.any_qual;
    all_qual = all_qual && c.all_qual;
  }
  memset(out, 0, sizeof *out);
  out->n = n;
  out->bases = (char *)malloc(nb + 64);
  out->names = (char *)malloc(nn + 1);
  out->off = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
  out->name_off = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
  const bool keep_qual = any_qual && all_qual;
  if (keep_qual) out->quals = (char *)malloc(nb + 1);
  if (!out->bases || !out->names || !out->off || !out->name_off || (keep_qual && !out->quals)) return -4;
  std::vector<uint64_t> r0(parts.size() + 1, 0), b0(parts.size() + 1, 0), n0(parts.size() + 1, 0);
  for (size_t i = 0; i < parts.size(); ++i) {
    r0[i + 1] = r0[i] + parts[i].len.size();
    b0[i + 1] = b0[i] + parts[i].bases.size();
    n0[i + 1] = n0[i] + parts[i].names.size();
  }
#pragma omp parallel for schedule(static, 1) num_threads((int)std::max<size_t>(1, std::min<size_t>(parts.size(), 64)))
  for (int64_t i = 0; i < (int64_t)parts.size(); ++i) {
    const ParsedChunk &c = parts[(size_t)i];
    memcpy(out->bases + b0[i], c.bases.data(), c.bases.size());
    if (keep_qual) memcpy(out->quals + b0[i], c.quals.data(), c.quals.size());
    memcpy(out->names + n0[i], c.names.data(), c.names.size());
    uint64_t b = b0[i], m = n0[i];
    for (size_t k = 0; k < c.len.size(); ++k) {
      out->off[r0[i] + k] = b;
      out->name_off[r0[i] + k] = m;
      b += c.len[k];
      m += c.name_len[k];
    }
  }
  out->off[n] = nb;
  out->name_off[n] = nn;
  memset(out->bases + nb, 0, 64);
  return 0;
}

void push_record(ParsedChunk &c, const char *name, size_t name_len, const char *seq, size_t len, const char *qual) {
  c.bases.append(seq, len);
  if (qual) {
    c.any_qual = true;
    c.quals.append(qual, len);
  } else {
    c.all_qual = false;
    c.quals.append(len, '\0');
  }
  c.names.append(name, name_len);
  c.len.push_back((uint32_t)len);
  c.name_len.push_back((uint32_t)name_len);
}

// start of the first 4-line FASTQ record at or after `from`: a line that begins with '@' whose second-next line
// begins with '+' (a quality line may begin with '@', but then the second-next line is a sequence line)
size_t next_fastq_record(const char *m, size_t len, size_t from) {
  size_t p = from;
  if (p > 0 && m[p - 1] != '\n') {
    const char *nl = (const char *)memchr(m + p, '\n', len - p);
    if (!nl) return len;
    p = (size_t)(nl - m) + 1;
  }
  while (p < len) {
    const char *l1 = (const char *)memchr(m + p, '\n', len - p);
    if (!l1) return len;
    const char *l2 = (const char *)memchr(l1 + 1, '\n', len - (size_t)(l1 + 1 - m));
    if (!l2) return len;
    if (m[p] == '@' && (size_t)(l2 + 1 - m) < len && l2[1] == '+') return p;
    p = (size_t)(l1 - m) + 1;
  }
  return len;
}

// strict 4-line FASTQ over [lo, hi) (record boundaries); false if anything else shows up
bool parse_fastq_range(const char *m, size_t lo, size_t hi, ParsedChunk &c) {
  size_t p = lo;
  while (p < hi) {
    if (m[p] == '\n') {  // blank line between records
      ++p;
      continue;
    }
    if (m[p] != '@') return false;
    const char *e0 = (const char *)memchr(m + p, '\n', hi - p);
    if (!e0) return false;
    const char *s1 = e0 + 1;
    const char *e1 = (const char *)memchr(s1, '\n', hi - (size_t)(s1 - m));
    if (!e1) return false;
    const char *s2 = e1 + 1;
    if (s2 >= m + hi || *s2 != '+') return false;
    const char *e2 = (const char *)memchr(s2, '\n', hi - (size_t)(s2 - m));
    if (!e2) return false;
    const char *s3 = e2 + 1;
    const char *e3 = (const char *)memchr(s3, '\n', hi - (size_t)(s3 - m));
    if (!e3) e3 = m + hi;  // last line of the file without a newline
    size_t name_len = 0;
    while (m + p + 1 + name_len < e0 && !isspace((unsigned char)m[p + 1 + name_len])) ++name_len;
    size_t sl = (size_t)(e1 - s1), ql = (size_t)(e3 - s3);
    if (sl > 1 && s1[sl - 1] == '\r') --sl;  // kseq strips a trailing CR
    if (ql > 1 && s3[ql - 1] == '\r') --ql;
    if (sl != ql) return false;                           // multi-line or truncated: let the exact parser decide
    if (sl > 0 && (s1[0] == '>' || s1[0] == '@' || s1[0] == '+')) return false;
    if (sl > 0) push_record(c, m + p + 1, name_len, s1, sl, s3);  // zero-length records are skipped
    p = (size_t)(e3 - m) + (e3 < m + hi ? 1 : 0);
  }
  return true;
}


// ---- a stretch of the input in memory for the multi-threaded 4-line FASTQ parser: the file's mapping, or (gzip) a
//      window inflated ahead of the parser ----
struct FastView {
  const char *m = nullptr;
  size_t len = 0, lo = 0;
  bool whole = true;  // m[len] is the end of the file (a window of a gzip stream usually is not)
};
size_t next_fastq_record(const char *m, size_t len, size_t from);
// start of the last record that begins in m[lo, len) (it may be cut off by the end of the window); len if none is found
size_t last_fastq_record(const char *m, size_t lo, size_t len) {
  size_t from = len > lo + (1u << 16) ? len - (1u << 16) : lo;
  size_t cur = next_fastq_record(m, len, from), last = cur;
  while (cur < len) {
    cur = next_fastq_record(m, len, cur + 1);
    if (cur < len) last = cur;
  }
  return last;
}
bool fast_view(fem_seqfile *f, uint64_t approx_bytes, FastView *v);
void fast_view_consumed(fem_seqfile *f, size_t upto);
void fast_view_abandon(fem_seqfile *f, size_t pos0);

bool fast_view(fem_seqfile *f, uint64_t approx_bytes, FastView *v) {
  if (!f->fast_ok || f->last_char != 0) return false;
  if (f->map) {
    v->m = f->map, v->len = f->map_len, v->lo = f->in->tell(), v->whole = true;
    return true;
  }
  if ((!f->gz && !f->zmap) || f->gz_err) return false;
  // gzip: drop what has been parsed, inflate until the window holds approx_bytes (+ room for the record that straddles
  // its end), or everything when approx_bytes == 0
  if (f->win_lo > 0) {
    memmove(f->win.data(), f->win.data() + f->win_lo, f->win_len - f->win_lo);
    f->win_len -= f->win_lo, f->win_lo = 0;
  }
  const size_t slack = 1u << 20;
  size_t target = approx_bytes ? (size_t)approx_bytes + slack : (size_t)64 << 20;
  while (f->zmap && !f->gz_eof) {  // BGZF: the blocks that fill the window, inflated side by side
    std::vector<BgzfBlock> blocks;
    std::vector<size_t> out_at;
    size_t total = f->win_len, z = f->zpos;
    while (total < target && z < f->zlen) {
      BgzfBlock b;
      if (!bgzf_block_at(f->zmap, f->zlen, z, &b)) {
        f->gz_err = true;  // cut off inside a block, or not BGZF any more
        f->in->set_error();
        return false;
      }
      blocks.push_back(b), out_at.push_back(total);
      total += b.isize, z = b.next;
    }
    if (f->win.size() < total) f->win.resize(total);
    bool ok = true;
#pragma omp parallel for schedule(dynamic, 4) num_threads(std::max(1, f->threads)) reduction(&& : ok)
    for (int64_t i = 0; i < (int64_t)blocks.size(); ++i)
      ok = bgzf_inflate(f->zmap, blocks[(size_t)i], f->win.data() + out_at[(size_t)i]) && ok;
    if (!ok) {
      f->gz_err = true;
      f->in->set_error();
      return false;
    }
    f->win_len = total, f->zpos = z;
    if (z >= f->zlen) f->gz_eof = true;
    if (approx_bytes || f->gz_eof) break;
    target *= 2;  // the whole file was asked for
  }
  while (f->gz && !f->gz_eof) {
    if (f->win.size() < target) f->win.resize(target);
    while (f->win_len < target && !f->gz_eof) {
      const int n = gzread(f->gz, f->win.data() + f->win_len, (unsigned)std::min<size_t>(target - f->win_len, 1u << 26));
      int zerr = Z_OK;
      if (n <= 0) (void)gzerror(f->gz, &zerr);  // a stream that stops short reads as 0 bytes + Z_BUF_ERROR, not as -1
      if (n < 0 || zerr == Z_BUF_ERROR || zerr == Z_DATA_ERROR) {
        f->gz_err = true;  // truncated or damaged stream: "Didn't reach the end of sequence file"
        f->in->set_error();
        return false;
      }
      if (n == 0) f->gz_eof = true;
      f->win_len += (size_t)n;
    }
    if (approx_bytes) break;
    target *= 2;  // the whole file was asked for
  }
  if (f->gz_eof) f->in->set_eof();  // (the sequential reader, should it take over, has nothing behind the window)
  v->m = f->win.data(), v->len = f->win_len, v->lo = 0, v->whole = f->gz_eof;
  return true;
}
void fast_view_consumed(fem_seqfile *f, size_t upto) {
  if (f->map) f->in->seek_mem(upto);
  else f->win_lo = upto;
}
// the window is not 4-line FASTQ: the sequential reader takes over where the fast one stood
void fast_view_abandon(fem_seqfile *f, size_t pos0) {
  if (f->map) {
    f->in->seek_mem(pos0);
  } else {
    f->in->prepend(f->win.data() + f->win_lo, f->win_len - f->win_lo);
    f->win_lo = f->win_len = 0;
    f->win.clear();
    f->win.shrink_to_fit();
  }
}

}  // namespace

namespace {
// A whole mapped FASTA file (a reference: few sequences, lines of 60-80 characters, gigabytes) read by all threads:
// headers found by a parallel scan, every sequence body cut into pieces at line starts, lengths counted, then the lines
// copied to their places.  kseq's rules (src/kseq.h:186-226) decide what a record is; anything this reader does not
// reproduce exactly — FASTQ, a carriage return, a body line that begins with '@' or '+' (kseq looks at the first
// character of every line) — makes it decline, and the sequential reader takes the file.
bool read_fasta_parallel(fem_seqfile *f, fem_seqset *out) {
  const char *m = f->map;
  const size_t len = f->map_len;
  size_t p0 = 0;
  while (p0 < len && m[p0] != '>' && m[p0] != '@') ++p0;  // kseq skips to the first header character, wherever it is
  if (p0 >= len || m[p0] != '>') return false;
  size_t piece_bytes = (size_t)8 << 20;
  if (const char *e = getenv("FEM_FASTA_PIECE")) piece_bytes = (size_t)std::max(16, atoi(e));  // (tests: many small pieces)
  const int nt = std::max(1, std::min(omp_get_max_threads(), 16));
  // ---- headers: '>' at the start of a line ----
  std::vector<size_t> hdr;
  {
    const size_t span = len - p0, n_chunk = std::max<size_t>(1, std::min<size_t>((size_t)nt * 4, span / (1u << 20) + 1));
    std::vector<std::vector<size_t>> found(n_chunk);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (int64_t c = 0; c < (int64_t)n_chunk; ++c) {
      size_t lo = p0 + span * (size_t)c / n_chunk;
      const size_t hi = p0 + span * ((size_t)c + 1) / n_chunk;
      while (lo < hi) {
        const char *g = (const char *)memchr(m + lo, '>', hi - lo);
        if (!g) break;
        const size_t at = (size_t)(g - m);
        if (at == p0 || m[at - 1] == '\n') found[(size_t)c].push_back(at);
        lo = at + 1;
      }
    }
    for (const auto &v : found) hdr.insert(hdr.end(), v.begin(), v.end());
  }
  const size_t n_rec = hdr.size();
  // ---- names and pieces ----
  struct Piece {
    size_t rec, lo, hi;
    uint64_t out_len = 0;
  };
  std::vector<Piece> pieces;
  std::vector<size_t> name_at(n_rec), name_len(n_rec), first_piece(n_rec + 1, 0);
  for (size_t r = 0; r < n_rec; ++r) {
    const size_t h = hdr[r], end = r + 1 < n_rec ? hdr[r + 1] : len;
    const char *nl = (const char *)memchr(m + h, '\n', end - h);
    const size_t line_end = nl ? (size_t)(nl - m) : end;
    size_t q = h + 1;
    while (q < line_end && !isspace((unsigned char)m[q])) ++q;
    name_at[r] = h + 1, name_len[r] = q - (h + 1);
    first_piece[r] = pieces.size();
    size_t lo = nl ? line_end + 1 : end;
    while (lo < end) {  // pieces begin at line starts
      size_t hi = std::min(end, lo + piece_bytes);
      if (hi < end) {
        const char *e = (const char *)memchr(m + hi, '\n', end - hi);
        hi = e ? (size_t)(e - m) + 1 : end;
      }
      pieces.push_back(Piece{r, lo, hi});
      lo = hi;
    }
  }
  first_piece[n_rec] = pieces.size();
  // ---- pass 1: what each piece contributes; anything irregular -> decline ----
  bool ok = true;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt) reduction(&& : ok)
  for (int64_t i = 0; i < (int64_t)pieces.size(); ++i) {
    Piece &pc = pieces[(size_t)i];
    if (memchr(m + pc.lo, '\r', pc.hi - pc.lo)) {
      ok = false;
      continue;
    }
    uint64_t n = 0;
    for (size_t p = pc.lo; p < pc.hi;) {
      const char *e = (const char *)memchr(m + p, '\n', pc.hi - p);
      const size_t le = e ? (size_t)(e - m) : pc.hi;
      if (le > p) {
        if (m[p] == '@' || m[p] == '+' || m[p] == '>') ok = false;
        n += le - p;
      }
      p = le + 1;
    }
    pc.out_len = n;
  }
  if (!ok) return false;
  // ---- layout (zero-length records are skipped, src/sequence_batch.c:50-52) ----
  std::vector<uint64_t> rec_len(n_rec, 0), piece_at(pieces.size(), 0);
  for (const Piece &pc : pieces) rec_len[pc.rec] += pc.out_len;
  uint64_t n_keep = 0, nb = 0, nn = 0;
  for (size_t r = 0; r < n_rec; ++r)
    if (rec_len[r]) ++n_keep, nb += rec_len[r], nn += name_len[r];
  memset(out, 0, sizeof *out);
  out->n = n_keep;
  out->bases = (char *)malloc(nb + 64);
  out->names = (char *)malloc(nn + 1);
  out->off = (uint64_t *)malloc((n_keep + 1) * sizeof(uint64_t));
  out->name_off = (uint64_t *)malloc((n_keep + 1) * sizeof(uint64_t));
  if (!out->bases || !out->names || !out->off || !out->name_off) {
    fem_seqset_free(out);
    return false;
  }
  uint64_t b = 0, nm = 0, k = 0;
  for (size_t r = 0; r < n_rec; ++r) {
    if (!rec_len[r]) continue;
    out->off[k] = b, out->name_off[k] = nm;
    memcpy(out->names + nm, m + name_at[r], name_len[r]);
    uint64_t at = b;
    for (size_t i = first_piece[r]; i < first_piece[r + 1]; ++i) piece_at[i] = at, at += pieces[i].out_len;
    b += rec_len[r], nm += name_len[r], ++k;
  }
  out->off[n_keep] = nb, out->name_off[n_keep] = nn;
  // ---- pass 2: the lines to their places ----
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
  for (int64_t i = 0; i < (int64_t)pieces.size(); ++i) {
    const Piece &pc = pieces[(size_t)i];
    if (!rec_len[pc.rec]) continue;
    char *w = out->bases + piece_at[(size_t)i];
    for (size_t p = pc.lo; p < pc.hi;) {
      const char *e = (const char *)memchr(m + p, '\n', pc.hi - p);
      const size_t le = e ? (size_t)(e - m) : pc.hi;
      memcpy(w, m + p, le - p);
      w += le - p;
      p = le + 1;
    }
  }
  memset(out->bases + nb, 0, 64);
  return true;
}
}  // namespace

int fem_seqfile_read(fem_seqfile *f, uint64_t max_seqs, fem_seqset *out) {
  if (!f || !out) return -1;
  if ((f->gz || f->zmap) && f->win_len > f->win_lo) {  // after batches read through the window: hand its rest to this reader
    f->fast_ok = false;
    fast_view_abandon(f, 0);
  }
  // From here on this reader's byte stream holds inflated bytes the window reader knows nothing of (it inflates from the
  // file itself): a compressed source stays with the sequential reader once it has been read this way.
  if (f->gz || f->zmap) f->fast_ok = false;
  if (f->map && max_seqs == 0 && f->last_char == 0 && f->in->tell() == 0 && read_fasta_parallel(f, out)) {
    f->in->seek_mem(f->map_len);
    return 0;
  }
  std::vector<ParsedChunk> parts(1);
  int rc = 0;
  while (max_seqs == 0 || parts[0].len.size() < max_seqs) {
    long len = f->next();
    if (len == 0) continue;  // zero-length records are skipped (src/sequence_batch.c:50-52)
    if (len < 0) {
      if (len != -1) rc = (int)len;  // "Didn't reach the end of sequence file, which might be corrupted!"
      break;
    }
    push_record(parts[0], f->name.data(), f->name.size(), f->seq.data(), f->seq.size(),
                f->qual.empty() ? nullptr : f->qual.data());
  }
  int frc = finish_seqset(parts, out);
  return frc ? frc : rc;
}

int fem_seqfile_read_bytes(fem_seqfile *f, uint64_t approx_bytes, int n_threads, fem_seqset *out) {
  if (!f || !out) return -1;
  if (n_threads < 1) n_threads = 1;
  const size_t pos0 = f->in->tell();
  f->threads = n_threads;
  // ---- mapped plain file, or inflated window of a gzip stream, at a record boundary: split it between threads ----
  FastView fv;
  if (fast_view(f, approx_bytes, &fv)) {
    const char *m = fv.m;
    const size_t len = fv.len;
    size_t lo = fv.lo;
    while (lo < len && m[lo] != '@' && m[lo] != '>') ++lo;  // kseq skips to the first header character
    if (lo >= len && fv.whole) {
      std::vector<ParsedChunk> none(1);
      fast_view_consumed(f, len);
      return finish_seqset(none, out);
    }
    size_t hi = approx_bytes == 0 || lo + approx_bytes >= len ? len : next_fastq_record(m, len, lo + (size_t)approx_bytes);
    if (!fv.whole && hi >= len) hi = last_fastq_record(m, lo, len);  // the record that straddles the window's end waits
    if (lo < len && hi > lo && (fv.whole || hi < len) && m[lo] == '@' && next_fastq_record(m, len, lo) == lo) {
      const size_t span = hi - lo;
      int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, span / (1u << 20) + 1));
      std::vector<size_t> cut((size_t)nt + 1, hi);
      cut[0] = lo;
      for (int t = 1; t < nt; ++t) cut[(size_t)t] = std::min(hi, next_fastq_record(m, len, lo + span * (size_t)t / (size_t)nt));
      for (int t = 1; t <= nt; ++t) cut[(size_t)t] = std::max(cut[(size_t)t], cut[(size_t)t - 1]);
      std::vector<ParsedChunk> parts((size_t)nt);
      bool ok = true;
#pragma omp parallel for schedule(static, 1) num_threads(nt) reduction(&& : ok)
      for (int t = 0; t < nt; ++t) ok = parse_fastq_range(m, cut[(size_t)t], cut[(size_t)t + 1], parts[(size_t)t]) && ok;
      if (ok) {
        fast_view_consumed(f, hi);
        return finish_seqset(parts, out);
      }
    }
    f->fast_ok = false;  // FASTA, multi-line FASTQ or malformed input: the exact sequential reader takes over
    fast_view_abandon(f, pos0);
  }
  // ---- sequential, kseq-exact ----
  if (f->gz || f->zmap) f->fast_ok = false;  // (see fem_seqfile_read: the window reader would skip what this one buffers)
  std::vector<ParsedChunk> parts(1);
  int rc = 0;
  while (approx_bytes == 0 || f->in->tell() - pos0 < approx_bytes) {
    long l = f->next();
    if (l == 0) continue;
    if (l < 0) {
      if (l != -1) rc = (int)l;
      break;
    }
    push_record(parts[0], f->name.data(), f->name.size(), f->seq.data(), f->seq.size(),
                f->qual.empty() ? nullptr : f->qual.data());
  }
  int frc = finish_seqset(parts, out);
  return frc ? frc : rc;
}

// ---- the same batch in two phases: scan + size, then copy straight into the caller's buffers ----
namespace {
struct RangeCount {
  uint64_t n = 0, bases = 0, names = 0;
  uint32_t max_len = 0, min_len = 0xFFFFFFFFu;
  bool ok = true;
};
// One strict 4-line FASTQ record at p (blank lines in front skipped).  false: not that shape (or end of range).
struct FqRec {
  const char *name, *seq, *qual;
  size_t name_len, len;
  size_t next;
};
// Positions of the next (up to) four '\n' in [p, hi): one sweep of 32-byte compares instead of four memchr calls on
// lines a few dozen bytes long (the call overhead was most of the parser's time).  Returns how many were found.
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline int newlines4_avx2(const char *m, size_t p, size_t hi, size_t out[4]) {
  int k = 0;
  const __m256i nl = _mm256_set1_epi8('\n');
  while (p + 32 <= hi) {
    uint32_t mask = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(m + p)), nl));
    while (mask) {
      out[k++] = p + (size_t)__builtin_ctz(mask);
      if (k == 4) return 4;
      mask &= mask - 1;
    }
    p += 32;
  }
  for (; p < hi; ++p)
    if (m[p] == '\n') {
      out[k++] = p;
      if (k == 4) return 4;
    }
  return k;
}
#endif
inline int newlines4(const char *m, size_t p, size_t hi, size_t out[4]) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) return newlines4_avx2(m, p, hi, out);
#endif
  int k = 0;
  while (k < 4 && p < hi) {
    const char *q = (const char *)memchr(m + p, '\n', hi - p);
    if (!q) break;
    out[k++] = (size_t)(q - m);
    p = (size_t)(q - m) + 1;
  }
  return k;
}
inline bool next_fq(const char *m, size_t &p, size_t hi, FqRec &r, bool &bad) {
  while (p < hi && m[p] == '\n') ++p;
  if (p >= hi) return false;
  if (m[p] != '@') {
    bad = true;
    return false;
  }
  size_t nl[4];
  const int k = newlines4(m, p, hi, nl);
  if (k < 3) return bad = true, false;
  const char *e0 = m + nl[0], *s1 = e0 + 1, *e1 = m + nl[1], *s2 = e1 + 1, *e2 = m + nl[2], *s3 = e2 + 1;
  if (s2 >= m + hi || *s2 != '+') return bad = true, false;
  const char *e3 = k == 4 ? m + nl[3] : m + hi;  // last line of the file without a newline
  size_t name_len = 0;
  while (m + p + 1 + name_len < e0 && !isspace((unsigned char)m[p + 1 + name_len])) ++name_len;
  size_t sl = (size_t)(e1 - s1), ql = (size_t)(e3 - s3);
  if (sl > 1 && s1[sl - 1] == '\r') --sl;  // kseq strips a trailing CR
  if (ql > 1 && s3[ql - 1] == '\r') --ql;
  if (sl != ql) return bad = true, false;  // multi-line or truncated: let the exact parser decide
  if (sl > 0 && (s1[0] == '>' || s1[0] == '@' || s1[0] == '+')) return bad = true, false;
  r.name = m + p + 1, r.name_len = name_len, r.seq = s1, r.qual = s3, r.len = sl;
  p = (size_t)(e3 - m) + (e3 < m + hi ? 1 : 0);
  return true;
}
}  // namespace

struct fem_batch_plan {
  bool fast = false;
  const char *m = nullptr;          // fast: file offset x is at m[x] (the file's window buffer, or its mapping)
  std::vector<size_t> cut;          // fast: byte ranges of the threads
  std::vector<RangeCount> count;    // fast: what each range holds
  size_t end = 0;                   // fast: where the stream continues
  fem_seqset held{};                // slow: the parsed batch
  int rc = 0;                       // slow: the reader's status (reported by plan)
};

int fem_seqfile_plan(fem_seqfile *f, uint64_t approx_bytes, int n_threads, fem_batch_plan **plan_out, fem_batch_shape *shape) {
  if (!f || !plan_out || !shape) return -1;
  if (n_threads < 1) n_threads = 1;
  *plan_out = nullptr;
  memset(shape, 0, sizeof *shape);
  fem_batch_plan *pl = new (std::nothrow) fem_batch_plan();
  if (!pl) return -4;
  const size_t pos0 = f->in->tell();
  f->threads = n_threads;
  FastView fv;
  if (fast_view(f, approx_bytes, &fv)) {
    // (Reading a plain file's window out of the page cache with pread by all threads, into a reusable buffer, was tried in
    // place of faulting the mapping in: 3 GB/s on tmpfs against 18 GB/s through the mapping.)
    const char *m = fv.m;
    const size_t len = fv.len;
    size_t lo = fv.lo;
    while (lo < len && m[lo] != '@' && m[lo] != '>') ++lo;  // kseq skips to the first header character
    if (lo >= len && fv.whole) {  // end of input: an empty batch
      fast_view_consumed(f, len);
      pl->fast = true, pl->m = m, pl->end = len, pl->cut.assign(2, len), pl->count.assign(1, RangeCount());
      shape->has_qual = 1;
      *plan_out = pl;
      return 0;
    }
    size_t hi = approx_bytes == 0 || lo + approx_bytes >= len ? len : next_fastq_record(m, len, lo + (size_t)approx_bytes);
    if (!fv.whole && hi >= len) hi = last_fastq_record(m, lo, len);  // the record that straddles the window's end waits
    if (lo < len && hi > lo && (fv.whole || hi < len) && m[lo] == '@' && next_fastq_record(m, len, lo) == lo) {
      const size_t span = hi - lo;
      const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, span / (1u << 18) + 1));
      pl->cut.assign((size_t)nt + 1, hi);
      pl->cut[0] = lo;
      for (int t = 1; t < nt; ++t) pl->cut[(size_t)t] = std::min(hi, next_fastq_record(m, len, lo + span * (size_t)t / (size_t)nt));
      for (int t = 1; t <= nt; ++t) pl->cut[(size_t)t] = std::max(pl->cut[(size_t)t], pl->cut[(size_t)t - 1]);
      pl->count.assign((size_t)nt, RangeCount());
      planner_pool().run(nt, [&](int t) {
        RangeCount c;
        size_t p = pl->cut[(size_t)t];
        const size_t h = pl->cut[(size_t)t + 1];
        FqRec r;
        bool bad = false;
        while (next_fq(m, p, h, r, bad)) {
          if (r.len == 0) continue;  // zero-length records are skipped (src/sequence_batch.c:50-52)
          ++c.n, c.bases += r.len, c.names += r.name_len;
          c.max_len = std::max<uint32_t>(c.max_len, (uint32_t)std::min<size_t>(r.len, 0xFFFFFFFFu));
          c.min_len = std::min<uint32_t>(c.min_len, (uint32_t)std::min<size_t>(r.len, 0xFFFFFFFFu));
        }
        c.ok = !bad;
        pl->count[(size_t)t] = c;
      });
      bool ok = true;
      uint32_t min_len = 0xFFFFFFFFu;
      for (const RangeCount &c : pl->count) ok = ok && c.ok;
      if (ok) {
        for (const RangeCount &c : pl->count) {
          shape->n_reads += c.n, shape->n_bases += c.bases, shape->n_name_bytes += c.names;
          shape->max_len = std::max(shape->max_len, c.max_len);
          min_len = std::min(min_len, c.min_len);
        }
        shape->min_len = shape->n_reads ? min_len : 0;
        shape->has_qual = 1;
        pl->fast = true, pl->m = m, pl->end = hi;
        fast_view_consumed(f, hi);
        *plan_out = pl;
        return 0;
      }
    }
    f->fast_ok = false;  // FASTA, multi-line FASTQ or malformed input: the exact sequential reader takes over
    fast_view_abandon(f, pos0);
    pl->cut.clear(), pl->count.clear();
  }
  pl->rc = fem_seqfile_read_bytes(f, approx_bytes, n_threads, &pl->held);
  if (pl->rc != 0 && pl->rc != -2 && pl->rc != -3) {  // allocation failure
    int rc = pl->rc;
    fem_batch_plan_free(pl);
    return rc;
  }
  shape->n_reads = pl->held.n;
  shape->n_bases = pl->held.n ? pl->held.off[pl->held.n] : 0;
  shape->n_name_bytes = pl->held.n ? pl->held.name_off[pl->held.n] : 0;
  shape->min_len = pl->held.n ? 0xFFFFFFFFu : 0u;
  for (uint64_t i = 0; i < pl->held.n; ++i) {
    const uint32_t l = (uint32_t)std::min<uint64_t>(pl->held.off[i + 1] - pl->held.off[i], 0xFFFFFFFFu);
    shape->max_len = std::max(shape->max_len, l), shape->min_len = std::min(shape->min_len, l);
  }
  shape->has_qual = pl->held.quals != nullptr || pl->held.n == 0;
  *plan_out = pl;
  return pl->rc;  // -2 / -3: "Didn't reach the end of sequence file"; the records read so far are still in the plan
}

int fem_seqfile_fill(fem_seqfile *f, fem_batch_plan *pl, int n_threads, char *bases, uint64_t *off, char *quals, char *names,
                     uint64_t *name_off) {
  if (!f || !pl || !bases || !off || !names || !name_off) return -1;
  if (n_threads < 1) n_threads = 1;
  if (!pl->fast) {
    const fem_seqset &h = pl->held;
    const uint64_t nb = h.n ? h.off[h.n] : 0, nn = h.n ? h.name_off[h.n] : 0;
    if (nb) memcpy(bases, h.bases, nb);
    if (quals && h.quals && nb) memcpy(quals, h.quals, nb);
    if (nn) memcpy(names, h.names, nn);
    for (uint64_t i = 0; i <= h.n; ++i) off[i] = h.n ? h.off[i] : 0, name_off[i] = h.n ? h.name_off[i] : 0;
    memset(bases + nb, 0, 64);
    fem_batch_plan_free(pl);
    return 0;
  }
  const char *m = pl->m;
  const int nt = (int)pl->count.size();
  std::vector<uint64_t> r0((size_t)nt + 1, 0), b0((size_t)nt + 1, 0), n0((size_t)nt + 1, 0);
  for (int t = 0; t < nt; ++t) {
    r0[(size_t)t + 1] = r0[(size_t)t] + pl->count[(size_t)t].n;
    b0[(size_t)t + 1] = b0[(size_t)t] + pl->count[(size_t)t].bases;
    n0[(size_t)t + 1] = n0[(size_t)t] + pl->count[(size_t)t].names;
  }
  parser_pool().run(nt, [&](int t) {
    size_t p = pl->cut[(size_t)t];
    const size_t h = pl->cut[(size_t)t + 1];
    uint64_t r = r0[(size_t)t], b = b0[(size_t)t], nm = n0[(size_t)t];
    FqRec rec;
    bool bad = false;
    while (next_fq(m, p, h, rec, bad)) {
      if (rec.len == 0) continue;
      off[r] = b, name_off[r] = nm;
      memcpy(bases + b, rec.seq, rec.len);
      if (quals) memcpy(quals + b, rec.qual, rec.len);
      memcpy(names + nm, rec.name, rec.name_len);
      ++r, b += rec.len, nm += rec.name_len;
    }
  });
  off[r0[(size_t)nt]] = b0[(size_t)nt];
  name_off[r0[(size_t)nt]] = n0[(size_t)nt];
  memset(bases + b0[(size_t)nt], 0, 64);
  fem_batch_plan_free(pl);
  return 0;
}

// The same with the bases written at two bits per base (fem_pack.h; include/fem_hip.h fem_dev_commit_stage_packed) into
// the staging buffer: no ASCII copy of the batch exists anywhere on the host.  Needs reads of one length (the plan's shape:
// min_len == max_len == read_len).  Returns 1 — and leaves the plan alive for fem_seqfile_fill — when the batch holds more
// non-ACGT characters than exc_cap.
static int fill_packed_impl(fem_seqfile *f, fem_batch_plan *pl, int n_threads, uint32_t read_len, uint8_t *codes, uint64_t exc_cap,
                            uint64_t *n_exc_out, char *quals, char *names, uint64_t *name_off, fem_read_refs *refs);

int fem_seqfile_fill_packed(fem_seqfile *f, fem_batch_plan *pl, int n_threads, uint32_t read_len, uint8_t *codes, uint64_t exc_cap,
                            uint64_t *n_exc_out, char *quals, char *names, uint64_t *name_off) {
  if (!names || !name_off) return -1;
  return fill_packed_impl(f, pl, n_threads, read_len, codes, exc_cap, n_exc_out, quals, names, name_off, nullptr);
}

// The same without a copy of anything but the packed bases: where read r's name, bases and qualities lie in the input goes
// into refs (arrays of the caller, n_reads entries each; refs->n and refs->read_len are set here).  Only for batches whose
// records sit in the file's own mapping (a plain, uncompressed 4-line FASTQ file: the mapping lives as long as the handle):
// 2 = not such a batch, the plan is still there (fem_seqfile_fill_packed or fem_seqfile_fill copy the fields instead).
int fem_seqfile_fill_packed_refs(fem_seqfile *f, fem_batch_plan *pl, int n_threads, uint32_t read_len, uint8_t *codes, uint64_t exc_cap,
                                 uint64_t *n_exc_out, fem_read_refs *refs) {
  if (!f || !pl || !refs || !refs->name || !refs->name_len || !refs->seq || !refs->qual) return -1;
  if (!pl->fast || !f->map || pl->m != f->map) return 2;
  if (!f->read_again) {
    // the formatter comes back to these records long after the parser's pass: under MADV_SEQUENTIAL (fem_seqfile_open) the
    // pages behind the parser are the first to go, and a file larger than the page cache would be read from disk twice
    (void)madvise((void *)f->map, f->map_len, MADV_NORMAL);
    f->read_again = true;
  }
  return fill_packed_impl(f, pl, n_threads, read_len, codes, exc_cap, n_exc_out, nullptr, nullptr, nullptr, refs);
}

static int fill_packed_impl(fem_seqfile *f, fem_batch_plan *pl, int n_threads, uint32_t read_len, uint8_t *codes, uint64_t exc_cap,
                            uint64_t *n_exc_out, char *quals, char *names, uint64_t *name_off, fem_read_refs *refs) {
  if (!f || !pl || !codes || !n_exc_out || read_len == 0) return -1;
  if (n_threads < 1) n_threads = 1;
  const uint32_t bpr = fempack::bytes_per_read(read_len);
  *n_exc_out = 0;
  std::vector<std::vector<uint64_t>> exc;
  uint64_t n_total = 0;
  if (!pl->fast) {
    const fem_seqset &h = pl->held;
    for (uint64_t i = 0; i < h.n; ++i)
      if (h.off[i + 1] - h.off[i] != read_len) return -1;
    exc.resize(1);
    for (uint64_t i = 0; i < h.n; ++i)
      fempack::pack_bases((const uint8_t *)h.bases + h.off[i], read_len, codes + i * bpr, i * (uint64_t)read_len, exc[0]);
    const uint64_t nb = h.n ? h.off[h.n] : 0, nn = h.n ? h.name_off[h.n] : 0;
    if (quals && h.quals && nb) memcpy(quals, h.quals, nb);
    if (nn) memcpy(names, h.names, nn);
    for (uint64_t i = 0; i <= h.n; ++i) name_off[i] = h.n ? h.name_off[i] : 0;
    n_total = h.n;
  } else {
    const char *m = pl->m;
    const int nt = (int)pl->count.size();
    std::vector<uint64_t> r0((size_t)nt + 1, 0), n0((size_t)nt + 1, 0);
    for (int t = 0; t < nt; ++t) {
      if (pl->count[(size_t)t].n && (pl->count[(size_t)t].min_len != read_len || pl->count[(size_t)t].max_len != read_len)) return -1;
      r0[(size_t)t + 1] = r0[(size_t)t] + pl->count[(size_t)t].n;
      n0[(size_t)t + 1] = n0[(size_t)t] + pl->count[(size_t)t].names;
    }
    exc.resize((size_t)nt);
    parser_pool().run(nt, [&](int t) {
      size_t p = pl->cut[(size_t)t];
      const size_t h = pl->cut[(size_t)t + 1];
      uint64_t r = r0[(size_t)t], nm = n0[(size_t)t];
      FqRec rec;
      bool bad = false;
      while (next_fq(m, p, h, rec, bad)) {
        if (rec.len == 0) continue;
        fempack::pack_bases((const uint8_t *)rec.seq, read_len, codes + r * bpr, r * (uint64_t)read_len, exc[(size_t)t]);
        if (refs) {  // nothing else is copied: the formatter takes the fields from the mapping
          refs->name[r] = rec.name, refs->name_len[r] = (uint32_t)rec.name_len;
          refs->seq[r] = rec.seq, refs->qual[r] = rec.qual;
        } else {
          name_off[r] = nm;
          if (quals) memcpy(quals + r * (uint64_t)read_len, rec.qual, rec.len);
          memcpy(names + nm, rec.name, rec.name_len);
        }
        ++r, nm += rec.name_len;
      }
    });
    n_total = r0[(size_t)nt];
    if (!refs) name_off[n_total] = n0[(size_t)nt];
  }
  uint64_t n_exc = 0;
  for (const auto &v : exc) n_exc += v.size();
  if (n_exc > exc_cap) return 1;  // (the plan stays: fem_seqfile_fill writes the characters)
  // the bytes between the codes and the 8-byte boundary the exception positions start at; positions, then the bytes
  const uint64_t exc_off = fempack::code_bytes(n_total, read_len);
  for (uint64_t i = n_total * bpr; i < exc_off; ++i) codes[i] = 0;
  uint32_t *exc_pos = (uint32_t *)(codes + exc_off);
  uint8_t *exc_chr = (uint8_t *)(exc_pos + n_exc);
  for (const auto &v : exc)
    for (uint64_t x : v) *exc_pos++ = (uint32_t)(x >> 8), *exc_chr++ = (uint8_t)x;
  *n_exc_out = n_exc;
  if (refs) refs->n = n_total, refs->read_len = read_len;
  fem_batch_plan_free(pl);
  return 0;
}

int fem_seqfile_plan_ahead_ok(fem_seqfile *f) { return f && f->map && f->fast_ok ? 1 : 0; }

void fem_batch_plan_free(fem_batch_plan *pl) {
  if (!pl) return;
  fem_seqset_free(&pl->held);
  delete pl;
}

void fem_seqset_free(fem_seqset *s) {
  if (!s) return;
  free(s->bases), free(s->off), free(s->quals), free(s->names), free(s->name_off);
  memset(s, 0, sizeof *s);
}

// ------------------------------------------------------------------------------------------------
// index files (reference src/index.c:100-168)
// ------------------------------------------------------------------------------------------------
int fem_index_save(const char *path, int32_t k, int32_t step, const uint32_t *lookup, uint64_t n_occ,
                   const uint64_t *occ) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  const size_t n_lookup = ((size_t)1 << (2 * k)) + 1;
  const size_t n = (size_t)n_occ;
  bool ok = fwrite(&k, sizeof(int32_t), 1, f) == 1 && fwrite(&step, sizeof(int32_t), 1, f) == 1 &&
            fwrite(lookup, sizeof(uint32_t), n_lookup, f) == n_lookup && fwrite(&n, sizeof(size_t), 1, f) == 1 &&
            (n == 0 || fwrite(occ, sizeof(uint64_t), n, f) == n);
  ok = (fclose(f) == 0) && ok;
  return ok ? 0 : -2;
}

int fem_index_load(const char *path, int32_t *k, int32_t *step, uint32_t **lookup, uint64_t *n_occ, uint64_t **occ) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  *lookup = nullptr, *occ = nullptr;
  int rc = 0;
  size_t n = 0, n_lookup = 0;
  if (fread(k, sizeof(int32_t), 1, f) != 1 || fread(step, sizeof(int32_t), 1, f) != 1) rc = -2;
  if (rc == 0 && (*k < 1 || *k > 16)) rc = -3;
  if (rc == 0) {
    n_lookup = ((size_t)1 << (2 * *k)) + 1;
    *lookup = (uint32_t *)malloc(n_lookup * sizeof(uint32_t));
    if (!*lookup)
      rc = -4;
    else if (fread(*lookup, sizeof(uint32_t), n_lookup, f) != n_lookup || fread(&n, sizeof(size_t), 1, f) != 1)
      rc = -2;
  }
  if (rc == 0) {
    *occ = (uint64_t *)malloc(std::max<size_t>(n, 1) * sizeof(uint64_t));
    if (!*occ) {
      rc = -4;
    } else if (n) {
      // the occurrence table is the bulk (8 GB for a 3 Gbp reference): copied out of a mapping of the file by all threads
      // (one fread stream: 1.6-1.9 s for 8 GB; a regular file that cannot be mapped still takes that way)
      const long at = ftell(f);
      struct stat st;
      void *mp = MAP_FAILED;
      const size_t want = (size_t)at + n * sizeof(uint64_t);
      if (at > 0 && fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && (size_t)st.st_size >= want)
        mp = mmap(nullptr, want, PROT_READ, MAP_PRIVATE, fileno(f), 0);
      if (mp != MAP_FAILED) {
        const char *src = (const char *)mp + at;
        const size_t bytes = n * sizeof(uint64_t), piece = (size_t)16 << 20, n_piece = (bytes + piece - 1) / piece;
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(1, std::min(omp_get_max_threads(), 16)))
        for (int64_t i = 0; i < (int64_t)n_piece; ++i)
          memcpy((char *)*occ + (size_t)i * piece, src + (size_t)i * piece, std::min(piece, bytes - (size_t)i * piece));
        munmap(mp, want);
      } else if (fread(*occ, sizeof(uint64_t), n, f) != n) {
        rc = -2;
      }
    }
  }
  if (rc == 0 && n && fseek(f, 0, SEEK_END) == 0) {  // (a file shorter than its header says: the mapping would have faulted)
    const long end = ftell(f);
    if (end >= 0 && (size_t)end < 8 + n_lookup * sizeof(uint32_t) + sizeof(size_t) + n * sizeof(uint64_t)) rc = -2;
  }
  fclose(f);
  if (rc != 0) {
    free(*lookup), free(*occ);
    *lookup = nullptr, *occ = nullptr;
    return rc;
  }
  *n_occ = n;
  return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// mapping tail (reference src/align.c:53-92, 279-544)
// ------------------------------------------------------------------------------------------------
namespace {

struct Hit {  // Mapping (src/utils.h:44-49) + its sort key
  uint64_t key;
  uint64_t cand;
  int16_t end;
  uint8_t ed, dir;
};

// radix_sort_mapping: klib's KRADIX_SORT_INIT(mapping, Mapping, MappingSortKey, 8) (src/ksort.h:101-151).
// <=64 records: insertion sort (stable).  More: in-place MSD radix sort, 8 bits per level starting at bit 56,
// each level a cycle-leader permutation (NOT stable), buckets of <=64 finished by insertion sort.  Equal keys
// are possible, so the exact permutation is part of the observable behaviour (it picks the primary record).
void insertion_by_key(Hit *beg, Hit *end) {
  for (Hit *i = beg + 1; i < end; ++i) {
    if (i->key < (i - 1)->key) {
      Hit tmp = *i, *j = i;
      for (; j > beg && tmp.key < (j - 1)->key; --j) *j = *(j - 1);
      *j = tmp;
    }
  }
}

void msd_radix_level(Hit *beg, Hit *end, int shift) {
  struct Span {
    Hit *b, *e;
  } bin[256];
  for (auto &s : bin) s.b = s.e = beg;
  for (Hit *i = beg; i != end; ++i) ++bin[(i->key >> shift) & 255].e;
  for (int k = 1; k < 256; ++k) {
    bin[k].e += bin[k - 1].e - beg;
    bin[k].b = bin[k - 1].e;
  }
  for (int k = 0; k < 256;) {
    Span &cur = bin[k];
    if (cur.b == cur.e) {
      ++k;
      continue;
    }
    int dst = (int)((cur.b->key >> shift) & 255);
    if (dst == k) {
      ++cur.b;
      continue;
    }
    Hit carry = *cur.b;
    do {  // follow the cycle until an element that belongs to bucket k comes back
      Hit displaced = *bin[dst].b;
      *bin[dst].b++ = carry;
      carry = displaced;
      dst = (int)((carry.key >> shift) & 255);
    } while (dst != k);
    *cur.b++ = carry;
  }
  bin[0].b = beg;
  for (int k = 1; k < 256; ++k) bin[k].b = bin[k - 1].e;
  if (shift) {
    int next = shift > 8 ? shift - 8 : 0;
    for (int k = 0; k < 256; ++k) {
      ptrdiff_t n = bin[k].e - bin[k].b;
      if (n > 64)
        msd_radix_level(bin[k].b, bin[k].e, next);
      else if (n > 1)
        insertion_by_key(bin[k].b, bin[k].e);
    }
  }
}

void sort_hits(std::vector<Hit> &h) {
  if (h.size() <= 64)
    insertion_by_key(h.data(), h.data() + h.size());
  else
    msd_radix_level(h.data(), h.data() + h.size(), 56);
}

struct OpRun {
  char op;
  int n;
};

void append_uint(std::string &s, unsigned v) {
  char tmp[12];
  int n = 0;
  do {
    tmp[n++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) s.push_back(tmp[--n]);
}

// generate_MD_tag (src/align.c:501-544)
void build_md(const char *ref_at_start, const char *text, const std::vector<uint32_t> &cigar, std::string &md) {
  md.clear();
  unsigned run = 0;
  size_t rp = 0, tp = 0;
  auto flush = [&] {
    if (run) {
      append_uint(md, run);
      run = 0;
    }
  };
  for (uint32_t c : cigar) {
    const uint32_t n = c >> 4;
    switch (c & 0xf) {
      case 0:  // M
        for (uint32_t i = 0; i < n; ++i, ++rp, ++tp) {
          if (ref_at_start[rp] == text[tp]) {
            ++run;
          } else {
            flush();
            md.push_back(ref_at_start[rp]);
          }
        }
        break;
      case 1:  // I
        tp += n;
        break;
      case 2:  // D
        flush();
        md.push_back('^');
        md.append(ref_at_start + rp, n);
        rp += n;
        break;
    }
  }
  flush();
}

struct Tracer {  // scratch reused across the mappings of one thread
  std::vector<uint32_t> d0, hp;
  std::vector<OpRun> runs;
};

// generate_alignment (src/align.c:279-499).  Returns the start offset inside pattern (>= 0), or -1 where the
// reference would have tripped one of its asserts.
int trace_alignment(int e, const char *pattern, const char *text, int len, int ed, int end, Tracer &tr,
                    std::vector<uint32_t> &cigar, std::string &md) {
  cigar.clear();
  int start = end - len + 1;
  if (start < 0) return -1;
  bool identical = true;
  for (int i = 0; i < len && identical; ++i) identical = text[i] == pattern[start + i];
  if (identical) {  // src/align.c:294-300
    cigar.push_back((uint32_t)len << 4);
    build_md(pattern + start, text, cigar, md);
    return start;
  }
  // re-run the recurrence keeping D0 and HP of every column (src/align.c:303-338)
  tr.d0.resize((size_t)len);
  tr.hp.resize((size_t)len);
  uint32_t peq[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * e; ++i) peq[code_of(pattern[i])] |= 1u << i;
  const uint32_t top = 1u << (2 * e);
  uint32_t vp = 0, vn = 0;
  for (int i = 0; i < len; ++i) {
    peq[code_of(pattern[i + 2 * e])] |= top;
    uint32_t x = peq[code_of(text[i])] | vn;
    uint32_t d0 = ((vp + (x & vp)) ^ vp) | x;
    uint32_t hn = vp & d0;
    uint32_t hp = vn | ~(vp | d0);
    x = d0 >> 1;
    vn = x & hp;
    vp = hn | ~(x | hp);
    tr.d0[(size_t)i] = d0;
    tr.hp[(size_t)i] = hp;
    for (uint32_t &q : peq) q >>= 1;
  }
  // walk back from (last read base, end) until `ed` errors are accounted for (src/align.c:340-440)
  enum Move { MATCH, MISMATCH, INSERT, DELETE };
  int bit = end - len + 1, t = len - 1, pe = end, n_err = 0;
  auto classify = [&]() -> Move {
    bool d = (tr.d0[(size_t)t] >> bit) & 1u;
    if (d && pattern[pe] == text[t]) return MATCH;
    if (!d) return MISMATCH;
    if ((tr.hp[(size_t)t] >> bit) & 1u) return INSERT;
    return DELETE;
  };
  tr.runs.clear();
  OpRun cur{'S', 1};
  switch (classify()) {  // the first step replaces the initial pseudo-run (src/align.c:345-368)
    case MATCH:
      --t, --pe;
      cur = {'M', 1};
      break;
    case MISMATCH:
      --t, --pe, ++n_err;
      cur = {'S', 1};
      break;
    case INSERT:
      --t, ++bit, ++n_err, ++start;
      cur = {'S', 1};
      break;
    case DELETE:
      return -1;  // assert(1 == 0)
  }
  auto extend = [&](char op) {
    if (cur.op == op) {
      ++cur.n;
    } else {
      tr.runs.push_back(cur);
      cur = {op, 1};
    }
  };
  while (t >= 0 && n_err != ed) {
    if (bit < 0 || bit > 31 || pe < 0) return -1;
    switch (classify()) {
      case MATCH:
        --t, --pe;
        extend('M');
        break;
      case MISMATCH:
        --t, --pe, ++n_err;
        if (cur.op == 'S')
          ++cur.n;  // read-end errors pile up in the pseudo-run (src/align.c:398-399)
        else
          extend('M');
        break;
      case INSERT:
        --t, ++bit, ++n_err, ++start;
        if (cur.op == 'S')
          ++cur.n;
        else
          extend('I');
        break;
      case DELETE:
        --bit, --pe, ++n_err, --start;
        extend('D');
        break;
    }
  }
  if (t >= 0) {  // everything left of the last error matches (src/align.c:445-455)
    if (cur.op == 'M') {
      cur.n += t + 1;
    } else {
      tr.runs.push_back(cur);
      cur = {'M', t + 1};
    }
  }
  tr.runs.push_back(cur);
  size_t first = 0;
  if (tr.runs[0].op == 'S') {  // the pseudo-run is added to the run that follows it (src/align.c:466-469)
    if (tr.runs.size() < 2) return -1;
    tr.runs[1].n += tr.runs[0].n;
    first = 1;
  }
  for (size_t i = tr.runs.size(); i-- > first;) {
    uint32_t op;
    switch (tr.runs[i].op) {
      case 'M': op = 0; break;
      case 'I': op = 1; break;
      case 'D': op = 2; break;
      default: return -1;
    }
    cigar.push_back(((uint32_t)tr.runs[i].n << 4) | op);
  }
  build_md(pattern + start, text, cigar, md);
  return start;
}

struct ReadView {
  const char *bases;
  uint32_t len;
};

// Mapping list of one read in verify_candidates' order: + strand then - strand, candidates ascending,
// rejected ones skipped (src/map.c:31-49, src/align.c:21-49).
void collect_hits(const fem_tail_input &in, uint64_t read, std::vector<Hit> &hits) {
  hits.clear();
  for (uint32_t dir = 0; dir < 2; ++dir) {
    const uint64_t slot = 2 * read + dir;
    const uint32_t b = in.cand_begin[slot], n = in.cand_count[slot];
    for (uint32_t i = b; i < b + n; ++i) {
      if (in.ed[i] == 0xFF) continue;
      Hit h;
      h.cand = in.cand[i], h.end = in.end[i], h.ed = in.ed[i], h.dir = (uint8_t)dir;
      h.key = ((uint64_t)h.ed << 60) | ((uint64_t)h.dir << 59) | (h.cand + (uint64_t)(int64_t)h.end);  // src/align.c:53
      hits.push_back(h);
    }
  }
}

void reverse_complement(const char *fwd, uint32_t len, std::string &out) {  // src/sequence_batch.h:90-98
  out.resize(len);
  for (uint32_t i = 0; i < len; ++i) out[i] = kCodeChar[3 ^ code_of(fwd[len - 1 - i])];
}

struct Record {
  uint16_t flag;
  uint32_t tid, pos0;
  uint8_t nm;
};

// htslib's seq_nt16_table followed by seq_nt16_str (third-party, un-vendored in the reference: restated from
// htslib's published tables): what SEQ looks like after a round trip through the BAM nibble encoding.
struct SeqRoundTrip {
  char t[256];
  SeqRoundTrip() {
    static const char iupac[] = "=ACMGRSVTWYHKDBN";
    for (int i = 0; i < 256; ++i) t[i] = 'N';
    for (int i = 0; i < 16; ++i) {
      t[(unsigned char)iupac[i]] = iupac[i];
      t[(unsigned char)tolower(iupac[i])] = iupac[i];
    }
    t['0'] = 'A', t['1'] = 'C', t['2'] = 'G', t['3'] = 'T';
  }
};
const SeqRoundTrip kSeqText;

// Growable text buffer of one formatting thread: records are written through a raw pointer after one capacity check.
struct TextBuf {
  char *p = nullptr;
  size_t n = 0, cap = 0;
  bool failed = false;
  bool fixed = false;  // a stretch of somebody else's buffer, sized from an upper bound: never reallocated
  bool room(size_t extra) {
    if (n + extra <= cap) return true;
    if (fixed) {
      failed = true;
      return false;
    }
    const size_t want = std::max(cap + cap / 2, n + extra + (size_t)(1u << 20));
    char *q = (char *)realloc(p, want);
    if (!q) {
      failed = true;
      return false;
    }
    p = q, cap = want;
    return true;
  }
};

inline char *put_uint(char *w, unsigned v) {
  char tmp[12];
  int k = 0;
  do {
    tmp[k++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (k) *w++ = tmp[--k];
  return w;
}
inline char *put_str(char *w, const char *s, size_t n) {
  memcpy(w, s, n);
  return w + n;
}

// One SAM line: QNAME FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL NM MD (src/align.c:546-632)
void append_sam_line(TextBuf &o, const fem_tail_ref &ref, const char *name, size_t name_len, const Record &rec,
                     const uint32_t *cigar, size_t n_cigar, const char *md, size_t n_md, bool primary, const char *fwd,
                     uint32_t len, const char *qual) {
  const size_t rname_len = (size_t)(ref.name_off[rec.tid + 1] - ref.name_off[rec.tid]);
  if (!o.room(name_len + rname_len + 2 * (size_t)len + n_md + 11 * n_cigar + 96)) return;
  char *w = o.p + o.n;
  w = put_str(w, name, name_len);
  *w++ = '\t';
  w = put_uint(w, rec.flag);
  *w++ = '\t';
  w = put_str(w, ref.names + ref.name_off[rec.tid], rname_len);
  *w++ = '\t';
  w = put_uint(w, rec.pos0 + 1u);
  w = put_str(w, "\t255\t", 5);
  if (n_cigar == 0) *w++ = '*';
  for (size_t i = 0; i < n_cigar; ++i) {
    w = put_uint(w, cigar[i] >> 4);
    *w++ = "MIDNSHP=XB"[cigar[i] & 0xf];
  }
  w = put_str(w, "\t*\t0\t0\t", 7);
  if (primary && len > 0) {  // only the primary record carries SEQ/QUAL (src/align.c:83-88)
    for (uint32_t i = 0; i < len; ++i) w[i] = kSeqText.t[(unsigned char)fwd[i]];  // original read (src/align.c:79)
    w += len;
    *w++ = '\t';
    if (qual)
      w = put_str(w, qual, len);
    else
      *w++ = '*';
  } else {
    w = put_str(w, "*\t*", 3);
  }
  w = put_str(w, "\tNM:i:", 6);
  w = put_uint(w, rec.nm);
  w = put_str(w, "\tMD:Z:", 6);
  w = put_str(w, md, n_md);
  *w++ = '\n';
  o.n = (size_t)(w - o.p);
}

// Concatenate the threads' pieces into one malloc'd buffer (each piece copied by its own thread); frees the pieces.
int join_parts(std::vector<TextBuf> &parts, int n_threads, char **text, uint64_t *text_len) {
  std::vector<size_t> at(parts.size() + 1, 0);
  bool failed = false;
  for (size_t i = 0; i < parts.size(); ++i) at[i + 1] = at[i] + parts[i].n, failed = failed || parts[i].failed;
  char *buf = failed ? nullptr : (char *)malloc(at.back() + 1);
  if (buf) {
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
    for (size_t i = 0; i < parts.size(); ++i)
      if (parts[i].n) memcpy(buf + at[i], parts[i].p, parts[i].n);
  }
  for (TextBuf &t : parts) free(t.p);
  if (!buf) return -4;
  *text = buf;
  *text_len = at.back();
  return 0;
}

template <typename Emit>
void process_read(int e, const fem_tail_ref &ref, const char *fwd, uint32_t len, std::vector<Hit> &hits, Tracer &tr,
                  std::string &rev, std::vector<uint32_t> &cigar, std::string &md, Emit &&emit) {
  sort_hits(hits);
  bool have_rev = false;
  for (size_t mi = 0; mi < hits.size(); ++mi) {
    const Hit &h = hits[mi];
    const char *text = fwd;
    if (h.dir) {
      if (!have_rev) {
        reverse_complement(fwd, len, rev);
        have_rev = true;
      }
      text = rev.data();
    }
    const uint32_t tid = (uint32_t)(h.cand >> 32);
    const char *pattern = ref.text + ref.off[tid] + (uint32_t)h.cand;
    int start = trace_alignment(e, pattern, text, (int)len, h.ed, h.end, tr, cigar, md);
    Record r;
    r.flag = (uint16_t)((h.dir ? 16 : 0) | (mi > 0 ? 256 : 0));  // BAM_FREVERSE, BAM_FSECONDARY (src/align.c:82-84)
    if (start < 0) {
      r.flag |= 0x8000;  // the reference would have asserted; never happens on valid data
      start = 0;
      cigar.clear();
      md.clear();
    }
    r.tid = tid;
    r.pos0 = (uint32_t)start + (uint32_t)h.cand;  // src/align.c:80
    r.nm = h.ed;
    emit(mi, r, cigar, md);
  }
}

}  // namespace

extern "C" {

int fem_tail_records(int32_t e, const fem_tail_ref *ref, const char *read_bases, const uint64_t *read_off,
                     const fem_tail_input *in, int n_threads, fem_records *out) {
  if (!ref || !in || !out || (in->n_reads && (!read_bases || !read_off))) return -1;
  if (n_threads < 1) n_threads = 1;
  const uint64_t n = in->n_reads;
  struct Part {
    std::vector<uint64_t> per_read;  // records of each read in this part
    std::vector<Record> rec;
    std::vector<uint64_t> cig_end, md_end;
    std::vector<uint32_t> cig;
    std::string md;
  };
  std::vector<Part> parts((size_t)n_threads);
#pragma omp parallel num_threads(n_threads)
  {
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
    Part &p = parts[(size_t)t];
    const uint64_t lo = n * (uint64_t)t / (uint64_t)nt, hi = n * (uint64_t)(t + 1) / (uint64_t)nt;
    std::vector<Hit> hits;
    Tracer tr;
    std::string rev, md;
    std::vector<uint32_t> cigar;
    for (uint64_t r = lo; r < hi; ++r) {
      collect_hits(*in, r, hits);
      p.per_read.push_back(hits.size());
      if (hits.empty()) continue;
      process_read(e, *ref, read_bases + read_off[r], (uint32_t)(read_off[r + 1] - read_off[r]), hits, tr, rev, cigar,
                   md, [&](size_t, const Record &rec, const std::vector<uint32_t> &cg, const std::string &m) {
                     p.rec.push_back(rec);
                     p.cig.insert(p.cig.end(), cg.begin(), cg.end());
                     p.cig_end.push_back(p.cig.size());
                     p.md += m;
                     p.md_end.push_back(p.md.size());
                   });
    }
  }
  std::vector<uint64_t> rec_off{0}, cig_off{0}, md_off{0};
  std::vector<uint16_t> flag;
  std::vector<uint32_t> tid, pos0, cig;
  std::vector<uint8_t> nm;
  std::string md;
  for (const Part &p : parts) {
    for (uint64_t c : p.per_read) rec_off.push_back(rec_off.back() + c);
    for (const Record &r : p.rec) flag.push_back(r.flag), tid.push_back(r.tid), pos0.push_back(r.pos0), nm.push_back(r.nm);
    for (uint64_t c : p.cig_end) cig_off.push_back(cig.size() + c);
    for (uint64_t c : p.md_end) md_off.push_back(md.size() + c);
    cig.insert(cig.end(), p.cig.begin(), p.cig.end());
    md += p.md;
  }
  while (rec_off.size() < n + 1) rec_off.push_back(rec_off.back());
  memset(out, 0, sizeof *out);
  out->n_records = flag.size();
  out->rec_off = dup_vec(rec_off);
  out->flag = dup_vec(flag), out->tid = dup_vec(tid), out->pos0 = dup_vec(pos0), out->nm = dup_vec(nm);
  out->cigar_off = dup_vec(cig_off), out->cigar = dup_vec(cig), out->md_off = dup_vec(md_off);
  out->md = (char *)malloc(md.size() + 1);
  if (out->md) memcpy(out->md, md.data(), md.size());
  return 0;
}

void fem_records_free(fem_records *r) {
  if (!r) return;
  free(r->rec_off), free(r->flag), free(r->tid), free(r->pos0), free(r->nm);
  free(r->cigar_off), free(r->cigar), free(r->md_off), free(r->md);
  memset(r, 0, sizeof *r);
}

int fem_sam_fill_quals(char *text, uint64_t text_len, const uint64_t *qual_at, uint64_t n_reads, const char *quals, const uint64_t *off,
                       uint32_t read_len, int n_threads) {
  if (!text || !qual_at || !quals) return -1;
  static SleepingPool pool;  // (its own: the parser's pool is busy with the next batch meanwhile)
  const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(1, n_threads), n_reads / 4096 + 1));
  std::vector<int> bad((size_t)nt, 0);
  pool.run(nt, [&](int t) {
    const uint64_t lo = n_reads * (uint64_t)t / (uint64_t)nt, hi = n_reads * (uint64_t)(t + 1) / (uint64_t)nt;
    for (uint64_t r = lo; r < hi; ++r) {
      const uint64_t at = qual_at[r];
      if (at == ~0ull) continue;
      const uint64_t from = off ? off[r] : r * (uint64_t)read_len, len = off ? off[r + 1] - off[r] : (uint64_t)read_len;
      if (at > text_len || len > text_len - at) {
        bad[(size_t)t] = 1;
        continue;
      }
      memcpy(text + at, quals + from, len);
    }
  });
  for (int b : bad)
    if (b) return -1;
  return 0;
}

int fem_sam_header(const fem_tail_ref *ref, char **text, uint64_t *text_len) {
  if (!ref || !text || !text_len) return -1;
  std::string s;
  for (uint32_t i = 0; i < ref->n_seq; ++i) {  // "@SQ\tSN:%s\tLN:%d\n" (src/output_queue.c:107)
    s += "@SQ\tSN:";
    s.append(ref->names + ref->name_off[i], ref->name_off[i + 1] - ref->name_off[i]);
    s += "\tLN:";
    s += std::to_string((int)ref->len[i]);
    s += "\n";
  }
  *text = (char *)malloc(s.size() + 1);
  if (!*text) return -4;
  memcpy(*text, s.data(), s.size());
  *text_len = s.size();
  return 0;
}

int fem_tail_sam(int32_t e, const fem_tail_ref *ref, const fem_seqset *reads, const fem_tail_input *in, int n_threads,
                 char **text, uint64_t *text_len) {
  if (!ref || !reads || !in || !text || !text_len) return -1;
  if (in->n_reads > reads->n) return -1;
  if (n_threads < 1) n_threads = 1;
  const uint64_t n = in->n_reads;
  std::vector<TextBuf> parts((size_t)n_threads);
#pragma omp parallel num_threads(n_threads)
  {
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
    TextBuf &o = parts[(size_t)t];
    const uint64_t lo = n * (uint64_t)t / (uint64_t)nt, hi = n * (uint64_t)(t + 1) / (uint64_t)nt;
    std::vector<Hit> hits;
    Tracer tr;
    std::string rev, md;
    std::vector<uint32_t> cigar;
    for (uint64_t r = lo; r < hi; ++r) {
      collect_hits(*in, r, hits);
      if (hits.empty()) continue;  // unmapped reads produce no record (src/map.c:50)
      const char *fwd = reads->bases + reads->off[r];
      const uint32_t len = (uint32_t)(reads->off[r + 1] - reads->off[r]);
      const char *qual = reads->quals ? reads->quals + reads->off[r] : nullptr;
      const char *name = reads->names + reads->name_off[r];
      const size_t name_len = (size_t)(reads->name_off[r + 1] - reads->name_off[r]);
      process_read(e, *ref, fwd, len, hits, tr, rev, cigar, md,
                   [&](size_t rank, const Record &rec, const std::vector<uint32_t> &cg, const std::string &m) {
                     append_sam_line(o, *ref, name, name_len, rec, cg.data(), cg.size(), m.data(), m.size(), rank == 0, fwd,
                                     len, qual);
                   });
    }
  }
  return join_parts(parts, n_threads, text, text_len);
}

int fem_records_sam(const fem_tail_ref *ref, const fem_seqset *reads, const fem_record_view *rv, int n_threads, char **text,
                    uint64_t *text_len) {
  if (!ref || !reads || !rv || !text || !text_len) return -1;
  if (rv->n_reads > reads->n) return -1;
  if (n_threads < 1) n_threads = 1;
  const uint64_t n = rv->n_reads;
  std::vector<TextBuf> parts((size_t)n_threads);
#pragma omp parallel num_threads(n_threads)
  {
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
    TextBuf &o = parts[(size_t)t];
    // threads take contiguous read ranges holding about the same number of records
    const uint64_t total = rv->rec_begin[n];
    auto cut = [&](uint64_t k) -> uint64_t {
      if (k == 0) return 0;
      if (k >= (uint64_t)nt) return n;
      const uint64_t target = total * k / (uint64_t)nt;
      return (uint64_t)(std::lower_bound(rv->rec_begin, rv->rec_begin + n, (uint32_t)target) - rv->rec_begin);
    };
    const uint64_t lo = cut((uint64_t)t), hi = cut((uint64_t)t + 1);
    if (hi > lo) (void)o.room((size_t)((rv->rec_begin[hi] - rv->rec_begin[lo]) * 260ull));
    for (uint64_t r = lo; r < hi; ++r) {
      const uint32_t b = rv->rec_begin[r], e_ = rv->rec_begin[r + 1];
      if (b == e_) continue;  // unmapped reads produce no record (src/map.c:50)
      const char *fwd = reads->bases + reads->off[r];
      const uint32_t len = (uint32_t)(reads->off[r + 1] - reads->off[r]);
      const char *qual = reads->quals ? reads->quals + reads->off[r] : nullptr;
      const char *name = reads->names + reads->name_off[r];
      const size_t name_len = (size_t)(reads->name_off[r + 1] - reads->name_off[r]);
      for (uint32_t j = b; j < e_; ++j) {
        Record rec;
        rec.flag = rv->flag[j], rec.tid = rv->tid[j], rec.pos0 = rv->pos0[j], rec.nm = rv->nm[j];
        append_sam_line(o, *ref, name, name_len, rec, rv->cigar + rv->cigar_off[j], rv->cigar_off[j + 1] - rv->cigar_off[j],
                        rv->md + rv->md_off[j], rv->md_off[j + 1] - rv->md_off[j], j == b, fwd, len, qual);
      }
    }
  }
  return join_parts(parts, n_threads, text, text_len);
}

}  // extern "C"

namespace {
// What a record line takes from its read: name, bases as they were read, qualities.
struct ReadFields {
  const char *name, *fwd, *qual;
  size_t name_len;
  uint32_t len;
};
// fem_records_sam_parts / fem_records_sam_refs: `read_of(r)` says where read r's fields are.
template <typename ReadOf>
int records_sam_parts_impl(const fem_tail_ref *ref, uint64_t n_reads_have, ReadOf read_of, const fem_record_view *rv, int n_threads,
                           char **buf, uint64_t *cap, fem_text_part *parts, uint64_t *n_asserted) {
  if (!ref || !rv || !buf || !cap || !parts) return -1;
  if (rv->n_reads > n_reads_have) return -1;
  if (n_threads < 1) n_threads = 1;
  const uint64_t n = rv->n_reads;
  const uint64_t total = n ? rv->rec_begin[n] : 0;
  size_t max_rname = 0;
  for (uint32_t i = 0; i < ref->n_seq; ++i) max_rname = std::max(max_rname, (size_t)(ref->name_off[i + 1] - ref->name_off[i]));
  std::vector<uint64_t> lo((size_t)n_threads + 1, n), need((size_t)n_threads + 1, 0);
  for (int k = 0; k <= n_threads; ++k) {  // contiguous read ranges holding about the same number of records
    if (k == 0) lo[0] = 0;
    else if (k < n_threads)
      lo[(size_t)k] = (uint64_t)(std::lower_bound(rv->rec_begin, rv->rec_begin + n, (uint32_t)(total * (uint64_t)k / (uint64_t)n_threads)) - rv->rec_begin);
  }
  uint64_t asserted = 0;
  bool failed = false;
#pragma omp parallel num_threads(n_threads) reduction(+ : asserted)
  {
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
    // an upper bound of this thread's text (what append_sam_line asks room for, record by record)
    for (int k = t; k < n_threads; k += nt) {
      uint64_t b = 0;
      for (uint64_t r = lo[(size_t)k]; r < lo[(size_t)k + 1]; ++r) {
        const uint32_t rb = rv->rec_begin[r], re = rv->rec_begin[r + 1];
        if (rb == re) continue;
        const ReadFields rf = read_of(r);
        b += (uint64_t)(re - rb) * (rf.name_len + max_rname + 2 * (uint64_t)rf.len + 96) + 11ull * (rv->cigar_off[re] - rv->cigar_off[rb]) + (rv->md_off[re] - rv->md_off[rb]);
      }
      need[(size_t)k + 1] = b;
    }
#pragma omp barrier
#pragma omp single
    {
      for (int k = 0; k < n_threads; ++k) need[(size_t)k + 1] += need[(size_t)k];
      if (need[(size_t)n_threads] + 1 > *cap) {
        const uint64_t want = need[(size_t)n_threads] + need[(size_t)n_threads] / 8 + (1u << 20);
        char *q = (char *)realloc(*buf, want);
        if (q) *buf = q, *cap = want;
        else failed = true;
      }
    }  // (implicit barrier)
    if (!failed) {
      for (int k = t; k < n_threads; k += nt) {
        TextBuf o;
        o.p = *buf + need[(size_t)k], o.n = 0, o.cap = (size_t)(need[(size_t)k + 1] - need[(size_t)k]);
        o.fixed = true;
        for (uint64_t r = lo[(size_t)k]; r < lo[(size_t)k + 1]; ++r) {
          const uint32_t b = rv->rec_begin[r], e_ = rv->rec_begin[r + 1];
          if (b == e_) continue;  // unmapped reads produce no record (src/map.c:50)
          const ReadFields rf = read_of(r);
          const char *fwd = rf.fwd, *qual = rf.qual, *name = rf.name;
          const uint32_t len = rf.len;
          const size_t name_len = rf.name_len;
          for (uint32_t j = b; j < e_; ++j) {
            Record rec;
            rec.flag = rv->flag[j], rec.tid = rv->tid[j], rec.pos0 = rv->pos0[j], rec.nm = rv->nm[j];
            if (rec.flag & 0x8000u) ++asserted, rec.flag &= 0x7FFFu;
            append_sam_line(o, *ref, name, name_len, rec, rv->cigar + rv->cigar_off[j], rv->cigar_off[j + 1] - rv->cigar_off[j],
                            rv->md + rv->md_off[j], rv->md_off[j + 1] - rv->md_off[j], j == b, fwd, len, qual);
          }
        }
        parts[k].offset = need[(size_t)k], parts[k].length = o.n;
      }
    }
  }
  if (failed) return -4;
  if (n_asserted) *n_asserted = asserted;
  return 0;
}
}  // namespace

extern "C" {

int fem_records_sam_parts(const fem_tail_ref *ref, const fem_seqset *reads, const fem_record_view *rv, int n_threads,
                          char **buf, uint64_t *cap, fem_text_part *parts, uint64_t *n_asserted) {
  if (!reads) return -1;
  auto read_of = [reads](uint64_t r) {
    ReadFields f;
    f.fwd = reads->bases + reads->off[r], f.len = (uint32_t)(reads->off[r + 1] - reads->off[r]);
    f.qual = reads->quals ? reads->quals + reads->off[r] : nullptr;
    f.name = reads->names + reads->name_off[r], f.name_len = (size_t)(reads->name_off[r + 1] - reads->name_off[r]);
    return f;
  };
  return records_sam_parts_impl(ref, reads->n, read_of, rv, n_threads, buf, cap, parts, n_asserted);
}

// The same for reads that were never copied: name, bases and qualities of read r are where the parser found them in the
// (still mapped) input file — fem_seqfile_fill_packed_refs.  Reads of one length.
int fem_records_sam_refs(const fem_tail_ref *ref, const fem_read_refs *reads, const fem_record_view *rv, int n_threads, char **buf,
                         uint64_t *cap, fem_text_part *parts, uint64_t *n_asserted) {
  if (!reads || (reads->n && (!reads->name || !reads->name_len || !reads->seq || !reads->qual))) return -1;
  auto read_of = [reads](uint64_t r) {
    ReadFields f;
    f.fwd = reads->seq[r], f.len = reads->read_len, f.qual = reads->qual[r];
    f.name = reads->name[r], f.name_len = reads->name_len[r];
    return f;
  };
  return records_sam_parts_impl(ref, reads->n, read_of, rv, n_threads, buf, cap, parts, n_asserted);
}

// ------------------------------------------------------------------------------------------------
// synthetic data; splitmix64 streams keyed by (seed, index)
// ------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {
struct SplitMix {
  uint64_t s;
  explicit SplitMix(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
};
inline uint64_t stream_seed(uint64_t seed, uint64_t tag, uint64_t index) {
  SplitMix m(seed ^ (tag * 0xD6E8FEB86659FD93ull) ^ (index * 0xA0761D6478BD642Full));
  m.next();
  return m.next();
}
}  // namespace

extern "C" {

void fem_synth_reference(uint64_t seed, uint32_t n_seq, const uint64_t *seq_off, const uint32_t *seq_len, char *out,
                         int n_threads) {
  static const char acgt[4] = {'A', 'C', 'G', 'T'};
  if (n_threads < 1) n_threads = 1;
  const uint64_t chunk = 1u << 20;  // every 1 Mi bases of a sequence form their own stream -> parallel + reproducible
  for (uint32_t s = 0; s < n_seq; ++s) {
    const uint64_t len = seq_len[s], n_chunks = (len + chunk - 1) / chunk;
    char *dst = out + seq_off[s];
#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (int64_t c = 0; c < (int64_t)n_chunks; ++c) {
      SplitMix rng(stream_seed(seed, 0x5EC0 + s, (uint64_t)c));
      const uint64_t lo = (uint64_t)c * chunk, hi = std::min(len, lo + chunk);
      for (uint64_t i = lo; i < hi;) {
        uint64_t bits = rng.next();
        for (int j = 0; j < 32 && i < hi; ++j, ++i, bits >>= 2) dst[i] = acgt[bits & 3];
      }
    }
  }
}

void fem_synth_reads(uint64_t seed, const char *ref_text, const uint64_t *seq_off, const uint32_t *seq_len,
                     uint32_t n_seq, uint64_t first_read, uint64_t n_reads, uint32_t L, int32_t e, char *bases_out,
                     int n_threads) {
  fem_synth_reads_ex(seed, ref_text, seq_off, seq_len, n_seq, first_read, n_reads, L, e, bases_out, nullptr, n_threads);
}

void fem_synth_reads_ex(uint64_t seed, const char *ref_text, const uint64_t *seq_off, const uint32_t *seq_len,
                        uint32_t n_seq, uint64_t first_read, uint64_t n_reads, uint32_t L, int32_t e, char *bases_out,
                        uint8_t *n_err_out, int n_threads) {
  static const char acgt[4] = {'A', 'C', 'G', 'T'};
  if (n_threads < 1) n_threads = 1;
  // sequences long enough to hold a read, weighted by the number of admissible start positions
  std::vector<uint64_t> cum(n_seq + 1, 0);
  const uint64_t span = (uint64_t)L + (uint64_t)e;
  for (uint32_t s = 0; s < n_seq; ++s) cum[s + 1] = cum[s] + (seq_len[s] > span ? seq_len[s] - span : 0);
  const uint64_t total = cum[n_seq];
#pragma omp parallel num_threads(n_threads)
  {
    std::vector<char> buf(span + 8);
#pragma omp for schedule(static)
    for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
      SplitMix rng(stream_seed(seed, 0x7EAD, first_read + (uint64_t)r));
      char *dst = bases_out + (uint64_t)r * L;
      if (total == 0) {
        for (uint32_t i = 0; i < L; ++i) dst[i] = acgt[rng.below(4)];
        if (n_err_out) n_err_out[r] = 0;
        continue;
      }
      const uint64_t pick = rng.below(total);
      const uint32_t s = (uint32_t)(std::upper_bound(cum.begin(), cum.end(), pick) - cum.begin() - 1);
      const uint64_t start = pick - cum[s];
      memcpy(buf.data(), ref_text + seq_off[s] + start, span);
      size_t cur = span;
      const int n_err = (int)rng.below((uint64_t)e + 1);
      if (n_err_out) n_err_out[r] = (uint8_t)n_err;
      for (int k = 0; k < n_err; ++k) {
        const uint64_t kind = rng.below(10);
        // a uniform interior offset of the READ: inside the first L bases, so that the truncation to L
        // never cuts an error off (round 1 drew from the whole L + e window: ~e/2L of the errors fell behind the cut)
        const size_t pos = 1 + (size_t)rng.below(L > 2 ? L - 2 : 1);
        if (kind < 6) {  // substitution by a different base
          const char old = buf[pos];
          char nb;
          do nb = acgt[rng.below(4)];
          while (nb == old);
          buf[pos] = nb;
        } else if (kind < 8) {  // insertion
          if (cur < buf.size()) {
            memmove(buf.data() + pos + 1, buf.data() + pos, cur - pos);
            buf[pos] = acgt[rng.below(4)];
            ++cur;
          }
        } else {  // deletion
          memmove(buf.data() + pos, buf.data() + pos + 1, cur - pos - 1);
          --cur;
        }
      }
      // cur >= span - e = L always holds
      if (rng.below(2)) {
        for (uint32_t i = 0; i < L; ++i) dst[i] = kCodeChar[3 ^ code_of(buf[L - 1 - i])];
      } else {
        memcpy(dst, buf.data(), L);
      }
    }
  }
}

// The same reads at two bits per base (fem_pack.h), written straight into `codes` (e.g. the pinned staging lent by
// fem_dev_acquire_stage): ceil(L / 4) bytes per read.  The generator draws A C G T only, so there are no exceptions.
void fem_synth_reads_packed(uint64_t seed, const char *ref_text, const uint64_t *seq_off, const uint32_t *seq_len, uint32_t n_seq,
                            uint64_t first_read, uint64_t n_reads, uint32_t L, int32_t e, uint8_t *codes, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  const uint32_t bpr = fempack::bytes_per_read(L);
  constexpr uint64_t kPiece = 4096;  // reads generated as characters and packed at a time (they stay in the core's cache)
  const int64_t n_pieces = (int64_t)((n_reads + kPiece - 1) / kPiece);
#pragma omp parallel num_threads(n_threads)
  {
    std::vector<char> tmp(kPiece * L + 64);
    std::vector<uint64_t> exc;
#pragma omp for schedule(dynamic, 1)
    for (int64_t pc = 0; pc < n_pieces; ++pc) {
      const uint64_t lo = (uint64_t)pc * kPiece, n = std::min<uint64_t>(kPiece, n_reads - lo);
      fem_synth_reads_ex(seed, ref_text, seq_off, seq_len, n_seq, first_read + lo, n, L, e, tmp.data(), nullptr, 1);
      if ((L & 3u) == 0u) {
        fempack::pack_bases((const uint8_t *)tmp.data(), n * L, codes + lo * bpr, lo * L, exc);
      } else {
        for (uint64_t i = 0; i < n; ++i) fempack::pack_bases((const uint8_t *)tmp.data() + i * L, L, codes + (lo + i) * bpr, (lo + i) * L, exc);
      }
    }
  }
  for (uint64_t i = n_reads * bpr; i < fempack::code_bytes(n_reads, L); ++i) codes[i] = 0;
}

int fem_synth_write_fastq(const char *path, const char *bases, uint32_t L, uint64_t n_reads, uint64_t first_index) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  std::vector<char> buf;
  buf.reserve(1u << 24);
  std::string qual(L, 'I');
  for (uint64_t i = 0; i < n_reads; ++i) {
    char name[32];
    int nl = snprintf(name, sizeof name, "@r%lu\n", (unsigned long)(first_index + i));
    buf.insert(buf.end(), name, name + nl);
    buf.insert(buf.end(), bases + i * L, bases + (i + 1) * L);
    buf.push_back('\n');
    buf.push_back('+');
    buf.push_back('\n');
    buf.insert(buf.end(), qual.begin(), qual.end());
    buf.push_back('\n');
    if (buf.size() > (1u << 24) - 1024) {
      if (fwrite(buf.data(), 1, buf.size(), f) != buf.size()) {
        fclose(f);
        return -2;
      }
      buf.clear();
    }
  }
  bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
  return (fclose(f) == 0 && ok) ? 0 : -2;
}

int fem_synth_write_fasta(const char *path, const char *text, const uint64_t *seq_off, const uint32_t *seq_len,
                          uint32_t n_seq) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  bool ok = true;
  for (uint32_t s = 0; s < n_seq && ok; ++s) {
    ok = fprintf(f, ">chr%u synthetic\n", s + 1) > 0;
    const char *p = text + seq_off[s];
    for (uint64_t i = 0; i < seq_len[s] && ok; i += 60) {
      size_t n = (size_t)std::min<uint64_t>(60, seq_len[s] - i);
      ok = fwrite(p + i, 1, n, f) == n && fputc('\n', f) != EOF;
    }
  }
  return (fclose(f) == 0 && ok) ? 0 : -2;
}

}  // extern "C"
