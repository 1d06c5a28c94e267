// tools/parse_probe.cc <fastq> <threads>: seconds of fem_seqfile_plan and fem_seqfile_fill_packed per batch of 250 MB
#include "../fem_amd/csrc/fem_host.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
extern "C" int fem_dev_packed_layout(uint64_t, uint32_t, uint32_t *, uint64_t *, uint64_t *);
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const int nt = argc > 2 ? atoi(argv[2]) : 1;
  for (int rep = 0; rep < 2; ++rep) {
    fem_seqfile *f = fem_seqfile_open(argv[1]);
    std::vector<char> codes(300u << 20), quals(300u << 20), names(100u << 20);
    std::vector<uint64_t> name_off(4u << 20);
    double t_plan = 0, t_fill = 0;
    uint64_t n = 0;
    for (;;) {
      fem_batch_plan *pl = nullptr;
      fem_batch_shape sh{};
      double t0 = now();
      if (fem_seqfile_plan(f, 250000000ull, nt, &pl, &sh) != 0 || !pl || sh.n_reads == 0) { fem_batch_plan_free(pl); break; }
      double t1 = now();
      uint64_t n_exc = 0;
      int rc = fem_seqfile_fill_packed(f, pl, nt, sh.max_len, (uint8_t *)codes.data(), 1u << 20, &n_exc, quals.data(), names.data(), name_off.data());
      double t2 = now();
      if (rc) { printf("fill rc %d\n", rc); break; }
      t_plan += t1 - t0, t_fill += t2 - t1, n += sh.n_reads;
    }
    printf("rep %d: %lu reads, %d threads: plan %.1f ns per read, fill %.1f ns per read (wall x threads)\n", rep, (unsigned long)n, nt, 1e9 * t_plan * nt / n, 1e9 * t_fill * nt / n);
    fem_seqfile_close(f);
  }
  return 0;
}
