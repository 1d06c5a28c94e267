// tools/mmap_write_probe.cc <dir>: how fast a file takes data through a shared mapping filled by N threads (against write())
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const size_t total = (size_t)3 << 30, piece = (size_t)300 << 20;
  char *src = (char *)malloc(piece);
  memset(src, 'x', piece);
  for (int mode = 0; mode < 2; ++mode)
    for (int nt : {1, 2, 4, 8, 16}) {
      if (mode == 0 && nt > 1) continue;
      const std::string path = dir + "/mmap_probe.bin";
      int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
      if (fd < 0) { perror("open"); return 1; }
      const double t0 = now();
      if (mode == 0) {
        for (size_t at = 0; at < total; at += piece) {
          size_t done = 0;
          while (done < piece) done += (size_t)write(fd, src + done, piece - done);
        }
      } else {
        if (ftruncate(fd, (off_t)total) != 0) { perror("ftruncate"); return 1; }
        for (size_t at = 0; at < total; at += piece) {  // a batch's text at a time, like FEM map's writer
          char *m = (char *)mmap(nullptr, piece, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)at);
          if (m == MAP_FAILED) { perror("mmap"); return 1; }
          std::vector<std::thread> ts;
          const size_t share = (piece / nt + 4095) & ~(size_t)4095;
          for (int t = 0; t < nt; ++t)
            ts.emplace_back([&, t] {
              const size_t lo = std::min(piece, (size_t)t * share), hi = std::min(piece, lo + share);
              if (hi > lo) memcpy(m + lo, src + lo, hi - lo);
            });
          for (auto &t : ts) t.join();
          munmap(m, piece);
        }
      }
      const double t1 = now();
      close(fd);
      const double t2 = now();
      printf("%s, %2d thread(s): %.2f GB/s (close %.2f s)\n", mode == 0 ? "write()" : "shared mapping", nt, total / (t1 - t0) / 1e9, t2 - t1);
      unlink(path.c_str());
    }
  return 0;
}
