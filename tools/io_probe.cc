// tools/io_probe.cc -- where does an output file of 640 MB in tmpfs spend its time?  (development tool; `hipcc -O2 -o tools/io_probe tools/io_probe.cc`)
//   ./io_probe [threads = 8] [dir = /dev/shm]
// Writes 640 MB in 16 MB tiles: fwrite by one thread; pwrite by N threads into ONE file (buffered writes serialise on the inode);
// N threads copying into a shared mapping of a file grown by ftruncate (page faults allocate in parallel), with and without
// MADV_POPULATE_WRITE; N threads writing N separate files (is the limit per file or global?); the same from a PAGE-LOCKED source
// (hipHostMalloc) that a device-to-host copy has just filled; and reads (pread by 1 / N threads).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <vector>
#include <string>
#include <chrono>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
using namespace std;
static double now() { return chrono::duration<double> (chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t TOTAL = size_t (640) << 20, TILE = size_t (16) << 20;

static void
run (const char *what, const unsigned char *src, int T, const string& dir)
{
  const string path = dir + "/io_probe.bin";
  {
    unlink (path.c_str()); FILE *f = fopen (path.c_str(), "w"); double t = now();
    for (size_t p = 0; p < TOTAL; p += TILE) fwrite (src, 1, TILE, f);
    fclose (f); printf ("%-10s fwrite, 1 thread:                      %7.1f ms\n", what, (now() - t) * 1e3);
  }
  {
    unlink (path.c_str()); int fd = open (path.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644); double t = now();
    vector<thread> th;
    for (int i = 0; i < T; i++) th.emplace_back ([=] { for (size_t p = i * TILE; p < TOTAL; p += TILE * T) if (pwrite (fd, src, TILE, p) < 0) perror ("pwrite"); });
    for (auto& x : th) x.join();
    close (fd); printf ("%-10s pwrite, %2d threads, one file:           %7.1f ms\n", what, T, (now() - t) * 1e3);
  }
  for (int pop = 0; pop < 2; pop++)
    {
      unlink (path.c_str()); int fd = open (path.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644); double t = now();
      if (ftruncate (fd, TOTAL)) perror ("ftruncate");
      unsigned char *m = (unsigned char *) mmap (0, TOTAL, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      const size_t piece = size_t (2) << 20;
      vector<thread> th;
      for (int i = 0; i < T; i++) th.emplace_back ([=] { for (size_t p = i * piece; p < TOTAL; p += piece * T) { if (pop) madvise (m + p, piece, MADV_POPULATE_WRITE); memcpy (m + p, src + (p % TILE), piece); madvise (m + p, piece, MADV_DONTNEED); } });
      for (auto& x : th) x.join();
      double t1 = now(); munmap (m, TOTAL); close (fd);
      printf ("%-10s shared mapping, %2d threads%s:  %7.1f ms (+ munmap %.1f)\n", what, T, pop ? ", populate" : "          ", (t1 - t) * 1e3, (now() - t1) * 1e3);
    }
  {
    double t = now();
    vector<thread> th;
    for (int i = 0; i < T; i++) th.emplace_back ([=] { string p2 = dir + "/io_probe_" + to_string (i) + ".bin"; unlink (p2.c_str()); int fd = open (p2.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
                                                       for (size_t p = 0; p < TOTAL / T; p += TILE) if (write (fd, src, TILE) < 0) perror ("write"); close (fd); });
    for (auto& x : th) x.join();
    printf ("%-10s write, %2d threads, %2d separate files:   %7.1f ms\n", what, T, T, (now() - t) * 1e3);
    for (int i = 0; i < T; i++) unlink ((dir + "/io_probe_" + to_string (i) + ".bin").c_str());
  }
  {
    unsigned char *dst = (unsigned char *) aligned_alloc (4096, TILE); double t = now();
    vector<thread> th;
    for (int i = 0; i < T; i++) th.emplace_back ([=] { for (size_t p = i * (TILE / T); p < TOTAL; p += TILE) memcpy (dst + (p % TILE), src + (p % TILE), TILE / T); });
    for (auto& x : th) x.join();
    printf ("%-10s plain memcpy into 16 MB of heap, %2d thr: %7.1f ms\n", what, T, (now() - t) * 1e3);
    free (dst);
  }
  unlink (path.c_str());
}

int
main (int argc, char **argv)
{
  const int T = argc > 1 ? atoi (argv[1]) : 8;
  const string dir = argc > 2 ? argv[2] : "/dev/shm";
  unsigned char *heap = (unsigned char *) aligned_alloc (4096, TILE);
  memset (heap, 1, TILE);
  run ("heap", heap, T, dir);
  unsigned char *pinned = nullptr, *dev = nullptr;
  if (hipHostMalloc ((void **) &pinned, TILE, hipHostMallocDefault) == hipSuccess && hipMalloc ((void **) &dev, TILE) == hipSuccess)
    {
      hipMemset (dev, 2, TILE);
      double t = now();
      for (int i = 0; i < 40; i++) hipMemcpy (pinned, dev, TILE, hipMemcpyDeviceToHost);
      printf ("40 x D2H of 16 MB into page-locked memory: %.1f ms\n", (now() - t) * 1e3);
      run ("pinned", pinned, T, dir);
      // reads: pread into page-locked memory
      const string path = dir + "/io_probe.bin";
      { FILE *f = fopen (path.c_str(), "w"); for (size_t p = 0; p < TOTAL; p += TILE) fwrite (heap, 1, TILE, f); fclose (f); }
      for (int tt : { 1, T })
        {
          int fd = open (path.c_str(), O_RDONLY); double t0 = now();
          vector<thread> th;
          for (int i = 0; i < tt; i++) th.emplace_back ([=] { for (size_t p = i * (TILE / tt); p < TOTAL; p += TILE) if (pread (fd, pinned + (p % TILE), TILE / tt, p) < 0) perror ("pread"); });
          for (auto& x : th) x.join();
          close (fd); printf ("pread into page-locked memory, %2d threads: %.1f ms\n", tt, (now() - t0) * 1e3);
        }
      unlink (path.c_str());
    }
  else
    printf ("(no GPU: page-locked source skipped)\n");
  return 0;
}
