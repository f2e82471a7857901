// tools/rss_probe.hip -- where does the resident set size of a process using libawm_hip.so come from?
// build: hipcc -O2 --offload-arch=gfx950 -o tools/rss_probe tools/rss_probe.hip -Laudiowmark_amd -lawm_hip -Wl,-rpath,'$ORIGIN/../audiowmark_amd'
#include "../include/awm_hip.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static long rss_mb() { FILE *f = fopen ("/proc/self/status", "r"); char line[256]; long kb = 0; while (fgets (line, sizeof line, f)) if (!strncmp (line, "VmRSS:", 6)) sscanf (line + 6, "%ld", &kb); fclose (f); return kb / 1024; }
#define STEP(name) printf ("%-44s %5ld MB\n", name, rss_mb())
__global__ void tiny (float *p) { p[threadIdx.x] = 1.f; }
int main (int argc, char **argv)
{
  setvbuf (stdout, nullptr, _IONBF, 0);
  STEP ("start");
  (void) hipSetDevice (0);                                   STEP ("hipSetDevice");
  float *p; (void) hipMalloc (&p, 1 << 20);                  STEP ("hipMalloc 1 MiB");
  tiny<<<1, 64>>> (p); (void) hipDeviceSynchronize();        STEP ("first kernel (default stream)");
  float *big; (void) hipMalloc (&big, size_t (1) << 30);
  (void) hipMemsetAsync (big, 0, size_t (1) << 30, 0); (void) hipDeviceSynchronize();   STEP ("hipMalloc + hipMemsetAsync 1 GiB");
  (void) hipMemcpyAsync (big + (1 << 20), big, 1 << 20, hipMemcpyDeviceToDevice, 0); (void) hipDeviceSynchronize();   STEP ("hipMemcpyAsync D2D");
  void *pin; (void) hipHostMalloc (&pin, 64 << 20, hipHostMallocDefault); memset (pin, 1, 64 << 20);      STEP ("hipHostMalloc 64 MiB (touched)");
  (void) hipMemcpyAsync (big, pin, 64 << 20, hipMemcpyHostToDevice, 0); (void) hipDeviceSynchronize();    STEP ("H2D from pinned, default stream");
  (void) hipMemcpyAsync (pin, big, 64 << 20, hipMemcpyDeviceToHost, 0); (void) hipDeviceSynchronize();    STEP ("D2H to pinned, default stream");
  hipEvent_t ev; (void) hipEventCreateWithFlags (&ev, hipEventDisableTiming); (void) hipEventRecord (ev, 0); (void) hipEventSynchronize (ev);  STEP ("event record + synchronize");
  std::thread ([&] { (void) hipEventSynchronize (ev); }).join();                                          STEP ("hipEventSynchronize on a second host thread");
  hipStream_t st; (void) hipStreamCreateWithFlags (&st, hipStreamNonBlocking);                            STEP ("hipStreamCreate (non-blocking)");
  (void) hipMemcpyAsync (big, pin, 64 << 20, hipMemcpyHostToDevice, st); (void) hipStreamSynchronize (st);  STEP ("H2D from pinned on that stream");
  (void) hipStreamWaitEvent (0, ev, 0); tiny<<<1, 64>>> (p); (void) hipDeviceSynchronize();               STEP ("stream wait event + kernel");
  static float hostbuf[1 << 16];
  (void) hipMemcpy (p, hostbuf, sizeof hostbuf, hipMemcpyHostToDevice);                                   STEP ("hipMemcpy from PAGEABLE memory");
  awm_ctx *ctx = nullptr;
  (void) awm_ctx_create_on_stream (0, nullptr, &ctx);                                                     STEP ("awm_ctx_create_on_stream (default stream)");
  if (argc > 2)
    {
      unsigned char key[16] = {};
      awm_raw_format rf = { 2, 44100, 16, 0, 0 };
      awm_set_quiet (1);
      (void) awm_add_watermark_file (ctx, key, "0123456789abcdef0011223344556677", argv[1], argv[2], &rf, &rf);   STEP ("awm_add_watermark_file");
      awm_pattern pat[256];
      (void) awm_get_watermark_file (ctx, key, argv[2], &rf, 256, pat);                                    STEP ("awm_get_watermark_file");
    }
  return 0;
}
