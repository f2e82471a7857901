// tools/rss_probe.hip -- where does the resident set size of a process using libawm_hip.so come from?
#include "../include/awm_hip.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
static long rss_mb() { FILE *f = fopen ("/proc/self/status", "r"); char line[256]; long kb = 0; while (fgets (line, sizeof line, f)) if (!strncmp (line, "VmRSS:", 6)) sscanf (line + 6, "%ld", &kb); fclose (f); return kb / 1024; }
__global__ void tiny (float *p) { p[threadIdx.x] = 1.f; }
int main()
{
  setvbuf (stdout, nullptr, _IONBF, 0);
  printf ("start                         %5ld MB\n", rss_mb());
  hipSetDevice (0);
  printf ("hipSetDevice                  %5ld MB\n", rss_mb());
  float *p; hipMalloc (&p, 1 << 20);
  printf ("hipMalloc 1 MiB               %5ld MB\n", rss_mb());
  tiny<<<1, 64>>> (p); hipDeviceSynchronize();
  printf ("first kernel                  %5ld MB\n", rss_mb());
  float *big; hipMalloc (&big, size_t (1) << 30);
  printf ("hipMalloc 1 GiB               %5ld MB\n", rss_mb());
  hipMemset (big, 0, size_t (1) << 30); hipDeviceSynchronize();
  printf ("hipMemset 1 GiB               %5ld MB\n", rss_mb());
  void *pin; hipHostMalloc (&pin, 64 << 20, hipHostMallocDefault);
  printf ("hipHostMalloc 64 MiB          %5ld MB\n", rss_mb());
  memset (pin, 1, 64 << 20);
  printf ("  touched                     %5ld MB\n", rss_mb());
  hipStream_t st; hipStreamCreateWithFlags (&st, hipStreamNonBlocking);
  printf ("hipStreamCreate               %5ld MB\n", rss_mb());
  tiny<<<1, 64, 0, st>>> (p); hipStreamSynchronize (st);
  printf ("kernel on that stream         %5ld MB\n", rss_mb());
  static float hostbuf[1 << 16];
  hipMemcpy (p, hostbuf, sizeof hostbuf, hipMemcpyHostToDevice);
  printf ("hipMemcpy pageable 256 KiB    %5ld MB\n", rss_mb());
  hipMemcpyAsync (p, hostbuf, sizeof hostbuf, hipMemcpyHostToDevice, st); hipStreamSynchronize (st);
  printf ("hipMemcpyAsync pageable       %5ld MB\n", rss_mb());
  hipMemcpy (hostbuf, p, sizeof hostbuf, hipMemcpyDeviceToHost);
  printf ("hipMemcpy D2H pageable        %5ld MB\n", rss_mb());
  hipEvent_t ev; hipEventCreate (&ev); hipEventRecord (ev, st); hipEventSynchronize (ev);
  printf ("event                         %5ld MB\n", rss_mb());
  awm_ctx *ctx = nullptr;
  awm_ctx_create (0, &ctx);
  printf ("awm_ctx_create                %5ld MB\n", rss_mb());
  awm_add_stream *s = nullptr;
  unsigned char key[16] = {};
  awm_add_stream_create (ctx, key, "0123456789abcdef0011223344556677", 2, 4096, &s);
  printf ("awm_add_stream_create         %5ld MB\n", rss_mb());
  const float *out[3]; size_t nout[3];
  awm_add_stream_push (s, 4096 * 1024, 0, out, nout);
  awm_add_stream_push (s, 4096 * 1024, 1, out, nout);
  awm_ctx_synchronize (ctx);
  printf ("two tiles pushed              %5ld MB\n", rss_mb());
  float *pcm; hipMalloc (&pcm, size_t (60) * 44100 * 2 * 4 * 5);
  hipMemset (pcm, 0, size_t (60) * 44100 * 2 * 4 * 5);
  awm_pattern pat[64];
  awm_get_watermark_d (ctx, key, pcm, size_t (60) * 44100 * 5, 2, 64, pat);
  printf ("get on 5 min                  %5ld MB\n", rss_mb());
  return 0;
}
