// tools/stream_create_probe.cc -- what does a HIP stream cost to create, one after the other and from several threads at once?
// (development tool: hipcc -O2 -o tools/stream_create_probe tools/stream_create_probe.cc)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k (int *p) { if (p) *p = 1; }
int main()
{
  int n = 0; hipGetDeviceCount (&n); hipSetDevice (0);
  int *d; hipMalloc ((void **) &d, 4);
  hipLaunchKernelGGL (k, dim3 (1), dim3 (64), 0, nullptr, d); hipDeviceSynchronize();
  hipStream_t s[24];
  for (int i = 0; i < 4; i++) { double t = now(); hipStreamCreateWithFlags (&s[i], hipStreamNonBlocking); printf ("sequential create %d: %.2f ms\n", i, now() - t); }
  for (int i = 0; i < 4; i++) { double t = now(); hipLaunchKernelGGL (k, dim3 (1), dim3 (64), 0, s[i], d); hipStreamSynchronize (s[i]); printf ("first launch + wait on stream %d: %.2f ms\n", i, now() - t); }
  { double t = now(); std::vector<std::thread> th; for (int i = 4; i < 12; i++) th.emplace_back ([&, i] { hipSetDevice (0); hipStreamCreateWithFlags (&s[i], hipStreamNonBlocking); });
    for (auto& x : th) x.join(); printf ("8 creates from 8 threads: %.2f ms\n", now() - t); }
  { double t = now(); std::vector<std::thread> th; for (int i = 4; i < 12; i++) th.emplace_back ([&, i] { hipSetDevice (0); hipLaunchKernelGGL (k, dim3 (1), dim3 (64), 0, s[i], d); hipStreamSynchronize (s[i]); });
    for (auto& x : th) x.join(); printf ("first launch + wait on those 8, from 8 threads: %.2f ms\n", now() - t); }
  for (int i = 12; i < 16; i++) { double t = now(); hipStreamCreateWithFlags (&s[i], hipStreamNonBlocking); printf ("sequential create %d: %.2f ms\n", i, now() - t); }
  void *h; double t = now(); hipHostMalloc (&h, 64 << 20, hipHostMallocDefault); printf ("hipHostMalloc 64 MB: %.2f ms\n", now() - t);
  t = now(); hipMemcpyAsync (h, d, 4, hipMemcpyDeviceToHost, s[0]); hipStreamSynchronize (s[0]); printf ("first D2H on stream 0: %.2f ms\n", now() - t);
  t = now(); hipMemcpyAsync (h, d, 4, hipMemcpyDeviceToHost, s[1]); hipStreamSynchronize (s[1]); printf ("first D2H on stream 1: %.2f ms\n", now() - t);
  t = now(); hipMemcpyAsync (h, d, 4, hipMemcpyDeviceToHost, s[1]); hipStreamSynchronize (s[1]); printf ("second D2H on stream 1: %.2f ms\n", now() - t);
  t = now(); hipMemcpyAsync (d, h, 4, hipMemcpyHostToDevice, s[2]); hipStreamSynchronize (s[2]); printf ("first H2D on stream 2: %.2f ms\n", now() - t);
  return 0;
}
