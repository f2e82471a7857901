// tools/graph_probe.hip -- what does a chain of 38 small dependent kernels cost when it is issued launch by launch, and as one
// hipGraphLaunch?  Three chains on three streams (the decode chains of three chunks), issued by one host thread.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/graph_probe tools/graph_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf (stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString (e_)); return 1; } } while (0)
struct Args { float *p; int n; int pad[24]; };
__global__ void small (Args a, int round)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = a.p[i];
  for (int k = 0; k < 400; k++)
    v = v * 1.0001f + 0.5f;
  a.p[(i * 7 + round) % a.n] = v;
}
static double now() { return std::chrono::duration<double> (std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  constexpr int S = 3, N = 38, GRID = 400;
  hipStream_t st[S]; float *buf[S]; hipGraphExec_t exec[S];
  for (int s = 0; s < S; s++)
    {
      CK (hipStreamCreateWithFlags (&st[s], hipStreamNonBlocking));
      CK (hipMalloc (&buf[s], GRID * 256 * 4));
      CK (hipMemset (buf[s], 0, GRID * 256 * 4));
    }
  auto chain = [&] (int s) { Args a { buf[s], GRID * 256, {} }; for (int r = 0; r < N; r++) hipLaunchKernelGGL (small, dim3 (GRID), dim3 (256), 0, st[s], a, r); };
  for (int s = 0; s < S; s++)
    {
      hipGraph_t g;
      CK (hipStreamBeginCapture (st[s], hipStreamCaptureModeThreadLocal));
      chain (s);
      CK (hipStreamEndCapture (st[s], &g));
      CK (hipGraphInstantiate (&exec[s], g, nullptr, nullptr, 0));
    }
  for (int mode = 0; mode < 2; mode++)
    for (int rep = 0; rep < 4; rep++)
      {
        CK (hipDeviceSynchronize());
        const double t0 = now();
        for (int s = 0; s < S; s++)
          if (mode == 0) chain (s); else CK (hipGraphLaunch (exec[s], st[s]));
        const double t1 = now();
        CK (hipDeviceSynchronize());
        const double t2 = now();
        printf ("%s: issue %.0f us, all three chains done after %.0f us\n", mode ? "graph launch " : "launch by launch", (t1 - t0) * 1e6, (t2 - t0) * 1e6);
      }
  return 0;
}
