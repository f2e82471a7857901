// tools/mfma_f64_probe.hip -- v_mfma_f64_4x4x4f64 on gfx950: which lane holds which element, in which order the products are added
// (is D = fma (a3, b3, fma (a2, b2, fma (a1, b1, fma (a0, b0, c))))?), and what it costs beside FP64 vector instructions.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe && tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>

__global__ void
one_mfma (const double *a, const double *b, const double *c, double *d)
{
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64 (a[l], b[l], c[l], 0, 0, 0);
}

template<int MODE> __global__ void __launch_bounds__ (256)
rate_kernel (double *out, double seed, int iters)
{
  double acc[12], v[12];
  const double m = seed * 0.999, c = seed * 1e-3;
  for (int i = 0; i < 12; i++)
    {
      acc[i] = seed + i + threadIdx.x;
      v[i] = seed - i;
    }
  for (int it = 0; it < iters; it++)
    {
      if (MODE == 0 || MODE == 2)
        {
#pragma unroll
          for (int i = 0; i < 12; i++)
            acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64 (m, c, acc[i], 0, 0, 0);
        }
      if (MODE == 1 || MODE == 2)
        {
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 12; i++)
              asm volatile ("v_fma_f64 %0, %0, %1, %2" : "+v" (v[i]) : "v" (m), "v" (c));
        }
    }
  double s = 0;
  for (int i = 0; i < 12; i++)
    s += acc[i] + v[i];
  if (s == 12345.678)
    out[threadIdx.x] = s;
}

int
main()
{
  double *a, *b, *c, *d;
  hipMallocManaged (&a, 64 * 8); hipMallocManaged (&b, 64 * 8); hipMallocManaged (&c, 64 * 8); hipMallocManaged (&d, 64 * 8);
  // layout: one-hot A lane la, one-hot B lane lb -> which D lanes answer
  int a_of[64][4], b_of[64][4];            // for D lane l and k: the A lane and B lane whose product it sums
  memset (a_of, -1, sizeof (a_of)); memset (b_of, -1, sizeof (b_of));
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++)
      {
        for (int i = 0; i < 64; i++) { a[i] = i == la; b[i] = i == lb; c[i] = 0; }
        hipLaunchKernelGGL (one_mfma, dim3 (1), dim3 (64), 0, 0, a, b, c, d);
        hipDeviceSynchronize();
        for (int l = 0; l < 64; l++)
          if (d[l] != 0)
            for (int k = 0; k < 4; k++)
              if (a_of[l][k] < 0) { a_of[l][k] = la; b_of[l][k] = lb; break; }
      }
  for (int l = 0; l < 64; l += 1)
    if (l < 20 || l % 16 == 0)
      printf ("D lane %2d = sum of A lanes {%d %d %d %d} x B lanes {%d %d %d %d}\n", l, a_of[l][0], a_of[l][1], a_of[l][2], a_of[l][3],
              b_of[l][0], b_of[l][1], b_of[l][2], b_of[l][3]);
  // order of the additions: random values, compare with the sequential chain from c in the discovered k order (ascending A lane)
  std::mt19937_64 rng (1);
  std::uniform_real_distribution<double> u (-1, 1);
  long same_seq = 0, same_rev = 0, same_sumfirst = 0, total = 0;
  for (int rep = 0; rep < 2000; rep++)
    {
      for (int i = 0; i < 64; i++) { a[i] = u (rng); b[i] = u (rng); c[i] = u (rng) * (rep % 3 == 0 ? 1e6 : 1); }
      hipLaunchKernelGGL (one_mfma, dim3 (1), dim3 (64), 0, 0, a, b, c, d);
      hipDeviceSynchronize();
      for (int l = 0; l < 64; l++)
        {
          double s = c[l];
          for (int k = 0; k < 4; k++) s = fma (a[a_of[l][k]], b[b_of[l][k]], s);
          double r = c[l];
          for (int k = 3; k >= 0; k--) r = fma (a[a_of[l][k]], b[b_of[l][k]], r);
          double p = 0;
          for (int k = 0; k < 4; k++) p = fma (a[a_of[l][k]], b[b_of[l][k]], p);
          p += c[l];
          same_seq += s == d[l]; same_rev += r == d[l]; same_sumfirst += p == d[l]; total++;
        }
    }
  printf ("of %ld results: equal to the chain from c, k ascending %ld | k descending %ld | products first, c last %ld\n", total, same_seq, same_rev, same_sumfirst);
  // rate
  double *out; hipMalloc (&out, 4096);
  const int iters = 2000;
  auto time = [&] (auto kernel, const char *name, double mfma_per_iter, double fma_per_iter) {
    for (int wps : {1, 2, 3, 4})
      {
        hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
        hipLaunchKernelGGL (kernel, dim3 (256 * wps), dim3 (256), 0, 0, out, 1.0, 10);
        hipEventRecord (e0);
        hipLaunchKernelGGL (kernel, dim3 (256 * wps), dim3 (256), 0, 0, out, 1.0, iters);
        hipEventRecord (e1); hipEventSynchronize (e1);
        float ms; hipEventElapsedTime (&ms, e0, e1);
        const double cycles = ms * 1e-3 * 2.4e9 / iters / wps;      // per wave and iteration, at 2.4 GHz nominal
        printf ("%-28s %d wave(s) per SIMD: %.3f ms, %.1f cycles per iteration and wave (%.0f mfma + %.0f v_fma_f64)\n", name, wps, ms, cycles, mfma_per_iter, fma_per_iter);
      }
  };
  time (rate_kernel<0>, "12 mfma_f64_4x4x4", 12, 0);
  time (rate_kernel<1>, "48 v_fma_f64", 0, 48);
  time (rate_kernel<2>, "12 mfma + 48 v_fma_f64", 12, 48);
  return 0;
}
