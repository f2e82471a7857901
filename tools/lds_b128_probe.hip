// ds_read_b128 from addresses that are not 16-byte aligned: does gfx950 deliver them (unaligned LDS access mode), are the values right, and
// at what rate?  (K5w reads aligned quads and repairs the alignment with DPP moves folded into its additions; an unaligned quad per lane
// would need neither.)   hipcc -O2 --offload-arch=gfx950 -o tools/lds_b128_probe tools/lds_b128_probe.hip && tools/lds_b128_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float4v __attribute__ ((ext_vector_type (4)));
__global__ void __launch_bounds__ (832)
probe (int off_bytes, int iters, float *out, long long *cycles, int check)
{
  __shared__ __attribute__ ((aligned (16))) float s[36 * 1024];
  for (int i = threadIdx.x; i < 36 * 1024; i += blockDim.x)
    s[i] = float (i);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned base = (unsigned) (uintptr_t) (__attribute__ ((address_space (3))) float *) s + wave * 8192 + lane * 16 + off_bytes;
  float4v acc = { 0, 0, 0, 0 };
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; it++)
    {
      float4v v[8];
#pragma unroll
      for (int k = 0; k < 8; k++)
        asm volatile ("ds_read_b128 %0, %1" : "=v" (v[k]) : "v" (base + k * 1024 + (it & 3) * 256));
      asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; k++)
        acc += v[k];
    }
  const long long t1 = wall_clock64();
  if (check)
    {
      float4v v;
      asm volatile ("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v" (v) : "v" (base) : "memory");
      acc = v;
    }
  float *o = out + ((size_t) blockIdx.x * blockDim.x + threadIdx.x) * 4;
  o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w;
  if (threadIdx.x == 0)
    cycles[blockIdx.x] = t1 - t0;
}
int main()
{
  const int wgs = 256, threads = 832, iters = 20000;
  float *out; long long *cyc;
  hipMalloc (&out, sizeof (float) * 4 * wgs * threads);
  hipMalloc (&cyc, sizeof (long long) * wgs);
  std::vector<float> h (4 * wgs * threads);
  std::vector<long long> hc (wgs);
  for (int off : { 0, 4, 8, 12, 0 })
    {
      hipLaunchKernelGGL (probe, dim3 (wgs), dim3 (threads), 0, 0, off, 10, out, cyc, 1);
      hipMemcpy (h.data(), out, sizeof (float) * h.size(), hipMemcpyDeviceToHost);
      bool ok = true;
      for (int t = 0; t < threads && ok; t++)
        for (int e = 0; e < 4; e++)
          ok = ok && h[4 * t + e] == float ((t >> 6) * 2048 + (t & 63) * 4 + off / 4 + e);
      hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
      hipEventRecord (e0);
      hipLaunchKernelGGL (probe, dim3 (wgs), dim3 (threads), 0, 0, off, iters, out, cyc, 0);
      hipEventRecord (e1); hipEventSynchronize (e1);
      float ms; hipEventElapsedTime (&ms, e0, e1);
      const double reads = double (wgs) * (threads / 64) * iters * 8;             // wave instructions
      printf ("offset %2d bytes: values %s, %.3f ms, %.2f ns per ds_read_b128 and compute unit = %.1f TB/s over 256 units\n", off, ok ? "right" : "WRONG", ms,
              ms * 1e6 / (reads / wgs), reads * 1024 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
