// tools/valu_rate.hip -- issue rate of plain and packed FP32 VALU instructions on gfx950 (wave64): which bound do the FFT kernels face?
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate && tools/valu_rate
// Every kernel runs ITER x 32 independent instructions per wave (16 accumulators, two rounds); waves per SIMD from the grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
constexpr int ITER = 4096;

template<int OP> __global__ void __launch_bounds__ (256)
rate_kernel (float *out, float seed)
{
  float a[16];
  v2f   p[16];
  const float m = seed * 0.999f, c = seed * 1e-3f;
  const v2f pm = {m, m}, pc = {c, c};
  for (int i = 0; i < 16; i++)
    {
      a[i] = seed + i + threadIdx.x;
      p[i] = (v2f) {a[i], a[i] + 0.5f};
    }
  for (int it = 0; it < ITER; it++)
    {
#define FMA(i)   asm volatile ("v_fma_f32 %0, %0, %1, %2" : "+v" (a[i]) : "v" (m), "v" (c));
#define ADD(i)   asm volatile ("v_add_f32 %0, %0, %1" : "+v" (a[i]) : "v" (c));
#define MUL(i)   asm volatile ("v_mul_f32 %0, %0, %1" : "+v" (a[i]) : "v" (m));
#define PKFMA(i) asm volatile ("v_pk_fma_f32 %0, %0, %1, %2" : "+v" (p[i]) : "v" (pm), "v" (pc));
#define PKADD(i) asm volatile ("v_pk_add_f32 %0, %0, %1" : "+v" (p[i]) : "v" (pc));
#define PKMUL(i) asm volatile ("v_pk_mul_f32 %0, %0, %1" : "+v" (p[i]) : "v" (pm));
#define PKADDSEL(i) asm volatile ("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v" (p[i]) : "v" (pc));
#define DPPWAVE(i) asm volatile ("v_add_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v" (a[i]) : "v" (a[(i + 1) & 15]));
#define DPPROW(i)  asm volatile ("v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v" (a[i]) : "v" (a[(i + 1) & 15]));
#define DPPQUAD(i) asm volatile ("v_add_f32_dpp %0, %1, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v" (a[i]) : "v" (a[(i + 1) & 15]));
#define F64FMA(i) asm volatile ("v_fma_f64 %0, %0, %1, %2" : "+v" (d[i]) : "v" (dm), "v" (dc));
      if (OP == 0) { R16 (FMA) R16 (FMA) }
      if (OP == 1) { R16 (ADD) R16 (ADD) }
      if (OP == 2) { R16 (MUL) R16 (MUL) }
      if (OP == 3) { R16 (PKFMA) R16 (PKFMA) }
      if (OP == 4) { R16 (PKADD) R16 (PKADD) }
      if (OP == 5) { R16 (PKMUL) R16 (PKMUL) }
      if (OP == 6) { R16 (PKADDSEL) R16 (PKADDSEL) }
      if (OP == 7) { R16 (FMA) R16 (PKFMA) }
      if (OP == 8) { R16 (DPPWAVE) R16 (DPPWAVE) }
      if (OP == 9) { R16 (DPPROW) R16 (DPPROW) }
      if (OP == 10) { R16 (DPPQUAD) R16 (DPPQUAD) }
    }
  float s = 0;
  for (int i = 0; i < 16; i++)
    s += a[i] + p[i].x + p[i].y;
  if (s == 12345.678f)
    out[threadIdx.x] = s;
}

template<int OP> __global__ void __launch_bounds__ (256)
rate_kernel_d (float *out, float seed)
{
  double d[16];
  const double dm = seed * 0.999, dc = seed * 1e-3;
  for (int i = 0; i < 16; i++)
    d[i] = seed + i + threadIdx.x;
  for (int it = 0; it < ITER; it++)
    {
      if (OP == 0) { R16 (F64FMA) R16 (F64FMA) }
    }
  double s = 0;
  for (int i = 0; i < 16; i++)
    s += d[i];
  if (s == 12345.678)
    out[threadIdx.x] = float (s);
}

template<class K> static void
run (const char *name, K kernel, int flops_per_lane_instr, float *out)
{
  for (int wps : {1, 2, 4, 8})
    {
      const int blocks = 256 * wps;               // 256 threads = one wave per SIMD of a CU
      hipEvent_t e0, e1;
      hipEventCreate (&e0); hipEventCreate (&e1);
      hipLaunchKernelGGL (kernel, dim3 (blocks), dim3 (256), 0, 0, out, 1.0f);
      hipDeviceSynchronize();
      hipEventRecord (e0);
      for (int r = 0; r < 5; r++)
        hipLaunchKernelGGL (kernel, dim3 (blocks), dim3 (256), 0, 0, out, 1.0f);
      hipEventRecord (e1);
      hipEventSynchronize (e1);
      float ms;
      hipEventElapsedTime (&ms, e0, e1);
      ms /= 5;
      const double instr_per_simd = double (ITER) * 32 * wps;
      const double ns_per_instr = ms * 1e6 / instr_per_simd;
      const double tflops = double (ITER) * 32 * blocks * 4 * 64 * flops_per_lane_instr / (ms * 1e-3) / 1e12;
      printf ("%-34s %d waves/SIMD  %7.3f ms  %6.3f ns per wave-instruction and SIMD (%4.2f cycles at 2.4 GHz)  %7.1f TFLOP/s\n",
              name, wps, ms, ns_per_instr, ns_per_instr * 2.4, tflops);
    }
}

int
main()
{
  float *out;
  hipMalloc (&out, 4096);
  run ("v_fma_f32", rate_kernel<0>, 2, out);
  run ("v_add_f32", rate_kernel<1>, 1, out);
  run ("v_mul_f32", rate_kernel<2>, 1, out);
  run ("v_pk_fma_f32", rate_kernel<3>, 4, out);
  run ("v_pk_add_f32", rate_kernel<4>, 2, out);
  run ("v_pk_mul_f32", rate_kernel<5>, 2, out);
  run ("v_pk_add_f32 op_sel neg_hi", rate_kernel<6>, 2, out);
  run ("v_fma_f32 + v_pk_fma_f32 (1:1)", rate_kernel<7>, 3, out);
  run ("v_add_f32_dpp wave_shl:1", rate_kernel<8>, 1, out);
  run ("v_add_f32_dpp row_shl:1", rate_kernel<9>, 1, out);
  run ("v_add_f32_dpp quad_perm", rate_kernel<10>, 1, out);
  run ("v_fma_f64", rate_kernel_d<0>, 2, out);
  return 0;
}
