// viterbi.hip -- K8: soft-decision Viterbi decoder for the K=15 convolutional code on gfx950.
// Replaces conv_decode_soft (reference src/convcode.cc:128-213).
//
// One 1024-thread workgroup decodes one coded block.  The 2^15 path metrics live in LDS
// (128 KiB of the CU's 160 KiB); per trellis step every thread updates 16 butterflies
//   { old[p], old[p + 2^14] } -> { new[2p], new[2p + 1] }
// reading the old metrics with unit stride across lanes and writing the new ones as float2 with
// unit stride (no bank conflicts), so the metrics never leave the CU.  Survivor decisions
// (1 bit per state and step, 4 KiB per step) go to an HBM workspace and are walked back by one
// lane at the end.
//
// Bit-exactness with the reference: the path metric of a transition is accumulated term by term
//   delta = old; for p in 0..rate-1: delta += (cbit[p] - sbit[p])^2      (float, no FMA)
// for BOTH predecessors, and the second predecessor (the one with the top state bit set, which the
// reference visits later) wins only on a strict "<"; unreachable states carry -1 exactly like the
// reference's StateEntry::delta.
#include "kernels.hh"

namespace awmk {

constexpr int V_ORDER = 15;
constexpr int V_STATES = 1 << V_ORDER;       // 32768
constexpr int V_THREADS = 1024;
constexpr int V_PER_THREAD = V_STATES / 2 / V_THREADS;   // 16 butterflies

struct ViterbiGen { unsigned g[12]; };

template<int RATE> __global__ void __launch_bounds__ (V_THREADS)
viterbi_kernel (const float *soft, ViterbiGen gen, int n_steps, unsigned int *decisions, int *bits_out, float *error_out)
{
  extern __shared__ __attribute__ ((aligned (16))) float s_metric[];   // V_STATES floats
  __shared__ float s_e0[12], s_e1[12];
  constexpr int rate = RATE;
  const int t = threadIdx.x;
  const long long blk = blockIdx.x;
  const float *coded = soft + blk * (long long) n_steps * rate;
  unsigned int *dec = decisions + blk * (long long) n_steps * V_THREADS;

  for (int i = t; i < V_STATES; i += V_THREADS)
    s_metric[i] = i == 0 ? 0.f : -1.f;
  __syncthreads();

  for (int step = 0; step < n_steps; step++)
    {
      if (t < rate)
        {
          const float c = coded[step * rate + t];
          const float d1 = __fsub_rn (c, 1.0f);
          s_e0[t] = __fmul_rn (c, c);          // (cbit - 0)^2
          s_e1[t] = __fmul_rn (d1, d1);        // (cbit - 1)^2
        }
      float old0[V_PER_THREAD], old1[V_PER_THREAD];
#pragma unroll
      for (int i = 0; i < V_PER_THREAD; i++)
        {
          old0[i] = s_metric[t + V_THREADS * i];
          old1[i] = s_metric[t + V_THREADS * i + V_STATES / 2];
        }
      __syncthreads();                         // all reads done (and s_e0/s_e1 visible)
      unsigned int word = 0;
#pragma unroll
      for (int i = 0; i < V_PER_THREAD; i++)
        {
          const unsigned p = t + V_THREADS * i;
          float out[2];
#pragma unroll
          for (int b = 0; b < 2; b++)
            {
              const unsigned ns = 2 * p + b;
              float d0 = old0[i], d1 = old1[i];
#pragma unroll
              for (int g = 0; g < rate; g++)
                {
                  const float e = (__popc (ns & gen.g[g]) & 1) ? s_e1[g] : s_e0[g];
                  d0 = __fadd_rn (d0, e);
                  d1 = __fadd_rn (d1, e);
                }
              const bool r0 = old0[i] >= 0.f, r1 = old1[i] >= 0.f;
              float best;
              unsigned choose1;
              if (r0 && r1)
                {
                  choose1 = d1 < d0;
                  best = choose1 ? d1 : d0;
                }
              else if (r1)
                {
                  choose1 = 1;
                  best = d1;
                }
              else
                {
                  choose1 = 0;
                  best = r0 ? d0 : -1.f;
                }
              out[b] = best;
              word |= choose1 << (2 * i + b);
            }
          reinterpret_cast<float2 *> (s_metric)[p] = make_float2 (out[0], out[1]);
        }
      dec[(long long) step * V_THREADS + t] = word;
      __syncthreads();
    }

  __threadfence();
  __syncthreads();
  if (t == 0)
    {
      error_out[blk] = s_metric[0] / float (n_steps * rate);
      unsigned state = 0;
      int *bits = bits_out + blk * (long long) (n_steps - V_ORDER);
      for (int step = n_steps - 1; step >= 0; step--)
        {
          if (step < n_steps - V_ORDER)
            bits[step] = state & 1;
          const unsigned p = state >> 1;                       // butterfly index
          const unsigned int word = __builtin_nontemporal_load (&dec[(long long) step * V_THREADS + (p & (V_THREADS - 1))]);
          const unsigned choose1 = (word >> (2 * (p >> 10) + (state & 1))) & 1;
          state = p | (choose1 << (V_ORDER - 1));
        }
    }
}

size_t
viterbi_workspace_bytes (long long coded_len, int rate, long long n_blocks)
{
  return size_t (coded_len / rate) * V_THREADS * sizeof (unsigned int) * size_t (n_blocks);
}

hipError_t
launch_viterbi (hipStream_t st, const float *soft, int rate, const unsigned *generators,
                long long coded_len, long long n_blocks, unsigned char *decisions_ws, int *bits_out, float *error_out)
{
  if (n_blocks <= 0)
    return hipSuccess;
  if ((rate != 6 && rate != 12) || coded_len % rate)
    return hipErrorInvalidValue;
  ViterbiGen gen;
  for (int i = 0; i < 12; i++)
    gen.g[i] = i < rate ? generators[i] : 0;
  const size_t lds = V_STATES * sizeof (float);
  hipError_t e = hipFuncSetAttribute (reinterpret_cast<const void *> (viterbi_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, int (lds));
  if (e == hipSuccess)
    e = hipFuncSetAttribute (reinterpret_cast<const void *> (viterbi_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, int (lds));
  if (e != hipSuccess)
    return e;
  unsigned int *dec = reinterpret_cast<unsigned int *> (decisions_ws);
  if (rate == 6)
    hipLaunchKernelGGL (viterbi_kernel<6>, dim3 (unsigned (n_blocks)), dim3 (V_THREADS), lds, st, soft, gen, int (coded_len / rate), dec, bits_out, error_out);
  else
    hipLaunchKernelGGL (viterbi_kernel<12>, dim3 (unsigned (n_blocks)), dim3 (V_THREADS), lds, st, soft, gen, int (coded_len / rate), dec, bits_out, error_out);
  return hipGetLastError();
}

} // namespace awmk
