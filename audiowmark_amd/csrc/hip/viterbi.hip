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

// generator polynomials (reference convcode.cc:42-46), A and B interleaved
__device__ constexpr unsigned V_GEN_AB[12] = { 066561, 075211, 071545, 054435, 063635, 052475, 063543, 075307, 052547, 045627, 067657, 051757 };

// BT: 0 = A block (even generators), 1 = B block (odd generators), 2 = AB (all twelve)
template<int BT> __device__ constexpr unsigned v_gen (int g) { return BT == 2 ? V_GEN_AB[g] : V_GEN_AB[2 * g + BT]; }
__device__ constexpr unsigned v_parity (unsigned v) { v ^= v >> 16; v ^= v >> 8; v ^= v >> 4; v ^= v >> 2; v ^= v >> 1; return v & 1; }

typedef float v2f __attribute__ ((ext_vector_type (2)));

// The successor state owned by (thread t, butterfly i, bit b) is ns = b | t << 1 | i << 11, so its expected code bit
// for generator g splits into a lane dependent part parity ((t << 1) & G) and a part that is a compile time constant
// after unrolling: parity ((i << 11 | b) & G).  Per trellis step a lane therefore needs just two candidate costs per
// generator (own parity / flipped parity); which one a given (i, b) takes is decided by the compiler.
template<int BT> __device__ __forceinline__ void
viterbi_body (const float *soft, int n_steps, unsigned int *decisions, int *bits_out, float *error_out, long long blk,
              float *s_metric, float *s_e0, float *s_e1)
{
  constexpr int rate = BT == 2 ? 12 : 6;
  const int t = threadIdx.x;
  const float *coded = soft + blk * (long long) n_steps * rate;
  unsigned int *dec = decisions + blk * (long long) n_steps * V_THREADS;

  for (int i = t; i < V_STATES; i += V_THREADS)
    s_metric[i] = i == 0 ? 0.f : -1.f;
  unsigned int lane_parity = 0;                     // bit g: parity ((t << 1) & G_g)
#pragma unroll
  for (int g = 0; g < rate; g++)
    lane_parity |= (unsigned (__popc ((unsigned (t) << 1) & v_gen<BT> (g)) & 1)) << g;
  __syncthreads();

  // After V_ORDER steps every state is reachable, and for finite input all path metrics are >= 0 from then on: the
  // reachability tests (metric >= 0, which a NaN also fails -- the reference then skips that predecessor) only matter
  // in the first V_ORDER steps or for NaN input.  normalize_soft_bits turns ALL bits of a block into NaN or none
  // (0 / 0 mean), so one look at the first value decides whether the plain compare-select may be used.
  const bool plain_ok = coded[0] == coded[0];
  for (int step = 0; step < n_steps; step++)
    {
      const bool plain = plain_ok && step >= V_ORDER;     // uniform
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
      float cost_same[rate], cost_flip[rate];  // branch cost if the constant part of the parity is 0 / 1
#pragma unroll
      for (int g = 0; g < rate; g++)
        {
          const float e0 = s_e0[g], e1 = s_e1[g];
          const bool lp = (lane_parity >> g) & 1;
          cost_same[g] = lp ? e1 : e0;
          cost_flip[g] = lp ? e0 : e1;
        }
      unsigned int word = 0;
#pragma unroll
      for (int i = 0; i < V_PER_THREAD; i++)
        {
          const unsigned p = t + V_THREADS * i;
          // running sums for predecessors {low, high}: .x / .y; one pair per successor bit; term by term like the reference
          v2f d0 = { old0[i], old1[i] }, d1 = d0;
#pragma unroll
          for (int g = 0; g < rate; g++)
            {
              constexpr unsigned G = 0;   // placeholder to keep the loop body uniform
              (void) G;
              const bool c0 = v_parity ((unsigned (i) << 11) & v_gen<BT> (g));
              const bool c1 = v_parity (((unsigned (i) << 11) | 1u) & v_gen<BT> (g));
              const float ea = c0 ? cost_flip[g] : cost_same[g];
              const float eb = c1 ? cost_flip[g] : cost_same[g];
              d0 += (v2f) { ea, ea };
              d1 += (v2f) { eb, eb };
            }
          unsigned c0, c1;
          float best0, best1;
          if (plain)
            {
              // strict "<": the low predecessor is visited first by the reference and keeps ties
              c0 = d0.y < d0.x;
              c1 = d1.y < d1.x;
              best0 = c0 ? d0.y : d0.x;
              best1 = c1 ? d1.y : d1.x;
            }
          else
            {
              const bool r0 = old0[i] >= 0.f, r1 = old1[i] >= 0.f;
              c0 = r0 ? (r1 && d0.y < d0.x) : (r1 ? 1u : 0u);
              c1 = r0 ? (r1 && d1.y < d1.x) : (r1 ? 1u : 0u);
              best0 = c0 ? d0.y : (r0 ? d0.x : -1.f);
              best1 = c1 ? d1.y : (r0 ? d1.x : -1.f);
            }
          word |= (c0 << (2 * i)) | (c1 << (2 * i + 1));
          reinterpret_cast<float2 *> (s_metric)[p] = make_float2 (best0, best1);
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

// one launch for all three code types: blocks [0, n0) decode A blocks, [n0, n0 + n1) B blocks, the rest AB blocks
struct ViterbiBatch
{
  const float  *soft[3];
  unsigned int *decisions[3];
  int          *bits[3];
  float        *error[3];
  int           n[3];
  int           n_steps;
};

__global__ void __launch_bounds__ (V_THREADS)
viterbi_kernel (ViterbiBatch b)
{
  extern __shared__ __attribute__ ((aligned (16))) float s_metric[];   // V_STATES floats
  __shared__ float s_e0[12], s_e1[12];
  int blk = blockIdx.x;
  if (blk < b.n[0])
    viterbi_body<0> (b.soft[0], b.n_steps, b.decisions[0], b.bits[0], b.error[0], blk, s_metric, s_e0, s_e1);
  else if (blk < b.n[0] + b.n[1])
    viterbi_body<1> (b.soft[1], b.n_steps, b.decisions[1], b.bits[1], b.error[1], blk - b.n[0], s_metric, s_e0, s_e1);
  else
    viterbi_body<2> (b.soft[2], b.n_steps, b.decisions[2], b.bits[2], b.error[2], blk - b.n[0] - b.n[1], s_metric, s_e0, s_e1);
}

size_t
viterbi_workspace_bytes (long long coded_len, int rate, long long n_blocks)
{
  return size_t (coded_len / rate) * V_THREADS * sizeof (unsigned int) * size_t (n_blocks);
}

hipError_t
launch_viterbi (hipStream_t st, const float *const soft[3], const long long n_blocks[3], long long n_steps,
                unsigned char *const decisions_ws[3], int *const bits_out[3], float *const error_out[3])
{
  const long long total = n_blocks[0] + n_blocks[1] + n_blocks[2];
  if (total <= 0)
    return hipSuccess;
  ViterbiBatch b;
  for (int i = 0; i < 3; i++)
    {
      b.soft[i] = soft[i];
      b.decisions[i] = reinterpret_cast<unsigned int *> (decisions_ws[i]);
      b.bits[i] = bits_out[i];
      b.error[i] = error_out[i];
      b.n[i] = int (n_blocks[i]);
    }
  b.n_steps = int (n_steps);
  const size_t lds = V_STATES * sizeof (float);
  hipError_t e = hipFuncSetAttribute (reinterpret_cast<const void *> (viterbi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int (lds));
  if (e != hipSuccess)
    return e;
  hipLaunchKernelGGL (viterbi_kernel, dim3 ((unsigned) total), dim3 (V_THREADS), lds, st, b);
  return hipGetLastError();
}

} // namespace awmk
