// viterbi.hip -- K8: soft-decision Viterbi decoder for the K = 15 convolutional code on gfx950.
// Replaces conv_decode_soft (reference src/convcode.cc:128-213).
//
// What the reference fixes: the path metric of a transition is accumulated term by term in float,
//   delta = old; for p in 0..rate-1: delta += (cbit[p] - sbit[p])^2
// for both predecessors of a state; the predecessor with the top state bit set (visited later) wins only on a strict "<";
// unreachable states carry -1.  So a decode costs 2 x 2^15 x rate sequential float additions per trellis step (143 steps:
// 56 M for an A or B block, 112 M for an AB block) whatever the implementation -- on ONE compute unit that is 0.4 / 0.8 ms
// (the round-1 kernel, one workgroup per decode), and a chunk has only ~50 decodes for 256 compute units.
//
// This version spreads every decode over the whole chip.  The shift-register trellis closes over short stretches: the 2^K
// states that share their low 15 - K bits at step T determine exactly the 2^K CONSECUTIVE states (those low bits shifted up)
// at step T + K, through a private K-step butterfly network.  So a lane loads such a group (16 metrics, stride 2048, for
// K = 4), runs 4 trellis steps entirely in registers -- no LDS, no barrier -- and stores 16 consecutive metrics and one
// decision word per step.  A ROUND (4 steps of all blocks of the batch) is one launch of 2048 lanes per block; the kernel
// boundary is the exchange.  143 steps = 35 rounds of 4 + 1 round of 3.  The expected code bit of a successor state splits
// into a lane part (parity of the group bits, per step) and a part that is a compile time constant after unrolling.
// (K = 5: 29 rounds of 10.6 us for eight AB decodes alone on the GPU, 3 600 straight-line instructions per lane; K = 4: 36
// rounds of about half that with twice the lanes -- 1.33 -> 1.25 ms per step of the 60 min bench.)
//
// One chain instead of two (rounds after step 15, finite input).  The code bits of a transition depend on the successor state
// only, so both predecessors add the SAME terms, and float addition is monotone: the chain that starts from min (old0, old1)
// ends at the new metric, whichever predecessor wins.  The decision is old1 < old0 unless rounding merges the two chains (the
// low predecessor then keeps the tie); chains that start further apart than 16 ulp of the largest possible sum cannot merge, a
// lane that is closer makes its wave repeat that step's decisions with both chains (a few percent of the wave-steps).  The
// additions are v_pk_add_f32 with op_sel picking the halves of ONE register pair { cost if the bit is as expected, cost if
// flipped } per generator: 768 -> 384 packed additions per lane and round for an AB block, 256 -> 82 registers.  Measured: no
// change for the 60 min bench (36 dependent launches of ~11 us each: launch bound), 0.273 -> 0.255 ms per clip in the clip batch
// (320 AB decodes per launch: throughput bound).  Tried and dropped: K = 5 with the lighter step (29 rounds, 125 registers:
// same time), one workgroup per block with the metrics in LDS (no launches per round, but 16 waves per decode on one compute
// unit: 1 ms per AB block, 2.5 x slower for both workloads).
//
// Survivors: lane L's decision words of a round describe the whole K-step history of the 2^K states it produced, so the
// trace back needs ONE 16-byte load per round (36 dependent loads instead of 143).
//
// Bit-exactness: sums and ties exactly as above; decoded bits and the error value are bit-identical to the oracle.
#include "kernels.hh"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <type_traits>
#include <utility>

namespace awmk {

namespace {

constexpr int V_ORDER = 15;
constexpr int V_STATES = 1 << V_ORDER;       // 32768
constexpr int V_WG = 256;

// generator polynomials (reference convcode.cc:42-46), A and B interleaved
__device__ constexpr unsigned V_GEN_AB[12] = { 066561, 075211, 071545, 054435, 063635, 052475, 063543, 075307, 052547, 045627, 067657, 051757 };

// BT: 0 = A block (even generators), 1 = B block (odd generators), 2 = AB (all twelve)
template<int BT> __device__ constexpr unsigned v_gen (int g) { return BT == 2 ? V_GEN_AB[g] : V_GEN_AB[2 * g + BT]; }
__device__ constexpr unsigned v_parity (unsigned v) { v ^= v >> 16; v ^= v >> 8; v ^= v >> 4; v ^= v >> 2; v ^= v >> 1; return v & 1; }

typedef float v2f __attribute__ ((ext_vector_type (2)));

/* d + { pair[a], pair[b] } in one v_pk_add_f32: the operand halves are picked by op_sel / op_sel_hi, so ONE register pair
 * { cost if the code bit is as the lane expects, cost if it is flipped } per generator serves every successor state (the
 * compiler builds a register pair per combination instead: 4 x 12 pairs for an AB block, and spills) */
__device__ __forceinline__ v2f
pk_add_sel (v2f d, v2f pair, bool a, bool b)
{
  v2f r;
  if (!a && !b)
    asm ("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v" (r) : "v" (d), "v" (pair));
  else if (a && !b)
    asm ("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v" (r) : "v" (d), "v" (pair));
  else if (!a && b)
    asm ("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,1]" : "=v" (r) : "v" (d), "v" (pair));
  else
    asm ("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v" (r) : "v" (d), "v" (pair));
  return r;
}

/* Device-coherent accesses (relaxed atomics at agent scope: global_load / global_store with sc1).  The one-launch kernel's workgroups
 * of a decode sit on different XCDs, whose L2s do not see each other's plain stores; an agent-scope fence fixes that by writing back
 * and invalidating the WHOLE L2 (buffer_wbl2 / buffer_inv sc1) -- with one fence pair per meeting the kernel took 2.3 x the time of the
 * 16-launch chain (2.13 against 0.92 ms per bench step).  So everything that crosses workgroups -- path metrics, decision words --
 * is moved with these instead and no cache is ever flushed; COH = false compiles to the plain accesses of the launch chain. */
typedef __attribute__ ((address_space (1))) float               *gfloat_ptr;       // (global, not flat: the workspace pointers come out of
typedef __attribute__ ((address_space (1))) unsigned long long  *gu64_ptr;         //  a runtime-indexed array and the compiler loses the space)
template<bool COH> __device__ __forceinline__ float
ld_metric (const float *p)
{
  if (COH)
    return __hip_atomic_load ((gfloat_ptr) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template<bool COH> __device__ __forceinline__ void
st_pair (float *p, float a, float b)                       // 8-byte aligned
{
  if (COH)
    {
      const unsigned long long v = (unsigned long long) __float_as_uint (a) | ((unsigned long long) __float_as_uint (b) << 32);
      __hip_atomic_store ((gu64_ptr) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  else
    *reinterpret_cast<float2 *> (p) = make_float2 (a, b);
}
template<bool COH> __device__ __forceinline__ void
st_quad (float *p, float a, float b, float c, float d)     // 16-byte aligned
{
  if (COH)
    {
      st_pair<true> (p, a, b);
      st_pair<true> (p + 2, c, d);
    }
  else
    *reinterpret_cast<float4 *> (p) = make_float4 (a, b, c, d);
}
template<bool COH> __device__ __forceinline__ void
st_words (unsigned int *p, unsigned int a, unsigned int b, unsigned int c, unsigned int d)     // 16-byte aligned
{
  if (COH)
    {
      __hip_atomic_store ((gu64_ptr) p, (unsigned long long) a | ((unsigned long long) b << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store ((gu64_ptr) p + 1, (unsigned long long) c | ((unsigned long long) d << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  else
    *reinterpret_cast<uint4 *> (p) = make_uint4 (a, b, c, d);
}
template<bool COH> __device__ __forceinline__ uint4
ld_words (const unsigned int *p)
{
  if (COH)
    {
      const unsigned long long lo = __hip_atomic_load ((gu64_ptr) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long hi = __hip_atomic_load ((gu64_ptr) p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return make_uint4 ((unsigned int) lo, (unsigned int) (lo >> 32), (unsigned int) hi, (unsigned int) (hi >> 32));
    }
  return *reinterpret_cast<const uint4 *> (p);
}

/* workspace of one block: [metrics A][metrics B][decision words of all rounds] */
constexpr size_t V_METRIC_BYTES = V_STATES * sizeof (float);

struct RoundPlan           // host side description of one launch
{
  int k;                   // trellis steps of the round (5 or 3)
  int step0;               // first step
  size_t dec_offset;       // words, from the start of the block's decision area
};

/* words per lane and round in the decision area (a power of two, so that a lane's words are one aligned load) */
constexpr int dec_words (int k) { return k <= 4 ? 4 : 8; }

template<class F, int... Is> __device__ __forceinline__ void
for_each_step (F& f, std::integer_sequence<int, Is...>)
{
  (f (std::integral_constant<int, Is + 1>()), ...);
}

/* K trellis steps in registers: m = metrics of the 2^K states { L + j 2^(15-K) } on entry, of the 2^K states { L 2^K + loc } on
 * return; words[i] = decisions of step i (bit loc: the predecessor with the top state bit set won) */
template<int BT, int K, bool PLAIN> __device__ __forceinline__ void
viterbi_steps (const float *coded, int step0, float (&m)[1 << K], unsigned int (&words)[K], int L)
{
  constexpr int rate = BT == 2 ? 12 : 6;
  constexpr int NS = 1 << K;                               // states per lane
  // one trellis step; the step number within the round is a compile time constant (the loop form is not unrolled by the compiler
  // once the additions are inline assembly, and everything below depends on it being constant)
  auto step = [&] (auto step_number)
    {
      constexpr int i = decltype (step_number)::value;
      // squared distances of this step's code bits to 0 and 1 (wave uniform), and which of them a lane's states take:
      // expected bit = parity ((L << i) & G)  ^  compile time part
      v2f cost[rate];                                          // .x: code bit as the lane part predicts, .y: flipped
      float cost_same[rate], cost_flip[rate];
#pragma unroll
      for (int g = 0; g < rate; g++)
        {
          const float c = coded[(step0 + i - 1) * rate + g];
          const float d1 = __fsub_rn (c, 1.0f);
          const float e0 = __fmul_rn (c, c), e1 = __fmul_rn (d1, d1);        // (cbit - 0)^2, (cbit - 1)^2
          const bool lp = __popc ((unsigned (L) << i) & v_gen<BT> (g)) & 1;
          cost_same[g] = lp ? e1 : e0;
          cost_flip[g] = lp ? e0 : e1;
          cost[g] = (v2f) { cost_same[g], cost_flip[g] };
        }
      float n[NS];
      unsigned int word = 0;
      // expected code bits of the successors loc = 2 jp + b: the state  (L << i) | (loc >> i) << (15 - K + i) | (loc & (2^i - 1)),
      // lane part above, compile time part here
      constexpr int hi_shift = V_ORDER - K;
      auto flips = [] (int loc, int ii, int g) {
        const unsigned st = (unsigned (loc >> ii) << (hi_shift + ii)) | unsigned (loc & ((1 << ii) - 1));
        return bool (v_parity (st & v_gen<BT> (g)));
      };
      if (PLAIN)
        {
          // Both predecessors of a successor add the SAME terms (the code bits depend on the successor state only), and a float
          // addition is monotone: the chain that starts from the smaller metric ends at the smaller or equal sum, so the new metric
          // is the chain of min (old0, old1) -- one chain instead of two, the two successors of a pair packed into one register
          // pair.  The decision "high predecessor wins" is old1 < old0 unless rounding merges the two chains (then the low one
          // keeps the tie, convcode.cc:168-185).  Chains that start further apart than 16 ulp of the largest possible sum cannot
          // merge (each of the <= 12 additions moves them together by at most one ulp); the rare lane that is closer makes its
          // wave repeat the step's decisions with both chains.
          float sum_max = 0.f;
#pragma unroll
          for (int g = 0; g < rate; g++)
            sum_max += fmaxf (cost_same[g], cost_flip[g]);
          bool risky = false;
#pragma unroll
          for (int jp = 0; jp < NS / 2; jp++)
            {
              const float old0 = m[jp], old1 = m[jp + NS / 2];              // predecessors without / with the top state bit
              const bool high = old1 < old0;
              const float lo = high ? old1 : old0, hi = high ? old0 : old1;
              v2f d = { lo, lo };                                            // .x: successor 2 jp, .y: successor 2 jp + 1
#pragma unroll
              for (int g = 0; g < rate; g++)
                d = pk_add_sel (d, cost[g], flips (2 * jp, i, g), flips (2 * jp + 1, i, g));
              n[2 * jp] = d.x;
              n[2 * jp + 1] = d.y;
              risky = risky || (high && !(__fsub_rn (hi, lo) > __fmul_rn (0x1p-19f, __fadd_rn (hi, sum_max))));
              word |= high ? (3u << (2 * jp)) : 0u;
            }
          if (__builtin_amdgcn_ballot_w64 (risky))
            {
              word = 0;
#pragma unroll
              for (int jp = 0; jp < NS / 2; jp++)
                {
                  const float old0 = m[jp], old1 = m[jp + NS / 2];
                  v2f d0 = { old0, old1 }, d1 = d0;
#pragma unroll
                  for (int g = 0; g < rate; g++)
                    {
                      d0 = pk_add_sel (d0, cost[g], flips (2 * jp, i, g), flips (2 * jp, i, g));
                      d1 = pk_add_sel (d1, cost[g], flips (2 * jp + 1, i, g), flips (2 * jp + 1, i, g));
                    }
                  // strict "<": the low predecessor is visited first by the reference and keeps ties
                  word |= (unsigned (d0.y < d0.x) << (2 * jp)) | (unsigned (d1.y < d1.x) << (2 * jp + 1));
                }
            }
        }
      else
        {
#pragma unroll
          for (int jp = 0; jp < NS / 2; jp++)
            {
              const float old0 = m[jp], old1 = m[jp + NS / 2];              // predecessors without / with the top state bit
              // running sums for predecessors {low, high}: .x / .y; one pair per successor bit; term by term like the reference
              v2f d0 = { old0, old1 }, d1 = d0;
#pragma unroll
              for (int g = 0; g < rate; g++)
                {
                  d0 = pk_add_sel (d0, cost[g], flips (2 * jp, i, g), flips (2 * jp, i, g));
                  d1 = pk_add_sel (d1, cost[g], flips (2 * jp + 1, i, g), flips (2 * jp + 1, i, g));
                }
              // states that cannot be reached from state 0 at step 0 carry -1 (also what a NaN metric turns into a skip)
              const bool r0 = old0 >= 0.f, r1 = old1 >= 0.f;
              const unsigned c0 = r0 ? (r1 && d0.y < d0.x) : (r1 ? 1u : 0u);
              const unsigned c1 = r0 ? (r1 && d1.y < d1.x) : (r1 ? 1u : 0u);
              word |= (c0 << (2 * jp)) | (c1 << (2 * jp + 1));
              n[2 * jp] = c0 ? d0.y : (r0 ? d0.x : -1.f);
              n[2 * jp + 1] = c1 ? d1.y : (r0 ? d1.x : -1.f);
            }
        }
      words[i - 1] = word;
#pragma unroll
      for (int j = 0; j < NS; j++)
        m[j] = n[j];
    };
  for_each_step (step, std::make_integer_sequence<int, K>());
}

/* a round through global memory: lane L of the block's 2^(15-K) */
template<int BT, int K, bool PLAIN, bool COH = false> __device__ __forceinline__ void
viterbi_round (const float *coded, int step0, const float *m_in, float *m_out, unsigned int *dec, int L)
{
  constexpr int NS = 1 << K;
  constexpr int GROUPS = V_STATES >> K;                    // lanes per block, stride of a lane's states on input
  float m[NS];
#pragma unroll
  for (int j = 0; j < NS; j++)
    m[j] = ld_metric<COH> (m_in + L + j * GROUPS);
  unsigned int words[K];
  viterbi_steps<BT, K, PLAIN> (coded, step0, m, words, L);
  float *out = m_out + (size_t) L * NS;
#pragma unroll
  for (int j = 0; j < NS / 4; j++)
    st_quad<COH> (out + 4 * j, m[4 * j], m[4 * j + 1], m[4 * j + 2], m[4 * j + 3]);
  constexpr int DW = dec_words (K);
  unsigned int *d = dec + (size_t) L * DW;
  st_words<COH> (d, words[0], words[1], words[2], K > 3 ? words[K > 3 ? 3 : 0] : 0u);
  if (DW == 8)
    st_words<COH> (d + 4, words[K > 4 ? 4 : 0], 0u, 0u, 0u);
}

// one launch for all three code types: blocks [0, n0) decode A blocks, [n0, n0 + n1) B blocks, the rest AB blocks
struct ViterbiBatch
{
  const float   *soft[3];
  unsigned char *ws[3];        // per type: blocks x block_ws_bytes
  int           *bits[3];
  float         *error[3];
  int            n[3];
  int            n_steps;
  size_t         block_ws_bytes;
};

template<int K, bool PLAIN> __global__ void __launch_bounds__ (V_WG)
viterbi_round_kernel (ViterbiBatch b, int step0, int parity_in, size_t dec_offset)
{
  int blk = blockIdx.y, t = 0;
  if (blk >= b.n[0]) { blk -= b.n[0]; t = 1; }
  if (t == 1 && blk >= b.n[1]) { blk -= b.n[1]; t = 2; }
  const int rate = t == 2 ? 12 : 6;
  const int L = blockIdx.x * V_WG + threadIdx.x;
  unsigned char *ws = b.ws[t] + (size_t) blk * b.block_ws_bytes;
  const float *m_in = reinterpret_cast<const float *> (ws + (parity_in ? V_METRIC_BYTES : 0));
  float *m_out = reinterpret_cast<float *> (ws + (parity_in ? 0 : V_METRIC_BYTES));
  unsigned int *dec = reinterpret_cast<unsigned int *> (ws + 2 * V_METRIC_BYTES) + dec_offset;
  const float *coded = b.soft[t] + (size_t) blk * b.n_steps * rate;
  // normalize_soft_bits turns ALL bits of a block into NaN or none (0 / 0 mean): such a block takes the checked path, where a
  // NaN sum fails every comparison exactly like in the reference
  const bool finite = coded[0] == coded[0];
  if (t == 0)
    {
      if (PLAIN && finite) viterbi_round<0, K, true> (coded, step0, m_in, m_out, dec, L);
      else                 viterbi_round<0, K, false> (coded, step0, m_in, m_out, dec, L);
    }
  else if (t == 1)
    {
      if (PLAIN && finite) viterbi_round<1, K, true> (coded, step0, m_in, m_out, dec, L);
      else                 viterbi_round<1, K, false> (coded, step0, m_in, m_out, dec, L);
    }
  else
    {
      if (PLAIN && finite) viterbi_round<2, K, true> (coded, step0, m_in, m_out, dec, L);
      else                 viterbi_round<2, K, false> (coded, step0, m_in, m_out, dec, L);
    }
}

/* THREE rounds of 4 steps in one launch.  The closure that lets a lane run 4 steps in registers holds one level up: the 4096
 * states that share their low 3 bits at step T determine the 4096 CONSECUTIVE states (those bits shifted to the top) at step
 * T + 12.  A workgroup of 256 lanes owns such a family (8 workgroups per decode): round 1 reads its metrics from global memory
 * (stride 8), the 16 x 256 results change hands inside the workgroup through LDS -- the lanes regroup by the next 4 address bits
 * -- round 2, LDS again, round 3, and 16 consecutive metrics per lane go back to global memory.  One launch and one trip through
 * L2 per 12 trellis steps instead of three: 143 steps = 11 launches of 12 + the rounds 4, 4, 3 of the single-round kernel (the
 * chain of a batch of decodes is bound by its dependent launches: 38 -> 16).  Decision words are stored exactly where the three
 * single rounds would put them (lane = the round's 11 bit group number), so the trace back does not change.
 *   round 1  group  L1 = lane << 3 | g                          in:  M[L1 + j 2048]
 *   round 2  group  L2 = (lane >> 4) << 7 | g << 4 | lane & 15  in:  round 1's result of lane' = j << 4 | lane >> 4, state lane & 15
 *   round 3  group  L3 = g << 8 | lane                          in:  round 2's result of lane' = j << 4 | lane >> 4, state lane & 15
 * LDS layout of an exchange: value (writer lane w, state loc) at loc * 324 + (w & 15) * 20 + (w >> 4): a reader's 16 values are
 * contiguous (four ds_read_b128), writes and reads are conflict free (324 = 4 mod 64, 20 a = 16 distinct multiples of 4 mod 64). */
constexpr int SUPER_LDS = 16 * 324;

template<int BT, int NPF, bool COH = false> __device__ __forceinline__ void
viterbi_super_round (const float *coded, int step0, const float *m_in, float *m_out, unsigned int *dec, const size_t (&dec_off)[3],
                     int g, int lane, float *lds_a, float *lds_b, bool in_perm, bool out_perm, bool first = false)
{
  /* Metric layout between two of these launches: PERMUTED, state s at (s & 7) * 4096 + (s >> 3) -- the family a workgroup reads
   * (states with the low bits g) is then one contiguous 16 KB block, a wave's 64 loads one 256-byte run; in the natural layout
   * they are 32 bytes apart, and as the producers ran on other XCDs (L2s are not shared) every launch fetched its 4.7 MB of metrics
   * as 41 MB of sectors from memory (PMC FETCH_SIZE).  The 16 consecutive states a lane produces scatter over the 8 families in
   * pairs (loc, loc + 8: adjacent places of family loc & 7): eight 8-byte stores per lane, 512 contiguous bytes per wave each.
   * The first launch reads what viterbi_init wrote (state 0 at index 0 in either layout), the last one writes the natural layout
   * for the single rounds and the trace back that follow. */
  float m[16];
  const int L1 = (lane << 3) | g;
#pragma unroll
  for (int j = 0; j < 16; j++)
    m[j] = first ? ((L1 | j) == 0 ? 0.f : -1.f)              // start state 0, everything else unreachable (convcode.cc:144-146)
                 : ld_metric<COH> (in_perm ? m_in + g * 4096 + ((j << 8) | lane) : m_in + L1 + j * 2048);
  const int hi = lane >> 4, lo = lane & 15;
#pragma unroll
  for (int r = 0; r < 3; r++)
    {
      const int L = r == 0 ? L1 : r == 1 ? ((hi << 7) | (g << 4) | lo) : ((g << 8) | lane);
      unsigned int words[4];
      if (r < NPF)
        viterbi_steps<BT, 4, false> (coded, step0 + 4 * r, m, words, L);
      else
        viterbi_steps<BT, 4, true> (coded, step0 + 4 * r, m, words, L);
      st_words<COH> (dec + dec_off[r] + (size_t) L * 4, words[0], words[1], words[2], words[3]);
      if (r < 2)
        {
          float *x = r == 0 ? lds_a : lds_b;
#pragma unroll
          for (int loc = 0; loc < 16; loc++)
            x[loc * 324 + lo * 20 + hi] = m[loc];
          __syncthreads();
          const float4 *src = reinterpret_cast<const float4 *> (x + lo * 324 + hi * 20);
#pragma unroll
          for (int q = 0; q < 4; q++)
            {
              const float4 v = src[q];
              m[4 * q] = v.x; m[4 * q + 1] = v.y; m[4 * q + 2] = v.z; m[4 * q + 3] = v.w;
            }
        }
    }
  if (out_perm)
    {
#pragma unroll
      for (int f = 0; f < 8; f++)
        st_pair<COH> (m_out + f * 4096 + (g << 9) + (lane << 1), m[f], m[f + 8]);
      return;
    }
  float *out = m_out + (size_t) ((g << 8) | lane) * 16;
#pragma unroll
  for (int q = 0; q < 4; q++)
    st_quad<COH> (out + 4 * q, m[4 * q], m[4 * q + 1], m[4 * q + 2], m[4 * q + 3]);
}

struct SuperOffsets { size_t off[3]; int in_perm, out_perm, first; };     // first: start from the initial metrics without reading them

template<int NPF> __global__ void __launch_bounds__ (V_WG)
viterbi_super_kernel (ViterbiBatch b, int step0, int parity_in, SuperOffsets so)
{
  __shared__ float lds_a[SUPER_LDS], lds_b[SUPER_LDS];
  int blk = blockIdx.y, t = 0;
  if (blk >= b.n[0]) { blk -= b.n[0]; t = 1; }
  if (t == 1 && blk >= b.n[1]) { blk -= b.n[1]; t = 2; }
  const int rate = t == 2 ? 12 : 6;
  const int g = blockIdx.x, lane = threadIdx.x;
  unsigned char *ws = b.ws[t] + (size_t) blk * b.block_ws_bytes;
  const float *m_in = reinterpret_cast<const float *> (ws + (parity_in ? V_METRIC_BYTES : 0));
  float *m_out = reinterpret_cast<float *> (ws + (parity_in ? 0 : V_METRIC_BYTES));
  unsigned int *dec = reinterpret_cast<unsigned int *> (ws + 2 * V_METRIC_BYTES);
  const float *coded = b.soft[t] + (size_t) blk * b.n_steps * rate;
  const bool finite = coded[0] == coded[0];              // (see viterbi_round_kernel)
  if (t == 0)
    {
      if (finite) viterbi_super_round<0, NPF> (coded, step0, m_in, m_out, dec, so.off, g, lane, lds_a, lds_b, so.in_perm != 0, so.out_perm != 0, so.first != 0);
      else        viterbi_super_round<0, 3> (coded, step0, m_in, m_out, dec, so.off, g, lane, lds_a, lds_b, so.in_perm != 0, so.out_perm != 0, so.first != 0);
    }
  else if (t == 1)
    {
      if (finite) viterbi_super_round<1, NPF> (coded, step0, m_in, m_out, dec, so.off, g, lane, lds_a, lds_b, so.in_perm != 0, so.out_perm != 0, so.first != 0);
      else        viterbi_super_round<1, 3> (coded, step0, m_in, m_out, dec, so.off, g, lane, lds_a, lds_b, so.in_perm != 0, so.out_perm != 0, so.first != 0);
    }
  else
    {
      if (finite) viterbi_super_round<2, NPF> (coded, step0, m_in, m_out, dec, so.off, g, lane, lds_a, lds_b, so.in_perm != 0, so.out_perm != 0, so.first != 0);
      else        viterbi_super_round<2, 3> (coded, step0, m_in, m_out, dec, so.off, g, lane, lds_a, lds_b, so.in_perm != 0, so.out_perm != 0, so.first != 0);
    }
}

__global__ void __launch_bounds__ (256)
viterbi_init_kernel (ViterbiBatch b)
{
  int blk = blockIdx.y, t = 0;
  if (blk >= b.n[0]) { blk -= b.n[0]; t = 1; }
  if (t == 1 && blk >= b.n[1]) { blk -= b.n[1]; t = 2; }
  float *m = reinterpret_cast<float *> (b.ws[t] + (size_t) blk * b.block_ws_bytes);
  const int i = blockIdx.x * 256 + threadIdx.x;
  m[i] = i == 0 ? 0.f : -1.f;                              // start state 0, everything else unreachable (convcode.cc:144-146)
}

constexpr int MAX_ROUNDS = 64;
struct TracePlan
{
  int n_rounds;
  unsigned char k[MAX_ROUNDS];
  int step0[MAX_ROUNDS];
  unsigned int dec_offset[MAX_ROUNDS];                     // words
  int final_parity;                                        // which metric buffer holds the last step's metrics
};

/* one lane walks the survivors of a decode back, round by round (one aligned load of the lane's decision words per round) */
template<bool COH = false> __device__ __forceinline__ void
viterbi_trace_one (const TracePlan& plan, int final_parity, const unsigned char *ws, int n_steps, int rate, int *bits, float *error)
{
  const float *metric = reinterpret_cast<const float *> (ws + (final_parity ? V_METRIC_BYTES : 0));
  const unsigned int *dec = reinterpret_cast<const unsigned int *> (ws + 2 * V_METRIC_BYTES);
  *error = ld_metric<COH> (metric) / float (n_steps * rate);   // convcode.cc:197-199: state 0 at the end
  unsigned state = 0;
  for (int r = plan.n_rounds - 1; r >= 0; r--)
    {
      const int K = plan.k[r];
      const unsigned L = state >> K;
      unsigned loc = state & ((1u << K) - 1);
      const int DW = dec_words (K);
      unsigned int w[5] = { 0, 0, 0, 0, 0 };
      const unsigned int *p = dec + plan.dec_offset[r] + (size_t) L * DW;
      const uint4 a = ld_words<COH> (p);
      w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
      if (DW == 8)
        w[4] = ld_words<COH> (p + 4).x;
      for (int i = K; i >= 1; i--)
        {
          const int step = plan.step0[r] + i - 1;
          if (step < n_steps - V_ORDER)
            bits[step] = loc & 1;                          // the input bit of this step is the state's low bit
          const unsigned choose_high = (w[i - 1] >> loc) & 1;
          loc = (loc >> 1) | (choose_high << (K - 1));
        }
      state = L + (loc << (V_ORDER - K));                  // state before the round
    }
}

/* The walk back with a WAVE instead of a lane: 36 dependent loads (one per round, each a trip to memory: the decision words were
 * written by workgroups on other XCDs) become 14.  The state before a round of 4 steps is  (state >> 4) | (choice << 11)  with the 4
 * choice bits read from the round's decision words, so of the NEXT round's 2048 lanes only 16 can hold the words that will be needed,
 * and of the one after 256: while the current round's words are on their way, lane c fetches candidate c of the next round and
 * every lane four candidates of the round after that; three rounds then resolve with register reads (v_readlane).  Same bits,
 * same order of decisions as the one-lane walk. */
template<bool COH> __device__ __forceinline__ void
viterbi_trace_wave (const TracePlan& plan, int final_parity, const unsigned char *ws, int n_steps, int rate, int *bits, float *error, int lane)
{
  const float *metric = reinterpret_cast<const float *> (ws + (final_parity ? V_METRIC_BYTES : 0));
  const unsigned int *dec = reinterpret_cast<const unsigned int *> (ws + 2 * V_METRIC_BYTES);
  if (lane == 0)
    *error = ld_metric<COH> (metric) / float (n_steps * rate);   // convcode.cc:197-199: state 0 at the end
  const int n_out = n_steps - V_ORDER;
  // one round of K steps from the words w: returns the K choice bits (the top bits of the state before the round)
  auto walk = [&] (const unsigned int (&w)[4], int K, unsigned loc, int step0) -> unsigned {
#pragma unroll
    for (int i = 4; i >= 1; i--)                           // (unrolled with a guard: a run-time index into w would put it in scratch)
      if (i <= K)
        {
          const int step = step0 + i - 1;
          if (step < n_out && lane == 0)
            bits[step] = loc & 1;                          // the input bit of this step is the state's low bit
          const unsigned choose_high = (w[i - 1] >> loc) & 1;
          loc = (loc >> 1) | (choose_high << (K - 1));
        }
    return loc;
  };
  auto pick = [] (const uint4& v, unsigned from_lane, unsigned int (&w)[4]) {
    w[0] = __builtin_amdgcn_readlane (v.x, from_lane);
    w[1] = __builtin_amdgcn_readlane (v.y, from_lane);
    w[2] = __builtin_amdgcn_readlane (v.z, from_lane);
    w[3] = __builtin_amdgcn_readlane (v.w, from_lane);
  };
  unsigned state = 0;
  int r = plan.n_rounds - 1;
  while (r >= 0)
    {
      const int K = plan.k[r];
      if (r >= 2 && K == 4 && plan.k[r - 1] == 4 && plan.k[r - 2] == 4)
        {
          const unsigned L0 = state >> 4;
          const uint4 a = ld_words<COH> (dec + plan.dec_offset[r] + (size_t) L0 * 4);
          const unsigned c1 = lane & 15;
          const uint4 b1 = ld_words<COH> (dec + plan.dec_offset[r - 1] + (size_t) ((L0 >> 4) | (c1 << 7)) * 4);
          uint4 c2[4];
#pragma unroll
          for (int q = 0; q < 4; q++)
            {
              const unsigned idx = lane + 64 * q, c = idx & 15, e = idx >> 4;
              c2[q] = ld_words<COH> (dec + plan.dec_offset[r - 2] + (size_t) ((L0 >> 8) | (c << 3) | (e << 7)) * 4);
            }
          unsigned int w[4];
          pick (a, 0, w);
          const unsigned la = __builtin_amdgcn_readfirstlane (walk (w, 4, state & 15, plan.step0[r]));
          const unsigned s1 = L0 | (la << 11);
          pick (b1, la, w);
          const unsigned lb = __builtin_amdgcn_readfirstlane (walk (w, 4, s1 & 15, plan.step0[r - 1]));
          const unsigned s2 = (s1 >> 4) | (lb << 11);
          const unsigned idx = la + 16 * lb;
          {
            // (all four quarters are read and the right one chosen among scalars: choosing the vector first makes c2 an indexed
            // array, i.e. scratch memory)
            unsigned int w0[4], w1[4], w2[4], w3[4];
            pick (c2[0], idx & 63, w0);
            pick (c2[1], idx & 63, w1);
            pick (c2[2], idx & 63, w2);
            pick (c2[3], idx & 63, w3);
            const unsigned qsel = idx >> 6;
#pragma unroll
            for (int i = 0; i < 4; i++)
              w[i] = qsel == 0 ? w0[i] : qsel == 1 ? w1[i] : qsel == 2 ? w2[i] : w3[i];
          }
          const unsigned lc = __builtin_amdgcn_readfirstlane (walk (w, 4, s2 & 15, plan.step0[r - 2]));
          state = (s2 >> 4) | (lc << 11);
          r -= 3;
          continue;
        }
      const unsigned L = state >> K;
      const uint4 a = ld_words<COH> (dec + plan.dec_offset[r] + (size_t) L * dec_words (K));      // (K <= 4 here: one 16-byte entry)
      unsigned int w[4];
      pick (a, 0, w);
      const unsigned l = __builtin_amdgcn_readfirstlane (walk (w, K, state & ((1u << K) - 1), plan.step0[r]));
      state = L | (l << (V_ORDER - K));
      r--;
    }
}

// FALLBACK: the walk back as a launch of its own (one lane per decode) -- only for chains without the sync workspace or whose last round is
// not a single round of <= 4 steps (awm_debug_set_viterbi_super (0), K = 5 rounds); the chains of the product end in viterbi_last_round_kernel.
__global__ void __launch_bounds__ (64)
viterbi_trace_kernel (ViterbiBatch b, TracePlan plan)
{
  const int total = b.n[0] + b.n[1] + b.n[2];
  int blk = blockIdx.x * 64 + threadIdx.x, t = 0;
  if (blk >= total)
    return;
  if (blk >= b.n[0]) { blk -= b.n[0]; t = 1; }
  if (t == 1 && blk >= b.n[1]) { blk -= b.n[1]; t = 2; }
  viterbi_trace_one (plan, plan.final_parity, b.ws[t] + (size_t) blk * b.block_ws_bytes, b.n_steps, t == 2 ? 12 : 6,
                     b.bits[t] + (size_t) blk * (b.n_steps - V_ORDER), &b.error[t][blk]);
}

/* ---- ONE launch per batch of decodes ---------------------------------------------------------------------------------------------
 * The chain above is 14 dependent launches (11 x 12 steps, 4, 4, 3 + walk back; 16 until the initial metrics and the walk back were folded in) of 18 - 30 us each for a chunk's ~37 decodes, and
 * what a batch costs is decided by the gaps between them: 0.9 ms per bench step from the kernels' own durations, 1.5 - 2.4 ms
 * between the HIP events on two different boxes.  The launches exist only because the 8 workgroups of a decode must exchange their
 * metrics every 12 steps -- a barrier among 8 workgroups, not across the grid.  So here a batch is one launch of 8 workgroups per
 * decode that stay resident for all 143 steps; between two exchanges they meet at a per-decode counter in global memory, the
 * metrics and decision words cross the XCDs through device-coherent loads and stores (ld_metric above: no cache is flushed), the
 * first round starts from the initial metrics without reading them, and the workgroup with family 0 walks the survivors back at
 * the end.
 *   Which (decode, family) a workgroup works on is NOT its blockIdx: it draws a ticket when it starts (atomic counter), decode =
 * ticket / 8, family = ticket % 8.  So the workgroups that are resident always hold the lowest tickets, the peers a waiting
 * workgroup needs are either resident or the very next ones to start, and at most one decode per launch (<= 7 workgroups) can be
 * waiting for workgroups that are not resident yet -- no deadlock whatever the batch size, the number of lanes launching such
 * kernels side by side, or other kernels holding compute units (a static blockIdx -> decode map can deadlock when two such
 * launches oversubscribe the chip: each XCD dispatches its share of a grid on its own).
 *   The counters are self-cleaning (zero when the launch ends: the family-0 workgroup clears its decode's counter after the last
 * wait, the last workgroup to leave clears ticket and exit counters), so a lane's sync block is zeroed once, when it is allocated.
 * A wait that lasts longer than ~2 s of wall time (a fault elsewhere) gives up and marks the batch: every decode then reports the
 * impossible error value -2 (a block of NaN soft bits ends at -1 / coded length, nothing lies below that) and the host fails the call instead of hanging the GPU.
 * sync block (unsigned int): [0] ticket, [1] workgroups that left, [2] failure mark, [16 (1 + d)] counter of decode d */
constexpr int SYNC_STRIDE = 16;

__device__ __forceinline__ void
decode_barrier (unsigned int *counter, unsigned int target, unsigned int *fail, bool wait = true)
{
  // every coherent store of this wave has been acknowledged (written through to where the other XCDs read it) before the wave arrives;
  // __syncthreads collects the workgroup's waves, then ONE lane counts the workgroup in.  No cache maintenance (see ld_metric).
  asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_fetch_add (counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!wait)                                               // (arrive only: the last meeting of a workgroup that has nothing left to read)
    return;
  if (threadIdx.x == 0)
    {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
      while (__hip_atomic_load (counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
        {
          __builtin_amdgcn_s_sleep (2);
          if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull || __hip_atomic_load (fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            {
              __hip_atomic_store (fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
        }
    }
  __syncthreads();
}

template<int BT> __device__ __forceinline__ void
viterbi_persistent_decode (const ViterbiBatch& b, const TracePlan& plan, int n_super, int blk, int g, int lane, unsigned int *counter,
                           unsigned int *fail, float *lds_a, float *lds_b)
{
  constexpr int t = BT, rate = BT == 2 ? 12 : 6;
  unsigned char *ws = b.ws[t] + (size_t) blk * b.block_ws_bytes;
  float *metric[2] = { reinterpret_cast<float *> (ws), reinterpret_cast<float *> (ws + V_METRIC_BYTES) };
  unsigned int *dec = reinterpret_cast<unsigned int *> (ws + 2 * V_METRIC_BYTES);
  const float *coded = b.soft[t] + (size_t) blk * b.n_steps * rate;
  const bool finite = coded[0] == coded[0];              // (see viterbi_round_kernel)
  int parity = 0;
  unsigned int meetings = 0;
  if (n_super == 0)
    {
      for (int i = g * 4096 + 2 * lane; i < (g + 1) * 4096; i += 2 * V_WG)
        st_pair<true> (metric[0] + i, i == 0 ? 0.f : -1.f, -1.f);
      decode_barrier (counter, 8 * ++meetings, fail);
    }
  for (int s = 0; s < n_super; s++)
    {
      const int r = 3 * s, step0 = plan.step0[r];
      const size_t off[3] = { plan.dec_offset[r], plan.dec_offset[r + 1], plan.dec_offset[r + 2] };
      int npf = 0;
      for (int q = 0; q < 3; q++)
        npf += plan.step0[r + q] < V_ORDER;
      if (!finite)
        npf = 3;
      const float *m_in = metric[parity];
      float *m_out = metric[parity ^ 1];
      const bool in_perm = s > 0, out_perm = s + 1 < n_super, first = s == 0;
      if (npf == 0)      viterbi_super_round<BT, 0, true> (coded, step0, m_in, m_out, dec, off, g, lane, lds_a, lds_b, in_perm, out_perm, first);
      else if (npf == 1) viterbi_super_round<BT, 1, true> (coded, step0, m_in, m_out, dec, off, g, lane, lds_a, lds_b, in_perm, out_perm, first);
      else if (npf == 2) viterbi_super_round<BT, 2, true> (coded, step0, m_in, m_out, dec, off, g, lane, lds_a, lds_b, in_perm, out_perm, first);
      else               viterbi_super_round<BT, 3, true> (coded, step0, m_in, m_out, dec, off, g, lane, lds_a, lds_b, in_perm, out_perm, first);
      parity ^= 1;
      decode_barrier (counter, 8 * ++meetings, fail, g == 0 || r + 3 < plan.n_rounds);
    }
  for (int r = 3 * n_super; r < plan.n_rounds; r++)
    {
      const int k = plan.k[r], step0 = plan.step0[r];
      const bool plain = step0 >= V_ORDER && finite;
      const float *m_in = metric[parity];
      float *m_out = metric[parity ^ 1];
      unsigned int *d = dec + plan.dec_offset[r];
      for (int L = g * V_WG + lane; L < (V_STATES >> k); L += 8 * V_WG)
        {
          if (k == 4 && plain)  viterbi_round<BT, 4, true, true> (coded, step0, m_in, m_out, d, L);
          else if (k == 4)      viterbi_round<BT, 4, false, true> (coded, step0, m_in, m_out, d, L);
          else if (plain)       viterbi_round<BT, 3, true, true> (coded, step0, m_in, m_out, d, L);
          else                  viterbi_round<BT, 3, false, true> (coded, step0, m_in, m_out, d, L);
        }
      parity ^= 1;
      decode_barrier (counter, 8 * ++meetings, fail, g == 0 || r + 1 < plan.n_rounds);       // after the last round only the tracer waits
    }
  if (g == 0 && lane < 64)
    {
      // (all eight have arrived for the last time and only this workgroup waited: nobody looks at the counter any more)
      if (lane == 0)
        __hip_atomic_store (counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float err = 0.f;
      bool wide = true;                                    // (the wave walk reads 16-byte entries: rounds of at most 4 steps)
      for (int r = 0; r < plan.n_rounds; r++)
        wide = wide && plan.k[r] <= 4;
      if (wide)
        viterbi_trace_wave<true> (plan, parity, ws, b.n_steps, rate, b.bits[t] + (size_t) blk * (b.n_steps - V_ORDER), &err, lane);
      else if (lane == 0)
        viterbi_trace_one<true> (plan, parity, ws, b.n_steps, rate, b.bits[t] + (size_t) blk * (b.n_steps - V_ORDER), &err);
      if (lane == 0)
        b.error[t][blk] = __hip_atomic_load (fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? -2.f : err;
    }
}

/* The LAST launch of the chain when it is a single round: the round itself with device-coherent stores, then the workgroup that
 * finishes last for a decode (a counter per decode in the lane's sync block, back at zero afterwards) walks that decode's survivors
 * back with its first wave -- the separate walk-back launch (one lane per decode, 36 dependent loads, ~30 us) disappears from the chain. */
template<int K, bool PLAIN> __global__ void __launch_bounds__ (V_WG)
viterbi_last_round_kernel (ViterbiBatch b, int step0, int parity_in, size_t dec_offset, TracePlan plan, unsigned int *sync)
{
  __shared__ int is_last;
  int blk = blockIdx.y, t = 0;
  if (blk >= b.n[0]) { blk -= b.n[0]; t = 1; }
  if (t == 1 && blk >= b.n[1]) { blk -= b.n[1]; t = 2; }
  const int rate = t == 2 ? 12 : 6;
  const int L = blockIdx.x * V_WG + threadIdx.x;
  unsigned char *ws = b.ws[t] + (size_t) blk * b.block_ws_bytes;
  const float *m_in = reinterpret_cast<const float *> (ws + (parity_in ? V_METRIC_BYTES : 0));
  float *m_out = reinterpret_cast<float *> (ws + (parity_in ? 0 : V_METRIC_BYTES));
  unsigned int *dec = reinterpret_cast<unsigned int *> (ws + 2 * V_METRIC_BYTES) + dec_offset;
  const float *coded = b.soft[t] + (size_t) blk * b.n_steps * rate;
  const bool finite = coded[0] == coded[0];              // (see viterbi_round_kernel)
  if (t == 0)
    {
      if (PLAIN && finite) viterbi_round<0, K, true, true> (coded, step0, m_in, m_out, dec, L);
      else                 viterbi_round<0, K, false, true> (coded, step0, m_in, m_out, dec, L);
    }
  else if (t == 1)
    {
      if (PLAIN && finite) viterbi_round<1, K, true, true> (coded, step0, m_in, m_out, dec, L);
      else                 viterbi_round<1, K, false, true> (coded, step0, m_in, m_out, dec, L);
    }
  else
    {
      if (PLAIN && finite) viterbi_round<2, K, true, true> (coded, step0, m_in, m_out, dec, L);
      else                 viterbi_round<2, K, false, true> (coded, step0, m_in, m_out, dec, L);
    }
  // my stores are through (acknowledged device-wide) before the workgroup counts itself in
  asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned int *counter = sync + SYNC_STRIDE * (1 + blockIdx.y);
  if (threadIdx.x == 0)
    is_last = __hip_atomic_fetch_add (counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == gridDim.x;
  __syncthreads();
  if (is_last && threadIdx.x < 64)
    {
      if (threadIdx.x == 0)
        __hip_atomic_store (counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float err = 0.f;
      viterbi_trace_wave<true> (plan, plan.final_parity, ws, b.n_steps, rate, b.bits[t] + (size_t) blk * (b.n_steps - V_ORDER), &err, threadIdx.x);
      if (threadIdx.x == 0)
        b.error[t][blk] = err;
    }
}

__global__ void __launch_bounds__ (V_WG)
viterbi_persistent_kernel (ViterbiBatch b, TracePlan plan, int n_super, unsigned int *sync)
{
  __shared__ float lds_a[SUPER_LDS], lds_b[SUPER_LDS];
  __shared__ unsigned int my_ticket;
  const int lane = threadIdx.x;
  if (lane == 0)
    my_ticket = __hip_atomic_fetch_add (&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned int ticket = my_ticket;
  const int decode = int (ticket >> 3), g = int (ticket & 7);
  int blk = decode, t = 0;
  if (blk >= b.n[0]) { blk -= b.n[0]; t = 1; }
  if (t == 1 && blk >= b.n[1]) { blk -= b.n[1]; t = 2; }
  unsigned int *counter = sync + SYNC_STRIDE * (1 + decode), *fail = sync + 2;
  if (t == 0)      viterbi_persistent_decode<0> (b, plan, n_super, blk, g, lane, counter, fail, lds_a, lds_b);
  else if (t == 1) viterbi_persistent_decode<1> (b, plan, n_super, blk, g, lane, counter, fail, lds_a, lds_b);
  else             viterbi_persistent_decode<2> (b, plan, n_super, blk, g, lane, counter, fail, lds_a, lds_b);
  __syncthreads();
  if (lane == 0)
    {
      // the last workgroup to leave clears the launch-wide counters (all tickets are drawn, all other exits are counted)
      const unsigned int left = __hip_atomic_fetch_add (&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (left + 1 == gridDim.x)
        {
          __hip_atomic_store (&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store (&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

constexpr int V_K = 4;            // trellis steps per full round (see the header)

std::vector<RoundPlan>
plan_rounds (int n_steps)
{
  // n_steps = V_K a + 3 b with the smallest b
  std::vector<RoundPlan> rounds;
  int threes = 0;
  while (threes < V_K && (n_steps - 3 * threes < 0 || (n_steps - 3 * threes) % V_K))
    threes++;
  if (threes == V_K || n_steps - 3 * threes < 0)
    return rounds;
  int step = 0;
  size_t off = 0;
  auto add = [&] (int k) {
    rounds.push_back ({ k, step, off });
    step += k;
    off += size_t (V_STATES >> k) * dec_words (k);
  };
  for (int i = 0; i < (n_steps - 3 * threes) / V_K; i++)
    add (V_K);
  for (int i = 0; i < threes; i++)
    add (3);
  return rounds;
}

}  // namespace

size_t
viterbi_workspace_bytes (long long coded_len, int rate, long long n_blocks)
{
  const auto rounds = plan_rounds (int (coded_len / rate));
  size_t words = 0;
  for (const auto& r : rounds)
    words += size_t (V_STATES >> r.k) * dec_words (r.k);
  const size_t per_block = (2 * V_METRIC_BYTES + words * sizeof (unsigned int) + 255) & ~size_t (255);
  return per_block * size_t (n_blocks);
}

int g_viterbi_super = 1;         // (debug toggle)
extern "C" void awm_debug_set_viterbi_super (int on) { g_viterbi_super = on; }
/* Which form runs.  The chain's 14 launches carry ~18 us of work each, so what a batch costs depends on how fast THIS host gets
 * dependent launches out: 0.92 ms per bench step on boxes where a dependent launch costs a few microseconds, 1.74 ms on the driver's
 * box of round 3 (same binary, every other kernel within 5 %).  The one-launch kernel costs 0.96 ms everywhere -- but it holds its
 * compute units for the whole batch (176 VGPRs, 41 KB of LDS per workgroup: no scan tile of another lane fits beside it), which makes the
 * 60 min STEP 5 % slower than with the chain on a fast-launch box (5.47 against 5.19 ms, alternating in one process,
 * profiles/r04/variants.txt).  So the choice is made per process from a measurement: awm_ctx_create times a chain of empty
 * dependent launches on the idle stream (probe_dependent_launch_us), and above ONE_LAUNCH_ABOVE_US per launch the batches take the
 * one-launch kernel.  awm_debug_set_viterbi_persistent (0 | 1) forces a form, -1 returns to the measurement. */
/* The measurement is LATCHED: the first context of the process that gets a valid probe through decides for the process (contexts are
 * created and decodes run from several host threads: both values are atomics, the probe itself runs under a mutex), so the form
 * cannot flip in mid-process because a helper context was created under a tracer or on a busy host.  A failed probe (-1) leaves the
 * question open for the next context; until then the chain runs. */
std::atomic<int>    g_viterbi_persistent { -1 };
std::atomic<double> g_dependent_launch_us { -1.0 };      // the latched measurement (-1: none yet)
std::mutex          g_probe_mutex;
constexpr double ONE_LAUNCH_ABOVE_US = 9.0;
extern "C" void awm_debug_set_viterbi_persistent (int on) { g_viterbi_persistent.store (on, std::memory_order_relaxed); }
extern "C" double awm_debug_dependent_launch_us (void) { return g_dependent_launch_us.load (std::memory_order_relaxed); }
static bool
use_one_launch()
{
  const int forced = g_viterbi_persistent.load (std::memory_order_relaxed);
  return forced > 0 || (forced < 0 && g_dependent_launch_us.load (std::memory_order_relaxed) > ONE_LAUNCH_ABOVE_US);
}
/* for error messages: which form the batches of this process take and why */
const char *
viterbi_form_description()
{
  static thread_local char text[160];
  const int forced = g_viterbi_persistent.load (std::memory_order_relaxed);
  snprintf (text, sizeof (text), "viterbi form: %s (%s; dependent launch probe %.1f us, one launch above %.1f us)",
            use_one_launch() ? "one launch per batch" : "launch chain", forced < 0 ? "chosen by the probe" : "forced by awm_debug_set_viterbi_persistent",
            g_dependent_launch_us.load (std::memory_order_relaxed), ONE_LAUNCH_ABOVE_US);
  return text;
}
extern "C" int awm_debug_viterbi_one_launch_in_use (void) { return use_one_launch() ? 1 : 0; }

namespace { __global__ void empty_kernel() {} }

/* microseconds per launch of a chain of empty kernels on an idle stream (issue + dispatch + completion hand-over of this host / driver) */
double
probe_dependent_launch_us (hipStream_t st)
{
  constexpr int N = 48;
  std::lock_guard<std::mutex> lock (g_probe_mutex);
  const double latched = g_dependent_launch_us.load (std::memory_order_relaxed);
  if (latched >= 0)
    return latched;
  if (hipStreamSynchronize (st) != hipSuccess)
    return -1;
  double best = -1;
  for (int rep = 0; rep < 3; rep++)                        // (the first repetition also pays for loading the kernel)
    {
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; i++)
        hipLaunchKernelGGL (empty_kernel, dim3 (1), dim3 (64), 0, st);
      if (hipStreamSynchronize (st) != hipSuccess)
        return -1;
      const double us = std::chrono::duration<double, std::micro> (std::chrono::steady_clock::now() - t0).count() / N;
      if (best < 0 || us < best)
        best = us;
    }
  if (best >= 0)
    g_dependent_launch_us.store (best, std::memory_order_relaxed);
  return best;
}

size_t
viterbi_sync_bytes (long long n_blocks)
{
  return size_t (1 + std::max (0ll, n_blocks)) * SYNC_STRIDE * sizeof (unsigned int);
}

hipError_t
launch_viterbi (hipStream_t st, const float *const soft[3], const long long n_blocks[3], long long n_steps,
                unsigned char *const decisions_ws[3], int *const bits_out[3], float *const error_out[3], unsigned int *sync_ws)
{
  const long long total = n_blocks[0] + n_blocks[1] + n_blocks[2];
  if (total <= 0)
    return hipSuccess;
  const auto rounds = plan_rounds (int (n_steps));
  if (rounds.empty() || rounds.size() > MAX_ROUNDS || total > 65535)
    return hipErrorInvalidValue;
  ViterbiBatch b;
  for (int i = 0; i < 3; i++)
    {
      b.soft[i] = soft[i];
      b.ws[i] = decisions_ws[i];
      b.bits[i] = bits_out[i];
      b.error[i] = error_out[i];
      b.n[i] = int (n_blocks[i]);
    }
  b.n_steps = int (n_steps);
  b.block_ws_bytes = viterbi_workspace_bytes (n_steps * 6, 6, 1);          // the layout does not depend on the rate
  TracePlan tp {};
  tp.n_rounds = int (rounds.size());
  for (size_t r = 0; r < rounds.size(); r++)
    {
      tp.k[r] = (unsigned char) rounds[r].k;
      tp.step0[r] = rounds[r].step0;
      tp.dec_offset[r] = (unsigned int) rounds[r].dec_offset;
    }
  if (sync_ws && use_one_launch())
    {
      // one launch: the leading triples of 4-step rounds as 12-step segments, the remaining rounds one by one
      int n_super = 0;
      while (V_K == 4 && 3 * n_super + 2 < int (rounds.size()) && rounds[3 * n_super].k == 4 && rounds[3 * n_super + 1].k == 4 && rounds[3 * n_super + 2].k == 4)
        n_super++;
      hipLaunchKernelGGL (viterbi_persistent_kernel, dim3 (8u * (unsigned) total), dim3 (V_WG), 0, st, b, tp, n_super, sync_ws);
      return hipGetLastError();
    }
  // (a chain that begins with a 12-step launch starts from the initial metrics inside it: one launch less)
  const bool fold_init = g_viterbi_super && V_K == 4 && rounds.size() >= 3 && rounds[0].k == V_K && rounds[1].k == V_K && rounds[2].k == V_K;
  if (!fold_init)
    hipLaunchKernelGGL (viterbi_init_kernel, dim3 (V_STATES / 256, (unsigned) total), dim3 (256), 0, st, b);
  int parity = 0;
  for (size_t r = 0; r < rounds.size(); )
    {
      const RoundPlan& rp = rounds[r];
      if (g_viterbi_super && r + 2 < rounds.size() && rounds[r].k == V_K && rounds[r + 1].k == V_K && rounds[r + 2].k == V_K && V_K == 4)
        {
          // three rounds in one launch (metrics exchanged through LDS); non-plain rounds = those that start before step V_ORDER
          int npf = 0;
          for (int q = 0; q < 3; q++)
            npf += rounds[r + q].step0 < V_ORDER;
          // (permuted metric layout between consecutive launches of this kind; the init kernel's output reads the same either way)
          const bool next_is_super = r + 5 < rounds.size() && rounds[r + 3].k == V_K && rounds[r + 4].k == V_K && rounds[r + 5].k == V_K;
          const SuperOffsets so { { rounds[r].dec_offset, rounds[r + 1].dec_offset, rounds[r + 2].dec_offset }, r > 0, next_is_super, r == 0 && fold_init };
          const dim3 grid (8, (unsigned) total);
          if (npf == 0)
            hipLaunchKernelGGL ((viterbi_super_kernel<0>), grid, dim3 (V_WG), 0, st, b, rp.step0, parity, so);
          else if (npf == 1)
            hipLaunchKernelGGL ((viterbi_super_kernel<1>), grid, dim3 (V_WG), 0, st, b, rp.step0, parity, so);
          else if (npf == 2)
            hipLaunchKernelGGL ((viterbi_super_kernel<2>), grid, dim3 (V_WG), 0, st, b, rp.step0, parity, so);
          else
            hipLaunchKernelGGL ((viterbi_super_kernel<3>), grid, dim3 (V_WG), 0, st, b, rp.step0, parity, so);
          parity ^= 1;                     // one trip through global memory for the three rounds
          r += 3;
          continue;
        }
      // after V_ORDER steps every state is reachable and, for finite input, all metrics are >= 0: plain compare-select
      const bool plain = rp.step0 >= V_ORDER;
      const dim3 grid ((V_STATES >> rp.k) / V_WG, (unsigned) total);
      if (sync_ws && r + 1 == rounds.size() && rp.k <= 4)
        {
          // the last round walks the survivors back itself (viterbi_last_round_kernel)
          tp.final_parity = parity ^ 1;
          if (rp.k == V_K && plain)
            hipLaunchKernelGGL ((viterbi_last_round_kernel<V_K, true>), grid, dim3 (V_WG), 0, st, b, rp.step0, parity, rp.dec_offset, tp, sync_ws);
          else if (rp.k == V_K)
            hipLaunchKernelGGL ((viterbi_last_round_kernel<V_K, false>), grid, dim3 (V_WG), 0, st, b, rp.step0, parity, rp.dec_offset, tp, sync_ws);
          else if (plain)
            hipLaunchKernelGGL ((viterbi_last_round_kernel<3, true>), grid, dim3 (V_WG), 0, st, b, rp.step0, parity, rp.dec_offset, tp, sync_ws);
          else
            hipLaunchKernelGGL ((viterbi_last_round_kernel<3, false>), grid, dim3 (V_WG), 0, st, b, rp.step0, parity, rp.dec_offset, tp, sync_ws);
          return hipGetLastError();
        }
      if (rp.k == V_K && plain)
        hipLaunchKernelGGL ((viterbi_round_kernel<V_K, true>), grid, dim3 (V_WG), 0, st, b, rp.step0, parity, rp.dec_offset);
      else if (rp.k == V_K)
        hipLaunchKernelGGL ((viterbi_round_kernel<V_K, false>), grid, dim3 (V_WG), 0, st, b, rp.step0, parity, rp.dec_offset);
      else if (plain)
        hipLaunchKernelGGL ((viterbi_round_kernel<3, true>), grid, dim3 (V_WG), 0, st, b, rp.step0, parity, rp.dec_offset);
      else
        hipLaunchKernelGGL ((viterbi_round_kernel<3, false>), grid, dim3 (V_WG), 0, st, b, rp.step0, parity, rp.dec_offset);
      parity ^= 1;
      r++;
    }
  tp.final_parity = parity;
  hipLaunchKernelGGL (viterbi_trace_kernel, dim3 ((unsigned) ((total + 63) / 64)), dim3 (64), 0, st, b, tp);
  return hipGetLastError();
}

} // namespace awmk
