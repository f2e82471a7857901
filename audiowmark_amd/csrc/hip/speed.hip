// speed.hip -- speed detection kernels for gfx950 (reference src/wmspeed.cc, src/resample.cc:96-125)
//
//   K12 resample_var_kernel    zita VResampler (arbitrary ratio, 256 phases, interpolated coefficients)
//   K13 speed_mags_kernel      SpeedSync::prepare_mags: FFT-512 hop 128 on the half-rate clip -> dB -> up / down sums per sync frame
//   K14 speed_compare_kernel   SpeedSync::compare / compare_bits: Q16 walk of every candidate block start over the magnitude matrix
//   K15 gather / energy        get_clip_locations sample subset, get_best_clip_location energies
//
// Layouts are chosen for the heavy kernel (K14): the magnitude matrix of a centre speed is [column][row] float2
// (umag, dmag) with the rows (time steps) contiguous, and the columns in the order [sync bit][frame of that bit, ascending],
// so that a thread (= one candidate block start) sums the frames of one sync bit in the reference's order with plain
// registers while its neighbours read neighbouring rows (coalesced).
#include "kernels.hh"
#include "awm_fft.hip.h"
#include <algorithm>

namespace awmk {

constexpr int SPEED_NB = 81, SPEED_MIN_BAND = 20;
constexpr int SPEED_COLS = 510;       // sync frames per block
constexpr int SPEED_TILE = 64;        // rows per workgroup of K13

/* ------------------------------------------------------------------------------------------------------------------
 * K12: output m of the resampler reads the input window that starts at floor (m * step / 256) -- zita accumulates the
 * phase in double (ph += step with wrap), here it comes from the exact product m * step (128 bit), which differs from
 * the accumulated value only by the rounding errors zita collects on the way (< 1e-9 phases per million outputs).
 * The arithmetic per output is zita's: c1[i] = a q1[i] + b q1[i + hl], c2[i] = a q2[i] + b q2[i - hl],
 * y = (1e-25 + sum_i (x1 c1[i] + x2 c2[i])) - 1e-25 with every product and sum rounded on its own.
 * ------------------------------------------------------------------------------------------------------------------ */
// phase and window start of output m
struct VarPhase { long long first; unsigned k; float af, bf; };
__device__ __forceinline__ VarPhase
var_phase (const SpeedCenterDev& cd, long long m)
{
  const int np = 256;
  const unsigned long long lo = (unsigned long long) m * cd.mant, hi = __umul64hi ((unsigned long long) m, cd.mant);
  long long b = (long long) ((hi << (64 - cd.shift)) | (lo >> cd.shift));
  const unsigned long long frac = lo & ((1ull << cd.shift) - 1);
  double ph = double (frac) * cd.frac_scale;
  if (ph >= np)                                                // only if the fraction rounds up to a whole step (ratios > 2)
    {
      ph = 0;
      b++;
    }
  VarPhase v;
  v.k = unsigned (ph);
  v.bf = float (ph - v.k);
  v.af = __fsub_rn (1.0f, v.bf);
  v.first = b - (cd.hl - 1);                                   // input frame of the first tap
  return v;
}

// one output frame, all channels; tab = coefficient rows with `stride` floats each (global or LDS)
template<int CT> __device__ __forceinline__ void
var_output (const SpeedCenterDev& cd, const float *tab, int stride, const float *in, int n_channels, long long m, float *out)
{
  const int C = CT ? CT : n_channels, hl = cd.hl, np = 256;
  const VarPhase v = var_phase (cd, m);
  const float *q1 = tab + stride * v.k, *q1n = q1 + stride;                 // rows k, k + 1         (zita: q1[i], q1[i + hl])
  const float *q2 = tab + stride * (np - v.k), *q2p = q2 - stride;          // rows np - k, np - k - 1 (q2[i], q2[i - hl])
  if (CT == 2)
    {
      const float2 *in2 = reinterpret_cast<const float2 *> (in);
      float s0 = 1e-25f, s1 = 1e-25f;
      if (__all (v.first >= 0 && v.first + 2 * hl <= cd.n_in))    // the whole wave is away from the ends: no bounds checks
        {
          const float2 *p1 = in2 + v.first, *p2 = in2 + v.first + 2 * hl - 1;
#pragma unroll 4
          for (int i = 0; i < hl; i++)
            {
              const float c1 = __fadd_rn (__fmul_rn (v.af, q1[i]), __fmul_rn (v.bf, q1n[i]));
              const float c2 = __fadd_rn (__fmul_rn (v.af, q2[i]), __fmul_rn (v.bf, q2p[i]));
              const float2 x1 = p1[i], x2 = p2[-i];
              s0 = __fadd_rn (s0, __fadd_rn (__fmul_rn (x1.x, c1), __fmul_rn (x2.x, c2)));
              s1 = __fadd_rn (s1, __fadd_rn (__fmul_rn (x1.y, c1), __fmul_rn (x2.y, c2)));
            }
          reinterpret_cast<float2 *> (out)[m] = make_float2 (__fsub_rn (s0, 1e-25f), __fsub_rn (s1, 1e-25f));
          return;
        }
      for (int i = 0; i < hl; i++)
        {
          const float c1 = __fadd_rn (__fmul_rn (v.af, q1[i]), __fmul_rn (v.bf, q1n[i]));
          const float c2 = __fadd_rn (__fmul_rn (v.af, q2[i]), __fmul_rn (v.bf, q2p[i]));
          const long long j1 = v.first + i, j2 = v.first + 2 * hl - 1 - i;
          const float2 x1 = (j1 >= 0 && j1 < cd.n_in) ? in2[j1] : make_float2 (0.f, 0.f);
          const float2 x2 = (j2 >= 0 && j2 < cd.n_in) ? in2[j2] : make_float2 (0.f, 0.f);
          s0 = __fadd_rn (s0, __fadd_rn (__fmul_rn (x1.x, c1), __fmul_rn (x2.x, c2)));
          s1 = __fadd_rn (s1, __fadd_rn (__fmul_rn (x1.y, c1), __fmul_rn (x2.y, c2)));
        }
      reinterpret_cast<float2 *> (out)[m] = make_float2 (__fsub_rn (s0, 1e-25f), __fsub_rn (s1, 1e-25f));
      return;
    }
  for (int c = 0; c < C; c++)
    {
      float sum = 1e-25f;
      for (int i = 0; i < hl; i++)
        {
          const float c1 = __fadd_rn (__fmul_rn (v.af, q1[i]), __fmul_rn (v.bf, q1n[i]));
          const float c2 = __fadd_rn (__fmul_rn (v.af, q2[i]), __fmul_rn (v.bf, q2p[i]));
          const long long j1 = v.first + i, j2 = v.first + 2 * hl - 1 - i;
          const float x1 = (j1 >= 0 && j1 < cd.n_in) ? in[j1 * C + c] : 0.f;
          const float x2 = (j2 >= 0 && j2 < cd.n_in) ? in[j2 * C + c] : 0.f;
          sum = __fadd_rn (sum, __fadd_rn (__fmul_rn (x1, c1), __fmul_rn (x2, c2)));
        }
      out[m * C + c] = __fsub_rn (sum, 1e-25f);
    }
}

/* stereo, the input window of the tile staged in LDS (zero extended at the ends of the stream): s_in[0] is input frame first0;
 * the same products and sums as var_output */
__device__ __forceinline__ void
var_output_staged (const SpeedCenterDev& cd, const float *tab, int stride, const VarPhase& v, const float2 *p1, long long m, float *out)
{
  const int hl = cd.hl, np = 256;
  const float *q1 = tab + stride * v.k, *q1n = q1 + stride;
  const float *q2 = tab + stride * (np - v.k), *q2p = q2 - stride;
  const float2 *p2 = p1 + 2 * hl - 1;
  float s0 = 1e-25f, s1 = 1e-25f;
#pragma unroll 4
  for (int i = 0; i < hl; i++)
    {
      const float c1 = __fadd_rn (__fmul_rn (v.af, q1[i]), __fmul_rn (v.bf, q1n[i]));
      const float c2 = __fadd_rn (__fmul_rn (v.af, q2[i]), __fmul_rn (v.bf, q2p[i]));
      const float2 x1 = p1[i], x2 = p2[-i];
      s0 = __fadd_rn (s0, __fadd_rn (__fmul_rn (x1.x, c1), __fmul_rn (x2.x, c2)));
      s1 = __fadd_rn (s1, __fadd_rn (__fmul_rn (x1.y, c1), __fmul_rn (x2.y, c2)));
    }
  reinterpret_cast<float2 *> (out)[m] = make_float2 (__fsub_rn (s0, 1e-25f), __fsub_rn (s1, 1e-25f));
}

constexpr int RV_TILE = 512;              // outputs per tile (one staged input window)
constexpr int RV_MAX_TAB = 12288;         // floats of LDS for the coefficient table (48 KiB): 257 rows of up to 47 floats
constexpr int RV_MAX_SPAN = 4096;         // input frames of a staged window (32 KiB)

/* Neighbouring outputs have unrelated phases, i.e. every lane reads its own four coefficient rows: from global memory that
 * is 64 different cache lines per load instruction (the first version of this kernel spent 58 % of the whole speed search
 * there).  The table (257 rows, row stride odd so that equal columns of different rows fall into different banks) is
 * therefore staged in LDS, once per workgroup, which then works through `tiles_per_wg` consecutive tiles (staging 4 - 9 k
 * floats took longer than the taps of one tile).  For stereo the input window of a tile can go through LDS as well (the
 * windows of neighbouring outputs overlap almost completely; read from global memory every tap pair is two load
 * instructions over 9 - 17 cache lines) -- off by default, see g_resample_var_mode. */
template<int CT> __global__ void __launch_bounds__ (256)
resample_var_kernel (VarResampleArgs a, int tiles_per_wg, int in_span)
{
  extern __shared__ float s_tab[];                           // lds_floats: sized for the largest table of the launch (occupancy); then the window
  const SpeedCenterDev cd = a.centers[blockIdx.y];
  if ((long long) blockIdx.x * tiles_per_wg * RV_TILE >= cd.n_out)
    return;
  const int stride = cd.stride, n_tab = 257 * stride;
  const bool in_lds = n_tab <= a.lds_floats;
  const bool staged = CT == 2 && in_lds && in_span > 0;
  float2 *s_in = reinterpret_cast<float2 *> (s_tab + ((a.lds_floats + 1) & ~1));
  if (in_lds)
    for (int i = threadIdx.x; i < n_tab; i += blockDim.x)
      s_tab[i] = cd.ctab[i];
  float *out = a.out + blockIdx.y * a.out_stride;
  for (int t = 0; t < tiles_per_wg; t++)
    {
      const long long tile0 = ((long long) blockIdx.x * tiles_per_wg + t) * RV_TILE;
      if (tile0 >= cd.n_out)
        break;                                                                 // (uniform)
      long long first0 = 0;
      if (staged)
        {
          first0 = var_phase (cd, tile0).first;                                // window starts do not decrease with m
          if (t)
            __syncthreads();                                                   // the previous tile's windows have been read
          const float2 *in2 = reinterpret_cast<const float2 *> (a.in);
          for (int i = threadIdx.x; i < in_span; i += 256)
            {
              const long long j = first0 + i;
              s_in[i] = (j >= 0 && j < cd.n_in) ? in2[j] : make_float2 (0.f, 0.f);
            }
        }
      if (in_lds && (staged || t == 0))
        __syncthreads();
      for (int q = 0; q < RV_TILE / 256; q++)
        {
          const long long m = tile0 + q * 256 + threadIdx.x;
          if (m >= cd.n_out)
            break;
          if (staged)
            {
              const VarPhase v = var_phase (cd, m);
              const long long rel = v.first - first0;
              if (rel >= 0 && rel + 2 * cd.hl <= in_span)                      // (always, by the launcher's bound; kept as a guard)
                var_output_staged (cd, s_tab, stride, v, s_in + rel, m, out);
              else
                var_output<CT> (cd, s_tab, stride, a.in, a.n_channels, m, out);
            }
          else if (in_lds)                                       // two copies of the loop: ds_read vs global_load addressing
            var_output<CT> (cd, s_tab, stride, a.in, a.n_channels, m, out);
          else
            var_output<CT> (cd, cd.ctab, stride, a.in, a.n_channels, m, out);
        }
    }
}

/* (measurement knob) bit 0: the stereo input window of a tile through LDS | bit 1: a workgroup keeps its table for several tiles.
 * Measured (tools/gpu_resample_var.py, profiles/r04/resample_var_modes.txt): the stretched copy of a 25 min chunk 1.38 / 1.47 / 1.17 / 1.19 ms
 * for modes 0 / 1 / 2 / 3, get --detect-speed of configs[2] 22.6 - 22.8 / 24.5 / 21.7 / 22.6 ms: keeping the table pays, the window in LDS
 * does not (the tap loop is 17 VALU + 3 LDS instructions per tap pair either way -- zita's 14 roundings per stereo tap pair -- and the
 * window costs a third of the workgroups per compute unit when the table is a large one, 35 KB for the half-rate pass).  Default 2. */
int g_resample_var_mode = 2;
extern "C" void awm_debug_set_resample_var_mode (int mode) { g_resample_var_mode = mode; }

hipError_t
launch_resample_var (hipStream_t st, const VarResampleArgs& args, long long max_n_out, int n_centers)
{
  if (max_n_out <= 0 || n_centers <= 0)
    return hipSuccess;
  VarResampleArgs a = args;
  // LDS for the largest table of the launch (a.max_stride): a small table (ratios near 1: 17 KiB) leaves room for 8 waves
  // per SIMD; tables beyond 48 KiB stay in global memory
  const int want = 257 * a.max_stride;
  a.lds_floats = want <= RV_MAX_TAB ? want : 0;
  const long long n_tiles = (max_n_out + RV_TILE - 1) / RV_TILE;
  const int tiles_per_wg = (g_resample_var_mode & 2) ? int (std::min<long long> (16, std::max<long long> (1, n_tiles * n_centers / 4096))) : 1;   // >= 4096 workgroups first
  const dim3 grid (unsigned ((n_tiles + tiles_per_wg - 1) / tiles_per_wg), unsigned (n_centers));
  const bool aligned = (reinterpret_cast<uintptr_t> (a.in) & 7) == 0 && (reinterpret_cast<uintptr_t> (a.out) & 7) == 0 && (a.out_stride & 1) == 0;
  // input frames the outputs of a tile read: the window start moves by floor ((RV_TILE - 1) * max_step) + 1 at most (+ 1: var_phase's
  // round-up case), plus one window (2 hl <= 2 max_stride)
  const long long span = (long long) ((RV_TILE - 1) * a.max_step) + 3 + 2LL * a.max_stride;
  const bool stage = (g_resample_var_mode & 1) && a.n_channels == 2 && aligned && a.lds_floats > 0 && a.max_step > 0 && span <= RV_MAX_SPAN;
  const int in_span = stage ? int (span) : 0;
  const size_t lds_bytes = size_t ((a.lds_floats + 1) & ~1) * sizeof (float) + size_t (in_span) * sizeof (float2);
  if (a.n_channels == 2 && aligned)
    hipLaunchKernelGGL (resample_var_kernel<2>, grid, dim3 (256), lds_bytes, st, a, tiles_per_wg, in_span);
  else
    hipLaunchKernelGGL (resample_var_kernel<0>, grid, dim3 (256), lds_bytes, st, a, tiles_per_wg, in_span);
  return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------------------------------
 * K13: one workgroup = 64 consecutive rows (hop 128 at half rate) of one centre speed.
 *   phase 1: a wave transforms a row: the 512 windowed samples of TWO channels ride in the real and imaginary part of one
 *            complex FFT-512 (X_a[k] = (Z[k] + conj Z[512-k]) / 2, X_b[k] = (Z[k] - conj Z[512-k]) / 2i), dB of the 81 bands
 *            summed over the channels in channel order (wmspeed.cc:232-246).  A channel whose frame is digital silence is
 *            not read out of the shared transform (the other channel's rounding noise would stand where the reference has
 *            exact zeros = -96 dB): it contributes -96 dB per band directly.
 *   phase 2: umag / dmag of the 510 sync frames (wmspeed.cc:247-257): a wave takes a column, its lanes are the 64 rows.
 *            The band list of the column is wave-uniform (scalar loads, scalar byte extraction); the dB tile has a row
 *            stride of 81 words, so the 64 lanes read 64 different banks, and a column's rows leave as one 512 byte store.
 *            (First version: thread = (column, row) over 16 rows with the band bytes in LDS: 1.45 ms per pass, bank
 *            conflicts and per-lane byte extraction; this one: see DESIGN.md.)
 * ------------------------------------------------------------------------------------------------------------------ */
typedef const unsigned int __attribute__ ((address_space (4))) *const_uint_ptr;

__global__ void __launch_bounds__ (256)
speed_mags_kernel (DevTables t, SpeedMagsArgs a)
{
  __shared__ float2 s_tw[512];
  __shared__ float  s_win[512];
  __shared__ float2 s_x[4][XBUF_ELEMS];
  __shared__ float  s_db[SPEED_TILE * SPEED_NB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const SpeedCenterDev cd = a.centers[blockIdx.y];
  const int row0 = blockIdx.x * SPEED_TILE;
  if (row0 >= cd.rows)
    return;
  fft512_load_twiddles (t.tw512, s_tw);
  for (int i = threadIdx.x; i < 512; i += blockDim.x)
    s_win[i] = a.window512[i];
  __syncthreads();

  const int C = a.n_channels;
  const float *sub = a.sub + blockIdx.y * a.sub_stride;
  float2 *xbuf = s_x[wave];
  for (int r = wave; r < SPEED_TILE; r += 4)
    {
      const int row = row0 + r;
      if (row >= cd.rows)
        break;
      const long long pos = (long long) row * 128;
      float acc0 = 0.f, acc1 = 0.f;                            // bands 20 + lane, 84 + lane
      for (int c0 = 0; c0 < C; c0 += 2)
        {
          const bool two = c0 + 1 < C;
          float2 z[8];
          bool nz0 = false, nz1 = false;
#pragma unroll
          for (int j = 0; j < 8; j++)
            {
              const int n = lane + 64 * j;
              const float x0 = sub[(pos + n) * C + c0];
              const float x1 = two ? sub[(pos + n) * C + c0 + 1] : 0.f;
              nz0 |= x0 != 0.f;
              nz1 |= x1 != 0.f;
              z[j] = make_float2 (__fmul_rn (x0, s_win[n]), __fmul_rn (x1, s_win[n]));
            }
          const bool live0 = __any (nz0), live1 = __any (nz1);
          fft512_forward (z, xbuf, s_tw, lane);
          xbuf[0 * 64 + lane] = z[0];
          xbuf[1 * 64 + lane] = z[1];
          xbuf[6 * 64 + lane] = z[6];
          xbuf[7 * 64 + lane] = z[7];
          wave_sync();
#pragma unroll
          for (int part = 0; part < 2; part++)
            {
              const int k = SPEED_MIN_BAND + 64 * part + lane;
              if (k <= 100)
                {
                  const float2 zk = xbuf[zpos (k)], zm = xbuf[zpos (512 - k)];
                  const float2 xa = make_float2 (0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                  const float2 xb = make_float2 (0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
                  float v = part ? acc1 : acc0;
                  v = __fadd_rn (v, live0 ? db_from_complex (xa) : -96.f);
                  if (two)
                    v = __fadd_rn (v, live1 ? db_from_complex (xb) : -96.f);
                  if (part)
                    acc1 = v;
                  else
                    acc0 = v;
                }
            }
          wave_sync();
        }
      s_db[r * SPEED_NB + lane] = acc0;
      if (lane < SPEED_NB - 64)
        s_db[r * SPEED_NB + 64 + lane] = acc1;
    }
  __syncthreads();

  float2 *mags = a.mags + blockIdx.y * a.mags_center_stride;
  const float *db = s_db + lane * SPEED_NB;                    // this lane's row of the tile
  const bool store = row0 + lane < cd.rows;
  const_uint_ptr cols = (const_uint_ptr) a.cols;
  for (int col = __builtin_amdgcn_readfirstlane (wave); col < SPEED_COLS; col += 4)
    {
      unsigned int cw[15];
#pragma unroll
      for (int i = 0; i < 15; i++)
        cw[i] = cols[col * 16 + i];
      float u = 0.f, d = 0.f;
#pragma unroll
      for (int i = 0; i < 30; i++)
        u = __fadd_rn (u, db[(cw[i >> 2] >> (8 * (i & 3))) & 0xff]);
#pragma unroll
      for (int i = 30; i < 60; i++)
        d = __fadd_rn (d, db[(cw[i >> 2] >> (8 * (i & 3))) & 0xff]);
      if (store)
        mags[(long long) col * a.ld + row0 + lane] = make_float2 (u, d);
    }
}

hipError_t
launch_speed_mags (hipStream_t st, const DevTables& t, const SpeedMagsArgs& a, int max_rows, int n_centers)
{
  if (max_rows <= 0 || n_centers <= 0)
    return hipSuccess;
  hipLaunchKernelGGL (speed_mags_kernel, dim3 (unsigned ((max_rows + SPEED_TILE - 1) / SPEED_TILE), unsigned (n_centers)), dim3 (256), 0, st, t, a);
  return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------------------------------
 * K14: per (centre, relative speed) pair and candidate block start ("state", offset -pad_start .. -1 in steps of
 * sync_search_step, scaled to Q16 by 1 / relative speed; wmspeed.cc:330-344): the state visits the sync frames of three
 * consecutive blocks (compare_bits<0..2>, :270-328): row = (offset + frame_offset) >> 16, used when the sum is not negative
 * and the row exists -- the reference's begin / end iterators are exactly this test because the frame offsets grow
 * monotonically.  Sums per sync bit are float, in the order block 0, 1, 2 and frame ascending; odd blocks swap up and down.
 * The best normalised quality over the states (:346-371) is all that survives: a 64 bit atomic max on the bits of the
 * (non-negative) double.
 *
 * Thread = one state for NS relative speeds of ONE centre at a time (registers: NS x (u, d, n) per sync bit).  The relative
 * speeds of a centre differ by < 1 %, i.e. a state's rows for them lie within a few dozen rows of each other: the NS
 * gathers of a column hit the same cache lines, so the matrix is read from HBM once per NS speeds.  (One workgroup per
 * (speed, state range) read 11.7 GB for 1 GB of matrices in the first pass and ran at HBM speed: 1.56 ms.)
 * ------------------------------------------------------------------------------------------------------------------ */
template<int NS> __global__ void __launch_bounds__ (256)
speed_compare_kernel (SpeedCompareArgs a)
{
  constexpr int BIT_COLS = 3 * 85;                             // columns of one sync bit in the three blocks
  __shared__ int2   s_fo[NS][BIT_COLS];                        // frame offsets in Q16 (whole rows x 8, fraction) of the current bit
  __shared__ double s_best[NS][4];
  const int center = blockIdx.y;
  // fold_groups = G: the G groups of speeds of a (state range, centre) are the workgroups x = 8 G (r / 8) + 8 g + r % 8 -- G workgroups
  // that are dispatched within 8 G places of each other AND on the same XCD (x mod 8, the grid's width is a multiple of 8), so that
  // the second group finds the rows of the matrix in that XCD's L2
  int range = blockIdx.x, group = blockIdx.z;
  if (a.fold_groups)
    {
      const int span = 8 * a.fold_groups, rem = range % span;
      group = rem >> 3;
      range = (range / span) * 8 + (rem & 7);
      if (range >= a.n_ranges)
        return;                                                // (uniform for the workgroup)
    }
  const SpeedCenterDev cd = a.centers[center];
  const int s0 = group * NS;
  const int ns = a.items_per_center - s0 < NS ? a.items_per_center - s0 : NS;       // speeds this workgroup has
  const SpeedItemDev *items = a.items + center * a.items_per_center + s0;
  const int state = range * blockDim.x + threadIdx.x;
  const int wave_state = range * blockDim.x + (threadIdx.x & ~63);
  const long long rows = cd.rows;
  const unsigned n_rows8 = unsigned (rows) * 8u;
  const bool active = state < a.pad_start && rows > 0;

  // per speed: this state's offset, and the interval of "global" frames (block * frames_per_block + frame) any lane of the
  // wave can use (offsets grow with the state, frame offsets with the frame; a frame of margin, the exact test is per lane)
  int off_rows8[NS], off_frac[NS];                            // whole rows x 8 (byte offset of a float2 row), Q16 fraction
  int g_lo = 0x7fffffff, g_hi = -0x7fffffff;
#pragma unroll
  for (int k = 0; k < NS; k++)
    {
      const SpeedItemDev it = items[k < ns ? k : 0];
      const auto offset_of = [&] (int st) {
        st = st < a.pad_start ? st : a.pad_start - 1;
        const double scaled = (st - a.pad_start) * it.q16_scale;
        return (int) scaled;
      };
      const int offset = offset_of (state);
      off_rows8[k] = (offset >> 16) * 8;
      off_frac[k] = offset & 0xffff;
      const long long o_min = offset_of (wave_state), o_max = offset_of (wave_state + 63), limit = rows << 16;
      const double steps_per_g = a.steps_per_frame * it.rel_speed_inv;
      const int lo = int (floor ((double (-o_max) / 65536.0 - 0.5) / steps_per_g)) - 1;
      const int hi = int (ceil ((double (limit - o_min) / 65536.0 - 0.5) / steps_per_g)) + 1;
      g_lo = lo < g_lo ? lo : g_lo;
      g_hi = hi > g_hi ? hi : g_hi;
    }
  const float2 *mags = a.mags + center * a.mags_center_stride;
  const unsigned ld = unsigned (a.ld);
  double q[NS];
  int total[NS];
#pragma unroll
  for (int k = 0; k < NS; k++)
    {
      q[k] = 0;
      total[k] = 0;
    }
  for (int bit = 0; bit < 6; bit++)
    {
      __syncthreads();                                         // the previous bit's offsets are still being read
      for (int e = threadIdx.x; e < NS * BIT_COLS; e += blockDim.x)
        {
          const int k = e / BIT_COLS, c = e - k * BIT_COLS;
          const int block = c / 85, j = c - block * 85;
          const int steps = (block * a.frames_per_block + a.col_frame[bit * a.rows_per_bit + j]) * a.steps_per_frame;
          double v = steps * items[k < ns ? k : 0].rel_speed_inv;
          v = v + 0.5;
          v = v * 65536.0;
          const long long fo = (long long) v;
          s_fo[k][c] = make_int2 (int (fo >> 16) * 8, int (fo & 0xffff));
        }
      __syncthreads();
      float u[NS], d[NS];
      int n[NS];
#pragma unroll
      for (int k = 0; k < NS; k++)
        {
          u[k] = d[k] = 0.f;
          n[k] = 0;
        }
      if (active)
        {
          const unsigned char *first = a.col_first + bit * (a.frames_per_block + 2);
          const float2 *mc = mags + (long long) bit * a.rows_per_bit * a.ld;
          for (int block = 0; block < 3; block++)
            {
              int f_lo = g_lo - block * a.frames_per_block, f_hi = g_hi - block * a.frames_per_block;
              f_lo = f_lo < 0 ? 0 : f_lo;
              f_hi = f_hi > a.frames_per_block - 1 ? a.frames_per_block - 1 : f_hi;
              if (f_lo > f_hi)
                continue;
              const int j_lo = __builtin_amdgcn_readfirstlane (first[f_lo]), j_hi = __builtin_amdgcn_readfirstlane (first[f_hi + 1]);
              const bool swap = block & 1;
              // branch free: the rows come through a buffer descriptor of the column (n_rows x 8 bytes), whose range check returns zeros
              // for a row outside the matrix (a negative row is a huge unsigned offset) -- x + 0.0f is x.  8 VALU instructions per
              // (column, speed) instead of 11 with a select for the index and two for the values.
              // The loads of column j + 1 are issued before the additions of column j: twice the loads in flight per wave (the column loop was
              // one round trip to L2 per column: NS loads, wait, 2 NS additions).  Two register sets, the loop unrolled by two (no copies), no
              // branch inside the loop body (at a join the wait counts would have to assume the shorter path and drain everything).
              // Measured (configs[2], 9 launches per call): 0.511 -> 0.450 ms per launch; three sets in flight (a column past the end loaded
              // from outside the descriptor's range, i.e. adding +0): 0.512 -- the compiler drains the queue inside the unrolled body.
              typedef decltype (__builtin_amdgcn_raw_buffer_load_b64 (__amdgpu_buffer_rsrc_t(), 0, 0, 0)) row_t;
              auto issue = [&] (int j, unsigned (&idx8)[NS], row_t (&m)[NS]) {
                const float2 *col = mc + unsigned (j) * ld;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc (const_cast<float2 *> (col), (short) 0, int (n_rows8), 0x00020000);
#pragma unroll
                for (int k = 0; k < NS; k++)
                  {
                    const int2 f = s_fo[k][block * 85 + j];                     // (whole rows x 8, fraction)
                    idx8[k] = unsigned (off_rows8[k] + f.x) + ((unsigned (off_frac[k] + f.y) >> 16) << 3);
                  }
#pragma unroll
                for (int k = 0; k < NS; k++)                                    // (all NS loads in flight)
                  m[k] = __builtin_amdgcn_raw_buffer_load_b64 (rs, int (idx8[k]), 0, 0);
              };
              auto accumulate = [&] (const unsigned (&idx8)[NS], const row_t (&m)[NS]) {
#pragma unroll
                for (int k = 0; k < NS; k++)
                  {
                    u[k] = __fadd_rn (u[k], __uint_as_float (swap ? m[k][1] : m[k][0]));
                    d[k] = __fadd_rn (d[k], __uint_as_float (swap ? m[k][0] : m[k][1]));
                    n[k] += idx8[k] < n_rows8;
                  }
              };
              unsigned idx_a[NS], idx_b[NS];
              row_t m_a[NS], m_b[NS];
              if (j_lo >= j_hi)                                                 // (no sync frame of this bit in the range)
                continue;
              int j = j_lo;
              issue (j, idx_a, m_a);
              for (; j + 1 < j_hi; j += 2)
                {
                  issue (j + 1, idx_b, m_b);
                  accumulate (idx_a, m_a);
                  issue (j + 2 < j_hi ? j + 2 : j_hi - 1, idx_a, m_a);          // (past the end: the last column again, not added)
                  accumulate (idx_b, m_b);
                }
              if (j < j_hi)                                                     // odd count: the last column is in set a
                accumulate (idx_a, m_a);
            }
        }
#pragma unroll
      for (int k = 0; k < NS; k++)
        {
          float raw;                                        // SyncFinder::bit_quality (reference syncfinder.cc:94-114)
          if (u[k] == 0 || d[k] == 0)
            raw = 0;
          else if (u[k] < d[k])
            raw = __fsub_rn (1.f, __fdiv_rn (u[k], d[k]));
          else
            raw = __fsub_rn (__fdiv_rn (d[k], u[k]), 1.f);
          const double rb = (bit & 1) ? double (raw) : -double (raw);
          q[k] += rb * n[k];
          total[k] += n[k];
        }
    }
#pragma unroll
  for (int k = 0; k < NS; k++)
    {
      double v = 0;
      if (total[k])
        {
          v = q[k] / total[k];
          v = v / a.min_delta / 2.9;                        // normalize_sync_quality
          v = fabs (v);
        }
      for (int o = 32; o > 0; o >>= 1)
        v = fmax (v, __shfl_xor (v, o));
      if ((threadIdx.x & 63) == 0)
        s_best[k][threadIdx.x >> 6] = v;
    }
  __syncthreads();
  if (threadIdx.x < ns)
    {
      const int k = threadIdx.x;
      const double best = fmax (fmax (s_best[k][0], s_best[k][1]), fmax (s_best[k][2], s_best[k][3]));
      if (best > 0)
        atomicMax (a.best + center * a.items_per_center + s0 + k, (unsigned long long) __double_as_longlong (best));
    }
}

// (A / B, tools/gpu_speed_compare_ab.py, round 6: 1 = all 11 relative speeds of a centre in one thread -- the matrix gathered once, but 179
// registers = two waves per SIMD for a kernel that lives on the latency of its gathers: 0.482 against 0.447 ms per launch; off)
int g_speed_compare_wide = 0;
extern "C" void awm_debug_set_speed_compare_wide (int on) { g_speed_compare_wide = on; }
// (A / B, same tool: 1 = the groups of six of a (state range, centre) as neighbours on one XCD instead of a grid dimension of their own --
// the second group's gathers find the first one's lines in that XCD's L2: 0.447 against 0.454 ms per launch, results identical; on)
int g_speed_compare_fold = 1;
extern "C" void awm_debug_set_speed_compare_fold (int on) { g_speed_compare_fold = on; }

hipError_t
launch_speed_compare (hipStream_t st, const SpeedCompareArgs& a, int n_items)
{
  if (n_items <= 0 || a.n_centers <= 0 || a.items_per_center <= 0)
    return hipSuccess;
  if (a.rows_per_bit != 85 || n_items != a.n_centers * a.items_per_center)
    return hipErrorInvalidValue;
  const unsigned ranges = unsigned ((a.pad_start + 255) / 256);
  const int per = a.items_per_center;
  // six speeds per thread where that still leaves enough workgroups to fill the chip (the first pass: 57 centres x 11 or 23
  // speeds); the small refinement passes get one speed per thread and more workgroups instead
  const unsigned groups6 = unsigned ((per + 5) / 6);
  // ... and ALL of a centre's relative speeds (11 in the first pass of a stereo stream) in one thread where they fit: the centre's
  // matrix is then gathered once instead of once per group of six (round 5's counters: 2.18 GB fetched per launch for 1.0 GB of
  // matrices, the two groups of a centre run on different XCDs)
  if (g_speed_compare_wide && per > 6 && per <= 12 && (long long) ranges * a.n_centers >= 1024)
    hipLaunchKernelGGL (speed_compare_kernel<12>, dim3 (ranges, unsigned (a.n_centers), 1), dim3 (256), 0, st, a);
  else if ((long long) ranges * a.n_centers * groups6 >= 1024 && g_speed_compare_fold && groups6 > 1)
    {
      SpeedCompareArgs f = a;
      f.fold_groups = int (groups6);
      f.n_ranges = int (ranges);
      hipLaunchKernelGGL (speed_compare_kernel<6>, dim3 ((ranges + 7) / 8 * 8 * groups6, unsigned (a.n_centers), 1), dim3 (256), 0, st, f);
    }
  else if ((long long) ranges * a.n_centers * groups6 >= 1024)
    hipLaunchKernelGGL (speed_compare_kernel<6>, dim3 (ranges, unsigned (a.n_centers), groups6), dim3 (256), 0, st, a);
  else
    hipLaunchKernelGGL (speed_compare_kernel<1>, dim3 (ranges, unsigned (a.n_centers), unsigned (per)), dim3 (256), 0, st, a);
  return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------------------------------
 * K15: get_clip_locations hashes a pseudo random subset of the samples (wmspeed.cc:533-553): the positions come from the
 * host's AES-CTR generator, the device only gathers.  get_best_clip_location (:555-577) compares the energies of the
 * candidate clips: float squares summed in double (the reference adds them one by one, here in a tree: the energies
 * agree to ~1e-15 relative, which only matters for clips of equal energy).
 * ------------------------------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__ (256)
gather_values_kernel (const float *in, const unsigned long long *pos, long long n, float *out)
{
  const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    out[i] = in[pos[i]];
}

hipError_t
launch_gather_values (hipStream_t st, const float *in, const unsigned long long *pos, long long n, float *out)
{
  if (n <= 0)
    return hipSuccess;
  hipLaunchKernelGGL (gather_values_kernel, dim3 (unsigned ((n + 255) / 256)), dim3 (256), 0, st, in, pos, n, out);
  return hipGetLastError();
}

__global__ void __launch_bounds__ (256)
energy_kernel (const float *in, const long long *range, double *out)
{
  __shared__ double s_part[4];
  const long long begin = range[2 * blockIdx.y], end = range[2 * blockIdx.y + 1];
  double e = 0;
  for (long long i = begin + (long long) blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (long long) gridDim.x * blockDim.x)
    {
      const float s = in[i];
      e += double (__fmul_rn (s, s));
    }
  for (int o = 32; o > 0; o >>= 1)
    e += __shfl_xor (e, o);
  if ((threadIdx.x & 63) == 0)
    s_part[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0)
    out[blockIdx.y * gridDim.x + blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

// out: [n_ranges][ENERGY_PARTS] partial sums (added up by the host in index order: deterministic)
hipError_t
launch_energy (hipStream_t st, const float *in, const long long *range, int n_ranges, double *out)
{
  if (n_ranges <= 0)
    return hipSuccess;
  hipLaunchKernelGGL (energy_kernel, dim3 (ENERGY_PARTS, unsigned (n_ranges)), dim3 (256), 0, st, in, range, out);
  return hipGetLastError();
}

} // namespace awmk
