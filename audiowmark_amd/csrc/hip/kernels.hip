// kernels.hip -- hand-written gfx950 kernels of the audiowmark spectral path.
//
//   K1 stft_full_kernel    FFTAnalyzer::run_fft / fft_range        (reference wmcommon.cc:91-141)
//   K2 add_mix_kernel      run_fft + apply_frame_mod + c2r + 3-frame windowed overlap-add + mix
//                          + per-second max|x|                     (reference wmadd.cc:61-84,215-250,297-317,564-565; limiter.cc:90-97)
//   K3 limiter_kernel      Limiter::process_block ramp             (reference limiter.cc:99-124)
//   K4 sync_db_kernel      SyncFinder::sync_fft (STFT -> dB, 81 bands) (reference syncfinder.cc:560-605)
//   K5 sync_scan_kernel    SyncFinder::sync_decode + bit_quality   (reference syncfinder.cc:80-153)
//   K5b local_mean_kernel  local mean of search_approx             (reference syncfinder.cc:234-254)
//   K7 soft_bits_kernel    mix_decode                              (reference wmget.cc:67-108)
//
// All FFT work is "one wavefront per frame-channel" (awm_fft.hip.h); workgroups are 4 independent
// waves that only share read-only LDS tables, so there is no barrier in any steady-state loop.
// Arithmetic whose rounding is visible in decisions downstream (dB conversion, the sequential
// float/double accumulations of sync_decode / mix_decode, the overlap-add and limiter expressions)
// follows the reference's operation order with explicit __fmul_rn/__fadd_rn (the reference is
// compiled for baseline x86-64: no FMA contraction).
#include "kernels.hh"
#include "awm_fft.hip.h"
#include <cstdlib>
#include <algorithm>
#include <type_traits>

namespace awmk {

constexpr int NB = 81;          // watermark bands: bins 20..100
constexpr int MIN_BAND = 20;
constexpr int WAVES = 4;        // waves per workgroup

// read-only, wave-uniform tables are addressed through the constant address space so that the compiler emits
// scalar loads (s_load_dwordx16) instead of 64 identical vector loads
typedef const int __attribute__ ((address_space (4))) *const_int_ptr;

__device__ __forceinline__ void
load_shared_tables (const DevTables& t, float2 *s_tw, float *s_win, float2 *s_twb)
{
  fft512_load_twiddles (t.tw512, s_tw);
  for (int i = threadIdx.x; i < 1024; i += blockDim.x)
    s_win[i] = t.window[i];
  if (s_twb)
    for (int i = threadIdx.x; i < NB; i += blockDim.x)
      s_twb[i] = t.tw1024[MIN_BAND + i];
}

/* ------------------------------------------------------------------------------------------
 * sample fetch: lane owns x[2 (lane + 64 j)] and x[2 (lane + 64 j) + 1], j = 0..7
 * ------------------------------------------------------------------------------------------ */

// stereo, `idx` = first sample (per channel) of the frame; avail = valid samples from idx (<= 1024)
__device__ __forceinline__ void
fetch_stereo (const float *pcm, long long idx, int avail, int lane, float (&l)[16], float (&r)[16])
{
  const float *p = pcm + idx * 2;
  if (avail >= 1024 && ((reinterpret_cast<uintptr_t> (p) & 15) == 0))
    {
      const float4 *p4 = reinterpret_cast<const float4 *> (p);
#pragma unroll
      for (int j = 0; j < 8; j++)
        {
          const float4 v = p4[lane + 64 * j];
          l[2 * j] = v.x; r[2 * j] = v.y; l[2 * j + 1] = v.z; r[2 * j + 1] = v.w;
        }
    }
  else if (avail >= 1024)
    {
      const float2 *p2 = reinterpret_cast<const float2 *> (p);
#pragma unroll
      for (int j = 0; j < 8; j++)
        {
          const float2 a = p2[2 * (lane + 64 * j)], b = p2[2 * (lane + 64 * j) + 1];
          l[2 * j] = a.x; r[2 * j] = a.y; l[2 * j + 1] = b.x; r[2 * j + 1] = b.y;
        }
    }
  else
    {
#pragma unroll
      for (int j = 0; j < 8; j++)
        {
          const int x = 2 * (lane + 64 * j);
          l[2 * j]     = x < avail     ? p[2 * x]     : 0.f;
          r[2 * j]     = x < avail     ? p[2 * x + 1] : 0.f;
          l[2 * j + 1] = x + 1 < avail ? p[2 * x + 2] : 0.f;
          r[2 * j + 1] = x + 1 < avail ? p[2 * x + 3] : 0.f;
        }
    }
}

// one channel `ch` of a C-channel interleaved stream
__device__ __forceinline__ void
fetch_channel (const float *pcm, long long idx, int avail, int C, int ch, int lane, float (&v)[16])
{
  const float *p = pcm + idx * C + ch;
  if (C == 1 && avail >= 1024 && ((reinterpret_cast<uintptr_t> (p) & 7) == 0))
    {
      const float2 *p2 = reinterpret_cast<const float2 *> (p);
#pragma unroll
      for (int j = 0; j < 8; j++)
        {
          const float2 a = p2[lane + 64 * j];
          v[2 * j] = a.x; v[2 * j + 1] = a.y;
        }
    }
  else
    {
#pragma unroll
      for (int j = 0; j < 8; j++)
        {
          const int x = 2 * (lane + 64 * j);
          v[2 * j]     = x < avail     ? p[(long long) x * C]       : 0.f;
          v[2 * j + 1] = x + 1 < avail ? p[(long long) (x + 1) * C] : 0.f;
        }
    }
}

// window + pack: frame[x] = sample * window[x] (float multiply, reference wmcommon.cc:106-110)
__device__ __forceinline__ void
window_pack (const float (&v)[16], const float *s_win, int lane, float2 (&z)[8])
{
  const float2 *w2 = reinterpret_cast<const float2 *> (s_win);
#pragma unroll
  for (int j = 0; j < 8; j++)
    {
      const float2 w = lds_ld (&w2[lane + 64 * j]);
      z[j] = make_float2 (__fmul_rn (v[2 * j], w.x), __fmul_rn (v[2 * j + 1], w.y));
    }
}

/* ==========================================================================================
 * K1: full 513-bin STFT
 * ========================================================================================== */
__global__ void __launch_bounds__ (64 * WAVES)
stft_full_kernel (DevTables t, const float *pcm, int C, long long start_index, long long hop, long long frame_count, float2 *out)
{
  __shared__ float2 s_tw[512];
  __shared__ float  s_win[1024];
  __shared__ float2 s_x[WAVES][XBUF_ELEMS];
  load_shared_tables (t, s_tw, s_win, nullptr);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long item = (long long) blockIdx.x * WAVES + wave;
  if (item >= frame_count * C)
    return;
  const long long f = item / C;
  const int ch = int (item % C);
  float2 *xbuf = s_x[wave];

  float v[16];
  fetch_channel (pcm, start_index + f * hop, 1024, C, ch, lane, v);
  float2 z[8];
  window_pack (v, s_win, lane, z);
  fft512_forward (z, xbuf, s_tw, lane);
#pragma unroll
  for (int kc = 0; kc < 8; kc++)
    xbuf[kc * 64 + lane] = z[kc];
  wave_sync();
  float2 *o = out + item * 513;
  for (int k = lane; k <= 512; k += 64)
    {
      const float2 zk = xbuf[zpos (k & 511)], zm = xbuf[zpos ((512 - k) & 511)];
      o[k] = real_split (zk, zm, t.tw1024[k]);
    }
}

hipError_t
launch_stft_full (hipStream_t st, const DevTables& t, const float *pcm, int n_channels,
                  long long start_index, long long hop, long long frame_count, float2 *out)
{
  const long long items = frame_count * n_channels;
  if (items <= 0)
    return hipSuccess;
  const unsigned grid = unsigned ((items + WAVES - 1) / WAVES);
  hipLaunchKernelGGL (stft_full_kernel, dim3 (grid), dim3 (64 * WAVES), 0, st, t, pcm, n_channels, start_index, hop, frame_count, out);
  return hipGetLastError();
}

/* ==========================================================================================
 * K2: fused add
 * ========================================================================================== */

__device__ __forceinline__ int
zdpos (int k)
{
  const int kc = k >> 6;                      // 0, 1, 6, 7 for the watermark bands and their mirrors
  return (kc < 2 ? kc : kc - 4) * 64 + ((k >> 3) & 7) + 8 * (k & 7);
}

// one frame-channel: windowed samples -> delta signal d (time domain, unnormalised c2r like FFTW)
__device__ __forceinline__ void
frame_delta (float2 (&z)[8], const int8_t *mod_row, float nd_up, float nd_down,
             float2 *xbuf, float2 *zd, const float2 *s_tw, const float2 *s_tw3, const float2 *s_twb, int lane)
{
  fft512_forward (z, xbuf, s_tw, lane);
  xbuf[0 * 64 + lane] = z[0];
  xbuf[1 * 64 + lane] = z[1];
  xbuf[6 * 64 + lane] = z[6];
  xbuf[7 * 64 + lane] = z[7];
  wave_sync();
#pragma unroll
  for (int pass = 0; pass < 2; pass++)
    {
      const int k = MIN_BAND + lane + 64 * pass;
      if (k <= 100)
        {
          const float2 w = s_twb[k - MIN_BAND];
          const float2 X = real_split (xbuf[zpos (k)], xbuf[zpos (512 - k)], w);
          const int mod = mod_row[k - MIN_BAND];
          float2 D = make_float2 (0.f, 0.f);
          if (mod)
            {
              // apply_frame_mod (reference wmadd.cc:61-84): D = X (|X|^e - 1) for |X| > 1e-7, e = -+ water_delta.
              // |X|^e = exp2 (e / 2 * log2 (|X|^2)) on the hardware's v_log_f32 / v_exp_f32 (1 ulp each; the exponent
              // e / 2 * log2 is at most ~0.3 in magnitude, so the factor is good to ~2e-7 relative and the watermark
              // signal -- 1 % of the spectrum -- to ~1e-8 of the sample scale; libm's hypotf + powf cost 250 VALU
              // instructions per pass here, a sixth of the kernel).  |X|^2 of a bin that passes the test is a normal float.
              const float abs2 = X.x * X.x + X.y * X.y;
              if (abs2 > 1e-14f)
                {
                  const float s = __builtin_amdgcn_exp2f (__builtin_amdgcn_logf (abs2) * (0.5f * (mod == 1 ? nd_up : nd_down))) - 1.0f;
                  D = make_float2 (X.x * s, X.y * s);
                }
            }
          // half-length spectrum of the c2r transform: Zd[k] = D + i D conj(W^k), Zd[512-k] = conj(D) + i conj(D conj(W^k))
          const float2 O = cmulc (D, w);
          zd[zdpos (k)]       = make_float2 (D.x - O.y, D.y + O.x);
          zd[zdpos (512 - k)] = make_float2 (D.x + O.y, O.x - D.y);
        }
    }
  wave_sync();
  const float2 zero = make_float2 (0.f, 0.f);
  z[0] = zd[0 * 64 + lane];
  z[1] = zd[1 * 64 + lane];
  z[2] = zero; z[3] = zero; z[4] = zero; z[5] = zero;
  z[6] = zd[2 * 64 + lane];
  z[7] = zd[3 * 64 + lane];
  wave_sync();
  fft512_inverse (z, xbuf, s_tw, s_tw3, lane);
}

// both channels of a stereo frame: the two transforms pipelined over the wave's one exchange tile (fft512_forward2 / fft512_inverse2);
// the frame_mod row is the same for both, the band edit of a bin runs for both at once
__device__ __forceinline__ void
frame_delta2 (float2 (&za)[8], float2 (&zb)[8], const int8_t *mod_row, float nd_up, float nd_down,
              float2 *xbuf, float2 *zd, const float2 *s_tw, const float2 *s_tw3, const float2 *s_twb, int lane)
{
  fft512_forward2 (za, zb, xbuf, s_tw, lane);
  // rows with the bands and their mirrors: a in rows 0, 1, 6, 7 of the tile (zpos), b in the rows between (zpos + 128 / - 128)
  lds_st (&xbuf[0 * 64 + lane], za[0]);
  lds_st (&xbuf[1 * 64 + lane], za[1]);
  lds_st (&xbuf[6 * 64 + lane], za[6]);
  lds_st (&xbuf[7 * 64 + lane], za[7]);
  lds_st (&xbuf[2 * 64 + lane], zb[0]);
  lds_st (&xbuf[3 * 64 + lane], zb[1]);
  lds_st (&xbuf[4 * 64 + lane], zb[6]);
  lds_st (&xbuf[5 * 64 + lane], zb[7]);
  wave_sync_pinned();
  auto zposb = [] (int k) { const int p = zpos (k); return p < 128 ? p + 128 : p - 128; };
  float2 Da[2], Db[2], Oa[2], Ob[2];
#pragma unroll
  for (int pass = 0; pass < 2; pass++)
    {
      const int k = MIN_BAND + lane + 64 * pass;
      Da[pass] = Db[pass] = Oa[pass] = Ob[pass] = make_float2 (0.f, 0.f);
      if (k <= 100)
        {
          const float2 w = s_twb[k - MIN_BAND];
          const float2 Xa = real_split (lds_ld (&xbuf[zpos (k)]), lds_ld (&xbuf[zpos (512 - k)]), w);
          const float2 Xb = real_split (lds_ld (&xbuf[zposb (k)]), lds_ld (&xbuf[zposb (512 - k)]), w);
          const int mod = mod_row[k - MIN_BAND];
          if (mod)
            {
              // apply_frame_mod (reference wmadd.cc:61-84), see frame_delta
              const float e = 0.5f * (mod == 1 ? nd_up : nd_down);
              const float abs2a = Xa.x * Xa.x + Xa.y * Xa.y, abs2b = Xb.x * Xb.x + Xb.y * Xb.y;
              if (abs2a > 1e-14f)
                {
                  const float s = __builtin_amdgcn_exp2f (__builtin_amdgcn_logf (abs2a) * e) - 1.0f;
                  Da[pass] = make_float2 (Xa.x * s, Xa.y * s);
                }
              if (abs2b > 1e-14f)
                {
                  const float s = __builtin_amdgcn_exp2f (__builtin_amdgcn_logf (abs2b) * e) - 1.0f;
                  Db[pass] = make_float2 (Xb.x * s, Xb.y * s);
                }
            }
          Oa[pass] = cmulc (Da[pass], w);
          Ob[pass] = cmulc (Db[pass], w);
        }
    }
  wave_sync_pinned();
  const float2 zero = make_float2 (0.f, 0.f);
#pragma unroll
  for (int pass = 0; pass < 2; pass++)
    {
      const int k = MIN_BAND + lane + 64 * pass;
      if (k <= 100)
        {
          lds_st (&zd[zdpos (k)],       make_float2 (Da[pass].x - Oa[pass].y, Da[pass].y + Oa[pass].x));
          lds_st (&zd[zdpos (512 - k)], make_float2 (Da[pass].x + Oa[pass].y, Oa[pass].x - Da[pass].y));
        }
    }
  wave_sync_pinned();
  za[0] = lds_ld (&zd[0 * 64 + lane]);
  za[1] = lds_ld (&zd[1 * 64 + lane]);
  za[2] = zero; za[3] = zero; za[4] = zero; za[5] = zero;
  za[6] = lds_ld (&zd[2 * 64 + lane]);
  za[7] = lds_ld (&zd[3 * 64 + lane]);
  wave_sync_pinned();
#pragma unroll
  for (int pass = 0; pass < 2; pass++)
    {
      const int k = MIN_BAND + lane + 64 * pass;
      if (k <= 100)
        {
          lds_st (&zd[zdpos (k)],       make_float2 (Db[pass].x - Ob[pass].y, Db[pass].y + Ob[pass].x));
          lds_st (&zd[zdpos (512 - k)], make_float2 (Db[pass].x + Ob[pass].y, Ob[pass].x - Db[pass].y));
        }
    }
  wave_sync_pinned();
  zb[0] = lds_ld (&zd[0 * 64 + lane]);
  zb[1] = lds_ld (&zd[1 * 64 + lane]);
  zb[2] = zero; zb[3] = zero; zb[4] = zero; zb[5] = zero;
  zb[6] = lds_ld (&zd[2 * 64 + lane]);
  zb[7] = lds_ld (&zd[3 * 64 + lane]);
  wave_sync_pinned();
  fft512_inverse2 (za, zb, xbuf, s_tw, s_tw3, lane);
}

__device__ __forceinline__ float
wave_max (float v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v = fmaxf (v, __shfl_xor (v, off));
  return v;
}

template<int CV, bool OPAQUE, bool PAIR = false> __device__ __forceinline__ void
add_mix_body (const DevTables& t, const AddMixArgs& a, long long frame_number0, int block_frames)
{
  __shared__ float2 s_tw[512];
  __shared__ float2 s_tw3[512];                            // twiddles of the inverse transform's last stage (awm_fft.hip.h)
  __shared__ float  s_win[1024];
  __shared__ float2 s_twb[NB];
  __shared__ float2 s_x[WAVES][XBUF_ELEMS];
  __shared__ float2 s_zd[WAVES][256];
  // the wave index is wave-uniform: in a scalar register, the frame loop's counters, bounds tests and base addresses run on the
  // scalar unit (the compiler cannot see that threadIdx.x >> 6 is the same for all lanes)
  const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane (threadIdx.x >> 6);
  int lane = lane0;
  load_shared_tables (t, s_tw, s_win, s_twb);
  fft512_load_twiddles_inverse (t.tw512, s_tw3);
  for (int i = lane; i < 256; i += 64)
    s_zd[wave][i] = make_float2 (0.f, 0.f);
  __syncthreads();

  const int C = a.n_channels;
  const int n_cg = CV == 2 ? 1 : C;
  const long long F = (a.n_frames + 1023) / 1024;
  const int L = a.frames_per_span;
  const long long n_spans = (F + L - 1) / L;
  const long long item = (long long) blockIdx.x * WAVES + wave;
  if (item >= n_spans * n_cg)
    return;
  const long long span = item / n_cg;
  const int ch0 = int (item % n_cg);
  const long long s = span * L, e = (s + L < F) ? s + L : F;
  float2 *xbuf = s_x[wave], *zd = s_zd[wave];

  // synthesis window pieces this lane needs (reference wmadd.cc:177-206): head = samples 2 lane, 2 lane + 1
  // of a frame (slots W1, W2); tail = samples 896 + 2 lane (+1) (slots W1, W0).  In between W1 == 1, W0 == W2 == 0.
  const float2 w1_head = reinterpret_cast<const float2 *> (t.synth + 1024)[lane];
  const float2 w2_head = reinterpret_cast<const float2 *> (t.synth + 2048)[lane];
  const float2 w1_tail = reinterpret_cast<const float2 *> (t.synth + 1024 + 896)[lane];
  const float2 w0_tail = reinterpret_cast<const float2 *> (t.synth + 896)[lane];

  float head2[CV][2], tail_s1[CV][2], tail_in[CV][2];
#pragma unroll
  for (int c = 0; c < CV; c++)
    head2[c][0] = head2[c][1] = tail_s1[c][0] = tail_s1[c][1] = tail_in[c][0] = tail_in[c][1] = 0.f;

  const int BS = a.limiter_block;
  const long long total_rows = 2LL * block_frames;

  // where frame m comes from (halo before / the span / halo after) and how much of it exists
  auto frame_source = [&] (long long m, int& avail) -> const float * {
    const float *src;
    if (m < 0)
      {
        src = a.halo_before;
        avail = src ? 1024 : 0;
      }
    else if (m >= F)
      {
        src = a.halo_after;
        avail = src ? 1024 : 0;
      }
    else
      {
        src = a.pcm_in + m * 1024 * C;
        const long long left = a.n_frames - m * 1024;
        avail = left < 1024 ? int (left) : 1024;
      }
    return src;
  };
  for (long long m = s - 1; m <= e; m++)
    {
      if (OPAQUE)
        {
          // make the lane index opaque per frame: the ~60 lane-dependent twiddle factors are then re-read from LDS
          // for every transform instead of being hoisted into registers for the whole span (occupancy over reuse)
          lane = lane0;
          asm volatile ("" : "+v" (lane));
        }
      int avail;
      const float *src = frame_source (m, avail);
      float in[CV][16];
      float2 d[CV][8];
      if (avail > 0)
        {
          if constexpr (CV == 2)
            fetch_stereo (src, 0, avail, lane, in[0], in[1]);
          else
            fetch_channel (src, 0, avail, C, ch0, lane, in[0]);
          const long long g = a.first_frame + m;                       // frame index in the whole stream
          const long long row = (frame_number0 + g) % total_rows;      // reference wmadd.cc:326-344
          const int8_t *mod_row = a.frame_mod + row * NB;
          if constexpr (CV == 2 && PAIR)
            {
              window_pack (in[0], s_win, lane, d[0]);
              window_pack (in[1], s_win, lane, d[1]);
              frame_delta2 (d[0], d[1], mod_row, a.neg_delta_up, a.neg_delta_down, xbuf, zd, s_tw, s_tw3, s_twb, lane);
            }
          else
            {
#pragma unroll
              for (int c = 0; c < CV; c++)
                {
                  window_pack (in[c], s_win, lane, d[c]);
                  frame_delta (d[c], mod_row, a.neg_delta_up, a.neg_delta_down, xbuf, zd, s_tw, s_tw3, s_twb, lane);
                }
            }
        }
      else
        {
#pragma unroll
          for (int c = 0; c < CV; c++)
            {
#pragma unroll
              for (int j = 0; j < 16; j++)
                in[c][j] = 0.f;
#pragma unroll
              for (int j = 0; j < 8; j++)
                d[c][j] = make_float2 (0.f, 0.f);
            }
        }

      if (a.delta_only)
        {
          // WatermarkGen::run alone: the mix below then adds 0 instead of the input
#pragma unroll
          for (int c = 0; c < CV; c++)
#pragma unroll
            for (int j = 0; j < 16; j++)
              in[c][j] = 0.f;
        }
      const bool own = m >= s && m < e;
      const bool own_prev = m - 1 >= s && m - 1 < e;
      float max0 = 0.f, max1 = 0.f, pmax0 = 0.f, pmax1 = 0.f;
      // limiter blocks touched by frame m / m - 1: offset inside the frame where the next block begins
      const long long gs_m = (a.first_frame + m) * 1024;
      const long long b0 = gs_m >= 0 ? gs_m / BS : 0;
      const long long bound = (b0 + 1) * BS - gs_m;
      const long long gs_p = gs_m - 1024;
      const long long pb0 = gs_p >= 0 ? gs_p / BS : 0;
      const long long pbound = (pb0 + 1) * BS - gs_p;

      // output frame m = d[m-1] W2 + d[m] W1 + d[m+1] W0 + in[m]   (reference wmadd.cc:228-238, 564-565)
      float o[CV][16];
#pragma unroll
      for (int c = 0; c < CV; c++)
        {
          // j = 0: head region, W1/W2 ramps
          const float s1x = __fadd_rn (head2[c][0], __fmul_rn (d[c][0].x, w1_head.x));
          const float s1y = __fadd_rn (head2[c][1], __fmul_rn (d[c][0].y, w1_head.y));
          o[c][0] = __fadd_rn (s1x, in[c][0]);
          o[c][1] = __fadd_rn (s1y, in[c][1]);
          head2[c][0] = __fmul_rn (d[c][0].x, w2_head.x);
          head2[c][1] = __fmul_rn (d[c][0].y, w2_head.y);
#pragma unroll
          for (int j = 1; j < 7; j++)
            {
              o[c][2 * j]     = __fadd_rn (d[c][j].x, in[c][2 * j]);
              o[c][2 * j + 1] = __fadd_rn (d[c][j].y, in[c][2 * j + 1]);
            }
          // j = 7 of the PREVIOUS frame gets its W0 contribution now
          o[c][14] = __fadd_rn (__fadd_rn (tail_s1[c][0], __fmul_rn (d[c][7].x, w0_tail.x)), tail_in[c][0]);
          o[c][15] = __fadd_rn (__fadd_rn (tail_s1[c][1], __fmul_rn (d[c][7].y, w0_tail.y)), tail_in[c][1]);
          tail_s1[c][0] = __fmul_rn (d[c][7].x, w1_tail.x);
          tail_s1[c][1] = __fmul_rn (d[c][7].y, w1_tail.y);
          tail_in[c][0] = in[c][14];
          tail_in[c][1] = in[c][15];
        }

      // stores (bounded by the span's sample count) + maxima.  Common case (all but the last frame of the stream, and 42 of 43
      // frames have no limiter block boundary inside): plain stores and ONE running maximum, no per-value bounds tests.
      const bool whole_m = (m + 1) * 1024 <= a.n_frames;
      if (own && whole_m && bound >= 1024)
        {
          const long long base = m * 1024;
#pragma unroll
          for (int j = 0; j < 7; j++)
            {
              const long long ls = base + 2 * (lane + 64 * j);
              if (CV == 2)
                *reinterpret_cast<float4 *> (a.out + ls * 2) = make_float4 (o[0][2 * j], o[CV - 1][2 * j], o[0][2 * j + 1], o[CV - 1][2 * j + 1]);
              else
                {
                  float *p = a.out + ls * C + ch0;
                  p[0] = o[0][2 * j];
                  p[C] = o[0][2 * j + 1];
                }
#pragma unroll
              for (int c = 0; c < CV; c++)
                max0 = fmaxf (max0, fmaxf (fabsf (o[c][2 * j]), fabsf (o[c][2 * j + 1])));
            }
        }
      else if (own)
        {
          const long long base = m * 1024;
#pragma unroll
          for (int j = 0; j < 7; j++)
            {
              const int x = 2 * (lane + 64 * j);
              const long long ls = base + x;
              if (CV == 2)
                {
                  float *p = a.out + ls * 2;
                  if (ls + 1 < a.n_frames)
                    *reinterpret_cast<float4 *> (p) = make_float4 (o[0][2 * j], o[CV - 1][2 * j], o[0][2 * j + 1], o[CV - 1][2 * j + 1]);
                  else if (ls < a.n_frames)
                    *reinterpret_cast<float2 *> (p) = make_float2 (o[0][2 * j], o[CV - 1][2 * j]);
                }
              else
                {
                  float *p = a.out + ls * C + ch0;
                  if (ls < a.n_frames)
                    p[0] = o[0][2 * j];
                  if (ls + 1 < a.n_frames)
                    p[C] = o[0][2 * j + 1];
                }
#pragma unroll
              for (int c = 0; c < CV; c++)
                {
                  const float v0 = ls < a.n_frames ? fabsf (o[c][2 * j]) : 0.f;
                  const float v1 = ls + 1 < a.n_frames ? fabsf (o[c][2 * j + 1]) : 0.f;
                  if (x < bound) max0 = fmaxf (max0, v0); else max1 = fmaxf (max1, v0);
                  if (x + 1 < bound) max0 = fmaxf (max0, v1); else max1 = fmaxf (max1, v1);
                }
            }
        }
      if (own_prev && m * 1024 <= a.n_frames && pbound >= 1024)
        {
          const long long ls = (m - 1) * 1024 + 896 + 2 * lane;
          if (CV == 2)
            *reinterpret_cast<float4 *> (a.out + ls * 2) = make_float4 (o[0][14], o[CV - 1][14], o[0][15], o[CV - 1][15]);
          else
            {
              float *p = a.out + ls * C + ch0;
              p[0] = o[0][14];
              p[C] = o[0][15];
            }
#pragma unroll
          for (int c = 0; c < CV; c++)
            pmax0 = fmaxf (pmax0, fmaxf (fabsf (o[c][14]), fabsf (o[c][15])));
        }
      else if (own_prev)
        {
          const int x = 896 + 2 * lane;
          const long long ls = (m - 1) * 1024 + x;
          if (CV == 2)
            {
              float *p = a.out + ls * 2;
              if (ls + 1 < a.n_frames)
                *reinterpret_cast<float4 *> (p) = make_float4 (o[0][14], o[CV - 1][14], o[0][15], o[CV - 1][15]);
              else if (ls < a.n_frames)
                *reinterpret_cast<float2 *> (p) = make_float2 (o[0][14], o[CV - 1][14]);
            }
          else
            {
              float *p = a.out + ls * C + ch0;
              if (ls < a.n_frames)
                p[0] = o[0][14];
              if (ls + 1 < a.n_frames)
                p[C] = o[0][15];
            }
#pragma unroll
          for (int c = 0; c < CV; c++)
            {
              const float v0 = ls < a.n_frames ? fabsf (o[c][14]) : 0.f;
              const float v1 = ls + 1 < a.n_frames ? fabsf (o[c][15]) : 0.f;
              if (x < pbound) pmax0 = fmaxf (pmax0, v0); else pmax1 = fmaxf (pmax1, v0);
              if (x + 1 < pbound) pmax0 = fmaxf (pmax0, v1); else pmax1 = fmaxf (pmax1, v1);
            }
        }
      if (a.block_max)
        {
          // Limiter::block_max (reference limiter.cc:90-97): max |x| per limiter block; non-negative
          // floats order like their bit patterns, so an integer atomic max does it
          if (own)
            {
              max0 = wave_max (max0);
              max1 = wave_max (max1);
              if (lane == 0)
                {
                  const long long i0 = b0 - a.first_block, i1 = i0 + 1;
                  if (i0 >= 0 && i0 < a.n_blocks && max0 > 0.f) atomicMax (a.block_max + i0, __float_as_uint (max0));
                  if (i1 >= 0 && i1 < a.n_blocks && max1 > 0.f) atomicMax (a.block_max + i1, __float_as_uint (max1));
                }
            }
          if (own_prev)
            {
              pmax0 = wave_max (pmax0);
              pmax1 = wave_max (pmax1);
              if (lane == 0)
                {
                  const long long i0 = pb0 - a.first_block, i1 = i0 + 1;
                  if (i0 >= 0 && i0 < a.n_blocks && pmax0 > 0.f) atomicMax (a.block_max + i0, __float_as_uint (pmax0));
                  if (i1 >= 0 && i1 < a.n_blocks && pmax1 > 0.f) atomicMax (a.block_max + i1, __float_as_uint (pmax1));
                }
            }
        }
    }
}

template<int CV> __global__ void __launch_bounds__ (64 * WAVES)
add_mix_kernel (DevTables t, AddMixArgs a, long long frame_number0, int block_frames)
{
  add_mix_body<CV, false> (t, a, frame_number0, block_frames);
}
// stereo: four waves per SIMD (122 registers, 39 KB of LDS per workgroup: four workgroups per CU)
template<int CV> __global__ void __launch_bounds__ (64 * WAVES) __attribute__ ((amdgpu_waves_per_eu (4, 4)))
add_mix_kernel_w4 (DevTables t, AddMixArgs a, long long frame_number0, int block_frames)
{
  add_mix_body<CV, true> (t, a, frame_number0, block_frames);
}
// the same with the two channels' transforms pipelined over the wave's one exchange tile (frame_delta2)
__global__ void __launch_bounds__ (64 * WAVES) __attribute__ ((amdgpu_waves_per_eu (4, 4)))
add_mix_pair_kernel (DevTables t, AddMixArgs a, long long frame_number0, int block_frames)
{
  add_mix_body<2, true, true> (t, a, frame_number0, block_frames);
}
/* a batch of clips in ONE launch (stereo): blockIdx.y = clip, its arguments from an array on the device (uniform: scalar loads); the grid's
 * x extent covers the clip with the most spans, the others' surplus workgroups return at once */
__global__ void __launch_bounds__ (64 * WAVES) __attribute__ ((amdgpu_waves_per_eu (4, 4)))
add_mix_pair_batch_kernel (DevTables t, const AddMixArgs *args, long long frame_number0, int block_frames)
{
  const AddMixArgs a = args[blockIdx.y];
  add_mix_body<2, true, true> (t, a, frame_number0, block_frames);
}
int add_mix_waves_per_simd() { return 4; }
int g_fft_pair = 1;              // (debug toggle: stereo add with frame_delta2)
extern "C" void awm_debug_set_fft_pair (int on) { g_fft_pair = on; }

hipError_t
launch_add_mix (hipStream_t st, const DevTables& t, const AddMixArgs& a)
{
  if (a.n_frames <= 0)
    return hipSuccess;
  const long long F = (a.n_frames + 1023) / 1024;
  const long long n_spans = (F + a.frames_per_span - 1) / a.frames_per_span;
  const bool stereo = a.n_channels == 2;
  const long long items = n_spans * (stereo ? 1 : a.n_channels);
  const unsigned grid = unsigned ((items + WAVES - 1) / WAVES);
  const int block_frames = a.block_frames;
  const long long frame_number0 = 2LL * block_frames - a.frames_pad_start;       // reference wmadd.cc:293-294
  // stereo: both channels in one wave, four waves per SIMD (122 registers; three waves + prefetch of the next frame: 3 - 5 % slower)
  if (stereo && g_fft_pair)
    hipLaunchKernelGGL (add_mix_pair_kernel, dim3 (grid), dim3 (64 * WAVES), 0, st, t, a, frame_number0, block_frames);
  else if (stereo)
    hipLaunchKernelGGL (add_mix_kernel_w4<2>, dim3 (grid), dim3 (64 * WAVES), 0, st, t, a, frame_number0, block_frames);
  else
    hipLaunchKernelGGL (add_mix_kernel<1>, dim3 (grid), dim3 (64 * WAVES), 0, st, t, a, frame_number0, block_frames);
  return hipGetLastError();
}

hipError_t
launch_add_mix_batch (hipStream_t st, const DevTables& t, const AddMixArgs *args_dev, int n_clips, long long max_spans, int block_frames, int frames_pad_start)
{
  if (n_clips <= 0 || max_spans <= 0)
    return hipSuccess;
  const long long frame_number0 = 2LL * block_frames - frames_pad_start;         // reference wmadd.cc:293-294
  hipLaunchKernelGGL (add_mix_pair_batch_kernel, dim3 (unsigned ((max_spans + WAVES - 1) / WAVES), unsigned (n_clips)), dim3 (64 * WAVES), 0, st,
                      t, args_dev, frame_number0, block_frames);
  return hipGetLastError();
}

/* ==========================================================================================
 * K3: limiter ramp (reference limiter.cc:99-124)
 * ========================================================================================== */
__device__ __forceinline__ float
limiter_scale (long long gs, const float *block_max, long long first_block, long long n_blocks, int BS, float ceiling)
{
  const long long b = gs / BS;
  const int i = int (gs - b * BS);
  auto M = [&] (long long bb) -> float {
    const long long k = bb - first_block;
    if (bb < 0 || k < 0 || k >= n_blocks)
      return ceiling;                       // block_max_last starts at the ceiling; blocks past the end are silent
    return fmaxf (block_max[k], ceiling);
  };
  const float m_last = M (b - 1), m_cur = M (b), m_next = M (b + 1);
  const float scale_start = __fdiv_rn (ceiling, fmaxf (m_last, m_cur));
  const float scale_end   = __fdiv_rn (ceiling, fmaxf (m_cur, m_next));
  const float scale_step  = __fdiv_rn (__fsub_rn (scale_end, scale_start), float (BS));
  return __fadd_rn (scale_start, __fmul_rn (float (i), scale_step));
}

// GENERIC FORM (what runs for 3+ channels, unaligned spans and the values behind the last whole float4; the stereo / mono streams of
// the bench take limiter_table_kernel + limiter_apply_kernel below): one value per thread
__global__ void __launch_bounds__ (256)
limiter_kernel (float *data, long long n_frames, int C, long long first_sample, const float *block_max,
                long long first_block, long long n_blocks, int BS, float ceiling)
{
  const long long n_values = n_frames * C;
  const long long stride = (long long) gridDim.x * blockDim.x;
  for (long long v = (long long) blockIdx.x * blockDim.x + threadIdx.x; v < n_values; v += stride)
    data[v] = __fmul_rn (data[v], limiter_scale (first_sample + v / C, block_max, first_block, n_blocks, BS, ceiling));
}

// 16 bytes per thread: C == 1 -> 4 frames, C == 2 -> 2 frames per float4
/* K3 in two steps.  The ramp of a limiter block (scale_start, scale_step) only depends on three block maxima, so it
 * is computed once per block (K3a) instead of once per sample (three IEEE divisions and a 64 bit integer division each);
 * K3b then streams the samples: every workgroup owns a run of 2048 float4 -- shorter than a limiter block, so it meets
 * at most two table entries, which it fetches with scalar loads -- and does one multiply-add per frame.  Runs whose
 * entries are (1, 0) are left untouched: x * 1.0f == x, the pass is the identity there (audio that never reaches the
 * ceiling costs no memory traffic at all).  The arithmetic per sample is unchanged (reference limiter.cc:99-124). */
__global__ void
limiter_table_kernel (float2 *tab, long long tab_first_block, long long n_tab, const float *block_max,
                      long long first_block, long long n_blocks, int BS, float ceiling)
{
  const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_tab)
    return;
  const long long b = tab_first_block + k;
  auto M = [&] (long long bb) -> float {
    const long long j = bb - first_block;
    if (bb < 0 || j < 0 || j >= n_blocks)
      return ceiling;
    return fmaxf (block_max[j], ceiling);
  };
  const float m_last = M (b - 1), m_cur = M (b), m_next = M (b + 1);
  const float scale_start = __fdiv_rn (ceiling, fmaxf (m_last, m_cur));
  const float scale_end   = __fdiv_rn (ceiling, fmaxf (m_cur, m_next));
  tab[k] = make_float2 (scale_start, __fdiv_rn (__fsub_rn (scale_end, scale_start), float (BS)));
}

constexpr int LIMITER_RUN = 2048;       // float4 per workgroup (8 per thread)

template<int C> __device__ __forceinline__ void
limiter_apply_run (float4 *data, long long n_vec, long long first_sample, const float2 *tab, long long tab_first_block, int BS)
{
  constexpr int FPV = 4 / C;            // frames per float4
  const long long base = (long long) blockIdx.x * LIMITER_RUN;
  const long long gs0 = first_sample + base * FPV;
  const long long b0 = gs0 / BS;                          // uniform: once per workgroup
  const int i0 = int (gs0 - b0 * BS);
  const float2 *tb = tab + (b0 - tab_first_block);
  const float2 t0 = tb[0], t1 = tb[1];
  if (t0.x == 1.f && t0.y == 0.f && t1.x == 1.f && t1.y == 0.f)
    return;
  auto scale = [&] (int i) -> float {
    const bool next = i >= BS;
    const float2 t = next ? t1 : t0;
    return __fadd_rn (t.x, __fmul_rn (float (next ? i - BS : i), t.y));
  };
  constexpr int U = LIMITER_RUN / 256;
  float4 v[U];
#pragma unroll
  for (int j = 0; j < U; j++)                   // all loads first: the stores below must not serialise them
    {
      const long long q = base + threadIdx.x + 256 * j;
      v[j] = q < n_vec ? data[q] : make_float4 (0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
  for (int j = 0; j < U; j++)
    {
      const int r = threadIdx.x + 256 * j;
      const int i = i0 + r * FPV;
      if (C == 2)
        {
          const float s0 = scale (i), s1 = scale (i + 1);
          v[j] = make_float4 (__fmul_rn (v[j].x, s0), __fmul_rn (v[j].y, s0), __fmul_rn (v[j].z, s1), __fmul_rn (v[j].w, s1));
        }
      else
        v[j] = make_float4 (__fmul_rn (v[j].x, scale (i)), __fmul_rn (v[j].y, scale (i + 1)), __fmul_rn (v[j].z, scale (i + 2)),
                            __fmul_rn (v[j].w, scale (i + 3)));
    }
#pragma unroll
  for (int j = 0; j < U; j++)
    {
      const long long q = base + threadIdx.x + 256 * j;
      if (q < n_vec)
        data[q] = v[j];
    }
}

template<int C> __global__ void __launch_bounds__ (256)
limiter_apply_kernel (float4 *data, long long n_vec, long long first_sample, const float2 *tab, long long tab_first_block, int BS)
{
  limiter_apply_run<C> (data, n_vec, first_sample, tab, tab_first_block, BS);
}

/* K3 for a batch of clips, every clip a stream of its own that starts at sample 0: blockIdx.y = clip */
__global__ void
limiter_table_batch_kernel (const LimiterClip *clips, int BS, float ceiling)
{
  const LimiterClip c = clips[blockIdx.y];
  const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= c.n_tab)
    return;
  auto M = [&] (long long bb) -> float {
    if (bb < 0 || bb >= c.n_blocks)
      return ceiling;
    return fmaxf (c.block_max[bb], ceiling);
  };
  const float m_last = M (k - 1), m_cur = M (k), m_next = M (k + 1);
  const float scale_start = __fdiv_rn (ceiling, fmaxf (m_last, m_cur));
  const float scale_end   = __fdiv_rn (ceiling, fmaxf (m_cur, m_next));
  c.tab[k] = make_float2 (scale_start, __fdiv_rn (__fsub_rn (scale_end, scale_start), float (BS)));
}

template<int C> __global__ void __launch_bounds__ (256)
limiter_apply_batch_kernel (const LimiterClip *clips, int BS)
{
  const LimiterClip c = clips[blockIdx.y];
  const long long n_values = c.n_frames * C, n_vec = n_values / 4;
  if ((long long) blockIdx.x * LIMITER_RUN < n_vec)
    limiter_apply_run<C> (reinterpret_cast<float4 *> (c.data), n_vec, 0, c.tab, 0, BS);
  // the values behind the last whole float4 (stereo: an odd number of frames), with the same table entries and arithmetic
  if (blockIdx.x == 0 && 4 * n_vec + threadIdx.x < n_values)
    {
      const long long v = 4 * n_vec + threadIdx.x, f = v / C, b = f / BS;
      const float2 t = c.tab[b];
      c.data[v] = __fmul_rn (c.data[v], __fadd_rn (t.x, __fmul_rn (float (int (f - b * BS)), t.y)));
    }
}

hipError_t
launch_limiter_batch (hipStream_t st, const LimiterClip *clips_dev, int n_clips, long long max_frames, int n_channels, int limiter_block, float ceiling)
{
  if (n_clips <= 0 || max_frames <= 0)
    return hipSuccess;
  if ((n_channels != 1 && n_channels != 2) || limiter_block < 4 * LIMITER_RUN)
    return hipErrorInvalidValue;
  const long long max_tab = limiter_tab_entries (max_frames, 0, limiter_block);
  hipLaunchKernelGGL (limiter_table_batch_kernel, dim3 (unsigned ((max_tab + 255) / 256), unsigned (n_clips)), dim3 (256), 0, st, clips_dev, limiter_block, ceiling);
  const long long max_vec = std::max<long long> (1, max_frames * n_channels / 4);
  const dim3 grid (unsigned ((max_vec + LIMITER_RUN - 1) / LIMITER_RUN), unsigned (n_clips));
  if (n_channels == 2)
    hipLaunchKernelGGL (limiter_apply_batch_kernel<2>, grid, dim3 (256), 0, st, clips_dev, limiter_block);
  else
    hipLaunchKernelGGL (limiter_apply_batch_kernel<1>, grid, dim3 (256), 0, st, clips_dev, limiter_block);
  return hipGetLastError();
}

size_t
limiter_tab_entries (long long n_frames, long long first_sample, int limiter_block)
{
  if (n_frames <= 0 || limiter_block <= 0)
    return 0;
  return size_t ((first_sample + n_frames - 1) / limiter_block - first_sample / limiter_block + 2);
}

hipError_t
launch_limiter (hipStream_t st, float *data, long long n_frames, int n_channels, long long first_sample,
                const float *block_max, long long first_block, long long n_blocks, int limiter_block, float ceiling,
                float2 *scale_tab, size_t scale_tab_entries)
{
  const long long n_values = n_frames * n_channels;
  if (n_values <= 0)
    return hipSuccess;
  const size_t need = limiter_tab_entries (n_frames, first_sample, limiter_block);
  // vector path: per-block (scale_start, scale_step) table + float4 apply; anything else (odd channel counts, unaligned
  // spans, limiter blocks shorter than a run, no table workspace) takes the scalar kernel below for all its frames
  const bool vec = (n_channels == 1 || n_channels == 2) && (reinterpret_cast<uintptr_t> (data) & 15) == 0
                && scale_tab && scale_tab_entries >= need && limiter_block >= 4 * LIMITER_RUN;
  const long long n_vec = vec ? n_values / 4 : 0;
  if (n_vec)
    {
      const long long tab_first = first_sample / limiter_block;
      hipLaunchKernelGGL (limiter_table_kernel, dim3 (unsigned ((need + 255) / 256)), dim3 (256), 0, st, scale_tab, tab_first, (long long) need,
                          block_max, first_block, n_blocks, limiter_block, ceiling);
      const unsigned grid = unsigned ((n_vec + LIMITER_RUN - 1) / LIMITER_RUN);
      if (n_channels == 2)
        hipLaunchKernelGGL (limiter_apply_kernel<2>, dim3 (grid), dim3 (256), 0, st, reinterpret_cast<float4 *> (data), n_vec, first_sample,
                            scale_tab, tab_first, limiter_block);
      else
        hipLaunchKernelGGL (limiter_apply_kernel<1>, dim3 (grid), dim3 (256), 0, st, reinterpret_cast<float4 *> (data), n_vec, first_sample,
                            scale_tab, tab_first, limiter_block);
    }
  const long long done = n_vec * 4;
  if (done < n_values)
    {
      // tail (or generic channel counts): scalar kernel on the remaining whole frames
      const long long done_frames = done / n_channels;
      long long blocks = (n_values - done + 255) / 256;
      if (blocks > 256 * 32)
        blocks = 256 * 32;
      hipLaunchKernelGGL (limiter_kernel, dim3 (unsigned (blocks)), dim3 (256), 0, st, data + done, n_frames - done_frames, n_channels,
                          first_sample + done_frames, block_max, first_block, n_blocks, limiter_block, ceiling);
    }
  return hipGetLastError();
}

/* ==========================================================================================
 * K10 / K11: other sample rates (kernels.hh ResampleArgs)
 * ========================================================================================== */
constexpr int RS_TILE = 1024;             // outputs per workgroup
constexpr int RS_MAX_TAB = 12288;         // floats of LDS for the coefficient table (48 KiB)

/* stereo, input window of the tile staged in LDS (zero extended): s_in[0] is input frame first0; same products and sums as below */
__device__ __forceinline__ void
resample_output_staged (const ResampleArgs& a, const float *tab, int stride, long long m, const float2 *s_in, unsigned int t_rel)
{
  // m step = tile0 step + t_rel - r0, r0 = (tile0 step) mod np: window start and phase relative to the tile in 32 bits (a 64-bit
  // division per output cost as much as the taps)
  const int hl = a.hl;
  const unsigned int b_rel = t_rel / (unsigned int) a.np;
  const int ph = int (t_rel - b_rel * (unsigned int) a.np);
  const float *c1 = tab + stride * ph;
  const float *c2 = tab + stride * (a.np - ph);
  const float2 *p1 = s_in + b_rel, *p2 = p1 + 2 * hl - 1;
  float s0 = 1e-20f, s1 = 1e-20f;
#pragma unroll 4
  for (int i = 0; i < hl; i++)
    {
      const float2 x1 = p1[i], x2 = p2[-i];
      s0 = __fadd_rn (s0, __fadd_rn (__fmul_rn (x1.x, c1[i]), __fmul_rn (x2.x, c2[i])));
      s1 = __fadd_rn (s1, __fadd_rn (__fmul_rn (x1.y, c1[i]), __fmul_rn (x2.y, c2[i])));
    }
  reinterpret_cast<float2 *> (a.out)[m] = make_float2 (__fsub_rn (s0, 1e-20f), __fsub_rn (s1, 1e-20f));
}

// one output frame m; tab: coefficient rows with `stride` floats each (LDS or global)
template<int CT> __device__ __forceinline__ void
resample_output (const ResampleArgs& a, const float *tab, int stride, long long m)
{
  const int C = CT ? CT : a.n_channels, hl = a.hl;
  const long long t = m * a.step;
  const long long b = t / a.np;
  const int ph = int (t - b * a.np);
  const float *c1 = tab + stride * ph;
  const float *c2 = tab + stride * (a.np - ph);
  const long long first = b - (hl - 1);                       // input frame of P[b]
  if (CT == 2)
    {
      const float2 *in2 = reinterpret_cast<const float2 *> (a.in);
      float s0 = 1e-20f, s1 = 1e-20f;
      if (__all (first >= 0 && first + 2 * hl <= a.n_in))      // the whole wave is away from the ends: no bounds checks
        {
          const float2 *p1 = in2 + first, *p2 = in2 + first + 2 * hl - 1;
#pragma unroll 4
          for (int i = 0; i < hl; i++)
            {
              const float2 x1 = p1[i], x2 = p2[-i];
              s0 = __fadd_rn (s0, __fadd_rn (__fmul_rn (x1.x, c1[i]), __fmul_rn (x2.x, c2[i])));
              s1 = __fadd_rn (s1, __fadd_rn (__fmul_rn (x1.y, c1[i]), __fmul_rn (x2.y, c2[i])));
            }
        }
      else
        for (int i = 0; i < hl; i++)
          {
            const long long j1 = first + i, j2 = first + 2 * hl - 1 - i;
            const float2 x1 = (j1 >= 0 && j1 < a.n_in) ? in2[j1] : make_float2 (0.f, 0.f);
            const float2 x2 = (j2 >= 0 && j2 < a.n_in) ? in2[j2] : make_float2 (0.f, 0.f);
            s0 = __fadd_rn (s0, __fadd_rn (__fmul_rn (x1.x, c1[i]), __fmul_rn (x2.x, c2[i])));
            s1 = __fadd_rn (s1, __fadd_rn (__fmul_rn (x1.y, c1[i]), __fmul_rn (x2.y, c2[i])));
          }
      reinterpret_cast<float2 *> (a.out)[m] = make_float2 (__fsub_rn (s0, 1e-20f), __fsub_rn (s1, 1e-20f));
      return;
    }
  for (int c = 0; c < C; c++)
    {
      float sum = 1e-20f;
      for (int i = 0; i < hl; i++)
        {
          const long long j1 = first + i, j2 = first + 2 * hl - 1 - i;
          const float x1 = (j1 >= 0 && j1 < a.n_in) ? a.in[j1 * C + c] : 0.f;
          const float x2 = (j2 >= 0 && j2 < a.n_in) ? a.in[j2 * C + c] : 0.f;
          sum = __fadd_rn (sum, __fadd_rn (__fmul_rn (x1, c1[i]), __fmul_rn (x2, c2[i])));
        }
      a.out[m * C + c] = __fsub_rn (sum, 1e-20f);
    }
}

/* Neighbouring outputs use different phases, i.e. different coefficient rows: read from global memory every load
 * instruction of a wave touches up to 64 cache lines (first version of this kernel: 41 ms for an hour of stereo at 48 kHz
 * down and up again).  The table ((np + 1) x hl, 10 - 30 KiB for the usual rates) is staged in LDS, with an odd row stride so
 * that equal columns of different rows fall into different banks. */
/* The windows of neighbouring outputs overlap almost completely (36 - 40 taps, the window start advances by ~1 frame per
 * output).  For stereo the input window of a whole tile (in_span frames, zero extended at the ends of the stream: no bounds
 * checks in the tap loop) is staged in LDS next to the table, and window start / phase are computed relative to the tile in
 * 32 bits. */
/* A workgroup keeps its table for `tiles_per_wg` consecutive tiles: staging the table took as long as the taps of one tile
 * (a dozen dependent global loads per thread for 4 outputs per thread). */
template<int CT> __global__ void __launch_bounds__ (256)
resample_kernel (ResampleArgs a, int lds_floats, int in_span, int tiles_per_wg)
{
  extern __shared__ float s_tab[];                           // lds_floats (launcher): the table, or nothing if it is too large; then the window
  const int stride = a.hl | 1;
  const bool in_lds = lds_floats > 0;
  const bool staged = CT == 2 && in_lds && in_span > 0;
  float2 *s_in = reinterpret_cast<float2 *> (s_tab + ((lds_floats + 1) & ~1));
  if (in_lds)
    {
      const int n = (a.np + 1) * a.hl;
      for (int i = threadIdx.x; i < n; i += 256)
        {
          const int r = i / a.hl;
          s_tab[r * stride + (i - r * a.hl)] = a.ctab[i];
        }
    }
  for (int t = 0; t < tiles_per_wg; t++)
    {
      const long long tile0 = ((long long) blockIdx.x * tiles_per_wg + t) * RS_TILE;
      if (tile0 >= a.n_out)
        break;                                                                 // (uniform)
      const long long b0 = (tile0 * a.step) / a.np;
      const unsigned int r0 = (unsigned int) (tile0 * a.step - b0 * a.np);
      const long long first0 = b0 - (a.hl - 1);                               // first input frame the tile's first output reads
      if (staged)
        {
          if (t)
            __syncthreads();                                                   // the previous tile's windows have been read
          const float2 *in2 = reinterpret_cast<const float2 *> (a.in);
          for (int i = threadIdx.x; i < in_span; i += 256)
            {
              const long long j = first0 + i;
              s_in[i] = (j >= 0 && j < a.n_in) ? in2[j] : make_float2 (0.f, 0.f);
            }
        }
      if (in_lds && (staged || t == 0))
        __syncthreads();
      for (int q = 0; q < RS_TILE / 256; q++)
        {
          const long long m = tile0 + q * 256 + threadIdx.x;
          if (m >= a.n_out)
            break;
          if (staged)
            resample_output_staged (a, s_tab, stride, m, s_in, r0 + (unsigned int) (q * 256 + threadIdx.x) * (unsigned int) a.step);
          else if (in_lds)
            resample_output<CT> (a, s_tab, stride, m);
          else
            resample_output<CT> (a, a.ctab, a.hl, m);
        }
    }
}

/* The fixed ratios 147 / 160 and 160 / 147 (48 <-> 44.1 kHz), stereo: outputs np apart have the SAME phase, i.e. the same two
 * coefficient rows, and windows exactly `step` input frames apart.  A thread therefore takes one phase, keeps its 2 hl coefficients in
 * registers for all the outputs it produces, and the tap loop is nothing but window reads (one ds_read2_b64 per tap pair: the windows
 * of neighbouring threads start 1 - 2 frames apart) and zita's eight unfused operations per tap pair -- no coefficient reads, no
 * address arithmetic (the generic kernel: 12 bytes of coefficients from LDS per tap pair beside the 16 bytes of samples, and a
 * third more VALU instructions for indices).  A workgroup = R replicas of the np phases; a tile = J rounds =
 * R J np outputs, whose R J step + 2 hl input frames are staged in LDS (zero extended at the ends of the stream); the coefficients
 * stay in registers over `tiles_per_wg` tiles.  Same products and sums, in the same order, as resample_output. */
/* Which thread takes which phase: the window of phase p starts at floor (p step / np).  Going DOWN (step > np) these starts skip a value
 * every ~11 phases, so 16 neighbouring phases span 17 - 18 frames: one 8-byte LDS word too many for 16 lanes' worth of banks, every
 * window read a two-way conflict (measured: 1.54 ms for an hour at 48 kHz against 0.96 ms for the way up, whose starts never skip).
 * There the threads are numbered by WINDOW START instead -- `step` slots per replica, the 13 starts that belong to no phase idle --
 * and neighbouring lanes read neighbouring words.  Going up, neighbouring phases share or succeed each other's start already. */
template<int NP, int STEP, int HL, int R, int J> __global__ void __launch_bounds__ (((STEP > NP ? STEP : NP) * R + 63) / 64 * 64)
resample_phase_kernel (ResampleArgs a, int tiles_per_wg)
{
  constexpr int SPAN = R * J * STEP + 2 * HL;                 // input frames of a tile
  constexpr int TILE = R * J * NP;                            // outputs of a tile
  constexpr int SLOTS = STEP > NP ? STEP : NP;                // threads per replica
  constexpr int WG = (SLOTS * R + 63) / 64 * 64;
  __shared__ float2 s_in[SPAN];
  const int tid = threadIdx.x;
  const int rep = tid < SLOTS * R ? tid / SLOTS : 0, slot = tid < SLOTS * R ? tid - rep * SLOTS : 0;
  const int p_of_slot = STEP > NP ? (slot * NP + STEP - 1) / STEP : slot;              // the phase whose window starts at frame `slot`, if any
  const bool active = tid < SLOTS * R && p_of_slot < NP && (STEP <= NP || (p_of_slot * STEP) / NP == slot);
  const int p = active ? p_of_slot : 0;
  const int ph = (p * STEP) % NP, b_p = (p * STEP) / NP;      // phase and window start of output p of a tile (tiles start at multiples of np)
  float c1[HL], c2[HL];
#pragma unroll
  for (int i = 0; i < HL; i++)
    {
      c1[i] = a.ctab[ph * HL + i];
      c2[i] = a.ctab[(NP - ph) * HL + i];
    }
  const float2 *in2 = reinterpret_cast<const float2 *> (a.in);
  float2 *out2 = reinterpret_cast<float2 *> (a.out);
  for (int t = 0; t < tiles_per_wg; t++)
    {
      const long long tile = (long long) blockIdx.x * tiles_per_wg + t;
      const long long tile0 = tile * TILE;
      if (tile0 >= a.n_out)
        break;                                                                 // (uniform)
      const long long first0 = tile * (R * J * STEP) - (HL - 1);              // input frame of s_in[0]
      if (t)
        __syncthreads();                                                       // the previous tile's windows have been read
      for (int i = tid; i < SPAN; i += WG)
        {
          const long long j = first0 + i;
          s_in[i] = (j >= 0 && j < a.n_in) ? in2[j] : make_float2 (0.f, 0.f);
        }
      __syncthreads();
      if (active)
        {
#pragma unroll 2
          for (int j = 0; j < J; j++)
            {
              const int q = rep + R * j;                                       // which group of np outputs of the tile
              const long long m = tile0 + p + NP * q;
              if (m >= a.n_out)
                break;
              const float2 *p1 = s_in + b_p + STEP * q, *p2 = p1 + 2 * HL - 1;
              float s0 = 1e-20f, s1 = 1e-20f;
#pragma unroll
              for (int i = 0; i < HL; i++)
                {
                  const float2 x1 = p1[i], x2 = p2[-i];
                  s0 = __fadd_rn (s0, __fadd_rn (__fmul_rn (x1.x, c1[i]), __fmul_rn (x2.x, c2[i])));
                  s1 = __fadd_rn (s1, __fadd_rn (__fmul_rn (x1.y, c1[i]), __fmul_rn (x2.y, c2[i])));
                }
              out2[m] = make_float2 (__fsub_rn (s0, 1e-20f), __fsub_rn (s1, 1e-20f));
            }
        }
    }
}

/* (measurement knob) 1 (default): the phase-per-thread kernel for stereo 48 <-> 44.1 kHz | 0: the generic kernel for every ratio */
int g_resample_phase = 1;
extern "C" void awm_debug_set_resample_phase (int on) { g_resample_phase = on; }

template<int NP, int STEP, int HL> static hipError_t
launch_resample_phase (hipStream_t st, const ResampleArgs& a)
{
  constexpr int R = 2, J = 8;
  const long long n_tiles = (a.n_out + R * J * NP - 1) / (R * J * NP);
  const int tiles_per_wg = int (std::min<long long> (8, std::max<long long> (1, n_tiles / 4096)));     // >= 4096 workgroups first
  const dim3 grid (unsigned ((n_tiles + tiles_per_wg - 1) / tiles_per_wg));
  hipLaunchKernelGGL ((resample_phase_kernel<NP, STEP, HL, R, J>), grid, dim3 (((STEP > NP ? STEP : NP) * R + 63) / 64 * 64), 0, st, a, tiles_per_wg);
  return hipGetLastError();
}

hipError_t
launch_resample (hipStream_t st, const ResampleArgs& a)
{
  if (a.n_out <= 0)
    return hipSuccess;
  const bool aligned = (reinterpret_cast<uintptr_t> (a.in) & 7) == 0 && (reinterpret_cast<uintptr_t> (a.out) & 7) == 0;
  if (g_resample_phase && a.n_channels == 2 && aligned)
    {
      if (a.np == 147 && a.step == 160 && a.hl == 18)
        return launch_resample_phase<147, 160, 18> (st, a);
      if (a.np == 160 && a.step == 147 && a.hl == 16)
        return launch_resample_phase<160, 147, 16> (st, a);
    }
  const long long n_tiles = (a.n_out + RS_TILE - 1) / RS_TILE;
  const int tiles_per_wg = int (std::min<long long> (8, std::max<long long> (1, n_tiles / 4096)));     // >= 4096 workgroups first
  const dim3 grid (unsigned ((n_tiles + tiles_per_wg - 1) / tiles_per_wg));
  // dynamic LDS of exactly the table size: 10 - 30 KiB for the usual rates leaves room for up to 8 waves per SIMD
  const int want = (a.np + 1) * (a.hl | 1);
  const int lds_floats = want <= RS_MAX_TAB ? want : 0;
  // input frames the RS_TILE outputs of a tile read: the window start moves by floor ((RS_TILE - 1) step / np) + 1 at most, plus one window
  const long long span = ((long long) (RS_TILE - 1) * a.step) / a.np + 2 + 2LL * a.hl;
  const bool stage = a.n_channels == 2 && aligned && lds_floats > 0 && span <= 4096 && (long long) RS_TILE * a.step + a.np < (1LL << 31);
  const int in_span = stage ? int (span) : 0;
  const size_t lds_bytes = size_t ((lds_floats + 1) & ~1) * sizeof (float) + size_t (in_span) * sizeof (float2);
  if (a.n_channels == 2 && aligned)
    hipLaunchKernelGGL (resample_kernel<2>, grid, dim3 (256), lds_bytes, st, a, lds_floats, in_span, tiles_per_wg);
  else
    hipLaunchKernelGGL (resample_kernel<0>, grid, dim3 (256), lds_bytes, st, a, lds_floats, in_span, tiles_per_wg);
  return hipGetLastError();
}

constexpr int MIX_RUN = 2048;            // values per thread group run: 256 threads x 8

/* `add --snr` (reference wmadd.cc:553-563): power of the input and of (mix - input) BEFORE the limiter, in double like the
 * reference's loop; acc[0] += sum (mix - orig)^2, acc[1] += sum orig^2 */
__global__ void __launch_bounds__ (256)
power_sums_kernel (const float *orig, const float *mixed, long long n_values, double *acc)
{
  __shared__ double s_d[4], s_s[4];
  double d2 = 0, s2 = 0;
  for (long long i = (long long) blockIdx.x * 256 + threadIdx.x; i < n_values; i += (long long) gridDim.x * 256)
    {
      const double o = orig[i], d = double (mixed[i]) - o;
      d2 += d * d;
      s2 += o * o;
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    {
      d2 += __shfl_xor (d2, off);
      s2 += __shfl_xor (s2, off);
    }
  if ((threadIdx.x & 63) == 0)
    {
      s_d[threadIdx.x >> 6] = d2;
      s_s[threadIdx.x >> 6] = s2;
    }
  __syncthreads();
  if (threadIdx.x == 0)
    {
      atomicAdd (acc, s_d[0] + s_d[1] + s_d[2] + s_d[3]);
      atomicAdd (acc + 1, s_s[0] + s_s[1] + s_s[2] + s_s[3]);
    }
}

hipError_t
launch_power_sums (hipStream_t st, const float *orig, const float *mixed, long long n_values, double *acc)
{
  if (n_values <= 0)
    return hipSuccess;
  const long long blocks = (n_values + 256 * 16 - 1) / (256 * 16);
  hipLaunchKernelGGL (power_sums_kernel, dim3 (unsigned (blocks < 4096 ? blocks : 4096)), dim3 (256), 0, st, orig, mixed, n_values, acc);
  return hipGetLastError();
}

__global__ void __launch_bounds__ (256)
mix_max_kernel (const float *orig, const float *wm, float *out, long long n_values, int C, unsigned int *block_max, long long n_blocks, int BS)
{
  __shared__ float s_m0[4], s_m1[4];
  const long long base = (long long) blockIdx.x * MIX_RUN;
  const long long f0 = base / C;                             // first frame this workgroup touches
  const long long b0 = f0 / BS;
  const long long bound = (b0 + 1) * BS * C;                  // first value of the next limiter block
  float m0 = 0.f, m1 = 0.f;
  for (int j = 0; j < MIX_RUN / 256; j++)
    {
      const long long v = base + threadIdx.x + 256LL * j;
      if (v < n_values)
        {
          const float r = __fadd_rn (wm[v], orig[v]);
          out[v] = r;
          if (v < bound)
            m0 = fmaxf (m0, fabsf (r));
          else
            m1 = fmaxf (m1, fabsf (r));
        }
    }
  m0 = wave_max (m0);
  m1 = wave_max (m1);
  if ((threadIdx.x & 63) == 0)
    {
      s_m0[threadIdx.x >> 6] = m0;
      s_m1[threadIdx.x >> 6] = m1;
    }
  __syncthreads();
  if (threadIdx.x == 0 && block_max)
    {
      m0 = fmaxf (fmaxf (s_m0[0], s_m0[1]), fmaxf (s_m0[2], s_m0[3]));
      m1 = fmaxf (fmaxf (s_m1[0], s_m1[1]), fmaxf (s_m1[2], s_m1[3]));
      if (b0 < n_blocks && m0 > 0.f) atomicMax (block_max + b0, __float_as_uint (m0));
      if (b0 + 1 < n_blocks && m1 > 0.f) atomicMax (block_max + b0 + 1, __float_as_uint (m1));
    }
}

hipError_t
launch_mix_max (hipStream_t st, const float *orig, const float *wm, float *out, long long n_frames, int n_channels,
                unsigned int *block_max, long long n_blocks, int limiter_block)
{
  const long long n_values = n_frames * n_channels;
  if (n_values <= 0)
    return hipSuccess;
  if ((long long) limiter_block * n_channels < MIX_RUN)
    return hipErrorInvalidValue;                               // a run may only straddle one block boundary
  hipLaunchKernelGGL (mix_max_kernel, dim3 (unsigned ((n_values + MIX_RUN - 1) / MIX_RUN)), dim3 (256), 0, st, orig, wm, out, n_values,
                      n_channels, block_max, n_blocks, limiter_block);
  return hipGetLastError();
}

__global__ void
fill_u32_kernel (unsigned int *p, unsigned int v, size_t n)
{
  const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    p[i] = v;
}
hipError_t
launch_fill_u32 (hipStream_t st, unsigned int *p, unsigned int v, size_t n)
{
  if (!n)
    return hipSuccess;
  hipLaunchKernelGGL (fill_u32_kernel, dim3 (unsigned ((n + 255) / 256)), dim3 (256), 0, st, p, v, n);
  return hipGetLastError();
}

/* ==========================================================================================
 * K4: STFT -> dB of 81 bands, band-major output tiles
 * ========================================================================================== */

/* first sample of a stream (kernels.hh SyncDbArgs: explicit list, or regular streams, optionally repeated per slice) */
__device__ __forceinline__ long long
sync_stream_base (const SyncDbArgs& a, long long stream)
{
  if (a.stream_base)
    return a.stream_base[stream];
  if (a.streams_per_slice > 0)
    return a.base0 + (stream % a.streams_per_slice) * a.base_stride + (stream / a.streams_per_slice) * a.slice_stride;
  return a.base0 + stream * a.base_stride;
}

/* non-silent value range that the skip rules of sync_fft use for this stream (syncfinder.cc:155-169, 578-590) */
__device__ __forceinline__ void
sync_stream_range (const SyncDbArgs& a, long long stream, long long& first, long long& last)
{
  first = a.first;
  last = a.last;
  if (a.stream_range)
    {
      const long long g = stream / a.range_div;
      const long long slice = a.range_index ? a.range_index[g] : g;
      first = a.stream_range[2 * slice];
      last = a.stream_range[2 * slice + 1];
      if (first < 0)
        first = last = 0x7fffffffffffffffLL;       // nothing but silence: every frame ends before `first`
    }
}

/* SPLIT (stereo, one output plane per channel -- the block decoder's fft_range): the interleaved samples are read ONCE and
 * both channels are transformed by the same wave; the tile holds the two planes side by side (<= 36 frames each). */

// TLD: row length of the output tile in LDS (frames per tile + 1).  33 keeps a workgroup at 38 KB so that FOUR of them
// (16 waves) fit on a CU; the full 73 is only needed by the table-driven refinement fallback (72 fine offsets).
template<int CV, bool SPLIT, int TLD> __global__ void __launch_bounds__ (64 * WAVES)
sync_db_kernel (DevTables t, SyncDbArgs a)
{
  constexpr int TILE_LD = TLD, TILE_MAX = TLD - 1, SPLIT_COLS = TILE_MAX / 2;
  __shared__ float2 s_tw[512];
  __shared__ float  s_win[1024];
  __shared__ float2 s_twb[NB];
  __shared__ float2 s_x[WAVES][XBUF_ELEMS];
  __shared__ float  s_tile[NB * TILE_LD];
  __shared__ char   s_have[TILE_MAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (a.silent_frames_are_zero && a.skip_unread_silent_tiles)
    {
      // (before the tables are loaded: workgroup uniform)  Blocks of padded slices for K7: a tile whose frames all lie in the padding's
      // silence, the two frames before and the two after it included, is never read -- K7 takes an item whose frame and both neighbours
      // are silent as (-96, -192, -96, -192) without a load -- so it is not written either.  Four fifths of a 30 s clip's blocks.
      long long stream = blockIdx.y, tile_idx = blockIdx.x;
      if (a.xcd_interleave)
        {
          const long long j = blockIdx.x >> 3;
          stream = j % a.n_streams;
          tile_idx = (j / a.n_streams) * 8 + (blockIdx.x & 7);
        }
      const long long base = sync_stream_base (a, stream);
      const int count = a.stream_count ? a.stream_count[stream] : a.count0;
      long long sil_first, sil_last;
      sync_stream_range (a, stream, sil_first, sil_last);
      const long long tile0 = tile_idx * a.tile_frames;
      if (tile0 >= count)
        return;
      const long long n_here = (count - tile0) < a.tile_frames ? count - tile0 : a.tile_frames;
      // (K7 reads the neighbours of every item whose frame or whose neighbour is live: frames up to TWO away from a live one)
      const long long idx_before = base + (tile0 - 2) * a.hop, idx_after = base + (tile0 + n_here + 1) * a.hop;
      const int C = a.n_channels;
      if ((idx_after + 1024) * C < sil_first || idx_before * C > sil_last)
        return;
    }
  load_shared_tables (t, s_tw, s_win, s_twb);
  __syncthreads();

  // Which (stream, tile).  The approximate search transforms the SAME samples once per shift (4 streams, 256 samples
  // apart): with xcd_interleave the grid is 1-D and workgroup id = 8 j + x runs tile 8 (j / n_streams) + x of stream
  // j % n_streams -- ids are dealt to the XCDs round robin, so the shifts of one tile run back to back on ONE XCD and
  // all but the first find the samples in its L2 (measured: 4x less HBM traffic, this kernel was HBM bound).
  long long stream = blockIdx.y, tile_idx = blockIdx.x;
  if (a.xcd_interleave)
    {
      const long long j = blockIdx.x >> 3;
      stream = j % a.n_streams;
      tile_idx = (j / a.n_streams) * 8 + (blockIdx.x & 7);
    }
  const int plane_ch = blockIdx.z;                       // only used in per-channel mode
  const long long base = sync_stream_base (a, stream);
  const int count = a.stream_count ? a.stream_count[stream] : a.count0;
  long long sil_first, sil_last;
  sync_stream_range (a, stream, sil_first, sil_last);
  const int TF = a.tile_frames;
  const long long tile0 = tile_idx * TF;
  if (tile0 >= count)
    return;
  const int C = a.n_channels;
  float2 *xbuf = s_x[wave];
  const int n_here = (count - tile0) < TF ? int (count - tile0) : TF;

  for (int ff = wave; ff < n_here; ff += WAVES)
    {
      const long long idx = base + (tile0 + ff) * a.hop;
      // skip rules of sync_fft (reference syncfinder.cc:578-590)
      const long long f_first = idx * C, f_last = (idx + 1024) * C;
      const bool outside = idx < 0 || idx + 1024 > a.n_frames;
      const bool skip = (f_last < sil_first) || (f_first > sil_last) || outside;
      float acc0 = 0.f, acc1 = 0.f;                      // bins 20 + lane, 84 + lane
      float split0 = 0.f, split1 = 0.f;                  // SPLIT: the same for channel 0 (acc0 / acc1 then hold channel 1)
      if (!skip)
        {
          if (CV == 2)
            {
              float in[2][16];
              fetch_stereo (a.pcm, idx, 1024, lane, in[0], in[1]);
              float2 za[8], zb[8];
              window_pack (in[0], s_win, lane, za);
              window_pack (in[1], s_win, lane, zb);
              fft512_forward2 (za, zb, xbuf, s_tw, lane);
              // bins 20 + lane, 84 + lane of both channels: the rows with the bins and their mirrors go through the tile, a then b
              const int k0 = MIN_BAND + lane, k1 = MIN_BAND + 64 + lane;
              const bool second = lane < NB - 64;
              float2 pa[4], pb[4];
              lds_st (&xbuf[0 * 64 + lane], za[0]);
              lds_st (&xbuf[1 * 64 + lane], za[1]);
              lds_st (&xbuf[6 * 64 + lane], za[6]);
              lds_st (&xbuf[7 * 64 + lane], za[7]);
              wave_sync();
              pa[0] = lds_ld (&xbuf[zpos (k0)]);
              pa[1] = lds_ld (&xbuf[zpos (512 - k0)]);
              pa[2] = lds_ld (&xbuf[zpos (second ? k1 : k0)]);
              pa[3] = lds_ld (&xbuf[zpos (512 - (second ? k1 : k0))]);
              wave_sync();
              lds_st (&xbuf[0 * 64 + lane], zb[0]);
              lds_st (&xbuf[1 * 64 + lane], zb[1]);
              lds_st (&xbuf[6 * 64 + lane], zb[6]);
              lds_st (&xbuf[7 * 64 + lane], zb[7]);
              wave_sync();
              pb[0] = lds_ld (&xbuf[zpos (k0)]);
              pb[1] = lds_ld (&xbuf[zpos (512 - k0)]);
              pb[2] = lds_ld (&xbuf[zpos (second ? k1 : k0)]);
              pb[3] = lds_ld (&xbuf[zpos (512 - (second ? k1 : k0))]);
              wave_sync();
              const float2 w0 = s_twb[lane], w1 = s_twb[second ? 64 + lane : lane];
              acc0 = __fadd_rn (acc0, db_from_complex (real_split (pa[0], pa[1], w0)));
              if (second)
                acc1 = __fadd_rn (acc1, db_from_complex (real_split (pa[2], pa[3], w1)));
              if (SPLIT)
                {
                  split0 = acc0;
                  split1 = acc1;
                  acc0 = acc1 = 0.f;
                }
              acc0 = __fadd_rn (acc0, db_from_complex (real_split (pb[0], pb[1], w0)));
              if (second)
                acc1 = __fadd_rn (acc1, db_from_complex (real_split (pb[2], pb[3], w1)));
            }
          else
            {
              const int c_begin = a.per_channel ? plane_ch : 0, c_end = a.per_channel ? plane_ch + 1 : C;
              for (int c = c_begin; c < c_end; c++)
                {
                  float in[16];
                  fetch_channel (a.pcm, idx, 1024, C, c, lane, in);
                  float2 z[8];
                  window_pack (in, s_win, lane, z);
                  fft512_forward (z, xbuf, s_tw, lane);
                  xbuf[0 * 64 + lane] = z[0];
                  xbuf[1 * 64 + lane] = z[1];
                  xbuf[6 * 64 + lane] = z[6];
                  xbuf[7 * 64 + lane] = z[7];
                  wave_sync();
                  {
                    const int k = MIN_BAND + lane;
                    acc0 = __fadd_rn (acc0, db_from_complex (real_split (xbuf[zpos (k)], xbuf[zpos (512 - k)], s_twb[lane])));
                  }
                  if (lane < NB - 64)
                    {
                      const int k = MIN_BAND + 64 + lane;
                      acc1 = __fadd_rn (acc1, db_from_complex (real_split (xbuf[zpos (k)], xbuf[zpos (512 - k)], s_twb[64 + lane])));
                    }
                  wave_sync();
                }
            }
        }
      else if (a.silent_frames_are_zero && !outside)
        {
          // every sample of the frame is zero: the spectrum is exactly zero, every band -96 dB (db_from_complex)
          const float zero_db = db_from_complex (make_float2 (0.f, 0.f));
          const int n_ch = CV == 2 ? 2 : (a.per_channel ? 1 : C);
          for (int c = 0; c < n_ch; c++)
            {
              if (SPLIT && c == 1)
                {
                  split0 = acc0;
                  split1 = acc1;
                  acc0 = acc1 = 0.f;
                }
              acc0 = __fadd_rn (acc0, zero_db);
              acc1 = __fadd_rn (acc1, zero_db);
            }
        }
      if (SPLIT)
        {
          s_tile[lane * TILE_LD + ff] = split0;
          s_tile[lane * TILE_LD + SPLIT_COLS + ff] = acc0;
          if (lane < NB - 64)
            {
              s_tile[(64 + lane) * TILE_LD + ff] = split1;
              s_tile[(64 + lane) * TILE_LD + SPLIT_COLS + ff] = acc1;
            }
        }
      else
        {
          s_tile[lane * TILE_LD + ff] = acc0;
          if (lane < NB - 64)
            s_tile[(64 + lane) * TILE_LD + ff] = acc1;
        }
      if (lane == 0)
        s_have[ff] = skip ? 0 : 1;
    }
  __syncthreads();
  if (SPLIT)
    {
      for (int p = 0; p < 2; p++)
        {
          float *out = a.out + stream * a.out_stream_stride + (long long) p * NB * a.ld + tile0;
          for (int i = threadIdx.x; i < NB * n_here; i += blockDim.x)
            {
              const int band = i / n_here, ff = i - band * n_here;
              out[band * a.ld + ff] = s_tile[band * TILE_LD + p * SPLIT_COLS + ff];
            }
        }
    }
  else
    {
      const long long plane = a.per_channel ? plane_ch : 0;
      float *out = a.out + stream * a.out_stream_stride + plane * NB * a.ld + tile0;
      for (int i = threadIdx.x; i < NB * n_here; i += blockDim.x)
        {
          const int band = i / n_here, ff = i - band * n_here;
          out[band * a.ld + ff] = s_tile[band * TILE_LD + ff];
        }
    }
  if (a.have && plane_ch == 0)
    for (int i = threadIdx.x; i < n_here; i += blockDim.x)
      a.have[stream * a.have_stream_stride + tile0 + i] = s_have[i];
}

hipError_t
launch_sync_db (hipStream_t st, const DevTables& t, const SyncDbArgs& a)
{
  if (a.n_streams <= 0 || a.tile_frames <= 0 || a.tile_frames > 72)
    return a.n_streams <= 0 ? hipSuccess : hipErrorInvalidValue;
  int max_count = a.count0;
  // with per-stream counts the caller passes the maximum in count0
  if (max_count <= 0)
    return hipSuccess;
  const unsigned tiles = unsigned ((max_count + a.tile_frames - 1) / a.tile_frames);
  // grid.y is limited to 65535: fold streams
  if (a.n_streams > 65535LL * 1)
    {
      // split into several launches over stream ranges
      SyncDbArgs b = a;
      long long done = 0;
      while (done < a.n_streams)
        {
          const long long n = (a.n_streams - done) > 65535 ? 65535 : (a.n_streams - done);
          b.n_streams = n;
          b.base0 = a.base0 + done * a.base_stride;
          b.stream_base = a.stream_base ? a.stream_base + done : nullptr;
          b.stream_count = a.stream_count ? a.stream_count + done : nullptr;
          b.out = a.out + done * a.out_stream_stride;
          b.have = a.have ? a.have + done * a.have_stream_stride : nullptr;
          hipError_t e = launch_sync_db (st, t, b);
          if (e != hipSuccess)
            return e;
          done += n;
        }
      return hipSuccess;
    }
  const bool small = a.tile_frames <= 32;
  if (a.n_channels == 2 && a.per_channel)
    {
      SyncDbArgs b = a;
      b.tile_frames = 16;                                        // two planes of 16 frames side by side in a 33-column tile
      const unsigned split_tiles = unsigned ((max_count + b.tile_frames - 1) / b.tile_frames);
      hipLaunchKernelGGL ((sync_db_kernel<2, true, 33>), dim3 (split_tiles, unsigned (a.n_streams), 1), dim3 (64 * WAVES), 0, st, t, b);
      return hipGetLastError();
    }
  const unsigned planes = a.per_channel ? unsigned (a.n_channels) : 1u;
  dim3 grid (tiles, unsigned (a.n_streams), planes);
  SyncDbArgs b = a;
  if (!a.stream_base && !a.stream_count && a.n_streams >= 2 && a.n_streams <= 8 && planes == 1)
    {
      b.xcd_interleave = 1;
      grid = dim3 (unsigned (((tiles + 7) / 8) * 8 * a.n_streams), 1, 1);
    }
  if (a.n_channels == 2 && !a.per_channel)
    {
      if (small)
        hipLaunchKernelGGL ((sync_db_kernel<2, false, 33>), grid, dim3 (64 * WAVES), 0, st, t, b);
      else
        hipLaunchKernelGGL ((sync_db_kernel<2, false, 73>), grid, dim3 (64 * WAVES), 0, st, t, b);
    }
  else
    {
      if (small)
        hipLaunchKernelGGL ((sync_db_kernel<1, false, 33>), grid, dim3 (64 * WAVES), 0, st, t, b);
      else
        hipLaunchKernelGGL ((sync_db_kernel<1, false, 73>), grid, dim3 (64 * WAVES), 0, st, t, b);
    }
  return hipGetLastError();
}

/* ==========================================================================================
 * K4s: refinement STFT by sliding DFT
 *
 * search_refine evaluates, for every candidate and every one of the 510 sync frames, the SAME 1024-sample window
 * advanced 65 times by 8 samples.  With the periodic von Hann window  w[n] = (1/256) (1/2 - 1/2 cos (2 pi n / N))
 * the windowed spectrum is  X[k] = (R[k]/2 - (R[k-1] + R[k+1])/4) / 256  where R is the plain DFT of the segment, and
 *   R'[k] = (R[k] + sum_{j<8} (x[s + N + j] - x[s + j]) e^{-2 pi i k j / N}) e^{+2 pi i 8 k / N}
 * advances R by 8 samples.  One wave carries bins 19..102 (two adjacent bins per lane, 42 lanes) of one stream through
 * all its fine offsets: one FFT at the first offset, then ~40 FP64 FMAs per lane, channel and offset instead of an
 * FFT.  The recursion runs in double precision (MI355X FP64 vector rate is half the FP32 rate), so no drift builds up:
 * the result is the double-precision DFT rounded to float -- the same definition the oracle's FFT uses.
 * dB conversion, channel sum, skip rules and output layout are exactly those of sync_db_kernel.
 * ========================================================================================== */
constexpr int SL_TILE = 16;                                   // fine offsets buffered in LDS between flushes

__device__ __forceinline__ double
dpp_from_lower_lane (double v)                                // lane i receives lane i - 1's value
{
  const int lo = __builtin_amdgcn_update_dpp (0, __double2loint (v), 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp (0, __double2hiint (v), 0x138, 0xf, 0xf, true);
  return __hiloint2double (hi, lo);
}
__device__ __forceinline__ double
dpp_from_upper_lane (double v)                                // lane i receives lane i + 1's value
{
  const int lo = __builtin_amdgcn_update_dpp (0, __double2loint (v), 0x130, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp (0, __double2hiint (v), 0x130, 0xf, 0xf, true);
  return __hiloint2double (hi, lo);
}

/* a value that is the same in every lane of the wave, moved to scalar registers: address arithmetic and comparisons on it then
 * run on the scalar unit instead of taking VALU issue slots (this kernel is bound by those) */
__device__ __forceinline__ long long
wave_uniform (long long v)
{
  const int lo = __builtin_amdgcn_readfirstlane (int (v)), hi = __builtin_amdgcn_readfirstlane (int (v >> 32));
  return (long long) (((unsigned long long) (unsigned int) hi << 32) | (unsigned int) lo);
}

/* a wave-uniform address in GLOBAL memory as a scalar base: accesses through it take the base from scalar registers and a 32 bit offset
 * per lane (global_load / global_store ... saddr) instead of a 64 bit address register pair per lane */
typedef __attribute__ ((address_space (1))) float global_float;
__device__ __forceinline__ global_float *
uniform_global (const float *p)
{
  return (global_float *) (unsigned long long) wave_uniform ((long long) (unsigned long long) p);
}

// STATUS: <1> is the product path for MONO streams; <2> ("form 0", round 2) only runs behind awm_debug_set_refine_form (0) as one of the
// three kernels test_refinement_kernel_forms holds against each other bit for bit -- stereo streams take sync_db_sliding4_kernel.
template<int CV> __global__ void __launch_bounds__ (64 * WAVES) __attribute__ ((amdgpu_waves_per_eu (3, 3)))
sync_db_sliding_kernel (DevTables t, SyncDbArgs a)
{
  constexpr int SCRATCH_FLOATS = XBUF_ELEMS * 4 > NB * SL_TILE ? XBUF_ELEMS * 4 : NB * SL_TILE;
  __shared__ __attribute__ ((aligned (16))) float s_scratch[WAVES][SCRATCH_FLOATS];    // FFT exchange tile (576 double2) / dB tile [offset][band]
  __shared__ unsigned char s_pos[WAVES][NB + 3];
  __shared__ __attribute__ ((aligned (16))) double s_delta[WAVES][SL_TILE * 8 * CV];     // sample differences of 16 steps
  __shared__ int s_nzd[WAVES][SL_TILE * CV];            // per transition and channel: non-zero samples entering minus leaving
  __shared__ int s_x0[WAVES][SL_TILE * CV];             // per step and channel: the first sample of the window is non-zero
  {
    // gathered output (refinement): which row of its 60 the stream's band b goes to; identity otherwise
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const long long s = (long long) blockIdx.x * WAVES + w;
    // (one key per clip: the tables of the stream's slice)
    const long long tslice = (a.tables_per_slice && s < a.n_streams) ? (a.range_index ? a.range_index[s / a.range_div] : s / a.range_div) : 0;
    for (int b = l; b < NB; b += 64)
      s_pos[w][b] = (a.band_pos && s < a.n_streams) ? a.band_pos[(tslice * a.rows_per_plane + s % a.rows_per_plane) * NB + b] : (unsigned char) b;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long stream = (long long) blockIdx.x * WAVES + wave;
  if (stream >= a.n_streams)
    return;
  // where this stream's rows go
  const long long perm_slice = a.tables_per_slice ? (a.range_index ? a.range_index[stream / a.range_div] : stream / a.range_div) : 0;
  const long long out_slot = wave_uniform (a.row_perm ? (stream / a.rows_per_plane) * a.rows_per_plane
                                                        + a.row_perm[perm_slice * a.rows_per_plane + stream % a.rows_per_plane] : stream);
  const long long base = wave_uniform (sync_stream_base (a, stream));
  const int count = __builtin_amdgcn_readfirstlane (a.stream_count ? a.stream_count[stream] : a.count0);
  if (count <= 0)
    return;
  long long sil_first, sil_last;
  sync_stream_range (a, stream, sil_first, sil_last);
  sil_first = wave_uniform (sil_first);
  sil_last = wave_uniform (sil_last);
  // every fine offset of this row inside the leading / trailing silence (syncfinder.cc:583-585; CLIP: the padding of a
  // padded clip): nothing to transform, the row is marked absent
  if (a.have && ((base + 8LL * (count - 1) + 1024) * CV < sil_first || base * CV > sil_last))
    {
      if (lane < count)
        a.have[out_slot * a.have_stream_stride + lane] = 0;
      if (lane == 0 && count > 64)
        a.have[out_slot * a.have_stream_stride + 64] = 0;
      return;
    }
  double2 *xbuf = reinterpret_cast<double2 *> (s_scratch[wave]);
  float *tile = s_scratch[wave];
  const int C = CV;
  const bool bins = lane < 42;                                // lane holds bins kA = 19 + 2 lane, kB = kA + 1
  const int kA = 19 + 2 * (bins ? lane : 0);

  // ---- first offset: plain (unwindowed) DFT bins from the wave FFT
  double2 R[CV][2];
  // Non-zero samples of the current window per channel.  A window of digital silence has an exactly zero spectrum in the
  // reference (-96 dB per band, wmcommon.hh:204-224); the recurrence below would arrive there with the rounding residue of
  // what it slid over (1e-15 of the previous content = -300 dB), so the bins are reset when the count reaches zero.
  // The same holds when the only non-zero sample sits at window position 0, whose Hann weight is exactly 0.
  int nz[CV];
  {
#pragma unroll
    for (int c = 0; c < CV; c++)
      {
        // one channel at a time (16 samples per lane live instead of 32: the double transform below needs the registers)
        float in[16];
        fetch_channel (a.pcm, base, 1024, CV, c, lane, in);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 16; j++)
          cnt += in[j] != 0.f;
        for (int o = 32; o > 0; o >>= 1)
          cnt += __shfl_xor (cnt, o);
        nz[c] = __builtin_amdgcn_readfirstlane (cnt);
        // in double: the recurrence carries the rounding of this transform through all fine offsets, and the Hann window is
        // applied in the frequency domain afterwards -- a float transform (1e-7 of the unwindowed content) shows where the
        // windowed content is tiny, e.g. for windows that slide into a gap of digital silence
        double2 z[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
          z[j] = make_double2 (double (in[2 * j]), double (in[2 * j + 1]));
        fft512_forward_d (z, xbuf, t.tw512d, lane);
        xbuf[0 * 64 + lane] = z[0];
        xbuf[1 * 64 + lane] = z[1];
        xbuf[6 * 64 + lane] = z[6];
        xbuf[7 * 64 + lane] = z[7];
        wave_sync();
#pragma unroll
        for (int b = 0; b < 2; b++)
          {
            const int k = kA + b;
            R[c][b] = real_split_d (xbuf[zpos (k)], xbuf[zpos (512 - k)], t.slide[(k - 19) * 9 + 1]);
          }
        wave_sync();
      }
  }
  // per-lane rotation constants
  double2 tw[2][8];                                           // e^{-2 pi i k j / N}, j = 1..7 (j = 0 is 1), then e^{+2 pi i 8 k / N}
  int k_tab = kA - 19;
  asm volatile ("" : "+v" (k_tab));                           // keep these 64 registers out of the FFT above (no hoisting)
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int j = 0; j < 8; j++)
      tw[b][j] = t.slide[(k_tab + b) * 9 + j + 1];

  // Sample feed.  The transition step -> step + 1 needs the 8 C samples entering the window and the 8 C leaving it.
  // Sixteen transitions are fetched at once (both blocks are contiguous: 128 C floats, 2 C per lane), a whole block
  // of steps before they are needed, turned into double differences and published in LDS, where every lane reads
  // them back as broadcasts: no global load is ever waited for inside the step loop.
  constexpr int FPL = 2 * CV;                                     // floats per lane and block
  float f_in[FPL], f_out[FPL];
  auto fetch_block = [&] (int q) {
    const long long s0 = base + 128LL * q;
#pragma unroll
    for (int i = 0; i < FPL; i++)
      {
        const int e = lane * FPL + i;                             // (transition * 8 + j) * C + c
        const int trans = SL_TILE * q + e / (8 * C);
        const bool need = trans + 1 < count;
        f_in[i] = need ? a.pcm[(s0 + 1024) * C + e] : 0.f;
        f_out[i] = trans < count ? a.pcm[s0 * C + e] : 0.f;       // the first 8 samples of the window of step `trans`
      }
  };
  auto publish_block = [&] () {
#pragma unroll
    for (int i = 0; i < FPL; i++)
      s_delta[wave][lane * FPL + i] = double (f_in[i]) - double (f_out[i]);
    // a lane holds FPL = 2 C consecutive values (transition * 8 + j) * C + c: two j of every channel; 4 lanes = one transition
#pragma unroll
    for (int c = 0; c < CV; c++)
      {
        int dn = (f_in[c] != 0.f) - (f_out[c] != 0.f) + (f_in[CV + c] != 0.f) - (f_out[CV + c] != 0.f);
        dn += __shfl_xor (dn, 1);
        dn += __shfl_xor (dn, 2);
        if ((lane & 3) == 0)
          {
            s_nzd[wave][(lane >> 2) * CV + c] = dn;
            s_x0[wave][(lane >> 2) * CV + c] = f_out[c] != 0.f;   // window position 0 of that step: weight 0 in the Hann window
          }
      }
  };
  fetch_block (0);
  unsigned long long have_mask = 0;      // offsets 0..63
  bool have_64 = false;                  // offset 64 (a candidate has at most 65 fine offsets)

  for (int step = 0; step < count; step++)
    {
      if (step % SL_TILE == 0)
        {
          publish_block();                                        // fetched one block of steps ago
          wave_sync();
          fetch_block (step / SL_TILE + 1);
        }
      // ---- output for this fine offset
      const long long idx = base + 8LL * step;
      const long long f_first = idx * C, f_last = (idx + 1024) * C;
      const bool skip = (f_last < sil_first) || (f_first > sil_last);
      float dbA = 0.f, dbB = 0.f;
      if (!skip)
        {
          if (step < 64)
            have_mask |= 1ULL << step;
          else
            have_64 = true;
#pragma unroll
          for (int c = 0; c < CV; c++)
            {
              // every sample that carries weight is zero (position 0 has none): exactly zero frame in the reference
              if (nz[c] - __builtin_amdgcn_readfirstlane (s_x0[wave][(step % SL_TILE) * CV + c]) == 0)
                {
                  dbA = __fadd_rn (dbA, -96.f);
                  dbB = __fadd_rn (dbB, -96.f);
                  continue;
                }
              // neighbours: R[kA - 1] lives in lane - 1 (its kB), R[kB + 1] in lane + 1 (its kA): DPP wave shifts
              const double2 up = make_double2 (dpp_from_lower_lane (R[c][1].x), dpp_from_lower_lane (R[c][1].y));
              const double2 dn = make_double2 (dpp_from_upper_lane (R[c][0].x), dpp_from_upper_lane (R[c][0].y));
              // X[k] = (R[k] / 2 - (R[k-1] + R[k+1]) / 4) / 256 = (2 R[k] - (R[k-1] + R[k+1])) / 1024: one rounding in the
              // subtraction either way, the power of two scalings are exact
              const float xa_re = float (fma (2.0, R[c][0].x, -(up.x + R[c][1].x))) * 0x1p-10f;
              const float xa_im = float (fma (2.0, R[c][0].y, -(up.y + R[c][1].y))) * 0x1p-10f;
              const float xb_re = float (fma (2.0, R[c][1].x, -(R[c][0].x + dn.x))) * 0x1p-10f;
              const float xb_im = float (fma (2.0, R[c][1].y, -(R[c][0].y + dn.y))) * 0x1p-10f;
              dbA = __fadd_rn (dbA, db_from_complex (make_float2 (xa_re, xa_im)));
              dbB = __fadd_rn (dbB, db_from_complex (make_float2 (xb_re, xb_im)));
            }
        }
      const int col = step % SL_TILE;
      if (bins)
        {
          const int bandA = kA - MIN_BAND, bandB = bandA + 1;           // -1 .. 80 / 0 .. 82
          // [offset][band] with the odd row length 81: the 42 writing lanes and the transposed flush are conflict free
          if (bandA >= 0 && bandA < NB)
            tile[col * NB + bandA] = dbA;
          if (bandB < NB)
            tile[col * NB + bandB] = dbB;
        }
      if (col == SL_TILE - 1 || step == count - 1)
        {
          wave_sync();
          const int t0 = step - col, n_cols = col + 1;
          float *out = a.out + out_slot * a.out_stream_stride + t0;
          for (int i = lane; i < NB * SL_TILE; i += 64)
            {
              const int band = i / SL_TILE, cc = i % SL_TILE;
              const int row = s_pos[wave][band];
              if (cc < n_cols && row != 255)
                out[row * a.ld + cc] = tile[cc * NB + band];
            }
          wave_sync();
        }
      // ---- advance by 8 samples
      if (step + 1 < count)
        {
          const double *dl = s_delta[wave] + (step % SL_TILE) * 8 * C;
#pragma unroll
          for (int c = 0; c < CV; c++)
            {
              double2 acc[2] = { R[c][0], R[c][1] };
#pragma unroll
              for (int j = 0; j < 8; j++)
                {
                  const double d = dl[j * C + c];
#pragma unroll
                  for (int b = 0; b < 2; b++)
                    {
                      if (j == 0)
                        acc[b].x = acc[b].x + d;
                      else
                        {
                          acc[b].x = fma (d, tw[b][j - 1].x, acc[b].x);
                          acc[b].y = fma (d, tw[b][j - 1].y, acc[b].y);
                        }
                    }
                }
#pragma unroll
              for (int b = 0; b < 2; b++)
                {
                  const double2 r = tw[b][7];
                  R[c][b] = make_double2 (acc[b].x * r.x - acc[b].y * r.y, acc[b].x * r.y + acc[b].y * r.x);
                }
              nz[c] += __builtin_amdgcn_readfirstlane (s_nzd[wave][(step % SL_TILE) * CV + c]);
              if (nz[c] == 0)
                R[c][0] = R[c][1] = make_double2 (0.0, 0.0);
            }
        }
      if (step % SL_TILE == SL_TILE - 1)
        wave_sync();                                              // all reads of this block's differences done
    }
  if (a.have && lane < count)
    a.have[out_slot * a.have_stream_stride + lane] = (have_mask >> lane) & 1;
  if (a.have && lane == 0 && count > 64)
    a.have[out_slot * a.have_stream_stride + 64] = have_64;
}

/* K4s for stereo with all but 8 lanes at work: a lane carries THREE adjacent bins of ONE channel (lanes 0..27: channel 0, bins
 * 19 + 3 l .. 21 + 3 l; lanes 28..55: channel 1) instead of two bins of both channels on 42 lanes -- 3 (bin, channel) pairs per lane
 * instead of 4 for the same 168 pairs: a quarter fewer FP64 instructions per fine offset and wave (the kernel is bound by VALU
 * issue).  The Hann neighbours of the inner bin are in the lane itself, the outer ones come from the adjacent lanes as before
 * (at the seam between the channels the neighbours are wrong, but only for bins 19 and 102, which are neighbours themselves and
 * never output).  The channels' dB values meet in the LDS tile ([offset][channel][band]) and are added when the tile is flushed:
 * 0 + db0 + db1 in the reference's order.  Everything else -- first transform in double, recurrence in double, non-zero sample
 * counts, skip rules, output layout -- is sync_db_sliding_kernel's.
 * STATUS: superseded by sync_db_sliding4_kernel (round 6, same values); kept behind awm_debug_set_refine_form (3) as the pinned
 * reference of test_refinement_kernel_forms and of tools/gpu_k4s_forms.py's before / after. */
__global__ void __launch_bounds__ (64 * WAVES) __attribute__ ((amdgpu_waves_per_eu (3, 3)))
sync_db_sliding3_kernel (DevTables t, SyncDbArgs a)
{
  constexpr int CV = 2, LPC = 28;                             // lanes per channel
  constexpr int TILE_FLOATS = 2 * NB * SL_TILE;
  constexpr int SCRATCH_FLOATS = XBUF_ELEMS * 4 > TILE_FLOATS ? XBUF_ELEMS * 4 : TILE_FLOATS;
  __shared__ __attribute__ ((aligned (16))) float s_scratch[WAVES][SCRATCH_FLOATS];    // FFT exchange tile (576 double2) / dB tile [offset][channel][band]
  __shared__ unsigned char s_pos[WAVES][NB + 3];
  __shared__ __attribute__ ((aligned (16))) double s_delta[WAVES][SL_TILE * 8 * CV];
  __shared__ int s_nzd[WAVES][SL_TILE * CV];
  __shared__ int s_x0[WAVES][SL_TILE * CV];
  {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const long long s = (long long) blockIdx.x * WAVES + w;
    const long long tslice = (a.tables_per_slice && s < a.n_streams) ? (a.range_index ? a.range_index[s / a.range_div] : s / a.range_div) : 0;
    for (int b = l; b < NB; b += 64)
      s_pos[w][b] = (a.band_pos && s < a.n_streams) ? a.band_pos[(tslice * a.rows_per_plane + s % a.rows_per_plane) * NB + b] : (unsigned char) b;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long stream = (long long) blockIdx.x * WAVES + wave;
  if (stream >= a.n_streams)
    return;
  const long long perm_slice = a.tables_per_slice ? (a.range_index ? a.range_index[stream / a.range_div] : stream / a.range_div) : 0;
  const long long out_slot = wave_uniform (a.row_perm ? (stream / a.rows_per_plane) * a.rows_per_plane
                                                        + a.row_perm[perm_slice * a.rows_per_plane + stream % a.rows_per_plane] : stream);
  const long long base = wave_uniform (sync_stream_base (a, stream));
  const int count = __builtin_amdgcn_readfirstlane (a.stream_count ? a.stream_count[stream] : a.count0);
  if (count <= 0)
    return;
  long long sil_first, sil_last;
  sync_stream_range (a, stream, sil_first, sil_last);
  sil_first = wave_uniform (sil_first);
  sil_last = wave_uniform (sil_last);
  if (a.have && ((base + 8LL * (count - 1) + 1024) * CV < sil_first || base * CV > sil_last))
    {
      if (lane < count)
        a.have[out_slot * a.have_stream_stride + lane] = 0;
      if (lane == 0 && count > 64)
        a.have[out_slot * a.have_stream_stride + 64] = 0;
      return;
    }
  double2 *xbuf = reinterpret_cast<double2 *> (s_scratch[wave]);
  float *tile = s_scratch[wave];
  const bool active = lane < 2 * LPC;
  const int ch = lane >= LPC ? 1 : 0;                         // (idle lanes ride along as channel 1, bins of lane 0)
  const int li = active ? lane - LPC * ch : 0;
  const int kA = 19 + 3 * li;

  // ---- first offset: plain (unwindowed) DFT bins from the wave FFT, in double (see sync_db_sliding_kernel)
  double2 R[3];
  int nz0 = 0, nz1 = 0;                                        // non-zero samples of the current window, per channel (wave uniform)
#pragma unroll
  for (int c = 0; c < CV; c++)
    {
      float in[16];
      fetch_channel (a.pcm, base, 1024, CV, c, lane, in);
      int cnt = 0;
#pragma unroll
      for (int j = 0; j < 16; j++)
        cnt += in[j] != 0.f;
      for (int o = 32; o > 0; o >>= 1)
        cnt += __shfl_xor (cnt, o);
      (c == 0 ? nz0 : nz1) = __builtin_amdgcn_readfirstlane (cnt);
      double2 z[8];
#pragma unroll
      for (int j = 0; j < 8; j++)
        z[j] = make_double2 (double (in[2 * j]), double (in[2 * j + 1]));
      fft512_forward_d (z, xbuf, t.tw512d, lane);
      xbuf[0 * 64 + lane] = z[0];
      xbuf[1 * 64 + lane] = z[1];
      xbuf[6 * 64 + lane] = z[6];
      xbuf[7 * 64 + lane] = z[7];
      wave_sync();
      if (ch == c)
        {
#pragma unroll
          for (int b = 0; b < 3; b++)
            {
              const int k = kA + b;
              R[b] = real_split_d (xbuf[zpos (k)], xbuf[zpos (512 - k)], t.slide[(k - 19) * 9 + 1]);
            }
        }
      wave_sync();
    }
  double2 tw[3][8];                                           // e^{-2 pi i k j / N}, j = 1..7, then e^{+2 pi i 8 k / N}
  int k_tab = kA - 19;
  asm volatile ("" : "+v" (k_tab));                           // keep these 96 registers out of the FFT above (no hoisting)
#pragma unroll
  for (int b = 0; b < 3; b++)
#pragma unroll
    for (int j = 0; j < 8; j++)
      tw[b][j] = t.slide[(k_tab + b) * 9 + j + 1];

  constexpr int FPL = 2 * CV;
  float f_in[FPL], f_out[FPL];
  auto fetch_block = [&] (int q) {
    const long long s0 = base + 128LL * q;
#pragma unroll
    for (int i = 0; i < FPL; i++)
      {
        const int e = lane * FPL + i;                             // (transition * 8 + j) * C + c
        const int trans = SL_TILE * q + e / (8 * CV);
        const bool need = trans + 1 < count;
        f_in[i] = need ? a.pcm[(s0 + 1024) * CV + e] : 0.f;
        f_out[i] = trans < count ? a.pcm[s0 * CV + e] : 0.f;
      }
  };
  auto publish_block = [&] () {
#pragma unroll
    for (int i = 0; i < FPL; i++)
      s_delta[wave][lane * FPL + i] = double (f_in[i]) - double (f_out[i]);
#pragma unroll
    for (int c = 0; c < CV; c++)
      {
        int dn = (f_in[c] != 0.f) - (f_out[c] != 0.f) + (f_in[CV + c] != 0.f) - (f_out[CV + c] != 0.f);
        dn += __shfl_xor (dn, 1);
        dn += __shfl_xor (dn, 2);
        if ((lane & 3) == 0)
          {
            s_nzd[wave][(lane >> 2) * CV + c] = dn;
            s_x0[wave][(lane >> 2) * CV + c] = f_out[c] != 0.f;
          }
      }
  };
  fetch_block (0);
  unsigned long long have_mask = 0;
  bool have_64 = false;

  for (int step = 0; step < count; step++)
    {
      if (step % SL_TILE == 0)
        {
          publish_block();
          wave_sync();
          fetch_block (step / SL_TILE + 1);
        }
      const long long idx = base + 8LL * step;
      const long long f_first = idx * CV, f_last = (idx + 1024) * CV;
      const bool skip = (f_last < sil_first) || (f_first > sil_last);
      const int col = step % SL_TILE;
      float db[3] = { 0.f, 0.f, 0.f };
      if (!skip)
        {
          if (step < 64)
            have_mask |= 1ULL << step;
          else
            have_64 = true;
          // every sample that carries weight is zero (position 0 has none): exactly zero frame in the reference (-96 dB per band)
          const int x0_0 = __builtin_amdgcn_readfirstlane (s_x0[wave][col * CV + 0]), x0_1 = __builtin_amdgcn_readfirstlane (s_x0[wave][col * CV + 1]);
          const bool zero_frame = ch ? (nz1 - x0_1 == 0) : (nz0 - x0_0 == 0);
          // neighbours of the outer bins: R[kA - 1] is the third bin of lane - 1, R[kC + 1] the first bin of lane + 1
          const double2 up = make_double2 (dpp_from_lower_lane (R[2].x), dpp_from_lower_lane (R[2].y));
          const double2 dn = make_double2 (dpp_from_upper_lane (R[0].x), dpp_from_upper_lane (R[0].y));
          // X[k] = (2 R[k] - (R[k-1] + R[k+1])) / 1024 (see sync_db_sliding_kernel)
          const float xa_re = float (fma (2.0, R[0].x, -(up.x + R[1].x))) * 0x1p-10f;
          const float xa_im = float (fma (2.0, R[0].y, -(up.y + R[1].y))) * 0x1p-10f;
          const float xb_re = float (fma (2.0, R[1].x, -(R[0].x + R[2].x))) * 0x1p-10f;
          const float xb_im = float (fma (2.0, R[1].y, -(R[0].y + R[2].y))) * 0x1p-10f;
          const float xc_re = float (fma (2.0, R[2].x, -(R[1].x + dn.x))) * 0x1p-10f;
          const float xc_im = float (fma (2.0, R[2].y, -(R[1].y + dn.y))) * 0x1p-10f;
          db[0] = zero_frame ? -96.f : db_from_complex (make_float2 (xa_re, xa_im));
          db[1] = zero_frame ? -96.f : db_from_complex (make_float2 (xb_re, xb_im));
          db[2] = zero_frame ? -96.f : db_from_complex (make_float2 (xc_re, xc_im));
        }
      if (active)
        {
          // [offset][channel][band], row length 2 x 81 = 162: the 56 writing lanes hit 56 different banks per store
#pragma unroll
          for (int b = 0; b < 3; b++)
            {
              const int band = kA + b - MIN_BAND;                         // -1 .. 82
              if (band >= 0 && band < NB)
                tile[(col * 2 + ch) * NB + band] = db[b];
            }
        }
      if (col == SL_TILE - 1 || step == count - 1)
        {
          wave_sync();
          const int t0 = step - col, n_cols = col + 1;
          float *out = a.out + out_slot * a.out_stream_stride + t0;
          for (int i = lane; i < NB * SL_TILE; i += 64)
            {
              const int band = i / SL_TILE, cc = i % SL_TILE;
              const int row = s_pos[wave][band];
              if (cc < n_cols && row != 255)
                out[row * a.ld + cc] = __fadd_rn (__fadd_rn (0.f, tile[(cc * 2 + 0) * NB + band]), tile[(cc * 2 + 1) * NB + band]);
            }
          wave_sync();
        }
      // ---- advance by 8 samples
      if (step + 1 < count)
        {
          const double *dl = s_delta[wave] + col * 8 * CV + ch;
          double2 acc[3] = { R[0], R[1], R[2] };
#pragma unroll
          for (int j = 0; j < 8; j++)
            {
              const double d = dl[j * CV];
#pragma unroll
              for (int b = 0; b < 3; b++)
                {
                  if (j == 0)
                    acc[b].x = acc[b].x + d;
                  else
                    {
                      acc[b].x = fma (d, tw[b][j - 1].x, acc[b].x);
                      acc[b].y = fma (d, tw[b][j - 1].y, acc[b].y);
                    }
                }
            }
#pragma unroll
          for (int b = 0; b < 3; b++)
            {
              const double2 r = tw[b][7];
              R[b] = make_double2 (acc[b].x * r.x - acc[b].y * r.y, acc[b].x * r.y + acc[b].y * r.x);
            }
          nz0 += __builtin_amdgcn_readfirstlane (s_nzd[wave][col * CV + 0]);
          nz1 += __builtin_amdgcn_readfirstlane (s_nzd[wave][col * CV + 1]);
          if ((ch ? nz1 : nz0) == 0)
            R[0] = R[1] = R[2] = make_double2 (0.0, 0.0);
        }
      if (col == SL_TILE - 1)
        wave_sync();
    }
  if (a.have && lane < count)
    a.have[out_slot * a.have_stream_stride + lane] = (have_mask >> lane) & 1;
  if (a.have && lane == 0 && count > 64)
    a.have[out_slot * a.have_stream_stride + 64] = have_64;
}

/* K4s for stereo, restructured (round 6).  The arithmetic of a fine offset is sync_db_sliding3_kernel's; what changes is how little
 * else a step costs, how the rows leave the chip, and -- behind U32 -- the precision of the UPDATE TERM only:
 *   - a step is straight-line code: the output of offset t (Hann combination of R_t, dB) and the update R_t -> R_{t+1} are independent
 *     and sit in one basic block, so the scheduler interleaves them (before: three exec-masked regions per step in between).  All the
 *     rules that are the same for the whole wave -- a window of digital silence, a row inside the padding of a clip, the bins' reset
 *     when the window runs empty, a power below 2^-96 that needs log2f's denormal scaling -- are decided on the SCALAR unit from masks
 *     that are built once per block of 16 offsets, and share ONE rarely taken branch to the careful form of the output;
 *   - the state is kept scaled by 2^-10 (exact: the first transform's result and the sample differences are scaled, everything
 *     downstream is linear), which removes the two scalings per bin and offset;
 *   - channel 1 lives in lanes 32..59 and hands its dB values to channel 0's lanes through ds_bpermute (the LDS crossbar, no VALU
 *     slot); the tile in LDS holds the SUM, and only the rows a sync frame's bit sums: [60 rows][32 offsets];
 *   - THE STORES.  Measured with the stores removed (tools/gpu_k4s_alone.py), the kernel of rounds 3 - 5 spent 108 of its 369 us
 *     on them: a flush of 16 offsets wrote 64 byte pieces, 4 bytes per lane, at a row stride of 288 bytes -- half cache lines that
 *     the L2 could not merge.  Now a row holds offsets 0..63 in 256 bytes (ld = 64, two whole 128 byte lines), a flush happens
 *     every 32 offsets and writes 16 bytes per lane: eight lanes complete a line, an instruction eight rows; offset 64 (a candidate
 *     has 65) goes to a compact array of its own (`tail`: 60 floats per stream, one coalesced store), which K5g's tail wave reads.
 * U32 = false: every floating point operation and its order is the old kernel's: the values are BIT-IDENTICAL (pinned by
 * tests/test_gpu_parity.py::test_refinement_kernel_forms).
 * U32 = true: the update term U[k] = sum_j d[j] W^{jk} -- 16 of the 19 double precision operations per bin and offset -- is
 * accumulated in FLOAT (d = x[s + N + j] - x[s + j] rounded to float, rotation folded into the table: T_j = W^{jk} rho_k as
 * float2), converted once and added to R rho in double: R' = R rho + U'.  The state R itself, the recurrence and the Hann combination
 * stay double.  Measured (profiles/r06/k4s_forms.txt): 14 % faster, 3e-3 dB off at most on stationary noise -- and WHOLE dB off
 * where a window slides into a gap of digital silence: the error is relative to the unwindowed content that passed through the
 * window, not to what is left in it.  Not the default. */
constexpr int S4_FLUSH = 32;                                  // offsets per flush: 128 bytes of a row
constexpr int S4_LD    = S4_FLUSH + 1;                        // tile row length: odd, the lanes' stores (one row each) take different banks
template<int ROWS, int DELTA_BYTES> struct S4Lds
{
  static constexpr int OFF_DUMMY = ROWS * S4_LD * 4;          // where the stores nobody reads go: word 3 lane + bin + column (no two lanes of a store collide)
  static constexpr int OFF_DELTA = ((OFF_DUMMY + (64 * 3 + S4_FLUSH) * 4 + 15) / 16) * 16;
  static constexpr int TOTAL     = OFF_DELTA + SL_TILE * 2 * 8 * DELTA_BYTES;
  static constexpr int BYTES     = TOTAL > XBUF_ELEMS * 16 ? TOTAL : XBUF_ELEMS * 16;   // the first transform's exchange tile lies over it all
};

template<bool U32, int ROWS> __device__ __forceinline__ void
sync_db_sliding4_body (const DevTables& t, const SyncDbArgs& a)
{
  constexpr int CV = 2, LPC = 28;
  typedef typename std::conditional<U32, float, double>::type delta_t;
  typedef S4Lds<ROWS, int (sizeof (delta_t))> L;
  __shared__ __attribute__ ((aligned (16))) unsigned char s_mem[WAVES][L::BYTES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane (threadIdx.x >> 6);
  const long long stream = (long long) blockIdx.x * WAVES + wave;
  if (stream >= a.n_streams)
    return;
  const long long tslice = a.tables_per_slice ? (a.range_index ? a.range_index[stream / a.range_div] : stream / a.range_div) : 0;
  const long long out_slot = wave_uniform (a.row_perm ? (stream / a.rows_per_plane) * a.rows_per_plane
                                                        + a.row_perm[tslice * a.rows_per_plane + stream % a.rows_per_plane] : stream);
  const long long base = wave_uniform (sync_stream_base (a, stream));
  const int count = __builtin_amdgcn_readfirstlane (a.stream_count ? a.stream_count[stream] : a.count0);
  if (count <= 0)
    return;
  long long sil_first, sil_last;
  sync_stream_range (a, stream, sil_first, sil_last);
  sil_first = wave_uniform (sil_first);
  sil_last = wave_uniform (sil_last);
  if (a.have && ((base + 8LL * (count - 1) + 1024) * CV < sil_first || base * CV > sil_last))
    {
      if (lane < count)
        a.have[out_slot * a.have_stream_stride + lane] = 0;
      if (lane == 0 && count > 64)
        a.have[out_slot * a.have_stream_stride + 64] = 0;
      return;
    }
  unsigned char *mem = s_mem[wave];
  double2 *xbuf = reinterpret_cast<double2 *> (mem);
  float *tile = reinterpret_cast<float *> (mem);                               // [row][S4_LD]
  delta_t *delta = reinterpret_cast<delta_t *> (mem + L::OFF_DELTA);           // [transition][channel][j]
  const int ch = lane >> 5;                                    // channel 0: lanes 0..27, channel 1: lanes 32..59
  const bool active = (lane & 31) < LPC;
  const int li = active ? (lane & 31) : 0;                     // (idle lanes ride along with the bins of their channel's first lane)
  const int kA = 19 + 3 * li;

  // ---- first offset: plain (unwindowed) DFT bins from the wave FFT, in double (see sync_db_sliding_kernel), scaled by 2^-10.
  // Both channels' samples come with one set of wide loads, the transform's twiddle factors are read once for both, the non-zero
  // counts are ballots (scalar population counts: no trips through the LDS crossbar).
  double2 R[3];
  int nz0 = 0, nz1 = 0;                                        // non-zero samples of the current window, per channel (wave uniform)
  {
    float in[CV][16];
    fetch_stereo (a.pcm, base, 1024, lane, in[0], in[1]);
    // (at four waves per SIMD -- the float form -- there is no room for the 56 registers of the two twiddle sets: read per channel)
    double2 tw1[U32 ? 1 : 7], tw2[U32 ? 1 : 7], ws[3];
    if constexpr (!U32)
      {
#pragma unroll
        for (int k = 1; k < 8; k++)
          {
            tw1[k - 1] = t.tw512d[lane * k];
            tw2[k - 1] = t.tw512d[8 * (lane & 7) * k];
          }
      }
#pragma unroll
    for (int b = 0; b < 3; b++)
      ws[b] = t.slide[(kA + b - 19) * 9 + 1];
#pragma unroll
    for (int j = 0; j < 16; j++)
      {
        nz0 += __builtin_popcountll (__builtin_amdgcn_ballot_w64 (in[0][j] != 0.f));
        nz1 += __builtin_popcountll (__builtin_amdgcn_ballot_w64 (in[1][j] != 0.f));
      }
#pragma unroll
    for (int c = 0; c < CV; c++)
      {
        double2 z[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
          z[j] = make_double2 (double (in[c][2 * j]), double (in[c][2 * j + 1]));
        if constexpr (U32)
          fft512_forward_d (z, xbuf, t.tw512d, lane);
        else
          fft512_forward_d (z, xbuf, tw1, tw2, lane);
        xbuf[0 * 64 + lane] = z[0];
        xbuf[1 * 64 + lane] = z[1];
        xbuf[6 * 64 + lane] = z[6];
        xbuf[7 * 64 + lane] = z[7];
        wave_sync();
        if (ch == c)
          {
#pragma unroll
            for (int b = 0; b < 3; b++)
              {
                const int k = kA + b;
                const double2 r = real_split_d (xbuf[zpos (k)], xbuf[zpos (512 - k)], ws[b]);
                R[b] = make_double2 (r.x * 0x1p-10, r.y * 0x1p-10);
              }
          }
        wave_sync();
      }
  }
  // ---- the lane's constants
  int k_tab = kA - 19;
  asm volatile ("" : "+v" (k_tab));                           // keep the tables' registers out of the transform above (no hoisting)
  double2 rho[3];                                             // e^{+2 pi i 8 k / N}
  double2 tw[U32 ? 1 : 3][U32 ? 1 : 7];                       // double form: e^{-2 pi i k j / N}, j = 1..7
  float2  tf[U32 ? 3 : 1][U32 ? 8 : 1];                       // float form: e^{-2 pi i k j / N} e^{+2 pi i 8 k / N}, j = 0..7
#pragma unroll
  for (int b = 0; b < 3; b++)
    {
      rho[b] = t.slide[(k_tab + b) * 9 + 8];
      if constexpr (U32)
        {
#pragma unroll
          for (int j = 0; j < 8; j++)
            tf[b][j] = t.slide32[(k_tab + b) * 8 + j];
        }
      else
        {
#pragma unroll
          for (int j = 0; j < 7; j++)
            tw[b][j] = t.slide[(k_tab + b) * 9 + j + 1];
        }
    }
  // where the lane's three dB values go: channel 0's lanes write the rows of their bands (the row of the output a band belongs to:
  // the 60 values a sync frame's bit sums in summation order -- gathered -- or the band itself), one column further per offset;
  // every other store (channel 1, idle lanes, bins 19 / 101 / 102, bands the frame does not use) keeps hitting a word of its own
  int tile_addr[3];
  {
    const unsigned char *pos = a.band_pos ? a.band_pos + (tslice * a.rows_per_plane + stream % a.rows_per_plane) * NB : nullptr;
#pragma unroll
    for (int b = 0; b < 3; b++)
      {
        const int band = kA + b - MIN_BAND;
        const bool in_range = active && ch == 0 && band >= 0 && band < NB;
        const int row = in_range ? (pos ? int (pos[band]) : band) : 255;
        const bool stores = row < ROWS;
        tile_addr[b] = stores ? row * S4_LD * 4 : L::OFF_DUMMY + (lane * 3 + b) * 4;
      }
  }
  int tile_col = 0;                                            // (wave uniform: column of the tile the next offset goes to, in bytes)
  const int partner = ((lane + 32) & 63) * 4;                 // ds_bpermute address of the same bins' other channel
  // Sample feed: the transition step -> step + 1 needs the 8 C samples entering the window and the 8 C leaving it.  Sixteen
  // transitions are fetched at once (both blocks are contiguous: 128 C floats, 2 C per lane), a whole block of steps ahead.
  constexpr int FPL = 2 * CV;
  float f_in[FPL], f_out[FPL];
  typedef float v2f __attribute__ ((ext_vector_type (2)));
  typedef __attribute__ ((address_space (1))) v2f global_v2f;
  auto fetch_block = [&] (int q) {
    // The block's two runs of samples start at wave-uniform addresses: scalar bases, one 32 bit lane offset -- no 64 bit address
    // registers to keep alive across the steps.  The loads are unconditional (a load behind a per-lane test makes the compiler
    // drain the memory counter -- the previous flush's stores -- before it may preset the register): lanes whose transition lies
    // beyond the stream read a valid place instead and publish_block drops what they got.
    if (SL_TILE * q >= count)
      return;
    const global_float *p_out = uniform_global (a.pcm + (base + 128LL * q) * CV);
    const int trans = SL_TILE * q + (lane >> 2);                  // lane * FPL + i = (transition * 8 + j) * C + c
    const unsigned at_out = trans < count ? unsigned (lane * FPL) : 0u;
    const unsigned at_in = trans + 1 < count ? unsigned (lane * FPL + 1024 * CV) : at_out;
#pragma unroll
    for (int i = 0; i < FPL; i += 2)
      {
        const v2f vo = *(const global_v2f *) (p_out + at_out + i), vi = *(const global_v2f *) (p_out + at_in + i);
        f_out[i] = vo.x; f_out[i + 1] = vo.y;                     // the first 8 samples of the window of step `trans`
        f_in[i] = vi.x;  f_in[i + 1] = vi.y;
      }
  };
  // per block of 16 offsets, one bit per offset at bit 4 * offset: the window carries no weighted non-zero sample (-96 dB exactly,
  // wmcommon.hh:204-224: position 0 has weight 0) | the window after the transition is all zeros (the bins restart from 0) | the
  // row lies in the padding of a padded clip (syncfinder.cc:583-585)
  unsigned long long zmask0 = 0, zmask1 = 0, rmask0 = 0, rmask1 = 0, skipmask = 0;
  auto publish_block = [&] (int t0) {
    const int tr = lane >> 2, jj = (lane & 3) * 2;
    {
      const bool need = t0 + tr + 1 < count, have_out = t0 + tr < count;
#pragma unroll
      for (int i = 0; i < FPL; i++)
        {
          f_in[i] = need ? f_in[i] : 0.f;
          f_out[i] = have_out ? f_out[i] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < FPL; i++)
      {
        delta_t d;
        if constexpr (U32)
          d = __fmul_rn (__fsub_rn (f_in[i], f_out[i]), 0x1p-10f);
        else
          d = (double (f_in[i]) - double (f_out[i])) * 0x1p-10;
        delta[(tr * 2 + (i & 1)) * 8 + jj + (i >> 1)] = d;
      }
#pragma unroll
    for (int c = 0; c < CV; c++)
      {
        int dn = (f_in[c] != 0.f) - (f_out[c] != 0.f) + (f_in[CV + c] != 0.f) - (f_out[CV + c] != 0.f);
        dn += __shfl_xor (dn, 1);
        dn += __shfl_xor (dn, 2);                                 // the transition's change of the non-zero count, in its four lanes
        int incl = dn;                                            // ... summed over the transitions up to this one
#pragma unroll
        for (int o = 4; o < 64; o <<= 1)
          {
            const int v = __shfl_up (incl, o);
            incl += lane >= o ? v : 0;
          }
        const int nz = c == 0 ? nz0 : nz1;
        const bool first_lane = (lane & 3) == 0;                  // holds sample j = 0 of the window of step tr: f_out[c]
        const unsigned long long z = __builtin_amdgcn_ballot_w64 (first_lane && nz + (incl - dn) - int (f_out[c] != 0.f) == 0);
        const unsigned long long r = __builtin_amdgcn_ballot_w64 (first_lane && nz + incl == 0);
        const int total = __builtin_amdgcn_readlane (incl, 63);
        if (c == 0) { zmask0 = z; rmask0 = r; nz0 += total; } else { zmask1 = z; rmask1 = r; nz1 += total; }
      }
    const long long idx = base + 8LL * (t0 + (lane >> 2));
    skipmask = __builtin_amdgcn_ballot_w64 ((lane & 3) == 0 && (((idx + 1024) * CV < sil_first) || (idx * CV > sil_last)));
  };
  fetch_block (0);
  unsigned long long have_mask = 0;      // offsets 0..63
  bool have_64 = false;                  // offset 64 (a candidate has at most 65 fine offsets)
  float *const out_base = a.out + out_slot * a.out_stream_stride;
  const int no_store = a.xcd_interleave & 8;                  // (measurement only: tools/gpu_k4s_alone.py)

  // Order of a block's end: the NEXT block's differences are published first (their loads were issued a block of steps ago, and so
  // were the previous flush's stores: the wait for the vector memory counter finds everything done), then the tile is flushed,
  // then the loads of the block after next go out.
  publish_block (0);
  wave_sync();
  fetch_block (1);
  for (int t0 = 0; t0 < count; t0 += SL_TILE)
    {
      const int n_cols = count - t0 < SL_TILE ? count - t0 : SL_TILE;
      for (int col = 0; col < n_cols; col++)
        {
          const int step = t0 + col;
          const unsigned sh = 4u * unsigned (col);
          const bool zero0 = (zmask0 >> sh) & 1, zero1 = (zmask1 >> sh) & 1, skip = (skipmask >> sh) & 1;
          // ---- the differences of this transition (two 16 / 32 byte broadcast reads)
          delta_t d[8];
          {
            const delta_t *dl = delta + (col * 2 + ch) * 8;
#pragma unroll
            for (int j = 0; j < 8; j++)
              d[j] = dl[j];
          }
          // ---- output for this fine offset: X[k] = (2 R[k] - (R[k-1] + R[k+1])) / 1024, the scaling is in R
          // neighbours of the outer bins: R[kA - 1] is the third bin of lane - 1, R[kC + 1] the first bin of lane + 1
          const double2 up = make_double2 (dpp_from_lower_lane (R[2].x), dpp_from_lower_lane (R[2].y));
          const double2 dn = make_double2 (dpp_from_upper_lane (R[0].x), dpp_from_upper_lane (R[0].y));
          float2 x[3];
          x[0] = make_float2 (float (fma (2.0, R[0].x, -(up.x + R[1].x))), float (fma (2.0, R[0].y, -(up.y + R[1].y))));
          x[1] = make_float2 (float (fma (2.0, R[1].x, -(R[0].x + R[2].x))), float (fma (2.0, R[1].y, -(R[0].y + R[2].y))));
          x[2] = make_float2 (float (fma (2.0, R[2].x, -(R[1].x + dn.x))), float (fma (2.0, R[2].y, -(R[1].y + dn.y))));
          float abs2[3], db[3];
#pragma unroll
          for (int b = 0; b < 3; b++)
            abs2[b] = __fadd_rn (__fmul_rn (x[b].x, x[b].x), __fmul_rn (x[b].y, x[b].y));
          // log2f scales arguments below the normal range; v_log_f32 alone is the same function from 2^-96 upwards: one test per
          // lane, one branch per wave -- shared with the wave-uniform special cases, which are all rare
          const bool small = fminf (fminf (abs2[0], abs2[1]), abs2[2]) < 0x1p-96f;
          if (__builtin_expect (__builtin_amdgcn_ballot_w64 (small) != 0 || zero0 || zero1 || skip, 0))
            {
              const bool zero_frame = ch ? zero1 : zero0;
#pragma unroll
              for (int b = 0; b < 3; b++)
                {
                  const float v = abs2[b] > 0 ? __fmul_rn (log2f (abs2[b]), 3.01029995663981f) : -96.f;
                  db[b] = skip ? 0.f : (zero_frame ? -96.f : v);
                }
            }
          else
            {
#pragma unroll
              for (int b = 0; b < 3; b++)
                db[b] = __fmul_rn (__builtin_amdgcn_logf (abs2[b]), 3.01029995663981f);
            }
          if (!skip)
            {
              if (step < 64)
                have_mask |= 1ULL << step;
              else
                have_64 = true;
            }
          // the other channel's values are asked for now and used behind the update below (the crossbar's latency is the update's time)
          float other[3];
#pragma unroll
          for (int b = 0; b < 3; b++)
            other[b] = __int_as_float (__builtin_amdgcn_ds_bpermute (partner, __float_as_int (db[b])));
          // ---- advance by 8 samples
          if (step + 1 < count)
            {
              if constexpr (U32)
                {
#pragma unroll
                  for (int b = 0; b < 3; b++)
                    {
                      float ux = __fmul_rn (d[0], tf[b][0].x), uy = __fmul_rn (d[0], tf[b][0].y);
#pragma unroll
                      for (int j = 1; j < 8; j++)
                        {
                          ux = fmaf (d[j], tf[b][j].x, ux);
                          uy = fmaf (d[j], tf[b][j].y, uy);
                        }
                      const double rx = R[b].x, ry = R[b].y;
                      R[b].x = fma (-ry, rho[b].y, fma (rx, rho[b].x, double (ux)));
                      R[b].y = fma (ry, rho[b].x, fma (rx, rho[b].y, double (uy)));
                    }
                }
              else
                {
#pragma unroll
                  for (int b = 0; b < 3; b++)
                    {
                      double ax = R[b].x + d[0], ay = R[b].y;
#pragma unroll
                      for (int j = 1; j < 8; j++)
                        {
                          ax = fma (d[j], tw[b][j - 1].x, ax);
                          ay = fma (d[j], tw[b][j - 1].y, ay);
                        }
                      R[b].x = fma (ax, rho[b].x, -(ay * rho[b].y));
                      R[b].y = fma (ax, rho[b].y, ay * rho[b].x);
                    }
                }
              const bool reset0 = (rmask0 >> sh) & 1, reset1 = (rmask1 >> sh) & 1;
              if (__builtin_expect (reset0 || reset1, 0))
                if (ch ? reset1 : reset0)
                  R[0] = R[1] = R[2] = make_double2 (0.0, 0.0);
            }
          // 0 + db0 + db1 in the reference's order (syncfinder.cc:594-599); channel 1's lanes compute a sum nobody reads.  (A skipped
          // row is 0 in both channels: the sum is the +0 the plain kernel writes.)
          __builtin_amdgcn_sched_barrier (0);
#pragma unroll
          for (int b = 0; b < 3; b++)
            *reinterpret_cast<float *> (mem + tile_addr[b] + tile_col) = __fadd_rn (__fadd_rn (0.f, db[b]), other[b]);
          tile_col += 4;
        }
      // Everything this wave has in flight in vector memory was issued a block of steps ago (the next block's samples, the previous
      // flush's stores): waiting for it HERE costs nothing -- and tells the compiler so, which otherwise drains the counter at the
      // worst place: behind the flush's stores, in front of the loads of the block after next.
      __builtin_amdgcn_s_waitcnt (0x0f70);                        // vmcnt (0)
      wave_sync();                                                // the block's columns are written, its differences have been read
      const int t_end = t0 + n_cols;                              // offsets done so far
      if (t_end < count)
        publish_block (t_end);
      // ---- flush: whole rows of 32 offsets (or what the stream has of them), 16 bytes per lane, eight lanes per 128 byte line
      if (t_end % S4_FLUSH == 0 || t_end == count)
        {
          const int f0 = (t_end - 1) / S4_FLUSH * S4_FLUSH, cols = t_end - f0;     // the flush covers offsets f0 .. f0 + cols - 1
          if (!no_store)
            {
              if (f0 == 64 && a.tail)
                {
                  // the 65th offset: ROWS values side by side in the compact array
                  global_float *tl = uniform_global (a.tail + out_slot * a.tail_stream_stride);
                  for (int row = lane; row < ROWS; row += 64)
                    tl[unsigned (row)] = tile[row * S4_LD];
                }
              else
                {
                  const int c4 = (lane & 7) * 4;
                  global_float *out = uniform_global (out_base + f0);                  // (the stores take a 32 bit offset from it)
                  const unsigned ld = unsigned (a.ld);
                  const bool vec = ((reinterpret_cast<uintptr_t> (out) | (ld * 4u)) & 15) == 0;
                  for (int row = lane >> 3; row < ROWS; row += 8)
                    {
                      const float *src = tile + row * S4_LD + c4;
                      const float v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
                      global_float *dst = out + (unsigned (row) * ld + unsigned (c4));
                      typedef float v4f __attribute__ ((ext_vector_type (4)));
                      if (vec && c4 + 3 < cols)
                        *(__attribute__ ((address_space (1))) v4f *) dst = (v4f) { v0, v1, v2, v3 };
                      else
                        {
                          if (c4 + 0 < cols) dst[0] = v0;
                          if (c4 + 1 < cols) dst[1] = v1;
                          if (c4 + 2 < cols) dst[2] = v2;
                          if (c4 + 3 < cols) dst[3] = v3;
                        }
                    }
                }
            }
          tile_col = 0;
          wave_sync();
        }
      fetch_block (t0 / SL_TILE + 2);
    }
  if (a.have && lane < count)
    a.have[out_slot * a.have_stream_stride + lane] = (have_mask >> lane) & 1;
  if (a.have && lane == 0 && count > 64)
    a.have[out_slot * a.have_stream_stride + 64] = have_64;
}

// (gathered: 60 rows; rows = the 81 bands: awm_debug_sync_db_sliding_d and nothing else)
__global__ void __launch_bounds__ (64 * WAVES) __attribute__ ((amdgpu_waves_per_eu (3, 3)))
sync_db_sliding4_kernel (DevTables t, SyncDbArgs a)
{
  sync_db_sliding4_body<false, 60> (t, a);
}
__global__ void __launch_bounds__ (64 * WAVES) __attribute__ ((amdgpu_waves_per_eu (4, 4)))
sync_db_sliding4f_kernel (DevTables t, SyncDbArgs a)
{
  sync_db_sliding4_body<true, 60> (t, a);
}
__global__ void __launch_bounds__ (64 * WAVES) __attribute__ ((amdgpu_waves_per_eu (2, 3)))
sync_db_sliding4_bands_kernel (DevTables t, SyncDbArgs a)
{
  sync_db_sliding4_body<false, NB> (t, a);
}
__global__ void __launch_bounds__ (64 * WAVES) __attribute__ ((amdgpu_waves_per_eu (2, 3)))
sync_db_sliding4f_bands_kernel (DevTables t, SyncDbArgs a)
{
  sync_db_sliding4_body<true, NB> (t, a);
}

/* which form of K4s runs for stereo streams (awm_debug_set_refine_form; the result of 0, 3 and 4 is the same to the last bit):
 *   0  sync_db_sliding_kernel<2>    two bins of both channels per lane (42 lanes)
 *   3  sync_db_sliding3_kernel      three bins of one channel per lane (56 lanes): rounds 3 - 5
 *   4  sync_db_sliding4_kernel      the same arithmetic, restructured (above)
 *   5  sync_db_sliding4f_kernel     the update term in float (above): NOT bit-identical, gated by the census of DESIGN.md section 4
 * (A form 6 -- the update term on the matrix cores, v_mfma_f64_4x4x4f64 with C = R, two streams x two channels per wave -- was built and
 * measured in round 6: bit-identical, 335 against 292 us for 12 750 streams alone.  FP64 matrix instructions run on the FP64 vector
 * pipeline of gfx950 and do 256 multiply-adds per ~18 cycles where v_fma_f64 does 64 per ~5: nothing to gain.  profiles/r06/k4s_form6.txt,
 * tools/mfma_f64_probe.hip; the kernel is in the history: "K4s form 6".) */
int g_refine_form = 4;
extern "C" void awm_debug_set_refine_form (int form) { g_refine_form = (form == 0 || form == 3 || form == 4 || form == 5) ? form : 4; }
extern "C" int  awm_debug_refine_form() { return g_refine_form; }
extern "C" void awm_debug_set_sliding3 (int on) { g_refine_form = on ? 3 : 0; }     // (rounds 3 - 5's toggle, kept for tools/gpu_variants.py)

bool sliding_rows_have_tail (int n_channels) { return n_channels == 2 && (g_refine_form == 4 || g_refine_form == 5); }

hipError_t
launch_sync_db_sliding (hipStream_t st, const DevTables& t, const SyncDbArgs& a)
{
  if (a.n_streams <= 0 || a.count0 <= 0)
    return hipSuccess;
  if (a.hop != 8 || a.count0 > 65 || a.per_channel || (a.n_channels != 1 && a.n_channels != 2))
    return hipErrorInvalidValue;
  const unsigned grid = unsigned ((a.n_streams + WAVES - 1) / WAVES);
  if (a.n_channels == 2 && g_refine_form == 5 && a.band_pos)
    hipLaunchKernelGGL (sync_db_sliding4f_kernel, dim3 (grid), dim3 (64 * WAVES), 0, st, t, a);
  else if (a.n_channels == 2 && g_refine_form == 5)
    hipLaunchKernelGGL (sync_db_sliding4f_bands_kernel, dim3 (grid), dim3 (64 * WAVES), 0, st, t, a);
  else if (a.n_channels == 2 && g_refine_form == 4 && a.band_pos)
    hipLaunchKernelGGL (sync_db_sliding4_kernel, dim3 (grid), dim3 (64 * WAVES), 0, st, t, a);
  else if (a.n_channels == 2 && g_refine_form == 4)
    hipLaunchKernelGGL (sync_db_sliding4_bands_kernel, dim3 (grid), dim3 (64 * WAVES), 0, st, t, a);
  else if (a.n_channels == 2 && g_refine_form == 3)
    hipLaunchKernelGGL (sync_db_sliding3_kernel, dim3 (grid), dim3 (64 * WAVES), 0, st, t, a);
  else if (a.n_channels == 2)
    hipLaunchKernelGGL (sync_db_sliding_kernel<2>, dim3 (grid), dim3 (64 * WAVES), 0, st, t, a);
  else
    hipLaunchKernelGGL (sync_db_sliding_kernel<1>, dim3 (grid), dim3 (64 * WAVES), 0, st, t, a);
  return hipGetLastError();
}

/* ==========================================================================================
 * K5: sync_decode for 64 candidates x 6 sync bits per workgroup
 * ========================================================================================== */
__global__ void __launch_bounds__ (384)
sync_scan_kernel (SyncScanArgs a)
{
  __shared__ float s_u[6][64], s_d[6][64];
  __shared__ int   s_n[6][64];
  const int lane = threadIdx.x;
  const int bit = __builtin_amdgcn_readfirstlane (threadIdx.y);     // one wave == one sync bit
  const long long plane = blockIdx.y;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md), so give every XCD one contiguous
  // range of candidate tiles -- the dB rows its resident workgroups re-read then stay inside that XCD's 4 MiB L2
  const long long n_tiles = (a.n_lanes + 63) / 64;
  const long long per_xcd = (n_tiles + 7) / 8;
  const long long tile = (long long) (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const bool tile_ok = (blockIdx.x >> 3) < per_xcd && tile < n_tiles;
  if (!tile_ok)
    return;                                               // uniform for the workgroup (grid is padded to 8 per XCD round)
  const long long cand = tile * 64 + lane;
  const long long n_valid = a.lane_count ? a.lane_count[plane] : a.n_lanes;
  const bool active = tile_ok && cand < n_valid;
  const float *db = a.db + plane * a.plane_stride + (active ? cand : 0);
  const char *have = a.have ? a.have + plane * a.have_plane_stride + (active ? cand : 0) : nullptr;
  const int R = a.table.rows_per_bit;
  const_int_ptr tab = (const_int_ptr) (a.table.packed + (size_t) bit * R * 64);

  // float accumulators, strictly sequential adds in the reference's order (syncfinder.cc:129-145)
  float umag = 0.f, dmag = 0.f;
  int n = 0;
  for (int r = 0; r < R; r++)
    {
      const_int_ptr tr = tab + r * 64;
      const long long row = tr[60];
      const bool present = have ? have[row * a.have_row_stride] != 0 : true;
      if (have && !__any (present))
        continue;                                         // CLIP: most rows of a padded clip lie in the silent padding
      const float *p = db + row * a.row_stride;
      float uv[30], dv[30];
#pragma unroll
      for (int i = 0; i < 30; i++)
        {
          uv[i] = p[tr[i] * a.band_stride];
          dv[i] = p[tr[30 + i] * a.band_stride];
        }
      if (present)
        {
#pragma unroll
          for (int i = 0; i < 30; i++)
            {
              umag = __fadd_rn (umag, uv[i]);
              dmag = __fadd_rn (dmag, dv[i]);
            }
          n++;
        }
    }
  s_u[bit][lane] = umag;
  s_d[bit][lane] = dmag;
  s_n[bit][lane] = n;
  __syncthreads();
  if (bit == 0 && active)
    {
      double q = 0;
      int total = 0;
      for (int b = 0; b < 6; b++)
        {
          const float um = s_u[b][lane], dm = s_d[b][lane];
          // SyncFinder::bit_quality (reference syncfinder.cc:94-114): float division and subtraction
          float raw;
          if (um == 0 || dm == 0)
            raw = 0;
          else if (um < dm)
            raw = __fsub_rn (1.f, __fdiv_rn (um, dm));
          else
            raw = __fsub_rn (__fdiv_rn (dm, um), 1.f);
          const double rb = (b & 1) ? double (raw) : -double (raw);
          q += rb * s_n[b][lane];
          total += s_n[b][lane];
        }
      if (total)
        q /= total;
      q = q / a.min_delta / 2.9;
      a.quality[plane * a.q_stride + cand] = q;
    }
}

hipError_t
launch_sync_scan (hipStream_t st, const SyncScanArgs& a)
{
  if (a.n_lanes <= 0 || a.n_planes <= 0)
    return hipSuccess;
  SyncScanArgs b = a;
  long long done = 0;
  while (done < a.n_planes)      // grid.y <= 65535
    {
      const long long n = (a.n_planes - done) > 65535 ? 65535 : (a.n_planes - done);
      b.db = a.db + done * a.plane_stride;
      b.have = a.have ? a.have + done * a.have_plane_stride : nullptr;
      b.lane_count = a.lane_count ? a.lane_count + done : nullptr;
      b.quality = a.quality + done * a.q_stride;
      const long long per_xcd = ((a.n_lanes + 63) / 64 + 7) / 8;
      const dim3 grid (unsigned (per_xcd * 8), unsigned (n));
      hipLaunchKernelGGL (sync_scan_kernel, grid, dim3 (64, 6), 0, st, b);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess)
        return e;
      done += n;
    }
  return hipSuccess;
}

/* K5g: the refinement's sync_decode over the gathered layout (kernels.hh GatheredScanArgs).  One WAVE = one (sync bit, up / down)
 * addition chain of one candidate x 64 fine offsets; it reads its 85 x 30 rows front to back (every value exactly once, 256 B
 * coalesced per row) and adds them in the reference's order.  The chains of a candidate are independent until the six bit
 * qualities are combined (sync_quality_kernel below), so they are separate single-wave workgroups: a batch of 25 candidates is
 * 300 waves spread over all CUs -- with the twelve chains in ONE workgroup (rounds 2 and 3a) 25 CUs pulled 8.8 MB each at the
 * ~55 GB/s one CU gets from HBM, whatever the number of loads in flight.  A wave hides the latency itself: the loads of the next
 * TWO sync frames are in flight while the 30 dependent adds of the current one issue.
 * The 65th fine offset: a second wave per chain would run for one lane; instead ONE tail wave per candidate takes the offsets past
 * the last full 64 for all twelve chains (lane = chain x offset, possible while 12 x offsets <= 64). */
template<bool HAVE> __global__ void __launch_bounds__ (64)
sync_scan_gathered_kernel (GatheredScanArgs a)
{
  const int lane = threadIdx.x;
  const long long plane = blockIdx.y;
  const int count = a.lane_count[plane];
  int chain, cand;
  if (int (blockIdx.x) < a.regular_waves)
    {
      chain = blockIdx.x % 12;
      cand = (blockIdx.x / 12) * 64 + lane;
    }
  else
    {
      chain = lane / a.tail_lanes;
      cand = a.tail_first + lane % a.tail_lanes;
    }
  const bool active = chain < 12 && cand < count;
  if (!__any (active))
    return;
  if (!active)
    chain = 0, cand = 0;
  const int bit = chain >> 1, down = chain & 1;
  const int R = a.rows_per_bit;
  const int ld = a.ld;
  // (forms 4 / 5 of K4s: a row holds fine offsets 0..63, offset 64 of every row lies in the compact `tail` array: the tail wave's
  // lanes walk it with a stride of one value instead of a row)
  const bool from_tail = a.tail && cand >= ld;
  const long long step = from_tail ? 1 : ld;
  const float *p = from_tail ? a.tail + plane * a.tail_plane_stride + ((long long) bit * R * 60 + down * 30)
                             : a.db + plane * a.plane_stride + ((long long) bit * R * 60 + down * 30) * ld + cand;
  const int hld = a.have_ld ? a.have_ld : ld;
  const char *hv = HAVE ? a.have + plane * a.have_plane_stride + (long long) bit * R * hld + cand : nullptr;

  float mag = 0.f;
  int n = 0;
  auto issue = [&] (int r, float (&v)[30], bool& present) {
    const float *q = p + (long long) r * 60 * step;
    present = HAVE ? hv[r * hld] != 0 : true;
    if (HAVE && !__any (present))
      return;                                             // nothing of this row is used by any candidate of the wave
#pragma unroll
    for (int i = 0; i < 30; i++)
      v[i] = q[i * step];
  };
  auto accumulate = [&] (const float (&v)[30], bool present) {
    if (present)
      {
#pragma unroll
        for (int i = 0; i < 30; i++)
          mag = __fadd_rn (mag, v[i]);
        n++;
      }
  };
  // rows r, r + 1 are in flight in v[0..1] when the loop body starts (60 loads: the wave's counter of outstanding loads holds 63);
  // the steady state has no branch and the scheduler may not move loads across the fences (the compiler counts outstanding
  // loads exactly only in straight-line code and in program order), rows past the end are re-reads of the last one
  float v[3][30];
  bool pr[3] = { false, false, false };
  const int last = R - 1;
  int r = 0;
  if (R > 0)
    {
#pragma unroll
      for (int k = 0; k < 2; k++)
        {
          issue (k < last ? k : last, v[k], pr[k]);
          __builtin_amdgcn_sched_barrier (0);
        }
      for (; r + 3 <= R; r += 3)
        {
#pragma unroll
          for (int k = 0; k < 3; k++)
            {
              const int nx = r + k + 2;
              issue (nx < last ? nx : last, v[(k + 2) % 3], pr[(k + 2) % 3]);
              __builtin_amdgcn_sched_barrier (0);
              accumulate (v[k], pr[k]);
              __builtin_amdgcn_sched_barrier (0);
            }
        }
#pragma unroll
      for (int k = 0; k < 2; k++)
        if (r + k < R)
          accumulate (v[k], pr[k]);
    }
  if (active)
    {
      a.chain_mag[(plane * 12 + chain) * 128 + cand] = mag;
      if (!down)
        a.chain_n[(plane * 6 + bit) * 128 + cand] = n;
    }
}

/* SyncFinder::sync_decode's tail for the refinement: six bit qualities -> one sync quality per (candidate, fine offset) */
__global__ void __launch_bounds__ (64)
sync_quality_kernel (GatheredScanArgs a)
{
  const long long plane = blockIdx.y;
  const int cand = blockIdx.x * 64 + threadIdx.x;
  if (cand >= a.lane_count[plane])
    return;
  double q = 0;
  int total = 0;
  for (int b = 0; b < 6; b++)
    {
      const float um = a.chain_mag[(plane * 12 + 2 * b) * 128 + cand], dm = a.chain_mag[(plane * 12 + 2 * b + 1) * 128 + cand];
      const int n = a.chain_n[(plane * 6 + b) * 128 + cand];
      float raw;                                      // SyncFinder::bit_quality (reference syncfinder.cc:94-114)
      if (um == 0 || dm == 0)
        raw = 0;
      else if (um < dm)
        raw = __fsub_rn (1.f, __fdiv_rn (um, dm));
      else
        raw = __fsub_rn (__fdiv_rn (dm, um), 1.f);
      const double rb = (b & 1) ? double (raw) : -double (raw);
      q += rb * n;
      total += n;
    }
  if (total)
    q /= total;
  q = q / a.min_delta / 2.9;
  a.quality[plane * a.q_stride + cand] = q;
}

hipError_t
launch_sync_scan_gathered (hipStream_t st, const GatheredScanArgs& a0)
{
  if (a0.n_lanes <= 0 || a0.n_planes <= 0)
    return hipSuccess;
  if (!a0.lane_count || a0.n_planes > 65535 || a0.n_lanes > 128 || !a0.chain_mag || !a0.chain_n)
    return hipErrorInvalidValue;
  GatheredScanArgs a = a0;
  const int full = a.n_lanes / 64, rem = a.n_lanes % 64;
  const bool tail = rem > 0 && rem * 12 <= 64;
  a.regular_waves = 12 * (tail || rem == 0 ? full : full + 1);
  a.tail_first = 64 * full;
  a.tail_lanes = tail ? rem : 1;
  const dim3 grid (unsigned (a.regular_waves + (tail ? 1 : 0)), unsigned (a.n_planes));
  if (a.have)
    hipLaunchKernelGGL (sync_scan_gathered_kernel<true>, grid, dim3 (64), 0, st, a);
  else
    hipLaunchKernelGGL (sync_scan_gathered_kernel<false>, grid, dim3 (64), 0, st, a);
  hipLaunchKernelGGL (sync_quality_kernel, dim3 (unsigned ((a.n_lanes + 63) / 64), unsigned (a.n_planes)), dim3 (64), 0, st, a);
  return hipGetLastError();
}

/* K5b: local mean (reference syncfinder.cc:234-254): 41-tap window without the 7 centre taps */
__global__ void __launch_bounds__ (256)
local_mean_kernel (const double *q, long long q_stride, long long S, double *raw_sorted, double *local_mean)
{
  const long long p = (long long) blockIdx.x * blockDim.x + threadIdx.x;
  const long long n_scores = 4 * S;
  if (p >= n_scores)
    return;
  q += (long long) blockIdx.y * 4 * q_stride;            // slices (batched clip search): independent score lists
  raw_sorted += (long long) blockIdx.y * n_scores;
  local_mean += (long long) blockIdx.y * n_scores;
  double avg = 0;
  int n = 0;
  for (int j = -20; j <= 20; j++)
    {
      if (j > -4 && j < 4)
        continue;
      const long long idx = p + j;
      if (idx >= 0 && idx < n_scores)
        {
          avg += q[(idx & 3) * q_stride + (idx >> 2)];
          n++;
        }
    }
  if (n > 0)
    avg /= n;
  raw_sorted[p] = q[(p & 3) * q_stride + (p >> 2)];
  local_mean[p] = avg;
}

hipError_t
launch_local_mean (hipStream_t st, const double *q, long long q_stride, long long n_start_frames, double *raw_sorted, double *local_mean,
                   int n_slices)
{
  const long long n = 4 * n_start_frames;
  if (n <= 0 || n_slices <= 0)
    return hipSuccess;
  hipLaunchKernelGGL (local_mean_kernel, dim3 (unsigned ((n + 255) / 256), unsigned (n_slices)), dim3 (256), 0, st, q, q_stride, n_start_frames,
                      raw_sorted, local_mean);
  return hipGetLastError();
}

/* ==========================================================================================
 * K7: mix_decode -- one thread per soft bit, double accumulators in the reference's order
 * GENERIC FORM: what runs when frames_per_bit x channels x 30 items of a bit do not fit soft_bits_wave_kernel's tile (3+ channels,
 * --frames-per-bit > 2) or behind awm_debug_set_soft_bits_generic; the streams of the bench take soft_bits_wave_kernel.
 * ========================================================================================== */
__global__ void __launch_bounds__ (128)
soft_bits_kernel (SoftBitsArgs a)
{
  const long long blk = blockIdx.y;
  const int bit = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_bits = a.n_data_frames / a.frames_per_bit;
  if (bit >= n_bits)
    return;
  const float *db = a.db + blk * a.block_stride;
  const int C = a.n_channels;
  const long long mix0 = a.block_slice ? (long long) a.block_slice[blk] * a.n_data_frames * 30 : 0;
  const int16_t *mix_frame = a.mix_frame + mix0;
  const uint8_t *mix_up = a.mix_up + mix0, *mix_down = a.mix_down + mix0;
  double umag = 0, dmag = 0;
  for (int f = bit * a.frames_per_bit; f < (bit + 1) * a.frames_per_bit; f++)
    for (int ch = 0; ch < C; ch++)
      {
        const float *plane = db + (long long) ch * NB * a.ld;
        for (int j = 0; j < 30; j++)
          {
            const int b = f * 30 + j;
            const int frame = mix_frame[b];
            // neighbours reflected at the block edges (reference wmget.cc:87-88)
            const int next = frame + 1 < a.block_frames ? frame + 1 : frame - 1;
            const int prev = frame - 1 >= 0 ? frame - 1 : frame + 1;
            const float *pu = plane + (long long) (mix_up[b] - MIN_BAND) * a.ld;
            const float *pd = plane + (long long) (mix_down[b] - MIN_BAND) * a.ld;
            umag += pu[frame];
            umag -= double (__fadd_rn (pu[prev], pu[next])) * 0.5;
            dmag += pd[frame];
            dmag -= double (__fadd_rn (pd[prev], pd[next])) * 0.5;
          }
      }
  a.out[blk * n_bits + bit] = float (umag - dmag);
}

/* K7 as one wave per soft bit.  A bit sums 30 mix entries x frames_per_bit x C terms, each of them three scattered dB values of
 * the block's matrix (table lookup first, then the value: two dependent trips to memory).  With one thread per bit the
 * 720 loads of a bit were issued almost one after the other (250 us for 31 000 threads on an otherwise idle GPU); here the
 * lanes of a wave fetch the terms in parallel and stage them in LDS, and lane 0 adds them up in the reference's order
 * (wmget.cc:67-108: double accumulators, terms in entry order), which keeps the result bit-identical. */
constexpr int SB_WAVES = 4;           // waves per workgroup
constexpr int SB_BPW = 4;             // bits per wave
constexpr int SB_MAX_ITEMS = 256;     // frames_per_bit * C * 30 terms per bit (stereo: 120); more -> one thread per bit kernel

/* One wave gathers the terms of SB_BPW bits into LDS (64 lanes, coalesced index reads, one term per lane and pass), then lanes
 * 0 .. SB_BPW - 1 each add up one bit's terms front to back in double, in the reference's order.  (Round 3a: one bit per wave, its sum
 * on lane 0 alone -- 600 double precision instructions per bit at 1 / 64 of the SIMD: the launch was bound by those additions.) */
__global__ void __launch_bounds__ (64 * SB_WAVES)
soft_bits_wave_kernel (SoftBitsArgs a, int groups_per_block)
{
  extern __shared__ __attribute__ ((aligned (16))) float4 s_item_dyn[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware order (1-D grid): workgroup id runs on XCD id % 8, and all workgroups of a block go to ONE XCD, so that the
  // block's dB matrix (1.4 MB for stereo) is fetched into one L2 instead of eight (PMC FETCH_SIZE: 278 -> 38 MB fetched per launch for
  // 53 MB of matrices; blockIdx.y = block spread every block over all eight)
  const long long slot = blockIdx.x >> 3;
  const long long blk = (slot / groups_per_block) * 8 + (blockIdx.x & 7);
  const int group = int (slot % groups_per_block);
  if (blk >= a.n_blocks)
    return;
  const int n_bits = a.n_data_frames / a.frames_per_bit;
  const int bit0 = (group * SB_WAVES + wave) * SB_BPW;
  if (bit0 >= n_bits)
    return;
  const float *db = a.db + blk * a.block_stride;
  const int C = a.n_channels;
  const long long mix0 = a.block_slice ? (long long) a.block_slice[blk] * a.n_data_frames * 30 : 0;
  const int16_t *mix_frame = a.mix_frame + mix0;
  const uint8_t *mix_up = a.mix_up + mix0, *mix_down = a.mix_down + mix0;
  const int n_items = a.frames_per_bit * C * 30;
  float4 *items = s_item_dyn + (size_t) wave * SB_BPW * n_items;
  const int n_here = min (SB_BPW, n_bits - bit0);
  // blocks of padded slices: the frames [live_lo, live_hi] of the block are the ones K4b transformed (its rule, sync_db_kernel: a frame is
  // skipped if it ends before the first or starts after the last value that is not silence); the others hold -96 dB everywhere
  int live_lo = 0, live_hi = 0x7fffffff;
  if (a.stream_range)
    {
      const long long sl = a.range_index[blk];
      const long long first = a.stream_range[2 * sl], last = a.stream_range[2 * sl + 1];
      live_lo = a.block_frames;                                // (first < 0: nothing but silence in the slice)
      live_hi = -1;
      if (first >= 0)
        {
          const long long base = a.block_base[blk];
          // frame f is live  <=>  (base + 1024 f + 1024) C >= first  and  (base + 1024 f) C <= last
          const long long lo_num = (first + C - 1) / C - 1024 - base, hi_num = last / C - base;   // 1024 f >= lo_num, 1024 f <= hi_num
          const long long lo = lo_num <= 0 ? 0 : (lo_num + 1023) / 1024, hi = hi_num < 0 ? -1 : hi_num / 1024;
          live_lo = int (lo < a.block_frames ? lo : a.block_frames);
          live_hi = int (hi < a.block_frames ? hi : a.block_frames - 1);
        }
    }
  const float zero_db = db_from_complex (make_float2 (0.f, 0.f));
  // the passes of the four bits are independent: unrolled, their index loads and then their value loads are in flight together (two
  // memory round trips per group of four passes instead of per pass)
  for (int i = lane; i < n_items; i += 64)
#pragma unroll
  for (int q = 0; q < SB_BPW; q++)
    {
      const int e = q * n_items + i;
      const int bit = bit0 + min (q, n_here - 1);            // (past the last bit: the last one again, into a slot nobody adds up -- no branch)
      // item order = summation order: frame of the bit, channel, entry
      const int fi = i / (C * 30), ch = (i / 30) % C, j = i % 30;
      const int b = (bit * a.frames_per_bit + fi) * 30 + j;
      const int frame = mix_frame[b];
      // neighbours reflected at the block edges (reference wmget.cc:87-88)
      const int next = frame + 1 < a.block_frames ? frame + 1 : frame - 1;
      const int prev = frame - 1 >= 0 ? frame - 1 : frame + 1;
      // the frame and both neighbours (as reflected at the block's edges) lie in the silence of the padding
      if (max (frame, max (prev, next)) < live_lo || min (frame, min (prev, next)) > live_hi)
        {
          items[e] = make_float4 (zero_db, __fadd_rn (zero_db, zero_db), zero_db, __fadd_rn (zero_db, zero_db));
          continue;
        }
      const float *plane = db + (long long) ch * NB * a.ld;
      const float *pu = plane + (long long) (mix_up[b] - MIN_BAND) * a.ld;
      const float *pd = plane + (long long) (mix_down[b] - MIN_BAND) * a.ld;
      items[e] = make_float4 (pu[frame], __fadd_rn (pu[prev], pu[next]), pd[frame], __fadd_rn (pd[prev], pd[next]));
    }
  wave_sync();
  if (lane < n_here)
    {
      const float4 *mine = items + lane * n_items;
      double umag = 0, dmag = 0;
      for (int i = 0; i < n_items; i++)
        {
          const float4 t = mine[i];
          umag = __dadd_rn (umag, double (t.x));
          umag = __dsub_rn (umag, __dmul_rn (double (t.y), 0.5));
          dmag = __dadd_rn (dmag, double (t.z));
          dmag = __dsub_rn (dmag, __dmul_rn (double (t.w), 0.5));
        }
      a.out[blk * n_bits + bit0 + lane] = float (__dsub_rn (umag, dmag));
    }
}

int g_soft_bits_generic = 0;     // (debug toggle: the one-thread-per-bit kernel for every shape)
extern "C" void awm_debug_set_soft_bits_generic (int on) { g_soft_bits_generic = on; }

hipError_t
launch_soft_bits (hipStream_t st, const SoftBitsArgs& a)
{
  if (a.n_blocks <= 0)
    return hipSuccess;
  const int n_bits = a.n_data_frames / a.frames_per_bit;
  if (!g_soft_bits_generic && a.frames_per_bit * a.n_channels * 30 <= SB_MAX_ITEMS)
    {
      const int groups = (n_bits + SB_WAVES * SB_BPW - 1) / (SB_WAVES * SB_BPW);
      const long long rounds = (a.n_blocks + 7) / 8;                   // 8 blocks (one per XCD) at a time
      const long long wgs = rounds * groups * 8;
      const size_t lds = size_t (SB_WAVES) * SB_BPW * (a.frames_per_bit * a.n_channels * 30) * sizeof (float4);
      if (wgs < (1LL << 31))
        {
          hipLaunchKernelGGL (soft_bits_wave_kernel, dim3 (unsigned (wgs)), dim3 (64 * SB_WAVES), lds, st, a, groups);
          return hipGetLastError();
        }
    }
  const dim3 grid (unsigned ((n_bits + 127) / 128), unsigned (a.n_blocks));
  hipLaunchKernelGGL (soft_bits_kernel, grid, dim3 (128), 0, st, a);
  return hipGetLastError();
}

/* scan_silence (reference syncfinder.cc:155-169) */
__global__ void __launch_bounds__ (256)
nonzero_range_kernel (const float *data, long long n_values, unsigned long long *result)
{
  const long long stride = (long long) gridDim.x * blockDim.x;
  unsigned long long first = ~0ULL, last = 0;
  for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += stride)
    if (data[i] != 0.f)
      {
        if ((unsigned long long) i < first) first = i;
        if ((unsigned long long) i + 1 > last) last = i + 1;
      }
  // one pair of atomics per workgroup (one per thread used to serialise half a million of them on two addresses:
  // 0.16 ms for a 30 s clip of noise)
  for (int o = 32; o > 0; o >>= 1)
    {
      const unsigned long long f = __shfl_xor (first, o), l = __shfl_xor (last, o);
      first = f < first ? f : first;
      last = l > last ? l : last;
    }
  __shared__ unsigned long long s_first[4], s_last[4];
  if ((threadIdx.x & 63) == 0)
    {
      s_first[threadIdx.x >> 6] = first;
      s_last[threadIdx.x >> 6] = last;
    }
  __syncthreads();
  if (threadIdx.x == 0)
    {
      for (int w = 1; w < 4; w++)
        {
          first = s_first[w] < first ? s_first[w] : first;
          last = s_last[w] > last ? s_last[w] : last;
        }
      if (first != ~0ULL)
        {
          atomicMin (result, first);
          atomicMax (result + 1, last);
        }
    }
}

/* K7b (kernels.hh SoftPrepArgs).  One workgroup per decode job.  The only order-sensitive step is the mean of |v| --
 * a double accumulated front to back (reference wmget.cc:52-55) -- which one thread does from LDS (<= 1716 adds). */
__global__ void __launch_bounds__ (256)
soft_prep_kernel (SoftPrepArgs a)
{
  __shared__ __attribute__ ((aligned (16))) float s_v[1728];
  __shared__ double s_mean;
  const SoftJobDev job = a.jobs[blockIdx.x];
  const int2 *src = a.src + job.src_off;
  const int nb = a.n_bits;
  const int *inv_order = a.inv_order + job.order_off;
  if (job.len > 1728)
    return;
  if (job.mode == 0)
    {
      const float *raw = a.raw + (long long) src[0].x * nb;
      for (int k = threadIdx.x; k < nb; k += 256)
        s_v[k] = raw[inv_order[k]];
    }
  else if (job.mode == 1)
    {
      for (int s = 0; s < job.n_src; s++)
        {
          const float *raw = a.raw + (long long) src[s].x * nb;
          const int half = src[s].y;
          for (int k = threadIdx.x; k < nb; k += 256)
            s_v[2 * k + half] = raw[inv_order[k]];
        }
    }
  else
    {
      const float div0 = float (job.norm0 > 1 ? job.norm0 : 1), div1 = float (job.norm1 > 1 ? job.norm1 : 1);
      for (int k = threadIdx.x; k < nb; k += 256)
        {
          const int i = inv_order[k];
          float acc0 = 0.f, acc1 = 0.f;                   // all_bits starts at zero and the blocks are added in list order
          int s = 0;
          for (; s + 8 <= job.n_src; s += 8)              // eight loads in flight, the additions in order behind them
            {
              float v[8];
              int half[8];
#pragma unroll
              for (int u = 0; u < 8; u++)
                {
                  const int2 sp = src[s + u];
                  v[u] = a.raw[(long long) sp.x * nb + i];
                  half[u] = sp.y;
                }
#pragma unroll
              for (int u = 0; u < 8; u++)
                {
                  if (half[u])
                    acc1 = __fadd_rn (acc1, v[u]);
                  else
                    acc0 = __fadd_rn (acc0, v[u]);
                }
            }
          for (; s < job.n_src; s++)
            {
              const float v = a.raw[(long long) src[s].x * nb + i];
              if (src[s].y)
                acc1 = __fadd_rn (acc1, v);
              else
                acc0 = __fadd_rn (acc0, v);
            }
          s_v[2 * k] = __fdiv_rn (acc0, div0);
          s_v[2 * k + 1] = __fdiv_rn (acc1, div1);
        }
    }
  __syncthreads();
  float *out = a.out + job.out_off;
  if (a.hard)
    {
      for (int k = threadIdx.x; k < job.len; k += 256)
        out[k] = s_v[k] > 0 ? 1.f : 0.f;
      return;
    }
  if (threadIdx.x == 0)
    {
      // front to back, one addition after the other (the order is the reference's); the VALUES come 32 at a time, so that the chain
      // waits for the adder and not for one LDS round trip per term
      double mean = 0;
      int k = 0;
      for (; k + 32 <= job.len; k += 32)
        {
          float4 q[8];
#pragma unroll
          for (int i = 0; i < 8; i++)
            q[i] = reinterpret_cast<const float4 *> (s_v + k)[i];
#pragma unroll
          for (int i = 0; i < 8; i++)
            {
              mean = __dadd_rn (mean, double (fabsf (q[i].x)));
              mean = __dadd_rn (mean, double (fabsf (q[i].y)));
              mean = __dadd_rn (mean, double (fabsf (q[i].z)));
              mean = __dadd_rn (mean, double (fabsf (q[i].w)));
            }
        }
      for (; k < job.len; k++)
        mean = __dadd_rn (mean, double (fabsf (s_v[k])));
      s_mean = __ddiv_rn (mean, double (job.len));
    }
  __syncthreads();
  const double mean = s_mean;
  for (int k = threadIdx.x; k < job.len; k += 256)
    out[k] = float (__dmul_rn (0.5, __dadd_rn (__ddiv_rn (double (s_v[k]), mean), 1.0)));
}

hipError_t
launch_soft_prep (hipStream_t st, const SoftPrepArgs& a)
{
  if (a.n_jobs <= 0)
    return hipSuccess;
  hipLaunchKernelGGL (soft_prep_kernel, dim3 (unsigned (a.n_jobs)), dim3 (256), 0, st, a);
  return hipGetLastError();
}

hipError_t
launch_nonzero_range (hipStream_t st, const float *data, long long n_values, unsigned long long *result)
{
  // result[0] = min index of a non-zero value (all ones if there is none), result[1] = max index + 1 (no host copy: async)
  hipError_t e = hipMemsetAsync (result, 0xff, sizeof (unsigned long long), st);
  if (e == hipSuccess)
    e = hipMemsetAsync (result + 1, 0, sizeof (unsigned long long), st);
  if (e != hipSuccess || n_values <= 0)
    return e;
  hipLaunchKernelGGL (nonzero_range_kernel, dim3 (1024), dim3 (256), 0, st, data, n_values, result);
  return hipGetLastError();
}

/* kernels.hh launch_clip_pad: grid (parts, clips); slice_values % 4 == 0, 16-byte stores */
__global__ void __launch_bounds__ (256)
clip_pad_kernel (const ClipSrc *src, float *dst, long long slice_values, long long margin_values, unsigned long long *range)
{
  const ClipSrc c = src[blockIdx.y];
  float4 *out = reinterpret_cast<float4 *> (dst + (long long) blockIdx.y * slice_values);
  const long long slice0 = (long long) blockIdx.y * slice_values;
  const long long stride = (long long) gridDim.x * blockDim.x;
  unsigned long long first = ~0ULL, last = 0;
  // only [pad_start - margin, pad_start + n_values + margin) of the slice is written: the consumers skip every frame that lies
  // outside the non-silent range found here (which is inside the clip) and read at most `margin` values beyond it
  long long q_lo = (c.pad_start - margin_values) / 4, q_hi = (c.pad_start + c.n_values + margin_values + 3) / 4;
  q_lo = q_lo < 0 ? 0 : q_lo;
  q_hi = q_hi > slice_values / 4 ? slice_values / 4 : q_hi;
  for (long long q = q_lo + (long long) blockIdx.x * blockDim.x + threadIdx.x; q < q_hi; q += stride)
    {
      const long long k = 4 * q - c.pad_start;              // source index of the quad's first value
      float v[4] = { 0.f, 0.f, 0.f, 0.f };
      if (k > -4 && k < c.n_values)
        {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (k + j >= 0 && k + j < c.n_values)
              v[j] = c.data[k + j];
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (v[j] != 0.f)
              {
                const unsigned long long at = (unsigned long long) (slice0 + 4 * q + j);
                first = at < first ? at : first;
                last = at + 1 > last ? at + 1 : last;
              }
        }
      out[q] = make_float4 (v[0], v[1], v[2], v[3]);
    }
  for (int o = 32; o > 0; o >>= 1)
    {
      const unsigned long long f = __shfl_xor (first, o), l = __shfl_xor (last, o);
      first = f < first ? f : first;
      last = l > last ? l : last;
    }
  __shared__ unsigned long long s_first[4], s_last[4];
  if ((threadIdx.x & 63) == 0)
    {
      s_first[threadIdx.x >> 6] = first;
      s_last[threadIdx.x >> 6] = last;
    }
  __syncthreads();
  if (threadIdx.x == 0)
    {
      for (int w = 1; w < 4; w++)
        {
          first = s_first[w] < first ? s_first[w] : first;
          last = s_last[w] > last ? s_last[w] : last;
        }
      if (first != ~0ULL)
        {
          atomicMin (range + 2 * blockIdx.y, first);          // all ones (= -1 as the signed value the consumers read) if nothing is found
          atomicMax (range + 2 * blockIdx.y + 1, last);
        }
    }
}

__global__ void
clip_range_init_kernel (unsigned long long *range, int n_clips)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_clips)
    {
      range[2 * i] = ~0ULL;
      range[2 * i + 1] = 0;
    }
}

/* (test knob) the slices are filled with NaNs before the padded copies are written: a consumer that reads a value the copy
 * kernel left out -- which would be stale data of an earlier group -- shows up as a changed result */
int g_clip_poison = 0;
extern "C" void awm_debug_set_clip_poison (int on) { g_clip_poison = on; }

hipError_t
launch_clip_pad (hipStream_t st, const ClipSrc *src, int n_clips, float *dst, long long slice_values, long long margin_values, long long *range)
{
  if (n_clips <= 0 || slice_values <= 0)
    return hipSuccess;
  if (slice_values % 4 || margin_values < 0 || (reinterpret_cast<uintptr_t> (dst) & 15))
    return hipErrorInvalidValue;
  if (g_clip_poison)
    if (hipError_t e = hipMemsetAsync (dst, 0xff, size_t (n_clips) * size_t (slice_values) * sizeof (float), st))
      return e;
  auto *r = reinterpret_cast<unsigned long long *> (range);
  hipLaunchKernelGGL (clip_range_init_kernel, dim3 (unsigned ((n_clips + 255) / 256)), dim3 (256), 0, st, r, n_clips);
  hipLaunchKernelGGL (clip_pad_kernel, dim3 (128, unsigned (n_clips)), dim3 (256), 0, st, src, dst, slice_values, margin_values, r);
  return hipGetLastError();
}

/* K5c: sync_select_local_maxima + sync_mask_avg_false_positives + the threshold part of
 * sync_select_threshold_and_n_best (reference syncfinder.cc:258-332, 364-383) evaluated per score.
 *
 * The reference walks the index-sorted scores sequentially ("a selected score makes its successor
 * ineligible"); since a successor can only qualify as a local maximum when it TIES with the selected
 * score, that rule is equivalent to: inside a run of consecutive qualifying positions take every
 * second one, starting with the first.  The false-positive mask looks at the selected maxima within
 * 23 search steps (index / 256) on either side -- exactly the scores p - 23 .. p + 23.
 * Survivors above `threshold` are appended (in no particular order) to `out`. */
__device__ __forceinline__ double
psel_absq (const double *raw, const double *mean, long long i)
{
  return fabs (raw[i] - mean[i]);
}
__device__ __forceinline__ bool
psel_cand (const double *raw, const double *mean, long long n, long long i)
{
  const double q = psel_absq (raw, mean, i);
  const double q_last = i > 0 ? psel_absq (raw, mean, i - 1) : 0;
  const double q_next = i + 1 < n ? psel_absq (raw, mean, i + 1) : 0;
  return q >= q_last && q >= q_next;
}
__device__ __forceinline__ bool
psel_is_max (const double *raw, const double *mean, long long n, long long i)
{
  if (!psel_cand (raw, mean, n, i))
    return false;
  int k = 0;
  while (i - 1 - k >= 0 && psel_cand (raw, mean, n, i - 1 - k))
    k++;
  return (k & 1) == 0;
}

__global__ void __launch_bounds__ (256)
peak_select_kernel (const double *raw, const double *mean, long long n, double threshold, unsigned int *count, int count_stride,
                    PeakOut *out, unsigned int cap)
{
  raw += (long long) blockIdx.y * n;                     // slices: independent score lists with their own counter and output list
  mean += (long long) blockIdx.y * n;
  count += (long long) blockIdx.y * count_stride;
  out += (long long) blockIdx.y * cap;
  const long long p = (long long) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || !psel_is_max (raw, mean, n, p))
    return;
  const double q = psel_absq (raw, mean, p);
  if (!(q > threshold))
    return;
  const int sign = raw[p] - mean[p] < 0 ? -1 : 1;
  for (int d = -23; d <= 23; d++)
    {
      const long long j = p + d;
      if (d == 0 || j < 0 || j >= n)
        continue;
      if (psel_absq (raw, mean, j) > q * 3 && (raw[j] - mean[j] < 0 ? -1 : 1) != sign && psel_is_max (raw, mean, n, j))
        return;                                  // masked by a much larger peak of the opposite sign
    }
  const unsigned int slot = atomicAdd (count, 1u);
  if (slot < cap)
    {
      out[slot].p = p;
      out[slot].raw = raw[p];
      out[slot].mean = mean[p];
    }
}

hipError_t
launch_peak_select (hipStream_t st, const double *raw, const double *mean, long long n, double threshold,
                    unsigned int *count, PeakOut *out, unsigned int cap)
{
  hipError_t e = hipMemsetAsync (count, 0, sizeof (unsigned int), st);
  if (e != hipSuccess || n <= 0)
    return e;
  hipLaunchKernelGGL (peak_select_kernel, dim3 (unsigned ((n + 255) / 256)), dim3 (256), 0, st, raw, mean, n, threshold, count, 0, out, cap);
  return hipGetLastError();
}

hipError_t
launch_peak_select_slices (hipStream_t st, const double *raw, const double *mean, long long n, double threshold,
                           unsigned int *count, int count_stride, PeakOut *out, unsigned int cap, int n_slices)
{
  if (n <= 0 || n_slices <= 0)
    return hipSuccess;
  hipLaunchKernelGGL (peak_select_kernel, dim3 (unsigned ((n + 255) / 256), unsigned (n_slices)), dim3 (256), 0, st, raw, mean, n, threshold,
                      count, count_stride, out, cap);
  return hipGetLastError();
}

/* K5d: the K largest |raw - mean| of an (unordered) peak list, per slice of the list.  Workgroup g owns slice g of
 * gridDim.x and writes its K best entries (order: quality descending, position ascending) to out[g * K ...];
 * unused slots get p = -1.  The global top K is a subset of the union, which the host merges (a few KB instead
 * of the whole list: the n_best fallback of syncfinder.cc:364-383 on unmarked or short material). */
__global__ void __launch_bounds__ (256)
peak_topk_kernel (const PeakOut *in, const unsigned int *count_ptr, int count_stride, unsigned int cap, PeakOut *out, int K)
{
  __shared__ double    s_q[256];
  __shared__ long long s_p[256];
  __shared__ long long s_i[256];
  in += (long long) blockIdx.y * cap;                    // lists (batched clip search)
  count_ptr += (long long) blockIdx.y * count_stride;
  out += (long long) blockIdx.y * gridDim.x * K;
  const unsigned int count = *count_ptr < cap ? *count_ptr : cap;
  const long long lo = (long long) count * blockIdx.x / gridDim.x;
  const long long hi = (long long) count * (blockIdx.x + 1) / gridDim.x;
  double last_q = 0;
  long long last_p = -1;
  bool have_last = false;
  for (int it = 0; it < K; it++)
    {
      double best_q = -1;
      long long best_p = 0, best_i = -1;
      for (long long i = lo + threadIdx.x; i < hi; i += 256)
        {
          const double q = fabs (in[i].raw - in[i].mean);
          const long long p = in[i].p;
          if (have_last && !(q < last_q || (q == last_q && p > last_p)))
            continue;                                   // already taken (or NaN: never selected)
          if (best_i < 0 || q > best_q || (q == best_q && p < best_p))
            {
              best_q = q;
              best_p = p;
              best_i = i;
            }
        }
      s_q[threadIdx.x] = best_q;
      s_p[threadIdx.x] = best_p;
      s_i[threadIdx.x] = best_i;
      __syncthreads();
      for (int step = 128; step > 0; step >>= 1)
        {
          if (threadIdx.x < step)
            {
              const int o = threadIdx.x + step;
              const bool take = s_i[o] >= 0 && (s_i[threadIdx.x] < 0 || s_q[o] > s_q[threadIdx.x]
                                                || (s_q[o] == s_q[threadIdx.x] && s_p[o] < s_p[threadIdx.x]));
              if (take)
                {
                  s_q[threadIdx.x] = s_q[o];
                  s_p[threadIdx.x] = s_p[o];
                  s_i[threadIdx.x] = s_i[o];
                }
            }
          __syncthreads();
        }
      const long long sel = s_i[0];
      if (sel >= 0)
        {
          last_q = s_q[0];
          last_p = s_p[0];
          have_last = true;
        }
      if (threadIdx.x == 0)
        {
          PeakOut o;
          if (sel >= 0)
            o = in[sel];
          else
            {
              o.p = -1;
              o.raw = o.mean = 0;
            }
          out[(long long) blockIdx.x * K + it] = o;
        }
      __syncthreads();
      if (sel < 0)
        {
          for (int rest = it + 1 + threadIdx.x; rest < K; rest += 256)
            {
              PeakOut o;
              o.p = -1;
              o.raw = o.mean = 0;
              out[(long long) blockIdx.x * K + rest] = o;
            }
          break;
        }
    }
}

hipError_t
launch_peak_topk (hipStream_t st, const PeakOut *in, const unsigned int *count, unsigned int cap, PeakOut *out, int k, int n_slices)
{
  if (k <= 0 || n_slices <= 0)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL (peak_topk_kernel, dim3 (n_slices), dim3 (256), 0, st, in, count, 0, cap, out, k);
  return hipGetLastError();
}

hipError_t
launch_peak_topk_lists (hipStream_t st, const PeakOut *in, const unsigned int *count, int count_stride, unsigned int cap, PeakOut *out,
                        int k, int n_slices, int n_lists)
{
  if (k <= 0 || n_slices <= 0 || n_lists <= 0)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL (peak_topk_kernel, dim3 (n_slices, n_lists), dim3 (256), 0, st, in, count, count_stride, cap, out, k);
  return hipGetLastError();
}

/* ==========================================================================================
 * PCM <-> float staging
 * ========================================================================================== */
__device__ __forceinline__ float
pcm_decode_value (const unsigned char *b, const PcmFormatDev& f)
{
  if (f.encoding == 2)
    {
      unsigned char tmp[8];
      for (int i = 0; i < f.width; i++)
        tmp[i] = b[f.big_endian ? f.width - 1 - i : i];
      if (f.width == 4)
        {
          float v;
          memcpy (&v, tmp, 4);
          return v;
        }
      double d;
      memcpy (&d, tmp, 8);
      return float (d);
    }
  unsigned int u = 0;
  for (int i = 0; i < f.width; i++)
    {
      const int significance = f.big_endian ? f.width - 1 - i : i;
      u |= (unsigned int) b[i] << (8 * (4 - f.width + significance));
    }
  if (f.encoding == 1)
    u ^= 0x80000000u;
  return __fmul_rn (float (int (u)), 1.0f / 2147483648.0f);
}

__device__ __forceinline__ void
pcm_encode_value (float v, unsigned char *b, const PcmFormatDev& f)
{
  if (f.encoding == 2)
    {
      const float c = v >= 1.f ? 1.f : (v <= -1.f ? -1.f : v);
      unsigned char tmp[8];
      if (f.width == 4)
        memcpy (tmp, &c, 4);
      else
        {
          const double d = c;
          memcpy (tmp, &d, 8);
        }
      for (int i = 0; i < f.width; i++)
        b[f.big_endian ? f.width - 1 - i : i] = tmp[i];
      return;
    }
  unsigned int u;
  if (f.direct16)
    {
      const float s = __fmul_rn (v, 32768.f);
      const int i = s >= 32767.f ? 32767 : (s <= -32768.f ? -32768 : int (s));
      u = (unsigned int) i << 16;
    }
  else
    {
      const float s = __fmul_rn (v, 2147483648.f);
      const int i = s >= 2147483648.f ? 2147483647 : (s <= -2147483648.f ? int (0x80000000u) : int (s));
      u = (unsigned int) i;
    }
  if (f.encoding == 1)
    u ^= 0x80000000u;
  for (int i = 0; i < f.width; i++)
    {
      const int significance = f.big_endian ? f.width - 1 - i : i;
      b[i] = (unsigned char) (u >> (8 * (4 - f.width + significance)));
    }
}

__global__ void __launch_bounds__ (256)
pcm_decode_kernel (const unsigned char *bytes, float *out, long long n_values, PcmFormatDev f)
{
  const long long stride = (long long) gridDim.x * blockDim.x;
  for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += stride)
    out[i] = pcm_decode_value (bytes + i * f.width, f);
}

// little-endian signed 16 bit: 4 values (8 bytes in, 16 bytes out) per thread
__global__ void __launch_bounds__ (256)
pcm_decode_s16le_kernel (const short4 *in, float4 *out, long long n_vec)
{
  const long long stride = (long long) gridDim.x * blockDim.x;
  for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride)
    {
      const short4 v = in[i];
      const float k = 1.0f / 32768.0f;
      out[i] = make_float4 (__fmul_rn (float (v.x), k), __fmul_rn (float (v.y), k), __fmul_rn (float (v.z), k), __fmul_rn (float (v.w), k));
    }
}

__global__ void __launch_bounds__ (256)
pcm_encode_kernel (const float *in, unsigned char *bytes, long long n_values, PcmFormatDev f)
{
  const long long stride = (long long) gridDim.x * blockDim.x;
  for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += stride)
    pcm_encode_value (in[i], bytes + i * f.width, f);
}

__global__ void __launch_bounds__ (256)
pcm_encode_s16le_kernel (const float4 *in, short4 *out, long long n_vec, int direct16)
{
  const long long stride = (long long) gridDim.x * blockDim.x;
  for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride)
    {
      const float4 v = in[i];
      const float x[4] = { v.x, v.y, v.z, v.w };
      short r[4];
#pragma unroll
      for (int j = 0; j < 4; j++)
        {
          if (direct16)
            {
              const float s = __fmul_rn (x[j], 32768.f);
              r[j] = short (s >= 32767.f ? 32767 : (s <= -32768.f ? -32768 : int (s)));
            }
          else
            {
              const float s = __fmul_rn (x[j], 2147483648.f);
              const int i32 = s >= 2147483648.f ? 2147483647 : (s <= -2147483648.f ? int (0x80000000u) : int (s));
              r[j] = short (i32 >> 16);
            }
        }
      out[i] = make_short4 (r[0], r[1], r[2], r[3]);
    }
}

static unsigned
pcm_grid (long long n)
{
  long long blocks = (n + 255) / 256;
  return unsigned (blocks > 256 * 16 ? 256 * 16 : (blocks < 1 ? 1 : blocks));
}

hipError_t
launch_pcm_decode (hipStream_t st, const unsigned char *bytes, float *out, long long n_values, PcmFormatDev f)
{
  if (n_values <= 0)
    return hipSuccess;
  long long done = 0;
  if (f.encoding == 0 && f.width == 2 && !f.big_endian && (reinterpret_cast<uintptr_t> (bytes) & 7) == 0 && (reinterpret_cast<uintptr_t> (out) & 15) == 0)
    {
      const long long n_vec = n_values / 4;
      if (n_vec)
        hipLaunchKernelGGL (pcm_decode_s16le_kernel, dim3 (pcm_grid (n_vec)), dim3 (256), 0, st, reinterpret_cast<const short4 *> (bytes),
                            reinterpret_cast<float4 *> (out), n_vec);
      done = n_vec * 4;
    }
  if (done < n_values)
    hipLaunchKernelGGL (pcm_decode_kernel, dim3 (pcm_grid (n_values - done)), dim3 (256), 0, st, bytes + done * f.width, out + done, n_values - done, f);
  return hipGetLastError();
}

hipError_t
launch_pcm_encode (hipStream_t st, const float *in, unsigned char *bytes, long long n_values, PcmFormatDev f)
{
  if (n_values <= 0)
    return hipSuccess;
  long long done = 0;
  if (f.encoding == 0 && f.width == 2 && !f.big_endian && (reinterpret_cast<uintptr_t> (bytes) & 7) == 0 && (reinterpret_cast<uintptr_t> (in) & 15) == 0)
    {
      const long long n_vec = n_values / 4;
      if (n_vec)
        hipLaunchKernelGGL (pcm_encode_s16le_kernel, dim3 (pcm_grid (n_vec)), dim3 (256), 0, st, reinterpret_cast<const float4 *> (in),
                            reinterpret_cast<short4 *> (bytes), n_vec, f.direct16);
      done = n_vec * 4;
    }
  if (done < n_values)
    hipLaunchKernelGGL (pcm_encode_kernel, dim3 (pcm_grid (n_values - done)), dim3 (256), 0, st, in + done, bytes + done * f.width, n_values - done, f);
  return hipGetLastError();
}

} // namespace awmk
