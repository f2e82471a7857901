// C ABI: kernel-level and pipeline-level entry points -- see include/awm_hip.h
#include "context.hh"
#include "syncfinder.hh"
#include "wmget.hh"
#include "wmspeed.hh"
#include "utils.hh"
#include "wmfile.hh"
#include <atomic>
#include <future>
#include <thread>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

using namespace awm;
namespace awm { Key capi_key (const uint8_t key[16]); }
namespace awm { extern int g_key_tables_on_device; int clip_key_tables_check (awm_ctx *ctx, const uint8_t *keys, size_t n_keys, long long mismatch_out[9]); }       // wmget.cc (also read by the clip batches of `get`)

namespace {

constexpr int LIMITER_BLOCK = Params::mark_sample_rate * int (Params::limiter_block_size_ms) / 1000;   // Limiter::set_block_size_ms
const float LIMITER_CEILING = float (Params::limiter_ceiling);

constexpr int MAX_FRAMES_PER_BIT = 8;

int
check_ctx (awm_ctx *ctx)
{
  if (!ctx)
    {
      set_error ("null context");
      return AWM_ERR_ARG;
    }
  hipError_t e = hipSetDevice (ctx->device);
  if (e != hipSuccess)
    {
      set_error ("hipSetDevice: " + hip_error_string (e));
      return AWM_ERR_HIP;
    }
  if (params().frames_per_bit < 1 || params().frames_per_bit > MAX_FRAMES_PER_BIT || params().payload_size != 128)
    {
      // --frames-per-bit (audiowmark.cc:675-679) sets the block's geometry -- 510 sync + 858 x frames_per_bit data frames --, which every
      // kernel takes as an argument; the tables that are BUILT on the device (K16 / K16g: clip batches with a key per clip) are laid
      // out for the default and hand over to the host builders otherwise.  The deprecated --short payloads are out of scope.
      set_error ("unsupported watermark parameters (frames_per_bit outside 1 .. 8 or a short payload)");
      return AWM_ERR_ARG;
    }
  return 0;
}

// prologue of every entry point that takes a context: the context's own settings (if it has any) are in force for the calling
// thread until the entry point returns, the device is selected, unsupported block geometries are refused
#define AWM_ENTER(ctx) \
  awm::ParamsBind params_bind__ ((ctx) ? (ctx)->own_params.get() : nullptr); \
  if (int rc__ = check_ctx (ctx)) return rc__

int
frames_per_span (awm_ctx *ctx, long long n_frames1024)
{
  // Every wave streams through L frames plus 2 halo frames.  All waves of one "round" (CUs x resident waves) start
  // and finish together, so pick the number of rounds k that minimises k * (L + 2) with L = ceil (F / (k * capacity)):
  // long spans amortise the halo, whole rounds avoid a half-empty tail.
  static int cus = 0;
  if (!cus)
    {
      hipDeviceProp_t prop;
      cus = 256;
      if (hipGetDeviceProperties (&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    }
  const int capacity = cus * 4 * awmk::add_mix_waves_per_simd();       // resident waves of the fused add kernel
  long long best_l = 4, best_cost = -1;
  for (int k = 1; k <= 16; k++)
    {
      long long l = (n_frames1024 + (long long) k * capacity - 1) / ((long long) k * capacity);
      l = std::max<long long> (l, 4);
      const long long cost = k * (l + 2);
      if (best_cost < 0 || cost < best_cost)
        {
          best_cost = cost;
          best_l = l;
        }
    }
  return int (std::min<long long> (best_l, 4096));
}

void
fill_pattern (const ResultSet::Pattern& p, awm_pattern& o)
{
  o.time = p.time;
  o.sync_index = p.sync_score.index;
  o.sync_quality = p.sync_score.quality;
  o.block_type = int (p.sync_score.block_type);
  o.type = int (p.type);
  o.decode_error = p.decode_error;
  o.speed = p.speed;
  o.n_bits = std::min<int> (p.bit_vec.size(), 128);
  for (int b = 0; b < o.n_bits; b++)
    o.bits[b] = p.bit_vec[b];
}

DeviceWav
make_wav (const float *pcm_d, size_t n_frames, int n_channels)
{
  DeviceWav w;
  w.data = pcm_d;
  w.n_frames = n_frames;
  w.n_channels = n_channels;
  w.sample_rate = Params::mark_sample_rate;
  return w;
}

} // namespace

extern "C" {

int
awm_stft_d (awm_ctx *ctx, const float *pcm_d, size_t n_frames, int n_channels, size_t start_index, size_t hop,
            size_t frame_count, float *out_d)
{
  AWM_ENTER (ctx);
  if (!pcm_d || !out_d || n_channels < 1)
    {
      set_error ("awm_stft_d: bad argument");
      return AWM_ERR_ARG;
    }
  if (frame_count && n_frames < start_index + (frame_count - 1) * hop + Params::frame_size)
    {
      set_error ("awm_stft_d: range exceeds the data");
      return AWM_ERR_ARG;
    }
  ProfScope ps (ctx, PROF_STFT, double (frame_count) * n_channels * 8200.0);
  AWM_HIP_CHECK (awmk::launch_stft_full (ctx->stream, ctx->tabs, pcm_d, n_channels, (long long) start_index, (long long) hop,
                                         (long long) frame_count, reinterpret_cast<float2 *> (out_d)));
  return 0;
}

int
awm_add_init_block_max_d (awm_ctx *ctx, float *block_max_d, size_t n_blocks)
{
  AWM_ENTER (ctx);
  unsigned int bits;
  std::memcpy (&bits, &LIMITER_CEILING, sizeof (bits));
  AWM_HIP_CHECK (awmk::launch_fill_u32 (ctx->stream, reinterpret_cast<unsigned int *> (block_max_d), bits, n_blocks));
  return 0;
}

static int
add_mix_impl (awm_ctx *ctx, const float *pcm_in_d, float *out_d, size_t n_frames, int n_channels,
              const int8_t *frame_mod_dev, double water_delta, size_t first_frame,
              const float *halo_before_d, const float *halo_after_d, float *block_max_d, size_t first_block, size_t n_blocks,
              WorkLane *lane = nullptr)
{
  hipStream_t st = lane ? lane->stream : ctx->stream;
  awmk::AddMixArgs a {};
  a.pcm_in = pcm_in_d;
  a.out = out_d;
  a.n_frames = (long long) n_frames;
  a.n_channels = n_channels;
  a.frame_mod = frame_mod_dev;
  // powf (mag, -params().water_delta * data_bit_sign): double product converted to float (reference wmadd.cc:79)
  a.neg_delta_up = float (-water_delta * 1);
  a.neg_delta_down = float (-water_delta * -1);
  a.first_frame = (long long) first_frame;
  a.halo_before = halo_before_d;
  a.halo_after = halo_after_d;
  a.block_max = reinterpret_cast<unsigned int *> (block_max_d);
  a.first_block = (long long) first_block;
  a.n_blocks = (long long) n_blocks;
  a.limiter_block = LIMITER_BLOCK;
  a.block_frames = int (mark_block_frame_count());
  a.frames_pad_start = int (Params::frames_pad_start);
  a.frames_per_span = frames_per_span (ctx, (long long) (n_frames + 1023) / 1024);
  {
    ProfScope ps (ctx, PROF_ADD_MIX, double (n_frames) * n_channels * 8.0, st);     // read + write every sample once
    AWM_HIP_CHECK (awmk::launch_add_mix (st, ctx->tabs, a));
  }
  if (ctx->snr_on)                    // add --snr: the mix before the limiter (reference wmadd.cc:553-563)
    AWM_HIP_CHECK (awmk::launch_power_sums (st, pcm_in_d, out_d, (long long) (n_frames * n_channels), ctx->ws_snr.as<double>()));
  return 0;
}

} // extern "C"
namespace awm {
int
add_mix_device (awm_ctx *ctx, const float *pcm_in_d, float *out_d, size_t n_frames, int n_channels, const int8_t *frame_mod_dev,
                double water_delta, size_t first_frame, const float *halo_before_d, const float *halo_after_d, float *block_max_d,
                size_t first_block, size_t n_blocks)
{
  return add_mix_impl (ctx, pcm_in_d, out_d, n_frames, n_channels, frame_mod_dev, water_delta, first_frame, halo_before_d, halo_after_d,
                       block_max_d, first_block, n_blocks);
}
}
extern "C" {

int
awm_ctx_snr_begin (awm_ctx *ctx)
{
  AWM_ENTER (ctx);
  if (int rc = ctx->ws_snr.reserve (2 * sizeof (double))) return rc;
  AWM_HIP_CHECK (hipMemsetAsync (ctx->ws_snr.ptr, 0, 2 * sizeof (double), ctx->stream));
  AWM_HIP_CHECK (hipStreamSynchronize (ctx->stream));          // (the lanes' streams start after this)
  ctx->snr_on = true;
  return 0;
}

int
awm_ctx_snr_end (awm_ctx *ctx, double *signal_power, double *delta_power)
{
  AWM_ENTER (ctx);
  if (!ctx->snr_on)
    {
      set_error ("awm_ctx_snr_end without awm_ctx_snr_begin");
      return AWM_ERR_ARG;
    }
  ctx->snr_on = false;
  double acc[2] = { 0, 0 };
  AWM_HIP_CHECK (hipDeviceSynchronize());
  AWM_HIP_CHECK (hipMemcpy (acc, ctx->ws_snr.ptr, sizeof (acc), hipMemcpyDeviceToHost));
  if (delta_power)
    *delta_power = acc[0];
  if (signal_power)
    *signal_power = acc[1];
  return 0;
}

int
awm_add_mix_d (awm_ctx *ctx, const float *pcm_in_d, float *out_d, size_t n_frames, int n_channels,
               const int8_t *frame_mod, double water_delta, size_t first_frame,
               const float *halo_before_d, const float *halo_after_d,
               float *block_max_d, size_t first_block, size_t n_blocks)
{
  AWM_ENTER (ctx);
  if (!pcm_in_d || !out_d || !frame_mod || n_channels < 1)
    {
      set_error ("awm_add_mix_d: bad argument");
      return AWM_ERR_ARG;
    }
  const size_t table_bytes = 2 * mark_block_frame_count() * Params::n_bands;
  if (int rc = ctx->ws_misc.reserve (table_bytes)) return rc;
  AWM_HIP_CHECK (hipMemcpyAsync (ctx->ws_misc.ptr, frame_mod, table_bytes, hipMemcpyHostToDevice, ctx->stream));
  return add_mix_impl (ctx, pcm_in_d, out_d, n_frames, n_channels, ctx->ws_misc.as<int8_t>(), water_delta, first_frame,
                       halo_before_d, halo_after_d, block_max_d, first_block, n_blocks);
}

int
awm_add_limit_d (awm_ctx *ctx, float *out_d, size_t n_frames, int n_channels, size_t first_sample,
                 const float *block_max_d, size_t first_block, size_t n_blocks)
{
  AWM_ENTER (ctx);
  const size_t tab_entries = awmk::limiter_tab_entries ((long long) n_frames, (long long) first_sample, LIMITER_BLOCK);
  if (int rc = ctx->ws_limit_tab.reserve ((tab_entries + 1) * sizeof (float2))) return rc;
  ProfScope ps (ctx, PROF_LIMITER, double (n_frames) * n_channels * 8.0);
  AWM_HIP_CHECK (awmk::launch_limiter (ctx->stream, out_d, (long long) n_frames, n_channels, (long long) first_sample, block_max_d,
                                       (long long) first_block, (long long) n_blocks, LIMITER_BLOCK, LIMITER_CEILING,
                                       ctx->ws_limit_tab.as<float2>(), tab_entries));
  return 0;
}

/* (measurement knob, tools/gpu_add_slabs.py) `add` in slabs of this many MB of output: the fused add of slab s, then the limiter for
 * everything whose look-ahead second is complete -- so that the limiter re-reads what the add has just written while it may still
 * be in the 256 MB memory-side cache.  0 (default): one add over the whole stream, then one limiter pass.  DESIGN.md section 3 has
 * the sweep: the fused add needs 4096 resident waves x long spans (the 2 halo frames per span are its overhead), a slab small enough
 * for the cache starves it. */
static int g_add_slab_mb = 0;
extern "C" void awm_debug_set_add_slab_mb (int mb) { g_add_slab_mb = mb < 0 ? 0 : mb; }

/* whole stream on one lane (stream + block maxima + limiter table of that lane; the context itself is lane 0) */
static int
add_full (awm_ctx *ctx, const float *pcm_in_d, float *out_d, size_t n_frames, int n_channels,
          const int8_t *frame_mod_dev, double water_delta, int use_limiter, WorkLane *lane = nullptr, ReadyMarks *marks = nullptr)
{
  if (!lane)
    lane = ctx;
  if (marks)
    {
      marks->disarm();
      marks->base = out_d;
      marks->n_frames = n_frames;
    }
  hipStream_t st = lane->stream;
  float *block_max = nullptr;
  const size_t n_blocks = n_frames / LIMITER_BLOCK + 2;
  if (use_limiter)
    {
      if (int rc = lane->ws_block_max.reserve (n_blocks * sizeof (float))) return rc;
      block_max = lane->ws_block_max.as<float>();
      unsigned int bits;
      std::memcpy (&bits, &LIMITER_CEILING, sizeof (bits));
      AWM_HIP_CHECK (awmk::launch_fill_u32 (st, reinterpret_cast<unsigned int *> (block_max), bits, n_blocks));
    }
  const size_t FRAME = Params::frame_size;
  size_t slab = g_add_slab_mb ? (size_t (g_add_slab_mb) << 20) / (n_channels * sizeof (float)) / FRAME * FRAME : 0;    // sample frames
  if (!use_limiter || pcm_in_d == out_d || slab < 64 * FRAME || n_frames < 2 * slab)
    slab = 0;
  if (slab)
    {
      // slabs of whole 1024-sample frames, the last one takes the rest; each one is a span with its neighbours' frames as halos
      // (add_mix_impl: the same entry the multi-GPU spans use -- output bit-identical to the whole-stream launch)
      const size_t tab_entries = awmk::limiter_tab_entries ((long long) n_frames, 0, LIMITER_BLOCK);
      if (int rc = lane->ws_limit_tab.reserve ((tab_entries + 1) * sizeof (float2))) return rc;
      const size_t n_slabs = n_frames / slab;
      size_t limited = 0;                                  // the limiter has run for samples [0, limited)
      for (size_t i = 0; i < n_slabs; i++)
        {
          const size_t a = i * slab, b = i + 1 == n_slabs ? n_frames : a + slab;
          if (int rc = add_mix_impl (ctx, pcm_in_d + a * n_channels, out_d + a * n_channels, b - a, n_channels, frame_mod_dev, water_delta,
                                     a / FRAME, a ? pcm_in_d + (a - FRAME) * n_channels : nullptr, b < n_frames ? pcm_in_d + b * n_channels : nullptr,
                                     block_max, 0, n_blocks, lane))
            return rc;
          // the ramp of limiter block k needs the maxima of k - 1, k, k + 1 (limiter.cc:99-124): complete once the add has passed (k + 2) BS
          const size_t upto = b == n_frames ? n_frames : (b / LIMITER_BLOCK >= 1 ? (b / LIMITER_BLOCK - 1) * size_t (LIMITER_BLOCK) : 0);
          if (upto > limited)
            {
              ProfScope ps (ctx, PROF_LIMITER, double (upto - limited) * n_channels * 8.0, st);
              // (the passes share the lane's ramp table: they run in stream order, a pass rebuilds the entries of its own blocks)
              AWM_HIP_CHECK (awmk::launch_limiter (st, out_d + limited * n_channels, (long long) (upto - limited), n_channels, (long long) limited, block_max, 0,
                                                   (long long) n_blocks, LIMITER_BLOCK, LIMITER_CEILING, lane->ws_limit_tab.as<float2>(), tab_entries));
              limited = upto;
            }
        }
      return 0;
    }
  if (int rc = add_mix_impl (ctx, pcm_in_d, out_d, n_frames, n_channels, frame_mod_dev, water_delta, 0, nullptr, nullptr,
                             block_max, 0, n_blocks, lane))
    return rc;
  // With marks (awm_add_get_watermark_d) the limiter runs in one pass per range of `get`'s chunk ends (every block maximum is known
  // after the mix: a pass needs nothing from the passes after it) and leaves an event behind each: the chunk that ends there may start.
  std::vector<size_t> ends;
  if (marks)
    for (const ChunkRange& c : plan_chunks (n_frames, n_channels))
      {
        size_t e = std::min (n_frames, (c.first_frame + c.n_frames + LIMITER_BLOCK - 1) / LIMITER_BLOCK * size_t (LIMITER_BLOCK));
        if (e < n_frames && n_frames - e < size_t (LIMITER_BLOCK))          // (no sliver of a last pass)
          e = n_frames;
        if (ends.empty() || e > ends.back())
          ends.push_back (e);
      }
  if (ends.empty() || ends.back() != n_frames)
    ends.push_back (n_frames);
  if (use_limiter)
    {
      const size_t tab_entries = awmk::limiter_tab_entries ((long long) n_frames, 0, LIMITER_BLOCK);
      if (int rc = lane->ws_limit_tab.reserve ((tab_entries + 1) * sizeof (float2))) return rc;
      size_t limited = 0;
      for (size_t e : ends)
        {
          {
            ProfScope ps (ctx, PROF_LIMITER, double (e - limited) * n_channels * 8.0, st);
            // (the passes share the lane's ramp table: they run in stream order, a pass rebuilds the entries of its own blocks)
            AWM_HIP_CHECK (awmk::launch_limiter (st, out_d + limited * n_channels, (long long) (e - limited), n_channels, (long long) limited, block_max, 0,
                                                 (long long) n_blocks, LIMITER_BLOCK, LIMITER_CEILING, lane->ws_limit_tab.as<float2>(), tab_entries));
          }
          limited = e;
          if (marks)
            {
              hipEvent_t ev = marks->next_event();
              if (!ev)
                {
                  set_error ("cannot create an event");
                  return AWM_ERR_HIP;
                }
              AWM_HIP_CHECK (hipEventRecord (ev, st));
              marks->marks.push_back ({ e, ev });
            }
        }
    }
  else if (marks)
    {
      hipEvent_t ev = marks->next_event();
      if (!ev)
        {
          set_error ("cannot create an event");
          return AWM_ERR_HIP;
        }
      AWM_HIP_CHECK (hipEventRecord (ev, st));
      marks->marks.push_back ({ n_frames, ev });
    }
  if (marks)
    marks->armed = true;
  return 0;
}

/* ResamplerImpl::create (reference resample.cc:233-270): zita's fixed-ratio Resampler if it takes the two rates, else
 * its VResampler with ratio new / old.  Both run as closed forms of the streaming classes (K10 / K12). */
struct RateConverter
{
  const ResampleTable *fixed = nullptr;
  VarResampleGeometry  var;
  double               ratio = 0;
  bool ok() const { return fixed || var.ok; }
  int  hl() const { return fixed ? fixed->hl : var.hl; }
  // input frame (of the stream without the leading null frames) where the window of output m starts + hl - 1
  size_t window_start (size_t m) const
  {
    return fixed ? size_t ((static_cast<unsigned __int128> (m) * unsigned (fixed->step)) / unsigned (fixed->np)) : var.window_start (m);
  }
  /* frames the streaming resampler delivers for n_in input frames when it is fed "hl - 1 null frames, the input, hl null
   * frames" (WavChunkLoader at EOF, wavchunkloader.cc:200-216): every m with window_start (m) <= n_in - 1 */
  size_t stream_frames (size_t n_in) const
  {
    if (fixed)
      return size_t ((static_cast<unsigned __int128> (n_in) * unsigned (fixed->np) + unsigned (fixed->step) - 1) / unsigned (fixed->step));
    return var.stream_frames (n_in);
  }
};

static RateConverter
rate_converter (awm_ctx *ctx, int rate_in, int rate_out)
{
  RateConverter rc;
  if (rate_in <= 0 || rate_out <= 0)
    return rc;
  rc.fixed = ctx->get_resample_table (rate_in, rate_out);
  if (!rc.fixed)
    {
      rc.ratio = double (rate_out) / rate_in;
      rc.var = var_resample_geometry (rc.ratio);
      if (!rc.var.ok)
        set_error (string_printf ("resampling from old_rate=%d to new_rate=%d not implemented", rate_in, rate_out));
    }
  return rc;
}

static int
resample_device (awm_ctx *ctx, const RateConverter& conv, const float *in_d, size_t n_in, int n_channels, float *out_d, size_t n_out)
{
  if (!conv.fixed)
    return resample_var_device (ctx, ctx, in_d, n_in, n_channels, conv.ratio, out_d, n_out);
  const ResampleTable& t = *conv.fixed;
  awmk::ResampleArgs ra {};
  ra.in = in_d;
  ra.n_in = (long long) n_in;
  ra.n_channels = n_channels;
  ra.ctab = t.ctab.as<float>();
  ra.hl = t.hl;
  ra.np = t.np;
  ra.step = t.step;
  ra.out = out_d;
  ra.n_out = (long long) n_out;
  ProfScope ps (ctx, PROF_RESAMPLE, double (n_in + n_out) * n_channels * 4.0);      // every input and output sample once
  AWM_HIP_CHECK (awmk::launch_resample (ctx->stream, ra));
  return 0;
}

size_t
awm_resample_frames (awm_ctx *ctx, size_t n_frames, int rate_in, int rate_out)
{
  if (check_ctx (ctx))
    return 0;
  const RateConverter conv = rate_converter (ctx, rate_in, rate_out);
  return conv.ok() ? conv.stream_frames (n_frames) : 0;
}

int
awm_resample_d (awm_ctx *ctx, const float *pcm_in_d, size_t n_frames, int n_channels, int rate_in, int rate_out,
                float *out_d, size_t n_out_frames)
{
  AWM_ENTER (ctx);
  const RateConverter conv = rate_converter (ctx, rate_in, rate_out);
  if (!conv.ok())
    return AWM_ERR_ARG;
  if ((n_frames && !pcm_in_d) || (n_out_frames && !out_d) || n_channels < 1)
    {
      set_error ("awm_resample_d: bad argument");
      return AWM_ERR_ARG;
    }
  return resample_device (ctx, conv, pcm_in_d, n_frames, n_channels, out_d, n_out_frames);
}

/* add_stream_watermark for a stream at another rate (WatermarkResampler, reference wmadd.cc:353-430 + 520-589): the
 * input is resampled to 44.1 kHz, the watermark SIGNAL is generated there frame by frame, resampled back and added to
 * the original; limiter blocks are one second at the input rate.  The reference feeds zero frames after the input until
 * everything is written, so every stage simply sees a zero extended input here. */
static int
add_full_rate (awm_ctx *ctx, const float *pcm_in_d, float *out_d, size_t n_frames, int C, const int8_t *frame_mod_dev,
               double water_delta, int use_limiter, int rate)
{
  const RateConverter down = rate_converter (ctx, rate, Params::mark_sample_rate);
  const RateConverter up = rate_converter (ctx, Params::mark_sample_rate, rate);
  if (!down.ok() || !up.ok())
    return AWM_ERR_ARG;
  if (!n_frames)
    return 0;
  // watermark frames (44.1 kHz) the last output sample can reach: window of output n ends at floor (n s / np) + hl (input index)
  const size_t last44 = up.window_start (n_frames - 1) + size_t (up.hl()) + 1;
  const size_t F = last44 / Params::frame_size + 1;              // watermark frames 0 .. F - 1 are needed
  const size_t n44 = (F + 1) * Params::frame_size;               // frame F's delta completes frame F - 1
  if (int rc = ctx->ws_rate_a.reserve (n44 * C * sizeof (float))) return rc;
  if (int rc = ctx->ws_rate_b.reserve (n44 * C * sizeof (float))) return rc;
  if (int rc = ctx->ws_rate_c.reserve (n_frames * C * sizeof (float))) return rc;
  float *x44 = ctx->ws_rate_a.as<float>(), *wm44 = ctx->ws_rate_b.as<float>(), *wm = ctx->ws_rate_c.as<float>();
  if (int rc = resample_device (ctx, down, pcm_in_d, n_frames, C, x44, n44)) return rc;
  {
    awmk::AddMixArgs a {};
    a.pcm_in = x44;
    a.out = wm44;
    a.n_frames = (long long) n44;
    a.n_channels = C;
    a.frame_mod = frame_mod_dev;
    a.neg_delta_up = float (-water_delta * 1);
    a.neg_delta_down = float (-water_delta * -1);
    a.limiter_block = LIMITER_BLOCK;
    a.block_frames = int (mark_block_frame_count());
    a.frames_pad_start = int (Params::frames_pad_start);
    a.frames_per_span = frames_per_span (ctx, (long long) (F + 1));
    a.delta_only = 1;
    ProfScope ps (ctx, PROF_ADD_MIX, double (n44) * C * 8.0);
    AWM_HIP_CHECK (awmk::launch_add_mix (ctx->stream, ctx->tabs, a));
  }
  // only frames 0 .. F - 1 of wm44 are complete; nothing later is read (n_in = F * 1024 makes the rest zero, unreachable anyway)
  if (int rc = resample_device (ctx, up, wm44, F * Params::frame_size, C, wm, n_frames)) return rc;
  const int lim_block = int (size_t (rate) * size_t (Params::limiter_block_size_ms) / 1000);
  const size_t n_blocks = n_frames / lim_block + 2;
  unsigned int *block_max = nullptr;
  if (use_limiter)
    {
      if (int rc = ctx->ws_block_max.reserve (n_blocks * sizeof (float))) return rc;
      block_max = ctx->ws_block_max.as<unsigned int>();
      if (int rc = awm_add_init_block_max_d (ctx, ctx->ws_block_max.as<float>(), n_blocks)) return rc;
    }
  AWM_HIP_CHECK (awmk::launch_mix_max (ctx->stream, pcm_in_d, wm, out_d, (long long) n_frames, C, block_max, (long long) n_blocks, lim_block));
  if (ctx->snr_on)
    AWM_HIP_CHECK (awmk::launch_power_sums (ctx->stream, pcm_in_d, out_d, (long long) (n_frames * C), ctx->ws_snr.as<double>()));
  if (use_limiter)
    {
      const size_t tab_entries = awmk::limiter_tab_entries ((long long) n_frames, 0, lim_block);
      if (int rc = ctx->ws_limit_tab.reserve ((tab_entries + 1) * sizeof (float2))) return rc;
      ProfScope ps (ctx, PROF_LIMITER, double (n_frames) * C * 8.0);
      AWM_HIP_CHECK (awmk::launch_limiter (ctx->stream, out_d, (long long) n_frames, C, 0, ctx->ws_block_max.as<float>(), 0, (long long) n_blocks,
                                           lim_block, LIMITER_CEILING, ctx->ws_limit_tab.as<float2>(), tab_entries));
    }
  return 0;
}

/* ---- add as a tile loop (reference add_stream_watermark, wmadd.cc:520-589: the stream is processed frame by frame with
 * WatermarkSynth holding one frame back (wmadd.cc:220-222) and the Limiter holding up to two blocks back (limiter.cc:51-64)).
 * Here a TILE of frames is in flight instead of a frame; what is carried from tile to tile is the same state:
 *   - tile t can be mixed once the first frame of tile t + 1 is there (the 3-frame overlap-add needs the next frame's spectrum)
 *     and needs the last frame of tile t - 1,
 *   - tile t can be limited once tile t + 1 is mixed (the ramp of a block needs the maximum of the following block).
 * Three input and three mix slots rotate; block maxima are kept for the whole stream (4 bytes per second). ------------------- */
struct awm_add_stream
{
  awm_ctx  *ctx = nullptr;
  int       C = 0;
  size_t    tile = 0;                 // samples per channel in a full tile (multiple of 1024)
  bool      limiter = true;
  bool      finished = false;
  long long t = 0;                    // tiles pushed so far
  size_t    len[3] = { 0, 0, 0 };     // samples per channel in the input slots
  DevBuffer table, in[3], mix[3], block_max;
  size_t    n_blocks = 0;             // block maxima allocated (and initialised)
  // a stream that starts `zero_frames` samples into its frame / limiter block grid (add_stream_watermark's zero_frames, reference
  // wmadd.cc:501-526): whole frames of zeros are only counted (WatermarkGen::skip, Limiter::skip), the rest is a prefix of zeros
  size_t    skipped = 0;              // zero_frames rounded down to whole 1024-sample frames
  size_t    prefix = 0;               // zero_frames % 1024: zeros in front of the caller's first sample inside tile 0
  size_t    first_block = 0;          // limiter block of sample `skipped`: block_max[0]
};

static int
add_stream_grow_blocks (awm_add_stream *s, size_t need)
{
  if (need <= s->n_blocks)
    return 0;
  awm_ctx *ctx = s->ctx;
  size_t cap = std::max<size_t> (4096, s->n_blocks * 2);
  while (cap < need)
    cap *= 2;
  DevBuffer bigger;
  if (int rc = bigger.reserve (cap * sizeof (float))) return rc;
  if (s->n_blocks)
    AWM_HIP_CHECK (hipMemcpyAsync (bigger.ptr, s->block_max.ptr, s->n_blocks * sizeof (float), hipMemcpyDeviceToDevice, ctx->stream));
  if (int rc = awm_add_init_block_max_d (ctx, bigger.as<float>() + s->n_blocks, cap - s->n_blocks)) return rc;
  AWM_HIP_CHECK (hipStreamSynchronize (ctx->stream));      // the old array may still be read by a queued kernel
  s->block_max.release();
  s->block_max = bigger;
  s->n_blocks = cap;
  return 0;
}

int
awm_add_stream_create_at (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, int n_channels, size_t tile_frames1024,
                          size_t zero_frames, awm_add_stream **out)
{
  AWM_ENTER (ctx);
  if (!out || n_channels < 1 || tile_frames1024 < 128)
    {
      set_error ("awm_add_stream_create: bad argument (a tile is at least 128 frames: the limiter looks one second ahead)");
      return AWM_ERR_ARG;
    }
  FrameModTable *fm = ctx->get_frame_mod (capi_key (key), payload_hex ? payload_hex : "");
  if (!fm)
    return AWM_ERR_ARG;
  auto s = std::make_unique<awm_add_stream>();
  s->ctx = ctx;
  s->C = n_channels;
  s->tile = tile_frames1024 * Params::frame_size;
  s->limiter = !params().test_no_limiter;
  s->prefix = zero_frames % Params::frame_size;
  s->skipped = zero_frames - s->prefix;
  s->first_block = s->skipped / LIMITER_BLOCK;
  const size_t table_bytes = 2 * mark_block_frame_count() * Params::n_bands;
  auto fail = [&] (int rc) { awm_add_stream_destroy (s.release()); return rc; };
  if (int rc = s->table.reserve (table_bytes)) return fail (rc);
  // own copy of the table: the context's cache may evict its entry while the stream lives
  if (hipMemcpyAsync (s->table.ptr, fm->dev.ptr, table_bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess
      || hipStreamSynchronize (ctx->stream) != hipSuccess)
    return fail (AWM_ERR_HIP);
  // (one frame of room behind a tile: with a prefix the caller's tile lies `prefix` frames into the slot, what hangs over is
  // carried to the next slot -- or, for the last tile, stays: the last tile may be up to prefix frames longer)
  const size_t room = s->prefix ? Params::frame_size : 0;
  for (int i = 0; i < 3; i++)
    {
      if (int rc = s->in[i].reserve ((s->tile + room) * s->C * sizeof (float))) return fail (rc);
      if (int rc = s->mix[i].reserve ((s->tile + room) * s->C * sizeof (float))) return fail (rc);
    }
  if (s->prefix)
    AWM_HIP_CHECK (hipMemsetAsync (s->in[0].ptr, 0, s->prefix * s->C * sizeof (float), ctx->stream));
  *out = s.release();
  return 0;
}

int
awm_add_stream_create (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, int n_channels, size_t tile_frames1024,
                       awm_add_stream **out)
{
  return awm_add_stream_create_at (ctx, key, payload_hex, n_channels, tile_frames1024, 0, out);
}

void
awm_add_stream_destroy (awm_add_stream *s)
{
  if (!s)
    return;
  (void) hipSetDevice (s->ctx->device);
  (void) hipStreamSynchronize (s->ctx->stream);
  s->table.release();
  s->block_max.release();
  for (int i = 0; i < 3; i++)
    {
      s->in[i].release();
      s->mix[i].release();
    }
  delete s;
}

float *
awm_add_stream_input (awm_add_stream *s)
{
  return s && !s->finished ? s->in[s->t % 3].as<float>() + s->prefix * s->C : nullptr;
}

int
awm_add_stream_push (awm_add_stream *s, size_t n_frames, int last, const float *out_d[3], size_t out_frames[3])
{
  if (!s || !out_d || !out_frames)
    {
      set_error ("awm_add_stream_push: bad argument");
      return AWM_ERR_ARG;
    }
  awm_ctx *ctx = s->ctx;
  AWM_ENTER (ctx);
  if (s->finished || n_frames > s->tile || (!last && n_frames != s->tile))
    {
      set_error ("awm_add_stream_push: every tile but the last one must be full, nothing may follow the last one");
      return AWM_ERR_ARG;
    }
  const int C = s->C;
  const size_t N = Params::frame_size;
  const long long t = s->t;
  int n_out = 0;
  for (int i = 0; i < 3; i++)
    {
      out_d[i] = nullptr;
      out_frames[i] = 0;
    }
  auto slot = [&] (long long k) { return int (k % 3); };
  auto mix_tile = [&] (long long k, bool has_next) -> int {
    const size_t n = s->len[slot (k)];
    if (!n)
      return 0;
    const float *before = k > 0 ? s->in[slot (k - 1)].as<float>() + (s->tile - N) * C : nullptr;
    const float *after = has_next ? s->in[slot (k + 1)].as<float>() : nullptr;
    return add_mix_impl (ctx, s->in[slot (k)].as<float>(), s->mix[slot (k)].as<float>(), n, C, s->table.as<int8_t>(), params().water_delta,
                         (s->skipped + size_t (k) * s->tile) / N, before, after, s->limiter ? s->block_max.as<float>() : nullptr,
                         s->first_block, s->n_blocks);
  };
  auto limit_tile = [&] (long long k) -> int {
    const size_t n = s->len[slot (k)];
    if (!n)
      return 0;
    if (s->limiter)
      if (int rc = awm_add_limit_d (ctx, s->mix[slot (k)].as<float>(), n, C, s->skipped + size_t (k) * s->tile, s->block_max.as<float>(),
                                    s->first_block, s->n_blocks))
        return rc;
    // (the zeros in front of the caller's first sample are not part of the output: reference wmadd.cc:574-580)
    const size_t cut = k == 0 ? s->prefix : 0;
    if (n > cut)
      {
        out_d[n_out] = s->mix[slot (k)].as<float>() + cut * C;
        out_frames[n_out++] = n - cut;
      }
    return 0;
  };

  // what the slot holds now: the prefix (tile 0: zeros; later: the frames that hung over the previous tile) and the caller's frames
  // (every tile before this one was full, so `prefix` frames always lie in front of the caller's)
  const size_t held = s->prefix + n_frames;
  const size_t len = last ? held : s->tile;                        // (the last tile keeps what hangs over: at most prefix frames)
  s->len[slot (t)] = len;
  if (!last && s->prefix)
    AWM_HIP_CHECK (hipMemcpyAsync (s->in[slot (t + 1)].ptr, s->in[slot (t)].as<float>() + s->tile * C, s->prefix * C * sizeof (float),
                                   hipMemcpyDeviceToDevice, ctx->stream));
  if (len && len < N)                 // a next tile shorter than a frame: the halo the previous tile reads is zero extended
    AWM_HIP_CHECK (hipMemsetAsync (s->in[slot (t)].as<float>() + len * C, 0, (N - len) * C * sizeof (float), ctx->stream));
  if (s->limiter)
    if (int rc = add_stream_grow_blocks (s, (s->skipped + size_t (t) * s->tile + len) / LIMITER_BLOCK + 2 - s->first_block)) return rc;
  if (t >= 1)
    if (int rc = mix_tile (t - 1, len > 0)) return rc;
  if (t >= 2)
    if (int rc = limit_tile (t - 2)) return rc;
  if (last)
    {
      if (int rc = mix_tile (t, false)) return rc;
      if (t >= 1)
        if (int rc = limit_tile (t - 1)) return rc;
      if (int rc = limit_tile (t)) return rc;
      s->finished = true;
    }
  s->t++;
  return n_out;
}

int
awm_add_d (awm_ctx *ctx, const float *pcm_in_d, float *out_d, size_t n_frames, int n_channels,
           const int8_t *frame_mod, double water_delta, int use_limiter)
{
  AWM_ENTER (ctx);
  if (!pcm_in_d || !out_d || !frame_mod || n_channels < 1)
    {
      set_error ("awm_add_d: bad argument");
      return AWM_ERR_ARG;
    }
  const size_t table_bytes = 2 * mark_block_frame_count() * Params::n_bands;
  if (int rc = ctx->ws_misc.reserve (table_bytes)) return rc;
  AWM_HIP_CHECK (hipMemcpyAsync (ctx->ws_misc.ptr, frame_mod, table_bytes, hipMemcpyHostToDevice, ctx->stream));
  return add_full (ctx, pcm_in_d, out_d, n_frames, n_channels, ctx->ws_misc.as<int8_t>(), water_delta, use_limiter);
}

static bool
pcm_format_ok (int bit_depth, int encoding)
{
  if (encoding == 2)
    return bit_depth == 32 || bit_depth == 64;
  return (encoding == 0 || encoding == 1) && (bit_depth == 8 || bit_depth == 16 || bit_depth == 24 || bit_depth == 32);
}

int
awm_pcm_decode_d (awm_ctx *ctx, const void *bytes_d, size_t n_values, int bit_depth, int encoding, int big_endian, float *out_d)
{
  AWM_ENTER (ctx);
  if (!pcm_format_ok (bit_depth, encoding))
    {
      set_error ("awm_pcm_decode_d: unsupported sample format");
      return AWM_ERR_ARG;
    }
  const awmk::PcmFormatDev f { bit_depth / 8, encoding, big_endian != 0, 0 };
  AWM_HIP_CHECK (awmk::launch_pcm_decode (ctx->stream, static_cast<const unsigned char *> (bytes_d), out_d, (long long) n_values, f));
  return 0;
}

int
awm_pcm_encode_d (awm_ctx *ctx, const float *in_d, size_t n_values, int bit_depth, int encoding, int big_endian, int direct16, void *bytes_d)
{
  AWM_ENTER (ctx);
  if (!pcm_format_ok (bit_depth, encoding))
    {
      set_error ("awm_pcm_encode_d: unsupported sample format");
      return AWM_ERR_ARG;
    }
  const bool d16 = direct16 && bit_depth == 16 && encoding == 0 && !big_endian;
  const awmk::PcmFormatDev f { bit_depth / 8, encoding, big_endian != 0, d16 };
  AWM_HIP_CHECK (awmk::launch_pcm_encode (ctx->stream, in_d, static_cast<unsigned char *> (bytes_d), (long long) n_values, f));
  return 0;
}

static int g_k4s_ablate = 0;
extern "C" void awm_debug_set_k4s_ablate (int f) { g_k4s_ablate = f; }
int
awm_debug_sync_db_sliding_d (awm_ctx *ctx, const float *pcm_d, size_t n_frames, int n_channels, const long long *base_d, size_t n_streams,
                             int count, int ld, float *out_d)
{
  AWM_ENTER (ctx);
  if (!pcm_d || !base_d || !out_d || count < 1 || count > 65 || (ld < count && ld != 64) || (n_channels != 1 && n_channels != 2))
    {
      set_error ("awm_debug_sync_db_sliding_d: bad argument");
      return AWM_ERR_ARG;
    }
  awmk::SyncDbArgs da {};
  da.pcm = pcm_d;
  da.n_frames = (long long) n_frames;
  da.n_channels = n_channels;
  da.stream_base = base_d;
  da.count0 = count;
  da.n_streams = (long long) n_streams;
  da.hop = Params::sync_search_fine;
  da.out = out_d;
  da.out_stream_stride = (long long) Params::n_bands * ld;
  da.ld = ld;
  da.first = 0;
  da.last = (long long) (n_frames * n_channels);
  da.tile_frames = ld;
  da.xcd_interleave = g_k4s_ablate;
  if (ld == 64)
    {
      // (measurement, tools/gpu_k4s_alone.py: the refinement's gathered layout with a synthetic table -- bands 0..59 are the rows --, rows
      // of 64 offsets at out_d[i][60][64] and the 65th values behind them at out_d + n_streams * 60 * 64)
      std::vector<unsigned char> pos (Params::n_bands, 255);
      for (int b = 0; b < 60; b++)
        pos[b] = (unsigned char) b;
      if (int rc = ctx->ws_misc.reserve (Params::n_bands + sizeof (int))) return rc;
      const int zero = 0;
      AWM_HIP_CHECK (hipMemcpyAsync (ctx->ws_misc.ptr, &zero, sizeof (int), hipMemcpyHostToDevice, ctx->stream));
      AWM_HIP_CHECK (hipMemcpyAsync (ctx->ws_misc.as<char>() + sizeof (int), pos.data(), pos.size(), hipMemcpyHostToDevice, ctx->stream));
      da.row_perm = ctx->ws_misc.as<int>();
      da.band_pos = ctx->ws_misc.as<unsigned char>() + sizeof (int);
      da.rows_per_plane = 1;
      da.out_stream_stride = 60 * 64;
      if (awmk::sliding_rows_have_tail (n_channels))
        {
          da.tail = out_d + n_streams * 60 * 64;
          da.tail_stream_stride = 60;
        }
      else if (count > 64)
        {
          set_error ("awm_debug_sync_db_sliding_d: rows of 64 offsets take 65 only with forms 4 / 5");
          return AWM_ERR_ARG;
        }
    }
  AWM_HIP_CHECK (awmk::launch_sync_db_sliding (ctx->stream, ctx->tabs, da));
  return 0;
}

int
awm_sync_fft_d (awm_ctx *ctx, const float *pcm_d, size_t n_frames, int n_channels, size_t index, size_t frame_count,
                const char *want_frames, size_t first, size_t last, float *db_out_d, char *have_out_d)
{
  AWM_ENTER (ctx);
  if (n_frames < index + frame_count * Params::frame_size)
    {
      set_error ("awm_sync_fft_d: read past end");       // the reference returns empty vectors here
      return AWM_ERR_ARG;
    }
  if (!frame_count)
    return 0;
  // band-major scratch, then transposed into the reference's [frame][81] layout by a strided copy
  const long long ld = (long long) ((frame_count + 63) & ~size_t (63));
  if (int rc = ctx->ws_db.reserve (size_t (ld) * Params::n_bands * sizeof (float))) return rc;
  if (int rc = ctx->ws_have.reserve (ld)) return rc;
  hipStream_t st = ctx->stream;
  std::vector<long long> bases;
  std::vector<int> counts;
  awmk::SyncDbArgs da {};
  da.pcm = pcm_d;
  da.n_frames = (long long) n_frames;
  da.n_channels = n_channels;
  da.per_channel = 0;
  da.hop = Params::frame_size;
  da.out = ctx->ws_db.as<float>();
  da.ld = ld;
  da.have = ctx->ws_have.as<char>();
  da.first = (long long) first;
  da.last = (long long) last;
  da.tile_frames = 32;
  AWM_HIP_CHECK (hipMemsetAsync (ctx->ws_db.ptr, 0, size_t (ld) * Params::n_bands * sizeof (float), st));
  AWM_HIP_CHECK (hipMemsetAsync (ctx->ws_have.ptr, 0, ld, st));
  if (!want_frames)
    {
      da.base0 = (long long) index;
      da.base_stride = 0;
      da.count0 = int (frame_count);
      da.n_streams = 1;
      da.out_stream_stride = 0;
      da.have_stream_stride = 0;
      AWM_HIP_CHECK (awmk::launch_sync_db (st, ctx->tabs, da));
    }
  else
    {
      // one single-frame stream per wanted frame, written at its own column
      for (size_t f = 0; f < frame_count; f++)
        if (want_frames[f])
          bases.push_back ((long long) (index + f * Params::frame_size));
      if (!bases.empty())
        {
          // process wanted frames one launch per contiguous run to keep column addressing simple
          size_t f = 0;
          while (f < frame_count)
            {
              if (!want_frames[f])
                {
                  f++;
                  continue;
                }
              size_t g = f;
              while (g < frame_count && want_frames[g])
                g++;
              awmk::SyncDbArgs run = da;
              run.base0 = (long long) (index + f * Params::frame_size);
              run.base_stride = 0;
              run.count0 = int (g - f);
              run.n_streams = 1;
              run.out = da.out + f;
              run.have = da.have + f;
              run.out_stream_stride = 0;
              run.have_stream_stride = 0;
              AWM_HIP_CHECK (awmk::launch_sync_db (st, ctx->tabs, run));
              f = g;
            }
        }
    }
  // transpose [81][ld] -> [frame_count][81]
  AWM_HIP_CHECK (hipMemcpy2DAsync (have_out_d, 1, ctx->ws_have.ptr, 1, 1, frame_count, hipMemcpyDeviceToDevice, st));
  for (int b = 0; b < Params::n_bands; b++)
    AWM_HIP_CHECK (hipMemcpy2DAsync (db_out_d + b, Params::n_bands * sizeof (float), ctx->ws_db.as<float>() + (long long) b * ld,
                                     sizeof (float), sizeof (float), frame_count, hipMemcpyDeviceToDevice, st));
  return 0;
}

int
awm_sync_search_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels, int clip_mode,
                   size_t max_out, uint64_t *index, double *quality, int *block_type)
{
  AWM_ENTER (ctx);
  SyncFinder sf (ctx);
  std::vector<SyncFinder::Score> scores;
  if (int rc = sf.search (capi_key (key), make_wav (pcm_d, n_frames, n_channels),
                          clip_mode ? SyncFinder::Mode::CLIP : SyncFinder::Mode::BLOCK, scores))
    return rc;
  for (size_t i = 0; i < scores.size() && i < max_out; i++)
    {
      index[i] = scores[i].index;
      quality[i] = scores[i].quality;
      block_type[i] = int (scores[i].block_type);
    }
  return int (scores.size());
}

long
awm_search_approx_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels, int clip_mode,
                     size_t max_out, uint64_t *index, double *raw_quality, double *local_mean)
{
  AWM_ENTER (ctx);
  KeyTables *kt = ctx->get_key_tables (capi_key (key));
  if (!kt)
    return AWM_ERR_HIP;
  SyncFinder sf (ctx);
  std::vector<SyncFinder::SearchScore> scores;
  const DeviceWav wav = make_wav (pcm_d, n_frames, n_channels);
  if (int rc = sf.prepare (wav, clip_mode ? SyncFinder::Mode::CLIP : SyncFinder::Mode::BLOCK))
    return rc;
  if (int rc = sf.search_approx (kt, wav, clip_mode ? SyncFinder::Mode::CLIP : SyncFinder::Mode::BLOCK, scores))
    return rc;
  for (size_t i = 0; i < scores.size() && i < max_out; i++)
    {
      index[i] = scores[i].index;
      raw_quality[i] = scores[i].raw_quality;
      local_mean[i] = scores[i].local_mean;
    }
  return long (scores.size());
}

int
awm_block_soft_bits_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels,
                       const uint64_t *index, size_t n_blocks, float *out, int *ok)
{
  AWM_ENTER (ctx);
  KeyTables *kt = ctx->get_key_tables (capi_key (key));
  if (!kt)
    return AWM_ERR_HIP;
  std::vector<size_t> idx (index, index + n_blocks);
  std::vector<std::vector<float>> raw;
  std::vector<char> okv;
  if (int rc = block_soft_bits (ctx, kt, make_wav (pcm_d, n_frames, n_channels), idx, raw, okv))
    return rc;
  const size_t n_bits = mark_data_frame_count() / params().frames_per_bit;
  for (size_t i = 0; i < n_blocks; i++)
    {
      ok[i] = okv[i];
      if (okv[i])
        std::copy (raw[i].begin(), raw[i].end(), out + i * n_bits);
      else
        std::fill (out + i * n_bits, out + (i + 1) * n_bits, 0.f);
    }
  return 0;
}

int
awm_viterbi_decode (awm_ctx *ctx, int block_type, const float *soft, size_t coded_len, size_t n, int *bits_out, float *error_out)
{
  AWM_ENTER (ctx);
  if (block_type < 0 || block_type > 2)
    return AWM_ERR_ARG;
  std::vector<std::vector<float>> in (n);
  for (size_t i = 0; i < n; i++)
    in[i].assign (soft + i * coded_len, soft + (i + 1) * coded_len);
  std::vector<std::vector<int>> bits;
  std::vector<float> errors;
  if (int rc = viterbi_decode (ctx, ConvBlockType (block_type), in, bits, errors))
    return rc;
  for (size_t i = 0; i < n; i++)
    {
      std::copy (bits[i].begin(), bits[i].end(), bits_out + i * bits[i].size());
      error_out[i] = errors[i];
    }
  return 0;
}

int
awm_add_watermark_d (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, const float *pcm_in_d, float *out_d,
                     size_t n_frames, int n_channels, int sample_rate)
{
  AWM_ENTER (ctx);
  FrameModTable *fm = ctx->get_frame_mod (capi_key (key), payload_hex ? payload_hex : "");
  if (!fm)
    return AWM_ERR_ARG;
  if (sample_rate != Params::mark_sample_rate)
    return add_full_rate (ctx, pcm_in_d, out_d, n_frames, n_channels, fm->dev.as<int8_t>(), params().water_delta, !params().test_no_limiter, sample_rate);
  return add_full (ctx, pcm_in_d, out_d, n_frames, n_channels, fm->dev.as<int8_t>(), params().water_delta, !params().test_no_limiter);
}

/* add_watermark followed by get_watermark of its output ("watermark, then verify") as ONE call: the library owns the order of the two
 * halves, so `get` may start chunk c as soon as the limiter has passed the chunk's last sample -- its first dB kernel (bound by FP32
 * issue) runs beside the limiter passes of the later ranges (bound by HBM) -- instead of behind the whole `add`, which is all that two
 * separate calls on the caller's stream can promise.  PCM and pattern list are those of the two calls (tests). */
int
awm_add_get_watermark_d (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, const float *pcm_in_d, float *out_d,
                         size_t n_frames, int n_channels, int sample_rate, size_t max_out, awm_pattern *out)
{
  AWM_ENTER (ctx);
  if (max_out && !out)
    {
      set_error ("awm_add_get_watermark_d: bad argument");
      return AWM_ERR_ARG;
    }
  FrameModTable *fm = ctx->get_frame_mod (capi_key (key), payload_hex ? payload_hex : "");
  if (!fm)
    return AWM_ERR_ARG;
  int rc;
  const bool hand_over = sample_rate == Params::mark_sample_rate && pcm_in_d != out_d && !g_add_slab_mb && n_frames > 0;
  if (sample_rate != Params::mark_sample_rate)
    rc = add_full_rate (ctx, pcm_in_d, out_d, n_frames, n_channels, fm->dev.as<int8_t>(), params().water_delta, !params().test_no_limiter, sample_rate);
  else
    rc = add_full (ctx, pcm_in_d, out_d, n_frames, n_channels, fm->dev.as<int8_t>(), params().water_delta, !params().test_no_limiter, nullptr,
                   hand_over ? &ctx->ready : nullptr);
  if (rc)
    {
      ctx->ready.disarm();
      return rc;
    }
  ResultSet rs;
  if (sample_rate != Params::mark_sample_rate)
    {
      // the reference's loader resamples to the watermark rate before anything else (wavchunkloader.cc:70-71)
      const size_t n44 = awm_resample_frames (ctx, n_frames, sample_rate, Params::mark_sample_rate);
      if (n_frames && !n44)
        return AWM_ERR_ARG;
      if (int r = ctx->ws_rate_c.reserve (std::max<size_t> (1, n44 * n_channels * sizeof (float)))) return r;
      if (int r = awm_resample_d (ctx, out_d, n_frames, n_channels, sample_rate, Params::mark_sample_rate, ctx->ws_rate_c.as<float>(), n44)) return r;
      rc = get_watermark_device (ctx, { capi_key (key) }, make_wav (ctx->ws_rate_c.as<float>(), n44, n_channels), rs);
    }
  else
    rc = get_watermark_device (ctx, { capi_key (key) }, make_wav (out_d, n_frames, n_channels), rs);
  ctx->ready.disarm();
  if (rc)
    return rc;
  for (size_t i = 0; i < rs.patterns.size() && i < max_out; i++)
    fill_pattern (rs.patterns[i], out[i]);
  return int (rs.patterns.size());
}

/* add_watermark for many independent inputs with one key and payload (BASELINE config 5: a batch of short clips).  A 30 s clip
 * is five launches of a few microseconds of work each: alone on a stream they run one after the other with the GPU mostly idle
 * (65 us per clip), so the clips are dealt to eight lanes.  Ordered after the work queued on the context's stream; the context's
 * stream is ordered after the batch. */
extern "C++" {
namespace {
std::vector<Key>
key_list_from (const uint8_t *keys, int n_keys)
{
  std::vector<Key> list;
  for (int k = 0; k < n_keys; k++)
    list.push_back (capi_key (keys + size_t (k) * Key::SIZE));
  return list;
}
// patterns of a multi-key `get` -> the C arrays; key_of_pattern[j] = position of pattern j's key in the list
int
fill_patterns_keys (const ResultSet& rs, const std::vector<Key>& list, size_t max_out, awm_pattern *out, int *key_of_pattern)
{
  for (size_t i = 0; i < rs.patterns.size() && i < max_out; i++)
    {
      fill_pattern (rs.patterns[i], out[i]);
      if (key_of_pattern)
        {
          key_of_pattern[i] = -1;
          for (size_t k = 0; k < list.size(); k++)
            if (rs.patterns[i].key == list[k])
              {
                key_of_pattern[i] = int (k);
                break;
              }
        }
    }
  return int (rs.patterns.size());
}
}
} // extern "C++"

/* Batches of clips, ONE launch per stage for many clips (fill of the block maxima, K2, K3a, K3b) instead of four launches per clip.
 * A 30 s clip is 1292 frames: alone it is 323 spans of four frames (+ two halo frames each: 1.5 x the transforms) and four launches of
 * 6 - 46 us -- 1024 clips on eight lanes were 26 ms of launches for ~8 ms of memory traffic.  Here the spans are sized for the whole batch
 * (frames_per_span of the batch's frames: ~20 frames per span, 1.1 x the transforms), blockIdx.y is the clip, its arguments come from an
 * array on the device.  Output bit-identical to the per-clip launches (a span's result does not depend on the span length).
 * (measurement knob) awm_debug_set_add_batched (0): the per-clip launches on eight lanes; (1) / (2, default): see add_batch_keys_tables_first */
static int g_add_batched = 2;              // 2: with a key per clip, the tables of all keys first (add_batch_keys_tables_first) | 1: a group's tables while the previous group is watermarked
extern "C" void awm_debug_set_add_batched (int on) { g_add_batched = on; }

static bool
add_clips_batchable (awm_ctx *ctx, size_t n_clips, const float *const *pcm_in_d, float *const *out_d, int n_channels)
{
  if (!g_add_batched || n_channels != 2 || ctx->snr_on || n_clips < 2)
    return false;
  for (size_t i = 0; i < n_clips; i++)
    if ((reinterpret_cast<uintptr_t> (pcm_in_d[i]) & 15) || (reinterpret_cast<uintptr_t> (out_d[i]) & 15))
      return false;
  return true;
}

/* Batched arguments of a whole call: staged once (page-locked), copied once; groups then launch on slices of the device arrays. */
struct AddBatchArgs
{
  awmk::AddMixArgs  *d_mix = nullptr;
  awmk::LimiterClip *d_lim = nullptr;
  float             *block_max = nullptr;
  size_t             nb_max = 0;
  std::vector<long long> spans, frames;
  int                L = 4;
};

/* tables: the device table of every clip, or ONE for all (known now; the kernels read them when a group runs) */
static int
add_batch_stage (awm_ctx *ctx, hipStream_t st, size_t n_clips, const float *const *pcm_in_d, float *const *out_d, const size_t *n_frames,
                 const std::vector<const int8_t *>& tables, int use_limiter, AddBatchArgs& b)
{
  const int C = 2;
  long long total_frames1024 = 0;
  size_t max_frames = 0;
  for (size_t i = 0; i < n_clips; i++)
    {
      total_frames1024 += (long long) (n_frames[i] + 1023) / 1024;
      max_frames = std::max (max_frames, n_frames[i]);
    }
  b.L = frames_per_span (ctx, total_frames1024);
  b.nb_max = max_frames / LIMITER_BLOCK + 2;
  const size_t tab_max = awmk::limiter_tab_entries ((long long) max_frames, 0, LIMITER_BLOCK) + 1;
  const size_t arg_bytes = n_clips * (sizeof (awmk::AddMixArgs) + sizeof (awmk::LimiterClip));
  if (int rc = ctx->ws_add_batch.reserve (arg_bytes)) return rc;
  if (int rc = ctx->ws_block_max.reserve (n_clips * b.nb_max * sizeof (float))) return rc;
  if (int rc = ctx->ws_limit_tab.reserve (n_clips * tab_max * sizeof (float2))) return rc;
  if (ctx->ev_add_batch)
    AWM_HIP_CHECK (hipEventSynchronize (ctx->ev_add_batch));            // (an earlier batch's staging has been copied)
  else
    AWM_HIP_CHECK (hipEventCreateWithFlags (&ctx->ev_add_batch, hipEventDisableTiming));
  if (int rc = ctx->pin_add_batch.reserve (arg_bytes)) return rc;
  auto *h_mix = ctx->pin_add_batch.as<awmk::AddMixArgs>();
  auto *h_lim = reinterpret_cast<awmk::LimiterClip *> (h_mix + n_clips);
  b.d_mix = ctx->ws_add_batch.as<awmk::AddMixArgs>();
  b.d_lim = reinterpret_cast<awmk::LimiterClip *> (b.d_mix + n_clips);
  b.block_max = ctx->ws_block_max.as<float>();
  float2 *tabs = ctx->ws_limit_tab.as<float2>();
  b.spans.assign (n_clips, 0);
  b.frames.assign (n_clips, 0);
  for (size_t i = 0; i < n_clips; i++)
    {
      awmk::AddMixArgs a {};
      a.pcm_in = pcm_in_d[i];
      a.out = out_d[i];
      a.n_frames = (long long) n_frames[i];
      a.n_channels = C;
      a.frame_mod = tables.size() == 1 ? tables[0] : tables[i];
      a.neg_delta_up = float (-params().water_delta * 1);
      a.neg_delta_down = float (-params().water_delta * -1);
      a.block_max = use_limiter ? reinterpret_cast<unsigned int *> (b.block_max + i * b.nb_max) : nullptr;
      a.n_blocks = (long long) (n_frames[i] / LIMITER_BLOCK + 2);
      a.limiter_block = LIMITER_BLOCK;
      a.block_frames = int (mark_block_frame_count());
      a.frames_pad_start = int (Params::frames_pad_start);
      a.frames_per_span = b.L;
      h_mix[i] = a;
      h_lim[i] = { out_d[i], (long long) n_frames[i], b.block_max + i * b.nb_max, a.n_blocks, tabs + i * tab_max,
                   (long long) awmk::limiter_tab_entries ((long long) n_frames[i], 0, LIMITER_BLOCK) };
      const long long F = (long long) (n_frames[i] + 1023) / 1024;
      b.spans[i] = (F + b.L - 1) / b.L;
      b.frames[i] = (long long) n_frames[i];
    }
  AWM_HIP_CHECK (hipMemcpyAsync (b.d_mix, h_mix, arg_bytes, hipMemcpyHostToDevice, st));
  AWM_HIP_CHECK (hipEventRecord (ctx->ev_add_batch, st));
  return 0;
}

/* clips [i0, i0 + n) of a staged batch on stream st */
static int
add_batch_run (awm_ctx *ctx, hipStream_t st, const AddBatchArgs& b, size_t i0, size_t n, int use_limiter)
{
  if (!n)
    return 0;
  long long max_spans = 0, max_frames = 0;
  double values = 0;
  for (size_t i = i0; i < i0 + n; i++)
    {
      max_spans = std::max (max_spans, b.spans[i]);
      max_frames = std::max (max_frames, b.frames[i]);
      values += double (b.frames[i]) * 2;
    }
  if (use_limiter)
    {
      unsigned int bits;
      std::memcpy (&bits, &LIMITER_CEILING, sizeof (bits));
      AWM_HIP_CHECK (awmk::launch_fill_u32 (st, reinterpret_cast<unsigned int *> (b.block_max + i0 * b.nb_max), bits, n * b.nb_max));
    }
  {
    ProfScope ps (ctx, PROF_ADD_MIX, values * 8.0, st);                            // read + write every sample once
    AWM_HIP_CHECK (awmk::launch_add_mix_batch (st, ctx->tabs, b.d_mix + i0, int (n), max_spans, int (mark_block_frame_count()), int (Params::frames_pad_start)));
  }
  if (use_limiter)
    {
      ProfScope ps (ctx, PROF_LIMITER, values * 8.0, st);
      AWM_HIP_CHECK (awmk::launch_limiter_batch (st, b.d_lim + i0, int (n), max_frames, 2, LIMITER_BLOCK, LIMITER_CEILING));
    }
  return 0;
}

int
awm_add_watermark_batch_d (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, size_t n_clips, const float *const *pcm_in_d,
                           float *const *out_d, const size_t *n_frames, int n_channels)
{
  AWM_ENTER (ctx);
  if (n_clips && (!pcm_in_d || !out_d || !n_frames || n_channels < 1))
    {
      set_error ("awm_add_watermark_batch_d: bad argument");
      return AWM_ERR_ARG;
    }
  FrameModTable *fm = ctx->get_frame_mod (capi_key (key), payload_hex ? payload_hex : "");
  if (!fm)
    return AWM_ERR_ARG;
  if (add_clips_batchable (ctx, n_clips, pcm_in_d, out_d, n_channels))
    {
      const int use_limiter = !params().test_no_limiter;
      AddBatchArgs batch;
      if (int rc = add_batch_stage (ctx, ctx->stream, n_clips, pcm_in_d, out_d, n_frames, { fm->dev.as<int8_t>() }, use_limiter, batch))
        return rc;
      constexpr size_t PER_LAUNCH = 4096;                    // (blockIdx.y <= 65535; a launch of 4096 clips fills the device many times over)
      for (size_t i0 = 0; i0 < n_clips; i0 += PER_LAUNCH)
        if (int rc = add_batch_run (ctx, ctx->stream, batch, i0, std::min (PER_LAUNCH, n_clips - i0), use_limiter))
          return rc;
      return 0;
    }
  constexpr int ADD_LANES = 8;
  const int n_lanes = int (std::min<size_t> (ADD_LANES, n_clips));
  std::vector<WorkLane *> lanes;
  for (int i = 0; i < n_lanes; i++)
    {
      WorkLane *l = ctx->lane (i);
      if (!l)
        {
          set_error ("cannot create a work lane (stream)");
          return AWM_ERR_HIP;
        }
      if (!l->ev_sync)
        AWM_HIP_CHECK (hipEventCreateWithFlags (&l->ev_sync, hipEventDisableTiming));
      lanes.push_back (l);
    }
  if (n_lanes > 1)
    {
      AWM_HIP_CHECK (hipEventRecord (ctx->ev_sync, ctx->stream));
      for (int i = 1; i < n_lanes; i++)
        AWM_HIP_CHECK (hipStreamWaitEvent (lanes[i]->stream, ctx->ev_sync, 0));
    }
  int rc = 0;
  for (size_t i = 0; i < n_clips && !rc; i++)
    rc = add_full (ctx, pcm_in_d[i], out_d[i], n_frames[i], n_channels, fm->dev.as<int8_t>(), params().water_delta, !params().test_no_limiter,
                   lanes[i % n_lanes]);
  for (int i = 1; i < n_lanes; i++)
    {
      AWM_HIP_CHECK (hipEventRecord (lanes[i]->ev_sync, lanes[i]->stream));
      AWM_HIP_CHECK (hipStreamWaitEvent (ctx->stream, lanes[i]->ev_sync, 0));
    }
  return rc;
}

/* the same with ONE KEY PER CLIP (BASELINE configs[4]: `--test-key k` per clip).  The frame_mod tables (361 KB per key; 2226 up / down
 * draws and three shuffles per key on the host: ~3 ms of one core) are built on host threads group by group while the device works on
 * the previous group, and live in one batch buffer instead of the context's per-key cache. */
/* (measurement knob) 1 (default): the frame_mod tables of a batch with one key per clip are built on the device (K16, hip/keytab.hip) |
 * 0: on host threads */
extern "C" void awm_debug_set_key_tables_on_device (int on) { g_key_tables_on_device = on; }

/* awm_add_watermark_batch_keys_d with the tables built by K16: groups of GROUP keys (one workgroup = one compute unit per key), two
 * table areas in turn -- the tables of group g + 1 are built (on a lane of their own) while the clips of group g are watermarked on the
 * add lanes; what the host contributes per key is the AES key schedule (176 bytes). */
/* Batched clips with a key per clip, everything on the context's stream: FIRST the tables of (up to 4096) keys, 256 per launch of K16, THEN the
 * clips in one launch per stage.  K16 holds 116 KB of LDS on every compute unit it runs on: beside it K2 runs one workgroup per unit
 * instead of four, and the first group's tables are waited for in any case -- building the next group's tables WHILE a group is
 * watermarked (the loop below, awm_debug_set_add_batched (1)) cost more than it hid once the clips of a group took 3 ms instead of 8. */
static int
add_batch_keys_tables_first (awm_ctx *ctx, const uint8_t *keys, const std::vector<int>& bits, size_t n_clips, const float *const *pcm_in_d,
                             float *const *out_d, const size_t *n_frames)
{
  constexpr size_t GROUP = 256, SUPER = 4096;
  const size_t table_bytes = awmk::key_table_bytes();
  const int use_limiter = !params().test_no_limiter;
  hipStream_t st = ctx->stream;
  // one page-locked block, one upload: [S-box 256][conv code of the payload, A then B: 2 x 858][key schedules: n x 176]
  const size_t aux_bytes = 256 + 2 * 858 + n_clips * 176;
  if (int rc = ctx->pin_keytab.reserve (aux_bytes)) return rc;
  if (int rc = ctx->ws_keytab_aux.reserve (aux_bytes)) return rc;
  if (int rc = ctx->ws_keytab.reserve (std::min (n_clips, SUPER) * table_bytes)) return rc;
  if (int rc = ctx->ws_keytab_scratch.reserve (GROUP * awmk::key_table_scratch_bytes())) return rc;
  unsigned char *aux = ctx->pin_keytab.as<unsigned char>();
  std::memcpy (aux, Aes128::sbox(), 256);
  for (int ab = 0; ab < 2; ab++)
    {
      const std::vector<int> coded = code_encode (ab ? ConvBlockType::b : ConvBlockType::a, bits);
      if (coded.size() != 858)
        {
          set_error ("conv code of unexpected size");
          return AWM_ERR_GENERIC;
        }
      for (size_t i = 0; i < coded.size(); i++)
        aux[256 + 858 * ab + i] = (unsigned char) (coded[i] & 1);
    }
  for (size_t i = 0; i < n_clips; i++)
    {
      Aes128 aes;
      aes.set_key (keys + 16 * i);
      std::memcpy (aux + 256 + 2 * 858 + 176 * i, aes.round_keys(), 176);
    }
  unsigned char *d_aux = ctx->ws_keytab_aux.as<unsigned char>();
  AWM_HIP_CHECK (hipMemcpyAsync (d_aux, aux, aux_bytes, hipMemcpyHostToDevice, st));
  hipEvent_t ev_aux = nullptr;
  struct Event { hipEvent_t& e; ~Event() { if (e) (void) hipEventDestroy (e); } } event { ev_aux };
  AWM_HIP_CHECK (hipEventCreateWithFlags (&ev_aux, hipEventDisableTiming));
  AWM_HIP_CHECK (hipEventRecord (ev_aux, st));
  std::vector<const int8_t *> tables (n_clips);
  for (size_t i = 0; i < n_clips; i++)
    tables[i] = ctx->ws_keytab.as<int8_t>() + (i % SUPER) * table_bytes;
  AddBatchArgs batch;
  if (int rc = add_batch_stage (ctx, st, n_clips, pcm_in_d, out_d, n_frames, tables, use_limiter, batch))
    return rc;
  for (size_t s0 = 0; s0 < n_clips; s0 += SUPER)
    {
      const size_t sn = std::min (SUPER, n_clips - s0);
      for (size_t g0 = s0; g0 < s0 + sn; g0 += GROUP)
        {
          const size_t gn = std::min (GROUP, s0 + sn - g0);
          awmk::KeyTableArgs ka {};
          ka.sbox = d_aux;
          ka.coded = d_aux + 256;
          ka.round_keys = d_aux + 256 + 2 * 858 + 176 * g0;
          ka.scratch = ctx->ws_keytab_scratch.as<unsigned char>();
          ka.scratch_slots = int (GROUP);
          ka.tables = reinterpret_cast<signed char *> (ctx->ws_keytab.as<int8_t>() + (g0 - s0) * table_bytes);
          ka.n_keys = (long long) gn;
          ProfScope ps (ctx, PROF_KEYTAB, double (gn) * table_bytes, st);              // the table out, once (360 KB per key)
          AWM_HIP_CHECK (awmk::launch_frame_mod_tables (st, ka));
        }
      if (int rc = add_batch_run (ctx, st, batch, s0, sn, use_limiter))
        return rc;
    }
  // (the staging block is the context's: its upload has to be through before the next call refills it)
  AWM_HIP_CHECK (hipEventSynchronize (ev_aux));
  return 0;
}

static int
add_batch_keys_device_tables (awm_ctx *ctx, const uint8_t *keys, const std::vector<int>& bits, size_t n_clips, const float *const *pcm_in_d,
                              float *const *out_d, const size_t *n_frames, int n_channels)
{
  constexpr size_t GROUP = 256;
  constexpr int ADD_LANES = 8;
  const size_t table_bytes = awmk::key_table_bytes();
  // a group of clips in one launch per stage on the context's stream (add_clips_batched above), or clip by clip on eight lanes
  const bool batched = add_clips_batchable (ctx, n_clips, pcm_in_d, out_d, n_channels);
  if (batched && g_add_batched == 2)
    return add_batch_keys_tables_first (ctx, keys, bits, n_clips, pcm_in_d, out_d, n_frames);
  const int use_limiter = !params().test_no_limiter;
  const int n_lanes = batched ? 1 : int (std::min<size_t> (ADD_LANES, n_clips));
  std::vector<WorkLane *> lanes;
  for (int i = 0; i <= n_lanes; i++)                       // lanes 0 .. n_lanes - 1 watermark, lane n_lanes builds tables
    {
      WorkLane *l = ctx->lane (i);
      if (!l)
        {
          set_error ("cannot create a work lane (stream)");
          return AWM_ERR_HIP;
        }
      if (!l->ev_sync)
        AWM_HIP_CHECK (hipEventCreateWithFlags (&l->ev_sync, hipEventDisableTiming));
      lanes.push_back (l);
    }
  hipStream_t table_stream = lanes[n_lanes]->stream;
  // one page-locked block, one upload: [S-box 256][conv code of the payload, A then B: 2 x 858][key schedules: n x 176]
  const size_t aux_bytes = 256 + 2 * 858 + n_clips * 176;
  if (int rc = ctx->pin_keytab.reserve (aux_bytes)) return rc;
  if (int rc = ctx->ws_keytab_aux.reserve (aux_bytes)) return rc;
  if (int rc = ctx->ws_keytab.reserve (2 * GROUP * table_bytes)) return rc;
  if (int rc = ctx->ws_keytab_scratch.reserve (GROUP * awmk::key_table_scratch_bytes())) return rc;
  unsigned char *aux = ctx->pin_keytab.as<unsigned char>();
  std::memcpy (aux, Aes128::sbox(), 256);
  for (int ab = 0; ab < 2; ab++)
    {
      const std::vector<int> coded = code_encode (ab ? ConvBlockType::b : ConvBlockType::a, bits);
      if (coded.size() != 858)
        {
          set_error ("conv code of unexpected size");
          return AWM_ERR_GENERIC;
        }
      for (size_t i = 0; i < coded.size(); i++)
        aux[256 + 858 * ab + i] = (unsigned char) (coded[i] & 1);
    }
  for (size_t i = 0; i < n_clips; i++)
    {
      Aes128 aes;
      aes.set_key (keys + 16 * i);
      std::memcpy (aux + 256 + 2 * 858 + 176 * i, aes.round_keys(), 176);
    }
  unsigned char *d_aux = ctx->ws_keytab_aux.as<unsigned char>();
  AWM_HIP_CHECK (hipMemcpyAsync (d_aux, aux, aux_bytes, hipMemcpyHostToDevice, ctx->stream));
  hipEvent_t ev_aux = nullptr, ev_tab[2] = { nullptr, nullptr };
  std::vector<hipEvent_t> lane_done (2 * size_t (n_lanes), nullptr);
  struct Events
  {
    hipEvent_t& a; hipEvent_t (&t)[2]; std::vector<hipEvent_t>& d;
    ~Events() { if (a) (void) hipEventDestroy (a); for (hipEvent_t e : t) if (e) (void) hipEventDestroy (e); for (hipEvent_t e : d) if (e) (void) hipEventDestroy (e); }
  } events { ev_aux, ev_tab, lane_done };
  AWM_HIP_CHECK (hipEventCreateWithFlags (&ev_aux, hipEventDisableTiming));
  for (auto& e : ev_tab)
    AWM_HIP_CHECK (hipEventCreateWithFlags (&e, hipEventDisableTiming));
  for (auto& e : lane_done)
    AWM_HIP_CHECK (hipEventCreateWithFlags (&e, hipEventDisableTiming));
  AddBatchArgs batch;
  if (batched)
    {
      // clip i of group g finds its table in area g mod 2, slot i of the group
      std::vector<const int8_t *> tables (n_clips);
      for (size_t i = 0; i < n_clips; i++)
        tables[i] = ctx->ws_keytab.as<int8_t>() + (((i / GROUP) & 1) * GROUP + i % GROUP) * table_bytes;
      if (int rc = add_batch_stage (ctx, ctx->stream, n_clips, pcm_in_d, out_d, n_frames, tables, use_limiter, batch))
        return rc;
    }
  AWM_HIP_CHECK (hipEventRecord (ev_aux, ctx->stream));    // (also orders everything behind the clips' producers on the context's stream)
  AWM_HIP_CHECK (hipStreamWaitEvent (table_stream, ev_aux, 0));
  for (int i = 1; i < n_lanes; i++)
    AWM_HIP_CHECK (hipStreamWaitEvent (lanes[i]->stream, ev_aux, 0));
  int rc = 0;
  for (size_t g0 = 0, gi = 0; g0 < n_clips && !rc; g0 += GROUP, gi++)
    {
      const size_t gn = std::min (GROUP, n_clips - g0);
      const int half = int (gi & 1);
      int8_t *dev = ctx->ws_keytab.as<int8_t>() + size_t (half) * GROUP * table_bytes;
      if (gi >= 2)                                          // the clips of group gi - 2 (same table area) are through on every lane
        for (int i = 0; i < n_lanes; i++)
          AWM_HIP_CHECK (hipStreamWaitEvent (table_stream, lane_done[size_t (half) * n_lanes + i], 0));
      awmk::KeyTableArgs ka {};
      ka.sbox = d_aux;
      ka.coded = d_aux + 256;
      ka.round_keys = d_aux + 256 + 2 * 858 + 176 * g0;
      ka.scratch = ctx->ws_keytab_scratch.as<unsigned char>();
      ka.scratch_slots = int (GROUP);
      ka.tables = reinterpret_cast<signed char *> (dev);
      ka.n_keys = (long long) gn;
      ProfScope ps (ctx, PROF_KEYTAB, double (gn) * table_bytes, table_stream);          // the table out, once (360 KB per key)
      AWM_HIP_CHECK (awmk::launch_frame_mod_tables (table_stream, ka));
      AWM_HIP_CHECK (hipEventRecord (ev_tab[half], table_stream));
      for (int i = 0; i < n_lanes; i++)
        AWM_HIP_CHECK (hipStreamWaitEvent (lanes[i]->stream, ev_tab[half], 0));
      if (batched)
        rc = add_batch_run (ctx, ctx->stream, batch, g0, gn, use_limiter);
      for (size_t i = 0; i < gn && !rc && !batched; i++)
        rc = add_full (ctx, pcm_in_d[g0 + i], out_d[g0 + i], n_frames[g0 + i], n_channels, dev + i * table_bytes, params().water_delta,
                       use_limiter, lanes[(g0 + i) % n_lanes]);
      for (int i = 0; i < n_lanes && !rc; i++)
        AWM_HIP_CHECK (hipEventRecord (lane_done[size_t (half) * n_lanes + i], lanes[i]->stream));
    }
  for (int i = 1; i < n_lanes; i++)
    {
      AWM_HIP_CHECK (hipEventRecord (lanes[i]->ev_sync, lanes[i]->stream));
      AWM_HIP_CHECK (hipStreamWaitEvent (ctx->stream, lanes[i]->ev_sync, 0));
    }
  AWM_HIP_CHECK (hipEventRecord (lanes[n_lanes]->ev_sync, table_stream));
  AWM_HIP_CHECK (hipStreamWaitEvent (ctx->stream, lanes[n_lanes]->ev_sync, 0));
  // (the staging block is the context's: its upload has to be through before the next call refills it)
  AWM_HIP_CHECK (hipEventSynchronize (ev_aux));
  return rc;
}

/* the tables K16 builds, copied to the host: n_keys x 2 x 2226 x 81 bytes (for tests: they must equal awm_tab_frame_mod key by key) */
int
awm_debug_frame_mod_tables_d (awm_ctx *ctx, const uint8_t *keys, size_t n_keys, const char *payload_hex, int8_t *tables_out)
{
  AWM_ENTER (ctx);
  const std::vector<int> bits = parse_payload (payload_hex ? payload_hex : "");
  if (bits.empty() || !keys || !tables_out || !params().mix || code_size (ConvBlockType::a, params().payload_size) != 858)
    {
      set_error ("awm_debug_frame_mod_tables_d: bad argument");
      return AWM_ERR_ARG;
    }
  constexpr size_t GROUP = 256;
  const size_t table_bytes = awmk::key_table_bytes();
  const size_t aux_bytes = 256 + 2 * 858 + GROUP * 176;
  if (int rc = ctx->ws_keytab_aux.reserve (aux_bytes)) return rc;
  if (int rc = ctx->ws_keytab.reserve (GROUP * table_bytes)) return rc;
  if (int rc = ctx->ws_keytab_scratch.reserve (GROUP * awmk::key_table_scratch_bytes())) return rc;
  std::vector<unsigned char> aux (aux_bytes);
  std::memcpy (aux.data(), Aes128::sbox(), 256);
  for (int ab = 0; ab < 2; ab++)
    {
      const std::vector<int> coded = code_encode (ab ? ConvBlockType::b : ConvBlockType::a, bits);
      for (size_t i = 0; i < 858 && i < coded.size(); i++)
        aux[256 + 858 * ab + i] = (unsigned char) (coded[i] & 1);
    }
  for (size_t g0 = 0; g0 < n_keys; g0 += GROUP)
    {
      const size_t gn = std::min (GROUP, n_keys - g0);
      for (size_t i = 0; i < gn; i++)
        {
          Aes128 aes;
          aes.set_key (keys + 16 * (g0 + i));
          std::memcpy (aux.data() + 256 + 2 * 858 + 176 * i, aes.round_keys(), 176);
        }
      unsigned char *d_aux = ctx->ws_keytab_aux.as<unsigned char>();
      AWM_HIP_CHECK (hipMemcpyAsync (d_aux, aux.data(), aux_bytes, hipMemcpyHostToDevice, ctx->stream));
      awmk::KeyTableArgs ka {};
      ka.sbox = d_aux;
      ka.coded = d_aux + 256;
      ka.round_keys = d_aux + 256 + 2 * 858;
      ka.scratch = ctx->ws_keytab_scratch.as<unsigned char>();
      ka.scratch_slots = int (GROUP);
      ka.tables = ctx->ws_keytab.as<signed char>();
      ka.n_keys = (long long) gn;
      AWM_HIP_CHECK (awmk::launch_frame_mod_tables (ctx->stream, ka));
      AWM_HIP_CHECK (hipMemcpyAsync (tables_out + g0 * table_bytes, ctx->ws_keytab.ptr, gn * table_bytes, hipMemcpyDeviceToHost, ctx->stream));
      AWM_HIP_CHECK (stream_wait (ctx->stream));
    }
  return 0;
}

int
awm_debug_clip_key_tables_check_d (awm_ctx *ctx, const uint8_t *keys, size_t n_keys, long long mismatch_out[9])
{
  AWM_ENTER (ctx);
  return clip_key_tables_check (ctx, keys, n_keys, mismatch_out);
}

int
awm_add_watermark_batch_keys_d (awm_ctx *ctx, const uint8_t *keys, const char *payload_hex, size_t n_clips, const float *const *pcm_in_d,
                                float *const *out_d, const size_t *n_frames, int n_channels)
{
  AWM_ENTER (ctx);
  if (n_clips && (!keys || !pcm_in_d || !out_d || !n_frames || n_channels < 1))
    {
      set_error ("awm_add_watermark_batch_keys_d: bad argument");
      return AWM_ERR_ARG;
    }
  const std::vector<int> bits = parse_payload (payload_hex ? payload_hex : "");
  if (bits.empty())
    {
      set_error (std::string ("cannot parse payload '") + (payload_hex ? payload_hex : "") + "'");
      return AWM_ERR_ARG;
    }
  const size_t table_bytes = 2 * mark_block_frame_count() * Params::n_bands;
  if (g_key_tables_on_device && params().mix && table_bytes == awmk::key_table_bytes()
      && code_size (ConvBlockType::a, params().payload_size) == 858 && n_clips)
    return add_batch_keys_device_tables (ctx, keys, bits, n_clips, pcm_in_d, out_d, n_frames, n_channels);
  // The host path (--linear, or the device path switched off): a key's table costs about a millisecond of one host core and is 360 KB; the device needs 25 us per clip.  So the tables are built
  // for SUPER clips at a time on up to 64 host threads, straight into one half of a page-locked staging block (the workers copy,
  // not the launching thread), go to the device in ONE copy per SUPER, and the next SUPER is built while the device works on this one.
  // (Round 3a: one task per group of 64 with the launching thread copying 23 MB per group into the staging block: 90 ms for 1024
  // clips where the device needs 25.)
  constexpr size_t SUPER = 128;
  constexpr int ADD_LANES = 8;
  // device side: TWO halves of SUPER tables, like the staging block -- the size does not grow with the batch.  Half h is refilled
  // (super s, s >= 2) only after every lane has finished the clips of super s - 2: the upload waits for the lanes' events on the device.
  if (int rc = ctx->ws_keytab.reserve (2 * SUPER * table_bytes)) return rc;
  if (int rc = ctx->pin_keytab.reserve (2 * SUPER * table_bytes)) return rc;
  const int n_lanes = int (std::min<size_t> (ADD_LANES, std::max<size_t> (1, n_clips)));
  std::vector<WorkLane *> lanes;
  for (int i = 0; i < n_lanes; i++)
    {
      WorkLane *l = ctx->lane (i);
      if (!l)
        {
          set_error ("cannot create a work lane (stream)");
          return AWM_ERR_HIP;
        }
      if (!l->ev_sync)
        AWM_HIP_CHECK (hipEventCreateWithFlags (&l->ev_sync, hipEventDisableTiming));
      lanes.push_back (l);
    }
  const std::vector<Key> key_list = key_list_from (keys, int (n_clips));
  ParamValues *const pv = &params();
  char *const pin_base = ctx->pin_keytab.as<char>();
  auto build_super = [&, pv] (size_t s0, int half) -> bool {
    const size_t sn = std::min (SUPER, n_clips - s0);
    char *pin = pin_base + size_t (half) * SUPER * table_bytes;
    const size_t n_threads = std::max<size_t> (1, std::min<size_t> ({ sn, size_t (64), size_t (std::max (1u, std::thread::hardware_concurrency())) }));
    std::atomic<size_t> next { 0 };
    std::atomic<bool> good { true };
    auto work = [&] {
      ParamsBind bind (pv);
      for (size_t i = next.fetch_add (1); i < sn; i = next.fetch_add (1))
        {
          const std::vector<int8_t> table = build_frame_mod_table (key_list[s0 + i], bits);
          if (table.size() != table_bytes)
            good = false;
          else
            std::memcpy (pin + i * table_bytes, table.data(), table_bytes);
        }
    };
    std::vector<std::thread> threads;
    for (size_t t = 1; t < n_threads; t++)
      threads.emplace_back (work);
    work();
    for (auto& t : threads)
      t.join();
    return good;
  };
  hipEvent_t ev_up[2] = { nullptr, nullptr };
  struct EvGuard { hipEvent_t (&ev)[2]; ~EvGuard() { for (hipEvent_t e : ev) if (e) (void) hipEventDestroy (e); } } guard { ev_up };
  for (auto& e : ev_up)
    AWM_HIP_CHECK (hipEventCreateWithFlags (&e, hipEventDisableTiming));
  std::vector<hipEvent_t> lane_done (2 * size_t (n_lanes), nullptr);                    // [half][lane]: the lane is through the clips of that half
  struct DoneGuard { std::vector<hipEvent_t>& ev; ~DoneGuard() { for (hipEvent_t e : ev) if (e) (void) hipEventDestroy (e); } } done_guard { lane_done };
  for (auto& e : lane_done)
    AWM_HIP_CHECK (hipEventCreateWithFlags (&e, hipEventDisableTiming));
  int rc = 0, last_half = -1;
  std::future<bool> next_built;
  struct FutureGuard { std::future<bool>& f; ~FutureGuard() { if (f.valid()) f.wait(); } } future_guard { next_built };     // (the task writes into the staging block)
  for (size_t s0 = 0, sidx = 0; s0 < n_clips && !rc; s0 += SUPER, sidx++)
    {
      const size_t sn = std::min (SUPER, n_clips - s0);
      const int half = int (sidx & 1);
      const bool built = next_built.valid() ? next_built.get() : build_super (s0, half);
      if (!built)
        {
          set_error ("frame_mod table of unexpected size");
          return AWM_ERR_GENERIC;
        }
      if (s0 + SUPER < n_clips)
        {
          if (sidx >= 1)
            AWM_HIP_CHECK (hipEventSynchronize (ev_up[half ^ 1]));                      // the upload out of the other half is done
          next_built = std::async (std::launch::async, build_super, s0 + SUPER, half ^ 1);     // while the device works on this one
        }
      int8_t *dev = ctx->ws_keytab.as<int8_t>() + size_t (half) * SUPER * table_bytes;
      if (sidx >= 2)
        for (int i = 1; i < n_lanes; i++)                                               // (lane 0 is ctx->stream itself)
          AWM_HIP_CHECK (hipStreamWaitEvent (ctx->stream, lane_done[size_t (half) * n_lanes + i], 0));
      AWM_HIP_CHECK (hipMemcpyAsync (dev, pin_base + size_t (half) * SUPER * table_bytes, sn * table_bytes, hipMemcpyHostToDevice, ctx->stream));
      AWM_HIP_CHECK (hipEventRecord (ev_up[half], ctx->stream));
      for (int i = 1; i < n_lanes; i++)
        AWM_HIP_CHECK (hipStreamWaitEvent (lanes[i]->stream, ev_up[half], 0));          // (also orders the lanes after the clips' producers)
      for (size_t i = 0; i < sn && !rc; i++)
        rc = add_full (ctx, pcm_in_d[s0 + i], out_d[s0 + i], n_frames[s0 + i], n_channels, dev + i * table_bytes, params().water_delta,
                       !params().test_no_limiter, lanes[(s0 + i) % n_lanes]);
      for (int i = 1; i < n_lanes && !rc; i++)
        AWM_HIP_CHECK (hipEventRecord (lane_done[size_t (half) * n_lanes + i], lanes[i]->stream));
      last_half = half;
    }
  // The staging block is the context's: the next call (or the clip `get` with per-clip keys, which stages its group tables through the
  // same block) may write into it as soon as this one returns -- so the last upload out of it has to be through.  Only the copy is
  // awaited, not the clips' kernels.
  if (last_half >= 0)
    AWM_HIP_CHECK (hipEventSynchronize (ev_up[last_half]));
  for (int i = 1; i < n_lanes; i++)
    {
      AWM_HIP_CHECK (hipEventRecord (lanes[i]->ev_sync, lanes[i]->stream));
      AWM_HIP_CHECK (hipStreamWaitEvent (ctx->stream, lanes[i]->ev_sync, 0));
    }
  return rc;
}

int
awm_get_watermark_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels,
                     size_t max_out, awm_pattern *out)
{
  AWM_ENTER (ctx);
  ResultSet rs;
  if (int rc = get_watermark_device (ctx, { capi_key (key) }, make_wav (pcm_d, n_frames, n_channels), rs))
    return rc;
  for (size_t i = 0; i < rs.patterns.size() && i < max_out; i++)
    fill_pattern (rs.patterns[i], out[i]);
  return int (rs.patterns.size());
}


int
awm_get_watermark_keys_d (awm_ctx *ctx, const uint8_t *keys, int n_keys, const float *pcm_d, size_t n_frames, int n_channels,
                          size_t max_out, awm_pattern *out, int *key_of_pattern)
{
  AWM_ENTER (ctx);
  if (n_keys < 0 || (n_keys && !keys) || (max_out && !out))
    {
      set_error ("awm_get_watermark_keys_d: bad argument");
      return AWM_ERR_ARG;
    }
  const std::vector<Key> list = key_list_from (keys, n_keys);
  ResultSet rs;
  if (int rc = get_watermark_device (ctx, list, make_wav (pcm_d, n_frames, n_channels), rs))
    return rc;
  return fill_patterns_keys (rs, list, max_out, out, key_of_pattern);
}

int
awm_get_watermark_batch_d (awm_ctx *ctx, const uint8_t key[16], size_t n_clips, const float *const *pcm_d, const size_t *n_frames,
                           int n_channels, int n_threads, size_t max_out_per_clip, awm_pattern *out, int *n_out)
{
  AWM_ENTER (ctx);
  if (n_clips && (!pcm_d || !n_frames || !n_out || (max_out_per_clip && !out)))
    {
      set_error ("awm_get_watermark_batch_d: bad argument");
      return AWM_ERR_ARG;
    }
  std::vector<DeviceWav> clips;
  for (size_t i = 0; i < n_clips; i++)
    clips.push_back (make_wav (pcm_d[i], n_frames[i], n_channels));
  std::vector<ResultSet> sets;
  if (int rc = get_watermark_batch_device (ctx, { capi_key (key) }, clips, sets, n_threads))
    return rc;
  for (size_t i = 0; i < n_clips; i++)
    {
      n_out[i] = int (sets[i].patterns.size());
      for (size_t j = 0; j < sets[i].patterns.size() && j < max_out_per_clip; j++)
        fill_pattern (sets[i].patterns[j], out[i * max_out_per_clip + j]);
    }
  return 0;
}

int
awm_get_watermark_batch_keys_d (awm_ctx *ctx, const uint8_t *keys, size_t n_clips, const float *const *pcm_d, const size_t *n_frames,
                                int n_channels, int n_threads, size_t max_out_per_clip, awm_pattern *out, int *n_out)
{
  AWM_ENTER (ctx);
  if (n_clips && (!keys || !pcm_d || !n_frames || !n_out || (max_out_per_clip && !out)))
    {
      set_error ("awm_get_watermark_batch_keys_d: bad argument");
      return AWM_ERR_ARG;
    }
  std::vector<DeviceWav> clips;
  for (size_t i = 0; i < n_clips; i++)
    clips.push_back (make_wav (pcm_d[i], n_frames[i], n_channels));
  const std::vector<Key> clip_keys = key_list_from (keys, int (n_clips));
  std::vector<ResultSet> sets;
  if (int rc = get_watermark_batch_device (ctx, {}, clips, sets, n_threads, &clip_keys))
    return rc;
  for (size_t i = 0; i < n_clips; i++)
    {
      n_out[i] = int (sets[i].patterns.size());
      for (size_t j = 0; j < sets[i].patterns.size() && j < max_out_per_clip; j++)
        fill_pattern (sets[i].patterns[j], out[i * max_out_per_clip + j]);
    }
  return 0;
}

int
awm_decode_chunks_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels,
                     int n_chunks, const uint64_t *first_frame, const uint64_t *chunk_frames, int first_is_stream_start,
                     size_t max_out, awm_pattern *out, int *chunk_of_pattern)
{
  AWM_ENTER (ctx);
  std::vector<ChunkRange> chunks;
  for (int i = 0; i < n_chunks; i++)
    {
      if (first_frame[i] + chunk_frames[i] > n_frames)
        {
          set_error ("awm_decode_chunks_d: chunk exceeds the buffer");
          return AWM_ERR_ARG;
        }
      chunks.push_back ({ size_t (first_frame[i]), size_t (chunk_frames[i]), 0.0 });
    }
  std::vector<ResultSet> sets;
  if (int rc = decode_chunks (ctx, { capi_key (key) }, make_wav (pcm_d, n_frames, n_channels), chunks, first_is_stream_start != 0, sets))
    return rc;
  size_t n = 0;
  for (size_t c = 0; c < sets.size(); c++)
    {
      auto& pats = sets[c].patterns;
      std::stable_sort (pats.begin(), pats.end(), [] (const ResultSet::Pattern& a, const ResultSet::Pattern& b) { return a.time < b.time; });
      for (const auto& p : pats)
        {
          if (n < max_out)
            {
              fill_pattern (p, out[n]);
              chunk_of_pattern[n] = int (c);
            }
          n++;
        }
    }
  return int (n);
}

/* ---- file level: the reference's add_watermark / get_watermark (wmcommon.hh:226-228) ------------------------------------ */
namespace {
// --input-format raw / --output-format raw with the --raw-* options for the duration of one call (the reference keeps
// these in the process-global Params, wmcommon.hh:79-83)
struct FormatScope
{
  Format old_in = params().input_format, old_out = params().output_format;
  RawFormat old_raw_in = StreamParams::raw_input_format, old_raw_out = StreamParams::raw_output_format;
  static bool
  apply (const awm_raw_format *f, Format& format, RawFormat& raw)
  {
    if (!f)
      {
        format = Format::AUTO;
        return true;
      }
    if (f->n_channels < 1 || f->sample_rate < 1 || !pcm_format_ok (f->bit_depth, f->encoding))
      return false;
    format = Format::RAW;
    raw.n_channels = f->n_channels;
    raw.sample_rate = f->sample_rate;
    raw.bit_depth = f->bit_depth;
    raw.encoding = f->encoding == 0 ? Encoding::SIGNED : f->encoding == 1 ? Encoding::UNSIGNED : Encoding::FLOAT;
    raw.endian = f->big_endian ? RawFormat::BIG : RawFormat::LITTLE;
    return true;
  }
  ~FormatScope()
  {
    params().input_format = old_in;
    params().output_format = old_out;
    StreamParams::raw_input_format = old_raw_in;
    StreamParams::raw_output_format = old_raw_out;
  }
};
}

int
awm_add_watermark_file (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, const char *in_path, const char *out_path,
                        const awm_raw_format *raw_in, const awm_raw_format *raw_out)
{
  AWM_ENTER (ctx);
  if (!payload_hex || !in_path || !out_path)
    {
      set_error ("awm_add_watermark_file: bad argument");
      return AWM_ERR_ARG;
    }
  FormatScope scope;
  if (!FormatScope::apply (raw_in, params().input_format, StreamParams::raw_input_format)
      || !FormatScope::apply (raw_out, params().output_format, StreamParams::raw_output_format))
    {
      set_error ("awm_add_watermark_file: unsupported raw format");
      return AWM_ERR_ARG;
    }
  file_fail_reset();
  return add_watermark (ctx, capi_key (key), in_path, out_path, payload_hex) ? file_fail_kind() : 0;
}

int
awm_add_stream_watermark_file (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, const char *in_path, const char *out_path,
                               const awm_raw_format *raw_in, const awm_raw_format *raw_out, size_t zero_frames)
{
  AWM_ENTER (ctx);
  if (!payload_hex || !in_path || !out_path)
    {
      set_error ("awm_add_stream_watermark_file: bad argument");
      return AWM_ERR_ARG;
    }
  FormatScope scope;
  if (!FormatScope::apply (raw_in, params().input_format, StreamParams::raw_input_format)
      || !FormatScope::apply (raw_out, params().output_format, StreamParams::raw_output_format))
    {
      set_error ("awm_add_stream_watermark_file: unsupported raw format");
      return AWM_ERR_ARG;
    }
  file_fail_reset();
  return add_watermark_at (ctx, capi_key (key), in_path, out_path, payload_hex, zero_frames) ? file_fail_kind() : 0;
}

int
awm_add_get_watermark_file (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, const char *in_path, const char *out_path,
                            const awm_raw_format *raw_in, const awm_raw_format *raw_out, size_t max_out, awm_pattern *out)
{
  AWM_ENTER (ctx);
  if (!payload_hex || !in_path || !out_path || (max_out && !out))
    {
      set_error ("awm_add_get_watermark_file: bad argument");
      return AWM_ERR_ARG;
    }
  FormatScope scope;
  if (!FormatScope::apply (raw_in, params().input_format, StreamParams::raw_input_format)
      || !FormatScope::apply (raw_out, params().output_format, StreamParams::raw_output_format))
    {
      set_error ("awm_add_get_watermark_file: unsupported raw format");
      return AWM_ERR_ARG;
    }
  ResultSet rs;
  file_fail_reset();
  if (add_get_watermark (ctx, capi_key (key), in_path, out_path, payload_hex, rs))
    return file_fail_kind();
  for (size_t i = 0; i < rs.patterns.size() && i < max_out; i++)
    fill_pattern (rs.patterns[i], out[i]);
  return int (rs.patterns.size());
}

int
awm_get_watermark_file (awm_ctx *ctx, const uint8_t key[16], const char *in_path, const awm_raw_format *raw_in,
                        size_t max_out, awm_pattern *out)
{
  AWM_ENTER (ctx);
  if (!in_path || (max_out && !out))
    {
      set_error ("awm_get_watermark_file: bad argument");
      return AWM_ERR_ARG;
    }
  FormatScope scope;
  if (!FormatScope::apply (raw_in, params().input_format, StreamParams::raw_input_format))
    {
      set_error ("awm_get_watermark_file: unsupported raw format");
      return AWM_ERR_ARG;
    }
  Error err;
  auto in_stream = AudioInputStream::create (in_path, err);
  if (err)
    {
      set_error (std::string ("error loading ") + in_path + ": " + err.message());
      return AWM_ERR_IO;
    }
  ResultSet rs;
  size_t n_values = 0;
  file_fail_reset();
  if (get_watermark_stream (ctx, { capi_key (key) }, in_stream.get(), false, rs, n_values, in_path))
    return file_fail_kind();
  for (size_t i = 0; i < rs.patterns.size() && i < max_out; i++)
    fill_pattern (rs.patterns[i], out[i]);
  return int (rs.patterns.size());
}

int
awm_get_watermark_keys_file (awm_ctx *ctx, const uint8_t *keys, int n_keys, const char *in_path, const awm_raw_format *raw_in,
                             size_t max_out, awm_pattern *out, int *key_of_pattern)
{
  AWM_ENTER (ctx);
  if (!in_path || n_keys < 0 || (n_keys && !keys) || (max_out && !out))
    {
      set_error ("awm_get_watermark_keys_file: bad argument");
      return AWM_ERR_ARG;
    }
  FormatScope scope;
  if (!FormatScope::apply (raw_in, params().input_format, StreamParams::raw_input_format))
    {
      set_error ("awm_get_watermark_keys_file: unsupported raw format");
      return AWM_ERR_ARG;
    }
  Error err;
  auto in_stream = AudioInputStream::create (in_path, err);
  if (err)
    {
      set_error (std::string ("error loading ") + in_path + ": " + err.message());
      return AWM_ERR_IO;
    }
  const std::vector<Key> list = key_list_from (keys, n_keys);
  ResultSet rs;
  size_t n_values = 0;
  file_fail_reset();
  if (get_watermark_stream (ctx, list, in_stream.get(), false, rs, n_values, in_path))
    return file_fail_kind();
  return fill_patterns_keys (rs, list, max_out, out, key_of_pattern);
}

int
awm_plan_chunks (size_t n_frames, size_t max_out, uint64_t *first_frame, uint64_t *chunk_frames, double *time_offset)
{
  const auto chunks = plan_chunks (n_frames, 1);
  for (size_t i = 0; i < chunks.size() && i < max_out; i++)
    {
      first_frame[i] = chunks[i].first_frame;
      chunk_frames[i] = chunks[i].n_frames;
      time_offset[i] = chunks[i].time_offset;
    }
  return int (chunks.size());
}

int
awm_merge_patterns (const uint8_t key[16], const awm_pattern *patterns, const int *chunk_count, int n_chunks,
                    size_t max_out, awm_pattern *out)
{
  const Key k = capi_key (key);
  ResultSet result;
  size_t pos = 0;
  for (int c = 0; c < n_chunks; c++)
    {
      ResultSet chunk;
      for (int i = 0; i < chunk_count[c]; i++, pos++)
        {
          const awm_pattern& p = patterns[pos];
          SyncFinder::Score score { size_t (p.sync_index), p.sync_quality, ConvBlockType (p.block_type) };
          chunk.add_pattern (k, p.time, score, std::vector<int> (p.bits, p.bits + p.n_bits), p.decode_error,
                             ResultSet::Type (p.type), p.speed);
        }
      result.merge (chunk);
    }
  result.sort ({ k });
  for (size_t i = 0; i < result.patterns.size() && i < max_out; i++)
    fill_pattern (result.patterns[i], out[i]);
  return int (result.patterns.size());
}

int
awm_decode_chunk_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels,
                    int first_chunk, size_t max_out, awm_pattern *out)
{
  AWM_ENTER (ctx);
  ResultSet rs;
  if (int rc = decode_chunk (ctx, rs, { capi_key (key) }, make_wav (pcm_d, n_frames, n_channels), first_chunk != 0))
    return rc;
  std::stable_sort (rs.patterns.begin(), rs.patterns.end(), [] (const ResultSet::Pattern& a, const ResultSet::Pattern& b) { return a.time < b.time; });
  for (size_t i = 0; i < rs.patterns.size() && i < max_out; i++)
    fill_pattern (rs.patterns[i], out[i]);
  return int (rs.patterns.size());
}

/* ---- speed detection (reference wmspeed.cc) ---------------------------------------------------------------------- */
static DeviceWav
make_wav_rate (const float *pcm_d, size_t n_frames, int n_channels, int rate)
{
  DeviceWav w = make_wav (pcm_d, n_frames, n_channels);
  w.sample_rate = rate;
  return w;
}

void
awm_set_speed_params (int detect_speed, int detect_speed_patient, double try_speed, double test_speed)
{
  ParamValues& g = global_params();
  g.detect_speed = detect_speed != 0;
  g.detect_speed_patient = detect_speed_patient != 0;
  g.try_speed = try_speed;
  g.test_speed = test_speed;
}

size_t
awm_resample_ratio_frames (size_t n_frames, int n_channels, int rate, double ratio, double max_in_seconds)
{
  size_t in_frames = n_frames;
  if (max_in_seconds > 0)
    in_frames = std::min<size_t> (in_frames * n_channels, n_channels * lrint (rate * max_in_seconds)) / n_channels;
  return size_t (lrint (in_frames * ratio));
}

int
awm_resample_ratio_d (awm_ctx *ctx, const float *pcm_in_d, size_t n_frames, int n_channels, int rate, double ratio,
                      double max_in_seconds, float *out_d, size_t n_out_frames)
{
  AWM_ENTER (ctx);
  if ((n_frames && !pcm_in_d) || (n_out_frames && !out_d) || n_channels < 1 || rate < 1)
    {
      set_error ("awm_resample_ratio_d: bad argument");
      return AWM_ERR_ARG;
    }
  DevBuffer tmp;
  size_t n_out = 0;
  int rc = resample_ratio_device (ctx, ctx, make_wav_rate (pcm_in_d, n_frames, n_channels, rate), ratio, max_in_seconds, tmp, &n_out);
  if (!rc)
    {
      const size_t n = std::min (n_out, n_out_frames) * n_channels * sizeof (float);
      if (n && hipMemcpyAsync (out_d, tmp.ptr, n, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
        rc = AWM_ERR_HIP;
      if (!rc && hipStreamSynchronize (ctx->stream) != hipSuccess)
        rc = AWM_ERR_HIP;
    }
  tmp.release();
  return rc;
}

int
awm_speed_clip_location_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels, int rate,
                           double seconds, int candidates, double *location)
{
  AWM_ENTER (ctx);
  return speed_clip_location (ctx, ctx, capi_key (key), make_wav_rate (pcm_d, n_frames, n_channels, rate), seconds, candidates, location);
}

int
awm_speed_mags_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels, int rate,
                  double clip_location, double center, double seconds, size_t max_rows, float *out)
{
  AWM_ENTER (ctx);
  std::vector<float> m;
  int rows = 0;
  if (int rc = speed_mags (ctx, ctx, capi_key (key), make_wav_rate (pcm_d, n_frames, n_channels, rate), clip_location, center, seconds, m, &rows))
    return rc;
  std::copy (m.begin(), m.begin() + std::min<size_t> (rows, max_rows) * 510 * 2, out);
  return rows;
}

int
awm_speed_scan_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels, int rate,
                  double clip_location, double seconds, double step, int n_steps, int n_center_steps,
                  const double *speeds, int n_speeds, size_t max_out, double *out_speed, double *out_quality)
{
  AWM_ENTER (ctx);
  std::vector<SpeedScore> scores;
  if (int rc = speed_scan (ctx, ctx, capi_key (key), make_wav_rate (pcm_d, n_frames, n_channels, rate), clip_location,
                           { seconds, step, n_steps, n_center_steps }, std::vector<double> (speeds, speeds + n_speeds), scores))
    return rc;
  std::sort (scores.begin(), scores.end(), [] (const SpeedScore& a, const SpeedScore& b) { return a.speed < b.speed; });
  for (size_t i = 0; i < scores.size() && i < max_out; i++)
    {
      out_speed[i] = scores[i].speed;
      out_quality[i] = scores[i].quality;
    }
  return int (scores.size());
}

int
awm_detect_speed_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels, int rate,
                    int patient, double *speed_out, double *quality_out)
{
  AWM_ENTER (ctx);
  const bool old_patient = params().detect_speed_patient;
  params().detect_speed_patient = patient != 0;
  std::vector<DetectSpeedResult> results;
  double speed = 0, quality = 0;
  const int rc = detect_speed (ctx, ctx, { capi_key (key) }, make_wav_rate (pcm_d, n_frames, n_channels, rate), nullptr, results, &speed, &quality);
  params().detect_speed_patient = old_patient;
  if (rc)
    return rc;
  if (speed_out)
    *speed_out = speed;
  if (quality_out)
    *quality_out = quality;
  return results.empty() ? 0 : 1;
}

} // extern "C"
