// C ABI: key-derived tables (pure host; usable without a GPU) -- see include/awm_hip.h
#include "context.hh"
#include "utils.hh"
#include "wmspeed.hh"
#include "random.hh"
#include <cstring>
#include <memory>

using namespace awm;

static Key
key_from_bytes (const uint8_t key[16])
{
  Key k;
  k.set_raw (key);
  return k;
}

namespace awm { Key capi_key (const uint8_t key[16]) { return key_from_bytes (key); } }

extern "C" {

int
awm_tab_up_down (const uint8_t key[16], int stream, int frame, int up[30], int down[30])
{
  UpDownGen gen (key_from_bytes (key), Random::Stream (stream));
  UpDownArray u, d;
  gen.get (frame, u, d);
  for (int i = 0; i < 30; i++)
    {
      up[i] = u[i];
      down[i] = d[i];
    }
  return 0;
}

int
awm_test_gen_noise (const uint8_t key[16], size_t n_values, float *out)
{
  // reference audiowmark.cc:399-417 (test-gen-noise): uniform [-1, 1) from the data_up_down stream, seed 0
  Random rng (key_from_bytes (key), 0, Random::Stream::data_up_down);
  for (size_t i = 0; i < n_values; i++)
    out[i] = float (rng.random_double() * 2 - 1);
  return 0;
}

int
awm_tab_bit_pos (const uint8_t key[16], int pos[AWM_BLOCK_FRAMES])
{
  BitPosGen gen (key_from_bytes (key));
  const int ns = mark_sync_frame_count(), nd = mark_data_frame_count();
  for (int f = 0; f < ns; f++)
    pos[f] = gen.sync_frame (f);
  for (int f = 0; f < nd; f++)
    pos[ns + f] = gen.data_frame (f);
  return ns + nd;
}

int
awm_tab_mix_entries (const uint8_t key[16], int *out)
{
  const auto entries = gen_mix_entries (key_from_bytes (key));
  for (size_t i = 0; i < entries.size(); i++)
    {
      out[3 * i] = entries[i].frame;
      out[3 * i + 1] = entries[i].up;
      out[3 * i + 2] = entries[i].down;
    }
  return int (entries.size());
}

int
awm_tab_bit_order (const uint8_t key[16], size_t n, unsigned *order)
{
  const auto o = bit_order (key_from_bytes (key), n);
  std::copy (o.begin(), o.end(), order);
  return 0;
}

int
awm_tab_frame_mod (const uint8_t key[16], const char *payload_hex, int8_t *out)
{
  const auto bits = parse_payload (payload_hex ? payload_hex : "");
  if (bits.empty())
    {
      set_error ("cannot parse payload");
      return AWM_ERR_ARG;
    }
  const auto table = build_frame_mod_table (key_from_bytes (key), bits);
  std::memcpy (out, table.data(), table.size());
  return int (table.size());
}

int
awm_tab_sync_bits (const uint8_t key[16], int clip_mode, int *out)
{
  const auto t = build_sync_table (key_from_bytes (key), clip_mode != 0);
  size_t o = 0;
  for (size_t r = 0; r < t.frame.size(); r++)
    {
      out[o++] = t.frame[r];
      for (int i = 0; i < 30; i++) out[o++] = t.up[r * 30 + i];
      for (int i = 0; i < 30; i++) out[o++] = t.down[r * 30 + i];
    }
  return t.rows_per_bit;
}

int
awm_tab_window (size_t n, float *out)
{
  const auto w = gen_normalized_window (n);
  std::copy (w.begin(), w.end(), out);
  return 0;
}

int
awm_tab_synth_window (float *out)
{
  const auto w = gen_synth_window();
  std::copy (w.begin(), w.end(), out);
  return 0;
}

int
awm_conv_encode (int block_type, const int *bits, size_t n, int *out)
{
  if (block_type < 0 || block_type > 2)
    return AWM_ERR_ARG;
  const auto r = conv_encode (ConvBlockType (block_type), std::vector<int> (bits, bits + n));
  std::copy (r.begin(), r.end(), out);
  return int (r.size());
}

/* ---- host side pieces of the speed detection (no GPU needed) ---------------------------------------------------- */
int
awm_speed_select_n_best (double *speed, double *quality, int count, int n)
{
  std::vector<SpeedScore> scores (count);
  for (int i = 0; i < count; i++)
    {
      scores[i].speed = speed[i];
      scores[i].quality = quality[i];
    }
  select_n_best_scores (scores, size_t (n));
  for (size_t i = 0; i < scores.size(); i++)
    {
      speed[i] = scores[i].speed;
      quality[i] = scores[i].quality;
    }
  return int (scores.size());
}

double
awm_speed_smooth_best (const double *speed, const double *quality, int count, double step, double distance)
{
  std::vector<SpeedScore> scores (count);
  for (int i = 0; i < count; i++)
    {
      scores[i].speed = speed[i];
      scores[i].quality = quality[i];
    }
  return count ? score_smooth_find_best (scores, step, distance) : 0;
}

size_t
awm_speed_clip_positions (const uint8_t key[16], size_t n_values, size_t max_out, uint64_t *positions)
{
  Random rng (key_from_bytes (key), 0, Random::Stream::speed_clip);
  size_t count = 0;
  for (size_t p = 0; p < n_values; p += rng() % 1000)
    {
      if (count < max_out)
        positions[count] = p;
      count++;
    }
  return count;
}

int
awm_speed_clip_candidates (const uint8_t key[16], const float *hashed_values, size_t n, int candidates, double *locations)
{
  unsigned char hash[20];
  sha1 (hashed_values, n * sizeof (float), hash);
  uint64_t seed = 0;
  for (int i = 0; i < 8; i++)
    seed = (seed << 8) | hash[i];
  Random rng (key_from_bytes (key), seed, Random::Stream::speed_clip);
  for (int c = 0; c < candidates; c++)
    locations[c] = rng.random_double();
  return 0;
}

void
awm_set_params (double water_delta, int mix, int frames_per_bit, int test_no_limiter,
                double sync_threshold2, int n_best, double chunk_size_min)
{
  ParamValues& g = global_params();
  g.water_delta = water_delta;
  g.mix = mix != 0;
  g.frames_per_bit = frames_per_bit;
  g.test_no_limiter = test_no_limiter != 0;
  g.sync_threshold2 = sync_threshold2;
  g.get_n_best = n_best;
  g.get_chunk_size = chunk_size_min;
}

/* ---- the whole parameter set, process-wide or per context (awm_hip.h: awm_params) ---- */
static void
params_to_c (const ParamValues& v, awm_params *p)
{
  p->struct_size = sizeof (awm_params);
  p->water_delta = v.water_delta;
  p->mix = v.mix;
  p->hard = v.hard;
  p->strict = v.strict;
  p->snr = v.snr;
  p->payload_size = int (v.payload_size);
  p->frames_per_bit = v.frames_per_bit;
  p->sync_threshold2 = v.sync_threshold2;
  p->get_n_best = v.get_n_best;
  p->get_chunk_size = v.get_chunk_size;
  p->detect_speed = v.detect_speed;
  p->detect_speed_patient = v.detect_speed_patient;
  p->try_speed = v.try_speed;
  p->test_speed = v.test_speed;
  p->test_cut = v.test_cut;
  p->test_no_sync = v.test_no_sync;
  p->test_no_limiter = v.test_no_limiter;
  p->test_truncate = v.test_truncate;
}

static int
params_from_c (const awm_params *p, ParamValues& v)
{
  if (!p || p->struct_size != sizeof (awm_params))
    {
      set_error ("awm_params: struct_size does not match this library (use awm_params_init)");
      return AWM_ERR_ARG;
    }
  if (!(p->water_delta >= 0) || p->payload_size < 1 || p->frames_per_bit < 1 || p->get_n_best < 1 || !(p->get_chunk_size > 0))
    {
      set_error ("awm_params: value out of range");
      return AWM_ERR_ARG;
    }
  v.water_delta = p->water_delta;
  v.mix = p->mix != 0;
  v.hard = p->hard != 0;
  v.strict = p->strict != 0;
  v.snr = p->snr != 0;
  v.payload_size = size_t (p->payload_size);
  v.frames_per_bit = p->frames_per_bit;
  v.sync_threshold2 = p->sync_threshold2;
  v.get_n_best = p->get_n_best;
  v.get_chunk_size = p->get_chunk_size;
  v.detect_speed = p->detect_speed != 0;
  v.detect_speed_patient = p->detect_speed_patient != 0;
  v.try_speed = p->try_speed;
  v.test_speed = p->test_speed;
  v.test_cut = p->test_cut;
  v.test_no_sync = p->test_no_sync != 0;
  v.test_no_limiter = p->test_no_limiter != 0;
  v.test_truncate = p->test_truncate;
  return 0;
}

void
awm_params_init (awm_params *p)
{
  if (p)
    params_to_c (ParamValues(), p);
}

int
awm_set_global_params (const awm_params *p)
{
  ParamValues v = global_params();          // (what awm_params does not carry -- formats, json output -- stays)
  if (int rc = params_from_c (p, v))
    return rc;
  global_params() = v;
  return 0;
}

int
awm_ctx_set_params (awm_ctx *ctx, const awm_params *p)
{
  if (!ctx)
    {
      set_error ("null context");
      return AWM_ERR_ARG;
    }
  if (!p)
    {
      ctx->own_params.reset();
      return 0;
    }
  auto v = std::make_unique<ParamValues> (ctx->own_params ? *ctx->own_params : global_params());
  if (int rc = params_from_c (p, *v))
    return rc;
  ctx->own_params = std::move (v);
  return 0;
}

int
awm_ctx_get_params (const awm_ctx *ctx, awm_params *out)
{
  if (!out)
    return AWM_ERR_ARG;
  params_to_c (ctx && ctx->own_params ? *ctx->own_params : global_params(), out);
  return 0;
}

/* -q / --quiet of the command line (reference audiowmark.cc:1020-1023): information messages off */
void
awm_set_quiet (int quiet)
{
  set_log_level (quiet ? Log::WARNING : Log::INFO);
}

} // extern "C"
