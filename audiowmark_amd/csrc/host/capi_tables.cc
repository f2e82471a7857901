// C ABI: key-derived tables (pure host; usable without a GPU) -- see include/awm_hip.h
#include "context.hh"
#include "utils.hh"
#include "wmspeed.hh"
#include "random.hh"
#include <cstring>

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
  Params::water_delta = water_delta;
  Params::mix = mix != 0;
  Params::frames_per_bit = frames_per_bit;
  Params::test_no_limiter = test_no_limiter != 0;
  Params::sync_threshold2 = sync_threshold2;
  Params::get_n_best = n_best;
  Params::get_chunk_size = chunk_size_min;
}

/* -q / --quiet of the command line (reference audiowmark.cc:1020-1023): information messages off */
void
awm_set_quiet (int quiet)
{
  set_log_level (quiet ? Log::WARNING : Log::INFO);
}

} // extern "C"
