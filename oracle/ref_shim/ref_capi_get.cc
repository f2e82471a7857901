/* oracle/ref_shim/ref_capi_get.cc -- wrappers around the reference's detect side.
 * Includes /root/reference/src/wmget.cc as-is (file-local mix_decode, BlockDecoder,
 * ClipDecoder, decode, ResultSet) and opens the private sections of SyncFinder /
 * ResultSet / WavChunkLoader for inspection only.  TEST INFRASTRUCTURE ONLY. */
#include <vector>
#include <string>
#include <memory>
#include <complex>
#include <algorithm>
#include <mutex>
#include <map>
#include <functional>
#include <random>
#include <array>
#include <thread>
#include <condition_variable>
#include <atomic>
#include <queue>
#include <unistd.h>

#define private public
#define protected public
#include "wmget.cc"          /* the reference translation unit, unmodified */
#undef private
#undef protected
#include "ref_capi.h"

Key awm_ref_make_key (const uint8_t k[16]);

namespace {
void
fill_patterns (const std::vector<ResultSet::Pattern>& patterns, size_t max_out, ref_pattern *out)
{
  for (size_t i = 0; i < patterns.size() && i < max_out; i++)
    {
      const auto& p = patterns[i];
      ref_pattern& o = out[i];
      o.time = p.time;
      o.sync_index = p.sync_score.index;
      o.sync_quality = p.sync_score.quality;
      o.block_type = int (p.sync_score.block_type);
      o.type = int (p.type);
      o.decode_error = p.decode_error;
      o.speed = p.speed;
      o.n_bits = std::min<int> (p.bit_vec.size(), 128);
      for (int b = 0; b < o.n_bits; b++) o.bits[b] = p.bit_vec[b];
    }
}
} // namespace

extern "C" {

int
ref_sync_bits (const uint8_t key[16], int clip_mode, int *out)
{
  auto sb = SyncFinder::get_sync_bits (awm_ref_make_key (key), clip_mode ? SyncFinder::Mode::CLIP : SyncFinder::Mode::BLOCK);
  size_t o = 0;
  for (auto& bit : sb)
    for (auto& fb : bit)
      {
        out[o++] = fb.frame;
        for (auto u : fb.up) out[o++] = u;
        for (auto d : fb.down) out[o++] = d;
      }
  return (int) sb[0].size();
}
int
ref_sync_fft (const float *samples, size_t n_values, int n_channels, size_t index, size_t frame_count,
              const char *want_frames, size_t first, size_t last, float *db_out, char *have_out)
{
  WavData wav (std::vector<float> (samples, samples + n_values), n_channels, 44100, 16);
  SyncFinder sf;
  sf.wav_data_first = first;
  sf.wav_data_last = last;
  std::vector<float> db; std::vector<char> have, want;
  if (want_frames) want.assign (want_frames, want_frames + frame_count);
  sf.sync_fft (wav, index, frame_count, db, have, want);
  std::copy (db.begin(), db.end(), db_out);
  std::copy (have.begin(), have.end(), have_out);
  return (int) have.size();
}
double
ref_sync_decode (const uint8_t key[16], int clip_mode, size_t start_frame, const float *db, size_t n_db, const char *have, size_t n_have)
{
  auto sb = SyncFinder::get_sync_bits (awm_ref_make_key (key), clip_mode ? SyncFinder::Mode::CLIP : SyncFinder::Mode::BLOCK);
  SyncFinder sf;
  return sf.sync_decode (sb, start_frame, std::vector<float> (db, db + n_db), std::vector<char> (have, have + n_have));
}
int
ref_sync_search (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int clip_mode,
                 size_t max_out, uint64_t *index, double *quality, int *block_type)
{
  WavData wav (std::vector<float> (samples, samples + n_values), n_channels, 44100, 16);
  SyncFinder sf;
  auto kr = sf.search ({ awm_ref_make_key (key) }, wav, clip_mode ? SyncFinder::Mode::CLIP : SyncFinder::Mode::BLOCK);
  const auto& scores = kr[0].sync_scores;
  for (size_t i = 0; i < scores.size() && i < max_out; i++)
    {
      index[i] = scores[i].index;
      quality[i] = scores[i].quality;
      block_type[i] = int (scores[i].block_type);
    }
  return (int) scores.size();
}
size_t
ref_search_approx (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int clip_mode,
                   size_t max_out, uint64_t *index, double *raw_quality, double *local_mean)
{
  WavData wav (std::vector<float> (samples, samples + n_values), n_channels, 44100, 16);
  SyncFinder sf;
  auto mode = clip_mode ? SyncFinder::Mode::CLIP : SyncFinder::Mode::BLOCK;
  if (clip_mode)
    sf.scan_silence (wav);
  else
    {
      sf.wav_data_first = 0;
      sf.wav_data_last = wav.samples().size();
    }
  Key k = awm_ref_make_key (key);
  std::vector<SyncFinder::SearchKeyResult> skr (1);
  skr[0].key = k;
  std::vector<std::vector<std::vector<SyncFinder::FrameBit>>> sync_bits { SyncFinder::get_sync_bits (k, mode) };
  sf.search_approx (skr, sync_bits, wav, mode);
  const auto& s = skr[0].scores;
  for (size_t i = 0; i < s.size() && i < max_out; i++)
    {
      index[i] = s[i].index;
      raw_quality[i] = s[i].raw_quality;
      local_mean[i] = s[i].local_mean;
    }
  return s.size();
}
int
ref_mix_decode (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, size_t index, float *out)
{
  std::vector<float> s (samples, samples + n_values);
  FFTAnalyzer a (n_channels);
  auto fft_out = a.fft_range (s, index, mark_sync_frame_count() + mark_data_frame_count());
  if (fft_out.empty()) return 0;
  auto r = mix_or_linear_decode (awm_ref_make_key (key), fft_out, n_channels);
  std::copy (r.begin(), r.end(), out);
  return (int) r.size();
}
int
ref_decode_chunk (const uint8_t key[16], const float *samples, size_t n_values, int n_channels,
                  int first_chunk, size_t max_out, ref_pattern *out)
{
  WavData wav (std::vector<float> (samples, samples + n_values), n_channels, 44100, 16);
  ResultSet rs;
  decode (rs, { awm_ref_make_key (key) }, wav, {}, first_chunk);
  /* deterministic order like ResultSet::merge (wmget.cc:288-316): by time */
  std::stable_sort (rs.patterns.begin(), rs.patterns.end(), [] (const ResultSet::Pattern& a, const ResultSet::Pattern& b) { return a.time < b.time; });
  fill_patterns (rs.patterns, max_out, out);
  return (int) rs.patterns.size();
}
int
ref_get (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, size_t max_out, ref_pattern *out)
{
  return ref_get_rate (key, samples, n_values, n_channels, 44100, max_out, out);
}
int
ref_get_rate (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int sample_rate, size_t max_out, ref_pattern *out)
{
  /* write the data as headerless float32 and read it back through the reference's own
   * RawInputStream + WavChunkLoader, then run the body of get_watermark (wmget.cc:971-1013) */
  char name[64];
  snprintf (name, sizeof (name), "/tmp/awm_refget_%d_%p.raw", (int) getpid(), (void *) samples);
  FILE *f = fopen (name, "wb");
  if (!f || fwrite (samples, sizeof (float), n_values, f) != n_values) return -1;
  fclose (f);

  const Format old_format = Params::input_format;
  Params::input_format = Format::RAW;
  Params::raw_input_format.set_channels (n_channels);
  Params::raw_input_format.set_sample_rate (sample_rate);      /* != 44100: WavChunkLoader resamples (wavchunkloader.cc:70-72) */
  Params::raw_input_format.set_bit_depth (32);
  Params::raw_input_format.set_encoding (Encoding::FLOAT);
  Params::raw_input_format.set_endian (RawFormat::LITTLE);

  std::vector<Key> key_list { awm_ref_make_key (key) };
  ResultSet result_set;
  bool first_chunk = true;
  int rc = 0;
  {
    WavChunkLoader loader (name);
    while (!loader.done())
      {
        Error err = loader.load_next_chunk();
        if (err) { rc = -2; break; }
        if (!loader.done())
          {
            ResultSet chunk_result_set;
            decode (chunk_result_set, key_list, loader.wav_data(), {}, first_chunk);
            chunk_result_set.apply_time_offset (loader.time_offset());
            result_set.merge (chunk_result_set);
            first_chunk = false;
          }
      }
  }
  unlink (name);
  Params::input_format = old_format;
  if (rc) return rc;
  result_set.sort (key_list);
  fill_patterns (result_set.patterns, max_out, out);
  return (int) result_set.patterns.size();
}

} /* extern "C" */
