/* oracle/ref_shim/ref_capi_add.cc -- wrappers around the reference's embed side.
 * Includes /root/reference/src/wmadd.cc as-is so its file-local helpers
 * (init_frame_mod_vec wmadd.cc:148-162, WatermarkGen :271-351 ...) can be called.
 * TEST INFRASTRUCTURE ONLY. */
#include <vector>
#include <string>
#include <memory>
#include <complex>
#include <algorithm>
#include <mutex>
#include <map>
#include <functional>
#include <random>
#include <regex>
#include <array>
#include <thread>
#include <condition_variable>
#include <unistd.h>

#include "wmadd.cc"          /* the reference translation unit, unmodified */
#include "ref_capi.h"

namespace {

Key
make_key (const uint8_t k[16])
{
  /* Key has no raw setter: write a temporary key file and use Key::load_key (random.cc:295-360)
   * -- but the key NAME then differs; name is irrelevant for the numeric path. For the all-zero key
   * keep the default-constructed Key exactly like the CLI does without --key. */
  Key key;
  bool zero = true;
  for (int i = 0; i < 16; i++) if (k[i]) zero = false;
  if (zero)
    return key;
  bool test_key = true;            /* --test-key N layout: big-endian u64 in the first 8 bytes */
  for (int i = 8; i < 16; i++) if (k[i]) test_key = false;
  if (test_key)
    {
      uint64_t v = 0;
      for (int i = 0; i < 8; i++) v = (v << 8) | k[i];
      key.set_test_key (v);
      return key;
    }
  char name[64];
  snprintf (name, sizeof (name), "/tmp/awm_refkey_%d_%p", (int) getpid(), (void *) &key);
  FILE *f = fopen (name, "w");
  fprintf (f, "key ");
  for (int i = 0; i < 16; i++) fprintf (f, "%02x", k[i]);
  fprintf (f, "\n");
  fclose (f);
  key.load_key (name);
  unlink (name);
  return key;
}

class MemInputStream : public AudioInputStream
{
  const float *m_data; size_t m_frames; int m_ch; int m_rate; size_t m_pos = 0;
public:
  MemInputStream (const float *d, size_t frames, int ch, int rate) : m_data (d), m_frames (frames), m_ch (ch), m_rate (rate) {}
  int bit_depth() const override { return 32; }
  int sample_rate() const override { return m_rate; }
  int n_channels() const override { return m_ch; }
  size_t n_frames() const override { return m_frames; }
  Encoding encoding() const override { return Encoding::FLOAT; }
  Error read_frames (std::vector<float>& samples, size_t count) override
  {
    size_t n = std::min (count, m_frames - m_pos);
    samples.assign (m_data + m_pos * m_ch, m_data + (m_pos + n) * m_ch);
    m_pos += n;
    return Error::Code::NONE;
  }
};
class MemOutputStream : public AudioOutputStream
{
  int m_ch; int m_rate;
public:
  std::vector<float> data;
  MemOutputStream (int ch, int rate) : m_ch (ch), m_rate (rate) {}
  int bit_depth() const override { return 32; }
  int sample_rate() const override { return m_rate; }
  int n_channels() const override { return m_ch; }
  Error write_frames (const std::vector<float>& frames) override { data.insert (data.end(), frames.begin(), frames.end()); return Error::Code::NONE; }
  Error close() override { return Error::Code::NONE; }
};
} // namespace

Key awm_ref_make_key (const uint8_t k[16]) { return make_key (k); }

extern "C" {

void
ref_set_params (double water_delta, int mix, int frames_per_bit, int test_no_limiter,
                double sync_threshold2, int n_best, double chunk_size_min)
{
  Params::water_delta = water_delta;
  Params::mix = mix;
  Params::frames_per_bit = frames_per_bit;
  Params::test_no_limiter = test_no_limiter;
  Params::sync_threshold2 = sync_threshold2;
  Params::get_n_best = n_best;
  Params::get_chunk_size = chunk_size_min;
}
void ref_set_quiet (int quiet) { set_log_level (quiet ? Log::WARNING : Log::INFO); }

void
ref_random_u64 (const uint8_t key[16], uint64_t seed, int stream, size_t n, uint64_t *out)
{
  Random rng (make_key (key), seed, Random::Stream (stream));
  for (size_t i = 0; i < n; i++) out[i] = rng();
}
void
ref_random_double (const uint8_t key[16], uint64_t seed, int stream, size_t n, double *out)
{
  Random rng (make_key (key), seed, Random::Stream (stream));
  for (size_t i = 0; i < n; i++) out[i] = rng.random_double();
}
void
ref_gen_noise (const uint8_t key[16], size_t n_values, float *out)
{
  /* audiowmark.cc:399-417 */
  Random rng (make_key (key), 0, /* there is no stream for this test */ Random::Stream::data_up_down);
  for (size_t i = 0; i < n_values; i++)
    out[i] = rng.random_double() * 2 - 1;
}
void
ref_up_down (const uint8_t key[16], int stream, int f, int up[30], int down[30])
{
  UpDownGen g (make_key (key), Random::Stream (stream));
  UpDownArray u, d;
  g.get (f, u, d);
  for (int i = 0; i < 30; i++) { up[i] = u[i]; down[i] = d[i]; }
}
void
ref_bit_pos (const uint8_t key[16], int *pos)
{
  BitPosGen g (make_key (key));
  const int ns = mark_sync_frame_count(), nd = mark_data_frame_count();
  for (int f = 0; f < ns; f++) pos[f] = g.sync_frame (f);
  for (int f = 0; f < nd; f++) pos[ns + f] = g.data_frame (f);
}
size_t
ref_mix_entries (const uint8_t key[16], int *out)
{
  auto e = gen_mix_entries (make_key (key));
  for (size_t i = 0; i < e.size(); i++) { out[3 * i] = e[i].frame; out[3 * i + 1] = e[i].up; out[3 * i + 2] = e[i].down; }
  return e.size();
}
void
ref_window (size_t n, float *out)
{
  auto w = FFTAnalyzer::gen_normalized_window (n);
  std::copy (w.begin(), w.end(), out);
}
size_t
ref_bit_order (const uint8_t key[16], size_t n, unsigned *order)
{
  /* randomize_bit_order (wmcommon.hh:165-185): recover `order` by encoding the identity */
  std::vector<unsigned> id (n);
  for (size_t i = 0; i < n; i++) id[i] = i;
  auto o = randomize_bit_order (make_key (key), id, /* encode */ true);
  std::copy (o.begin(), o.end(), order);
  return n;
}
size_t
ref_conv_encode (int block_type, const int *bits, size_t n, int *out)
{
  auto r = conv_encode (ConvBlockType (block_type), std::vector<int> (bits, bits + n));
  std::copy (r.begin(), r.end(), out);
  return r.size();
}
size_t
ref_conv_decode_soft (int block_type, const float *coded, size_t n, int *out, float *error_out)
{
  auto r = conv_decode_soft (ConvBlockType (block_type), std::vector<float> (coded, coded + n), error_out);
  std::copy (r.begin(), r.end(), out);
  return r.size();
}
int
ref_frame_mod (const uint8_t key[16], const char *payload_hex, int ab, uint8_t *out)
{
  auto bitvec = parse_payload (payload_hex);
  if (bitvec.empty()) return -1;
  std::vector<std::vector<FrameMod>> v;
  init_frame_mod_vec (make_key (key), v, ab, bitvec);
  for (size_t f = 0; f < v.size(); f++)
    for (size_t b = 0; b < v[f].size(); b++)
      out[f * (Params::max_band + 1) + b] = uint8_t (v[f][b]);
  return (int) v.size();
}
void
ref_ifft (size_t n, const float *spect, float *out)
{
  FFTProcessor p (n);
  std::vector<std::complex<float>> in (n / 2 + 1);
  for (size_t i = 0; i < in.size(); i++) in[i] = std::complex<float> (spect[2 * i], spect[2 * i + 1]);
  auto r = p.ifft (in);
  std::copy (r.begin(), r.end(), out);
}
int
ref_fft_range (const float *samples, size_t n_values, int n_channels, size_t start_index, size_t frame_count, float *out)
{
  std::vector<float> s (samples, samples + n_values);
  FFTAnalyzer a (n_channels);
  auto r = a.fft_range (s, start_index, frame_count);
  if (r.empty()) return 0;
  size_t o = 0;
  for (auto& v : r)
    for (auto c : v) { out[o++] = c.real(); out[o++] = c.imag(); }
  return (int) r.size();
}
int
ref_add_at (const uint8_t key[16], const float *samples, size_t n_frames, int n_channels, int sample_rate,
            const char *payload_hex, size_t zero_frames, float *out, size_t *out_frames)
{
  /* add_stream_watermark with its zero_frames argument (wmadd.cc:448, 501-526): what hls.cc:279 passes for a segment */
  MemInputStream in (samples, n_frames, n_channels, sample_rate);
  MemOutputStream os (n_channels, sample_rate);
  int rc = add_stream_watermark (make_key (key), &in, &os, payload_hex, zero_frames);
  if (out_frames) *out_frames = os.data.size() / n_channels;
  std::copy (os.data.begin(), os.data.end(), out);
  return rc;
}
int
ref_add (const uint8_t key[16], const float *samples, size_t n_frames, int n_channels, int sample_rate,
         const char *payload_hex, float *out, size_t *out_frames, double *snr_db)
{
  (void) snr_db;
  return ref_add_at (key, samples, n_frames, n_channels, sample_rate, payload_hex, 0, out, out_frames);
}


/* RawConverter::create + from_raw / to_raw (rawconverter.cc:73-286) for one raw format; returns 0, -1 if the format is refused */
int
ref_raw_convert (int bit_depth, int encoding, int big_endian, int to_raw, const void *in, void *out, size_t n_values)
{
  RawFormat fmt (2, 44100, bit_depth);
  fmt.set_encoding (encoding == 0 ? Encoding::SIGNED : encoding == 1 ? Encoding::UNSIGNED : Encoding::FLOAT);
  fmt.set_endian (big_endian ? RawFormat::BIG : RawFormat::LITTLE);
  Error err;
  std::unique_ptr<RawConverter> conv (RawConverter::create (fmt, err));
  if (err || !conv)
    return -1;
  if (to_raw)
    conv->to_raw ((const float *) in, (unsigned char *) out, n_values);
  else
    conv->from_raw ((const unsigned char *) in, (float *) out, n_values);
  return 0;
}
} /* extern "C" */
