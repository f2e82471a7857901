/* oracle/awm_oracle.cc -- CPU restatement of the audiowmark spectral watermark path.
 *
 * TEST INFRASTRUCTURE ONLY (see awm_oracle.h).  Plain single-file C++, no GPU, no third party
 * library.  Every function names the reference file:line (relative to the reference's src/)
 * whose behaviour it restates.  The FFT is the only arithmetic that cannot be pinned bit for
 * bit (the reference uses FFTW, absent here): it is evaluated in double precision and rounded
 * to float once, everything else follows the reference's operation order and types.
 *
 * Compile with -ffp-contract=off (oracle/Makefile): the reference is built for baseline
 * x86-64 where float expressions are never fused.
 */
#include "awm_oracle.h"
#include "zita_restated.h"
#include "sha1.h"
#include "aes128.h"

#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

using std::vector;
typedef std::complex<float> cfloat;

/* ---- parameters (wmcommon.hh:33-89, wmcommon.cc:27-46) --------------------------------- */
struct P
{
  static constexpr int frame_size = 1024, bands_per_frame = 30, max_band = 100, min_band = 20, n_bands = 81;
  static constexpr int sync_bits = 6, sync_frames_per_bit = 85, sync_search_step = 256, sync_search_fine = 8;
  static constexpr int frames_pad_start = 250, mark_sample_rate = 44100, payload_size = 128;
  static int    frames_per_bit;
  static double water_delta, sync_threshold2, chunk_size_min;
  static bool   mix, test_no_limiter;
  static int    n_best, threads;
};
int    P::frames_per_bit = 2;
double P::water_delta = 0.01, P::sync_threshold2 = 0.35, P::chunk_size_min = 30;
bool   P::mix = true, P::test_no_limiter = false;
int    P::n_best = 8, P::threads = 0;

/* ---- PRNG (random.cc:97-161, random.hh:73-113) ----------------------------------------- */
enum Stream { data_up_down = 1, sync_up_down = 2, speed_clip = 3, mix_stream = 4, bit_order_stream = 5, frame_position = 6 };

class Rng
{
  AwmAes128 aes;
  uint8_t   ctr[16];
  uint64_t  buf[32];
  int       pos = 32;
public:
  typedef uint64_t result_type;
  Rng (const uint8_t key[16], uint64_t seed_value, int stream) { aes.set_key (key); seed (seed_value, stream); }
  void
  seed (uint64_t s, int stream)                      /* random.cc:116-136 */
  {
    uint8_t plain[16] = { 0 };
    for (int i = 0; i < 8; i++)
      plain[i] = uint8_t (s >> (56 - 8 * i));        /* big endian */
    plain[8] = uint8_t (stream);
    aes.encrypt_block (plain, ctr);                  /* ECB of the seed block gives the CTR start value */
    pos = 32;
  }
  uint64_t
  operator()()                                       /* random.cc:144-161: 256-byte refills, big-endian words */
  {
    if (pos == 32)
      {
        for (int blk = 0; blk < 16; blk++)
          {
            uint8_t ks[16];
            aes.encrypt_block (ctr, ks);
            for (int k = 15; k >= 0; k--)
              if (++ctr[k]) break;
            for (int w = 0; w < 2; w++)
              {
                uint64_t v = 0;
                for (int b = 0; b < 8; b++) v = (v << 8) | ks[8 * w + b];
                buf[2 * blk + w] = v;
              }
          }
        pos = 0;
      }
    return buf[pos++];
  }
  static constexpr uint64_t min() { return 0; }
  static constexpr uint64_t max() { return UINT64_MAX; }
  template<class T> void
  shuffle (vector<T>& v)                             /* random.hh:102-113 */
  {
    for (size_t i = 0; i < v.size(); i++)
      std::swap (v[i], v[i + (*this)() % (v.size() - i)]);
  }
};

/* ---- convolutional code (convcode.cc:42-125, 128-213) ---------------------------------- */
const unsigned ab_generators[12] = { 066561, 075211, 071545, 054435, 063635, 052475, 063543, 075307, 052547, 045627, 067657, 051757 };
constexpr int conv_order = 15;

vector<unsigned>
generators (int block_type)                          /* 0 a, 1 b, 2 ab; convcode.cc:77-98 */
{
  vector<unsigned> g;
  for (int i = 0; i < 12; i++)
    if (block_type == 2 || (i & 1) == block_type)
      g.push_back (ab_generators[i]);
  return g;
}
size_t code_size (int block_type, size_t msg) { return (msg + conv_order) * (block_type == 2 ? 12 : 6); }

vector<int>
conv_encode (int block_type, const vector<int>& in_bits)
{
  auto gens = generators (block_type);
  vector<int> out;
  unsigned reg = 0;
  for (size_t i = 0; i < in_bits.size() + conv_order; i++)
    {
      reg = (reg << 1) | (i < in_bits.size() ? in_bits[i] : 0);
      for (unsigned g : gens)
        out.push_back (__builtin_parity (reg & g));
    }
  return out;
}

vector<int>
conv_decode_soft (int block_type, const vector<float>& coded, float *error_out)
{
  /* full trellis over 2^15 states; delta < 0 marks "not reachable"; ties keep the first predecessor */
  auto gens = generators (block_type);
  const size_t rate = gens.size(), n_steps = coded.size() / rate;
  const unsigned n_states = 1u << conv_order, mask = n_states - 1;
  vector<float> cur (n_states, -1.f), next (n_states);
  vector<vector<uint8_t>> from_hi (n_steps, vector<uint8_t> (n_states / 8));
  vector<uint16_t> out_bits (n_states);              /* parity pattern per state */
  for (unsigned s = 0; s < n_states; s++)
    {
      unsigned pat = 0;
      for (size_t p = 0; p < rate; p++)
        pat |= unsigned (__builtin_parity (s & gens[p])) << p;
      out_bits[s] = pat;
    }
  cur[0] = 0;
  for (size_t step = 0; step < n_steps; step++)
    {
      std::fill (next.begin(), next.end(), -1.f);
      const float *c = &coded[step * rate];
      for (unsigned state = 0; state < n_states; state++)
        {
          if (!(cur[state] >= 0))
            continue;
          for (int bit = 0; bit < 2; bit++)
            {
              const unsigned ns = ((state << 1) | bit) & mask;
              float delta = cur[state];
              for (size_t p = 0; p < rate; p++)
                {
                  const float sbit = (out_bits[ns] >> p) & 1;
                  delta += (c[p] - sbit) * (c[p] - sbit);
                }
              if (delta < next[ns] || next[ns] < 0)
                {
                  next[ns] = delta;
                  if (state & (n_states >> 1))
                    from_hi[step][ns >> 3] |= 1 << (ns & 7);
                  else
                    from_hi[step][ns >> 3] &= ~(1 << (ns & 7));
                }
            }
        }
      cur.swap (next);
    }
  if (error_out)
    *error_out = cur[0] / coded.size();
  vector<int> bits (n_steps);
  unsigned state = 0;
  for (size_t step = n_steps; step-- > 0;)
    {
      bits[step] = state & 1;
      const unsigned hi = (from_hi[step][state >> 3] >> (state & 7)) & 1;
      state = (state >> 1) | (hi << (conv_order - 1));
    }
  bits.resize (n_steps - conv_order);
  return bits;
}

/* ---- key-derived tables ------------------------------------------------------------------ */
size_t data_frame_count() { return code_size (0, P::payload_size) * P::frames_per_bit; }   /* wmcommon.cc:167-171 */
size_t sync_frame_count() { return P::sync_bits * P::sync_frames_per_bit; }                /* wmcommon.cc:173-177 */
size_t block_frame_count() { return data_frame_count() + sync_frame_count(); }

void
up_down (Rng& rng, int stream, int f, int up[30], int down[30])            /* wmcommon.hh:107-122 */
{
  vector<int> bands (P::n_bands);
  for (int i = 0; i < P::n_bands; i++)
    bands[i] = P::min_band + i;
  rng.seed (f, stream);
  rng.shuffle (bands);
  for (int i = 0; i < 30; i++)
    {
      up[i] = bands[i];
      down[i] = bands[30 + i];
    }
}

vector<int>
bit_pos (const uint8_t key[16])                                            /* wmcommon.cc:143-165 */
{
  vector<int> pos (block_frame_count());
  for (size_t i = 0; i < pos.size(); i++)
    pos[i] = i;
  Rng rng (key, 0, frame_position);
  rng.shuffle (pos);
  return pos;                      /* sync_frame (f) = pos[f], data_frame (f) = pos[f + 510] */
}

struct MixEntry { int frame, up, down; };
vector<MixEntry>
mix_entries (const uint8_t key[16])                                        /* wmcommon.cc:179-202 */
{
  const auto pos = bit_pos (key);
  vector<MixEntry> e;
  Rng rng (key, 0, data_up_down);
  for (size_t f = 0; f < data_frame_count(); f++)
    {
      int up[30], down[30];
      up_down (rng, data_up_down, f, up, down);
      for (int i = 0; i < 30; i++)
        e.push_back ({ pos[f + sync_frame_count()], up[i], down[i] });
    }
  Rng mix_rng (key, 0, mix_stream);
  mix_rng.shuffle (e);
  return e;
}

vector<unsigned>
bit_order (const uint8_t key[16], size_t n)                                /* wmcommon.hh:165-185 */
{
  vector<unsigned> order (n);
  for (size_t i = 0; i < n; i++)
    order[i] = i;
  Rng rng (key, 0, bit_order_stream);
  rng.shuffle (order);
  return order;
}
template<class T> vector<T>
randomize_bit_order (const uint8_t key[16], const vector<T>& v, bool encode)
{
  const auto order = bit_order (key, v.size());
  vector<T> out (v.size());
  for (size_t i = 0; i < v.size(); i++)
    {
      if (encode) out[i] = v[order[i]];
      else        out[order[i]] = v[i];
    }
  return out;
}

vector<int>
parse_payload (const char *hex)                                            /* wmcommon.cc:210-238, utils.cc:95-112 */
{
  vector<int> bits;
  for (const char *c = hex; *c; c++)
    {
      int v;
      if (*c >= '0' && *c <= '9') v = *c - '0';
      else if (*c >= 'a' && *c <= 'f') v = *c - 'a' + 10;
      else if (*c >= 'A' && *c <= 'F') v = *c - 'A' + 10;
      else return {};
      for (int s = 3; s >= 0; s--)
        bits.push_back ((v >> s) & 1);
    }
  if (bits.empty() || bits.size() > size_t (P::payload_size))
    return {};
  vector<int> out (P::payload_size);
  for (size_t i = 0; i < out.size(); i++)
    out[i] = bits[i % bits.size()];
  return out;
}

/* frame_mod[2226][101] in {0 KEEP, 1 UP, 2 DOWN}: wmadd.cc:49-59, 86-162 */
vector<vector<uint8_t>>
frame_mod_table (const uint8_t key[16], const vector<int>& payload, int ab)
{
  vector<vector<uint8_t>> fm (block_frame_count(), vector<uint8_t> (P::max_band + 1, 0));
  const auto fec = randomize_bit_order (key, conv_encode (ab, payload), true);
  const auto pos = bit_pos (key);
  auto mark = [] (vector<uint8_t>& row, const int up[30], const int down[30], int bit) {
    for (int i = 0; i < 30; i++) row[up[i]] = bit ? 1 : 2;
    for (int i = 0; i < 30; i++) row[down[i]] = bit ? 2 : 1;
  };
  Rng sync_rng (key, 0, sync_up_down);
  for (size_t f = 0; f < sync_frame_count(); f++)                          /* mark_sync */
    {
      int up[30], down[30];
      up_down (sync_rng, sync_up_down, f, up, down);
      mark (fm[pos[f]], up, down, (f / P::sync_frames_per_bit + ab) & 1);
    }
  if (P::mix)                                                              /* mark_data */
    {
      const auto e = mix_entries (key);
      for (size_t b = 0; b < e.size(); b++)
        {
          const int bit = fec[(b / 30) / P::frames_per_bit];
          fm[e[b].frame][e[b].up] = bit ? 1 : 2;
          fm[e[b].frame][e[b].down] = bit ? 2 : 1;
        }
    }
  else
    {
      Rng data_rng (key, 0, data_up_down);
      for (size_t f = 0; f < data_frame_count(); f++)
        {
          int up[30], down[30];
          up_down (data_rng, data_up_down, f, up, down);
          mark (fm[pos[f + sync_frame_count()]], up, down, fec[f / P::frames_per_bit]);
        }
    }
  return fm;
}

struct FrameBit { int frame; int up[30], down[30]; };
vector<vector<FrameBit>>
sync_bits_table (const uint8_t key[16], bool clip)                         /* syncfinder.cc:30-77 */
{
  const auto pos = bit_pos (key);
  const int blocks = clip ? 2 : 1;
  vector<vector<FrameBit>> out;
  Rng rng (key, 0, sync_up_down);
  for (int bit = 0; bit < P::sync_bits; bit++)
    {
      vector<FrameBit> rows;
      for (int f = 0; f < P::sync_frames_per_bit; f++)
        {
          int up[30], down[30];
          up_down (rng, sync_up_down, f + bit * P::sync_frames_per_bit, up, down);
          for (int b = 0; b < blocks; b++)
            {
              FrameBit fb;
              fb.frame = pos[f + bit * P::sync_frames_per_bit] + b * int (block_frame_count());
              for (int i = 0; i < 30; i++)
                {
                  fb.up[i] = (b == 0 ? up[i] : down[i]) - P::min_band;      /* 2nd block: inverted pattern */
                  fb.down[i] = (b == 0 ? down[i] : up[i]) - P::min_band;
                }
              std::sort (fb.up, fb.up + 30);
              std::sort (fb.down, fb.down + 30);
              rows.push_back (fb);
            }
        }
      std::sort (rows.begin(), rows.end(), [] (const FrameBit& a, const FrameBit& b) { return a.frame < b.frame; });
      out.push_back (rows);
    }
  return out;
}

/* ---- FFT (fft.cc:82-118), evaluated in double, rounded once ---------------------------- */
struct FFT
{
  int n, h;
  vector<int> rev;
  vector<std::complex<double>> tw, split;
  explicit FFT (int n_) : n (n_), h (n_ / 2)
  {
    int bits = 0;
    while ((1 << bits) < h) bits++;
    rev.resize (h);
    for (int i = 0; i < h; i++)
      {
        int r = 0;
        for (int b = 0; b < bits; b++)
          if (i & (1 << b)) r |= 1 << (bits - 1 - b);
        rev[i] = r;
      }
    for (int k = 0; k < h / 2; k++)
      tw.push_back (std::polar (1.0, -2 * M_PI * k / h));
    for (int k = 0; k <= h; k++)
      split.push_back (std::polar (1.0, -2 * M_PI * k / n));
  }
  void
  cfft (vector<std::complex<double>>& a) const
  {
    for (int len = 2; len <= h; len <<= 1)
      for (int i = 0; i < h; i += len)
        for (int j = 0; j < len / 2; j++)
          {
            const auto u = a[i + j], v = a[i + j + len / 2] * tw[j * (h / len)];
            a[i + j] = u + v;
            a[i + j + len / 2] = u - v;
          }
  }
  void
  r2c (const float *in, cfloat *out) const           /* forward, sign -1, bins 0..n/2 */
  {
    vector<std::complex<double>> z (h);
    for (int i = 0; i < h; i++)
      z[rev[i]] = { in[2 * i], in[2 * i + 1] };
    cfft (z);
    for (int k = 0; k <= h; k++)
      {
        const auto zk = z[k % h], zc = std::conj (z[(h - k) % h]);
        const auto x = 0.5 * (zk + zc) + split[k] * (std::complex<double> (0, -0.5) * (zk - zc));
        out[k] = cfloat (float (x.real()), float (x.imag()));
      }
  }
  void
  c2r (const cfloat *in, float *out) const           /* backward, sign +1, unnormalised like FFTW */
  {
    vector<std::complex<double>> z (h);
    for (int k = 0; k < h; k++)
      {
        std::complex<double> xk (in[k].real(), in[k].imag()), xm (in[h - k].real(), in[h - k].imag());
        if (k == 0)
          {
            xk = { in[0].real(), 0 };
            xm = { in[h].real(), 0 };
          }
        const auto e = xk + std::conj (xm), o = (xk - std::conj (xm)) * std::conj (split[k]);
        z[rev[k]] = std::conj (e + std::complex<double> (0, 1) * o);
      }
    cfft (z);
    for (int i = 0; i < h; i++)
      {
        out[2 * i] = float (z[i].real());
        out[2 * i + 1] = float (-z[i].imag());
      }
  }
};
const FFT& fft1024() { static FFT f (1024); return f; }

vector<float>
normalized_window (size_t n)                                               /* wmcommon.cc:68-89, wmcommon.hh:187-193 */
{
  vector<float> w (n);
  double weight = 0;
  for (size_t i = 0; i < n; i++)
    {
      const double half = n / 2.0, x = (i - half) / half;
      const double v = fabs (x) > 1 ? 0 : 0.5 * cos (x * M_PI) + 0.5;
      w[i] = v;
      weight += v;
    }
  for (size_t i = 0; i < n; i++)
    w[i] *= 2.0 / weight;
  return w;
}
const vector<float>& window1024() { static vector<float> w = normalized_window (1024); return w; }

vector<float>
synth_window()                                                             /* wmadd.cc:177-206 */
{
  vector<float> w (3 * P::frame_size);
  for (size_t i = 0; i < w.size(); i++)
    {
      const double overlap = 0.1;
      double norm_pos = (double (i) - P::frame_size) / P::frame_size;
      if (norm_pos > 0.5)
        norm_pos = 1 - norm_pos;
      double tri;
      if (norm_pos < -overlap) tri = 0;
      else if (norm_pos < overlap) tri = 0.5 + norm_pos / (2 * overlap);
      else tri = 1;
      w[i] = (cos (tri * M_PI + M_PI) + 1) * 0.5;
    }
  return w;
}

/* run_fft (wmcommon.cc:91-121): window, de-interleave, r2c; out[ch][513] */
void
run_fft (const float *samples, int C, size_t start, vector<vector<cfloat>>& out)
{
  const auto& win = window1024();
  out.assign (C, vector<cfloat> (513));
  float frame[1024];
  for (int ch = 0; ch < C; ch++)
    {
      for (int x = 0; x < 1024; x++)
        frame[x] = samples[(start + x) * C + ch] * win[x];
      fft1024().r2c (frame, out[ch].data());
    }
}

inline float
db_from_complex (cfloat v)                                                 /* wmcommon.hh:204-224 */
{
  const float abs2 = v.real() * v.real() + v.imag() * v.imag();
  if (abs2 > 0)
    return log2f (abs2) * 3.01029995663981f;
  return -96;
}

/* ---- add (wmadd.cc:61-84, 215-250, 297-344, 448-618; limiter.cc:45-124) ---------------- */
static vector<float> limiter_process (const vector<float>& mixed, size_t n_frames, int C, size_t limiter_block);

/* the watermark signal alone (WatermarkGen::run, wmadd.cc:297-317, for every 1024-sample frame of `in`, zero extended to
 * total_frames frames): wm[v] for v < (total_frames - 1) * 1024 * C */
vector<float>
watermark_signal (const uint8_t key[16], const float *in, size_t n_frames, int C, const vector<int>& payload, size_t total_frames)
{
  const size_t N = P::frame_size, block = block_frame_count();
  vector<vector<uint8_t>> fm[2] = { frame_mod_table (key, payload, 0), frame_mod_table (key, payload, 1) };
  const auto swin = synth_window();
  vector<float> wm (total_frames ? (total_frames - 1) * N * C : 0, 0.f);
  vector<float> synth (3 * N * C, 0.f);                /* WatermarkSynth::synth_samples */
  vector<float> frame_in (N * C), delta (N);
  for (size_t m = 0; m < total_frames; m++)
    {
      for (size_t i = 0; i < N * C; i++)
        {
          const size_t v = m * N * C + i;
          frame_in[i] = v < n_frames * C ? in[v] : 0.f;
        }
      vector<vector<cfloat>> spect;
      run_fft (frame_in.data(), C, 0, spect);
      const size_t fnum = (2 * block - P::frames_pad_start + m) % (2 * block);      /* wmadd.cc:293-294, 326-344 */
      const vector<uint8_t>& mod = fnum >= block ? fm[1][fnum - block] : fm[0][fnum];
      std::copy (synth.begin() + N * C, synth.end(), synth.begin());
      std::fill (synth.begin() + 2 * N * C, synth.end(), 0.f);
      for (int ch = 0; ch < C; ch++)
        {
          vector<cfloat> d (513);
          for (size_t i = 0; i < mod.size(); i++)                           /* apply_frame_mod */
            {
              if (!mod[i])
                continue;
              const int sign = mod[i] == 1 ? 1 : -1;
              const float mag = std::abs (spect[ch][i]);
              if (mag > 1e-7f)
                {
                  const float mag_factor = powf (mag, -P::water_delta * sign);
                  d[i] = spect[ch][i] * (mag_factor - 1);
                }
            }
          fft1024().c2r (d.data(), delta.data());
          for (int slot = 0; slot < 3; slot++)
            for (size_t x = 0; x < N; x++)
              synth[(slot * N + x) * C + ch] += delta[x] * swin[slot * N + x];
        }
      if (m == 0)
        continue;                                      /* first call emits nothing (1 frame latency) */
      std::copy (synth.begin(), synth.begin() + N * C, wm.begin() + (m - 1) * N * C);
    }
  return wm;
}

vector<float>
add_watermark (const uint8_t key[16], const float *in, size_t n_frames, int C, const vector<int>& payload)
{
  const size_t N = P::frame_size, block = block_frame_count();

  vector<vector<uint8_t>> fm[2] = { frame_mod_table (key, payload, 0), frame_mod_table (key, payload, 1) };
  const auto swin = synth_window();
  const size_t F = (n_frames + N - 1) / N;
  /* the reference keeps feeding zero frames until everything has been written (wmadd.cc:539-546);
   * the limiter needs one more 1-second block after the last sample */
  const size_t limiter_block = P::mark_sample_rate;
  const size_t total_frames = F + 1 + (2 * limiter_block) / N + 1;
  vector<float> mixed (total_frames * N * C, 0.f);
  vector<float> synth (3 * N * C, 0.f);                /* WatermarkSynth::synth_samples */
  vector<float> frame_in (N * C), delta (N);
  for (size_t m = 0; m < total_frames; m++)
    {
      /* input frame m (zero padded) */
      for (size_t i = 0; i < N * C; i++)
        {
          const size_t v = m * N * C + i;
          frame_in[i] = v < n_frames * C ? in[v] : 0.f;
        }
      vector<vector<cfloat>> spect;
      run_fft (frame_in.data(), C, 0, spect);
      const size_t fnum = (2 * block - P::frames_pad_start + m) % (2 * block);      /* wmadd.cc:293-294, 326-344 */
      const vector<uint8_t>& mod = fnum >= block ? fm[1][fnum - block] : fm[0][fnum];
      /* WatermarkSynth::run: shift the three frame slots, add this frame's delta into all of them */
      std::copy (synth.begin() + N * C, synth.end(), synth.begin());
      std::fill (synth.begin() + 2 * N * C, synth.end(), 0.f);
      for (int ch = 0; ch < C; ch++)
        {
          vector<cfloat> d (513);
          for (size_t i = 0; i < mod.size(); i++)                           /* apply_frame_mod */
            {
              if (!mod[i])
                continue;
              const int sign = mod[i] == 1 ? 1 : -1;
              const float mag = std::abs (spect[ch][i]);
              if (mag > 1e-7f)
                {
                  const float mag_factor = powf (mag, -P::water_delta * sign);
                  d[i] = spect[ch][i] * (mag_factor - 1);
                }
            }
          fft1024().c2r (d.data(), delta.data());
          for (int slot = 0; slot < 3; slot++)
            for (size_t x = 0; x < N; x++)
              synth[(slot * N + x) * C + ch] += delta[x] * swin[slot * N + x];
        }
      if (m == 0)
        continue;                                      /* first call emits nothing (1 frame latency) */
      for (size_t i = 0; i < N * C; i++)
        {
          const size_t v = (m - 1) * N * C + i;
          const float orig = v < n_frames * C ? in[v] : 0.f;
          mixed[v] = synth[i] + orig;                  /* samples[i] += orig_samples[i], wmadd.cc:564-565 */
        }
    }
  vector<float> out (n_frames * C);
  if (P::test_no_limiter)
    {
      std::copy (mixed.begin(), mixed.begin() + n_frames * C, out.begin());
      return out;
    }
  return limiter_process (mixed, n_frames, C, limiter_block);
}

/* Limiter (limiter.cc:90-124): blocks of one second, ceiling 0.99; `mixed` extends (zero filled) at least two blocks past n_frames */
static vector<float>
limiter_process (const vector<float>& mixed, size_t n_frames, int C, size_t limiter_block)
{
  vector<float> out (n_frames * C);
  const float ceiling = 0.99;
  const size_t n_blocks = n_frames / limiter_block + 2;
  auto block_max = [&] (size_t b) {
    float m = ceiling;
    for (size_t x = b * limiter_block * C; x < (b + 1) * limiter_block * C && x < mixed.size(); x++)
      m = std::max (m, fabsf (mixed[x]));
    return m;
  };
  float last = ceiling, cur = block_max (0);
  for (size_t b = 0; b < n_blocks && b * limiter_block < n_frames; b++)
    {
      const float next = block_max (b + 1);
      const float scale_start = ceiling / std::max (last, cur);
      const float scale_end = ceiling / std::max (cur, next);
      const float scale_step = (scale_end - scale_start) / limiter_block;
      for (size_t i = 0; i < limiter_block; i++)
        {
          const float scale = scale_start + i * scale_step;
          for (int c = 0; c < C; c++)
            {
              const size_t v = (b * limiter_block + i) * C + c;
              if (v < out.size())
                out[v] = mixed[v] * scale;
            }
        }
      last = cur;
      cur = next;
    }
  return out;
}

/* ---- sync search (syncfinder.cc) --------------------------------------------------------- */
struct Wav { const float *s; size_t n_values; int C; int rate = 44100; size_t frames() const { return n_values / C / P::frame_size; } };

struct SyncCtx { size_t first = 0, last = 0; };

/* sync_fft (syncfinder.cc:560-605) */
bool
sync_fft (const Wav& w, const SyncCtx& sc, size_t index, size_t frame_count, const char *want, vector<float>& db, vector<char>& have)
{
  db.clear();
  have.clear();
  if (w.n_values < (index + frame_count * P::frame_size) * w.C)
    return false;
  db.assign (frame_count * P::n_bands, 0.f);
  have.assign (frame_count, 0);
  vector<vector<cfloat>> spect;
  for (size_t f = 0; f < frame_count; f++)
    {
      const size_t f_first = (index + f * P::frame_size) * w.C, f_last = (index + (f + 1) * P::frame_size) * w.C;
      if ((want && !want[f]) || f_last < sc.first || f_first > sc.last)
        continue;
      run_fft (w.s, w.C, index + f * P::frame_size, spect);
      for (int ch = 0; ch < w.C; ch++)
        for (int i = P::min_band; i <= P::max_band; i++)
          db[f * P::n_bands + i - P::min_band] += db_from_complex (spect[ch][i]);
      have[f] = 1;
    }
  return true;
}

double
bit_quality (float umag, float dmag, int bit)                              /* syncfinder.cc:94-114 */
{
  double raw;
  if (umag == 0 || dmag == 0) raw = 0;
  else if (umag < dmag)       raw = 1 - umag / dmag;
  else                        raw = dmag / umag - 1;
  return (bit & 1) ? raw : -raw;
}

double
sync_decode (const vector<vector<FrameBit>>& bits, size_t start, const float *db, const char *have)   /* syncfinder.cc:116-153 */
{
  double q = 0;
  int total = 0;
  for (size_t bit = 0; bit < bits.size(); bit++)
    {
      float umag = 0, dmag = 0;
      int n = 0;
      for (const auto& fb : bits[bit])
        if (have[start + fb.frame])
          {
            const size_t base = (start + fb.frame) * P::n_bands;
            for (int i = 0; i < 30; i++)
              {
                umag += db[base + fb.up[i]];
                dmag += db[base + fb.down[i]];
              }
            n++;
          }
      q += bit_quality (umag, dmag, bit) * n;
      total += n;
    }
  if (total)
    q /= total;
  return q / std::min (P::water_delta, 0.080) / 2.9;                       /* normalize_sync_quality */
}

void
parallel_for (size_t n, const std::function<void (size_t)>& fn)
{
  int nt = P::threads > 0 ? P::threads : int (std::thread::hardware_concurrency());
  nt = std::max (1, std::min<int> (nt, 16));
  if (nt == 1 || n < 2)
    {
      for (size_t i = 0; i < n; i++) fn (i);
      return;
    }
  vector<std::thread> th;
  for (int t = 0; t < nt; t++)
    th.emplace_back ([&, t] { for (size_t i = t; i < n; i += nt) fn (i); });
  for (auto& t : th)
    t.join();
}

struct SearchScore { size_t index; double raw, mean; double absq() const { return fabs (raw - mean); } };

vector<SearchScore>
search_approx (const vector<vector<FrameBit>>& bits, const Wav& w, const SyncCtx& sc, bool clip)       /* syncfinder.cc:171-256 */
{
  vector<SearchScore> scores;
  const long total = long (block_frame_count()) * (clip ? 2 : 1);
  const long n_db = long (w.frames()) - 1;                                 /* sync_fft_parallel drops the last frame (:632) */
  for (size_t shift = 0; shift < size_t (P::frame_size); shift += P::sync_search_step)
    {
      if (n_db <= 0)
        break;
      vector<float> db (size_t (n_db) * P::n_bands);
      vector<char> have (n_db);
      parallel_for ((n_db + 255) / 256, [&] (size_t job) {
        const size_t f0 = job * 256, cnt = std::min<size_t> (256, n_db - f0);
        vector<float> d;
        vector<char> h;
        sync_fft (w, sc, shift + f0 * P::frame_size, cnt, nullptr, d, h);
        std::copy (d.begin(), d.end(), db.begin() + f0 * P::n_bands);
        std::copy (h.begin(), h.end(), have.begin() + f0);
      });
      const long n_start = std::max<long> (0, n_db - total);               /* (start + total) * 81 < db.size() */
      vector<double> q (n_start);
      parallel_for ((n_start + 255) / 256, [&] (size_t job) {
        for (size_t s = job * 256; s < std::min<size_t> (n_start, (job + 1) * 256); s++)
          q[s] = sync_decode (bits, s, db.data(), have.data());
      });
      for (long s = 0; s < n_start; s++)
        scores.push_back ({ s * P::frame_size + shift, q[s], 0 });
    }
  std::sort (scores.begin(), scores.end(), [] (const SearchScore& a, const SearchScore& b) { return a.index < b.index; });
  for (int i = 0; i < int (scores.size()); i++)                            /* local mean, :234-254 */
    {
      double avg = 0;
      int n = 0;
      for (int j = -20; j <= 20; j++)
        if (std::abs (j) >= 4 && i + j >= 0 && i + j < int (scores.size()))
          {
            avg += scores[i + j].raw;
            n++;
          }
      if (n > 0)
        avg /= n;
      scores[i].mean = avg;
    }
  return scores;
}

void
select_local_maxima (vector<SearchScore>& s)                               /* syncfinder.cc:258-281 */
{
  vector<SearchScore> sel;
  for (size_t i = 0; i < s.size(); i++)
    {
      const double q = s[i].absq(), q_last = i > 0 ? s[i - 1].absq() : 0, q_next = i + 1 < s.size() ? s[i + 1].absq() : 0;
      if (q >= q_last && q >= q_next)
        {
          sel.push_back (s[i]);
          i++;
        }
    }
  s = sel;
}

void
mask_false_positives (vector<SearchScore>& s)                              /* syncfinder.cc:292-332 */
{
  const int mask_distance = 23;
  vector<SearchScore> out;
  auto sign = [] (const SearchScore& x) { return x.raw - x.mean < 0 ? -1 : 1; };
  for (int i = 0; i < int (s.size()); i++)
    {
      bool mask = false;
      for (int d = -mask_distance; d <= mask_distance; d++)
        {
          const int j = i + d;
          if (i != j && j >= 0 && j < int (s.size()))
            {
              const int distance = std::abs (int (s[i].index) - int (s[j].index)) / P::sync_search_step;
              if (distance <= mask_distance && s[j].absq() > s[i].absq() * 3 && sign (s[j]) != sign (s[i]))
                mask = true;
            }
        }
      if (!mask)
        out.push_back (s[i]);
    }
  s = out;
}

void
select_threshold_n_best (vector<SearchScore>& s, double threshold)         /* syncfinder.cc:364-383 */
{
  std::sort (s.begin(), s.end(), [] (const SearchScore& a, const SearchScore& b) { return a.absq() > b.absq(); });
  int i = 0;
  while (i < int (s.size()) && s[i].absq() > threshold)
    i++;
  if (i >= P::n_best)
    s.resize (i);
  else if (int (s.size()) > P::n_best)
    s.resize (P::n_best);
}

void
search_refine (const vector<vector<FrameBit>>& bits, const uint8_t key[16], const Wav& w, const SyncCtx& sc, bool clip,
               vector<SearchScore>& scores)                                /* syncfinder.cc:393-458 */
{
  const auto pos = bit_pos (key);
  const int block = block_frame_count(), total = block * (clip ? 2 : 1);
  vector<char> want (total, 0);
  for (size_t f = 0; f < sync_frame_count(); f++)
    {
      want[pos[f]] = 1;
      if (clip)
        want[block + pos[f]] = 1;
    }
  vector<SearchScore> refined (scores.size());
  parallel_for (scores.size(), [&] (size_t c) {
    const SearchScore& s = scores[c];
    double best_q = s.raw;
    size_t best_index = s.index;
    const int start = std::max (int (s.index) - P::sync_search_step, 0), end = int (s.index) + P::sync_search_step;
    vector<float> db;
    vector<char> have;
    for (int fine = start; fine <= end; fine += P::sync_search_fine)
      if (sync_fft (w, sc, fine, total, want.data(), db, have))
        {
          const double q = sync_decode (bits, 0, db.data(), have.data());
          if (fabs (q - s.mean) > fabs (best_q - s.mean))
            {
              best_q = q;
              best_index = fine;
            }
        }
    refined[c] = { best_index, best_q, s.mean };
  });
  std::stable_sort (refined.begin(), refined.end(), [] (const SearchScore& a, const SearchScore& b) { return a.index < b.index; });
  scores = refined;
}

struct Score { size_t index; double quality; int block_type; };

void
scan_silence (const Wav& w, SyncCtx& sc)                                   /* syncfinder.cc:155-169 */
{
  sc.first = 0;
  while (sc.first < w.n_values && w.s[sc.first] == 0)
    sc.first++;
  sc.last = w.n_values;
  while (sc.last > sc.first && w.s[sc.last - 1] == 0)
    sc.last--;
}

vector<Score>
sync_search (const uint8_t key[16], const Wav& w, bool clip, vector<SearchScore> *approx_out = nullptr)   /* syncfinder.cc:487-558 */
{
  SyncCtx sc;
  if (clip)
    scan_silence (w, sc);
  else
    sc.last = w.n_values;
  const auto bits = sync_bits_table (key, clip);
  auto scores = search_approx (bits, w, sc, clip);
  if (approx_out)
    {
      *approx_out = scores;
      return {};
    }
  select_local_maxima (scores);
  mask_false_positives (scores);
  select_threshold_n_best (scores, P::sync_threshold2 * 0.75);
  if (clip)                                                                /* sync_select_truncate_n */
    {
      std::sort (scores.begin(), scores.end(), [] (const SearchScore& a, const SearchScore& b) { return a.absq() > b.absq(); });
      const size_t n_max = std::max (P::n_best, 5);
      if (scores.size() > n_max)
        scores.resize (n_max);
    }
  search_refine (bits, key, w, sc, clip, scores);
  select_threshold_n_best (scores, P::sync_threshold2);
  std::sort (scores.begin(), scores.end(), [] (const SearchScore& a, const SearchScore& b) { return a.index < b.index; });
  vector<Score> out;
  for (const auto& s : scores)
    {
      const double q = s.raw - s.mean;
      out.push_back ({ s.index, fabs (q), q > 0 ? 0 : 1 });
    }
  return out;
}

/* ---- soft bits, block / clip decoding (wmget.cc) ----------------------------------------- */
bool
fft_range (const Wav& w, size_t start, size_t count, vector<vector<cfloat>>& out)                     /* wmcommon.cc:123-141 */
{
  out.clear();
  if (w.n_values < (start + count * P::frame_size) * w.C)
    return false;
  vector<vector<cfloat>> one;
  for (size_t f = 0; f < count; f++)
    {
      run_fft (w.s, w.C, start + f * P::frame_size, one);
      for (auto& v : one)
        out.push_back (v);
    }
  return true;
}

vector<float>
mix_decode (const vector<MixEntry>& e, const vector<vector<cfloat>>& fft_out, int C)                  /* wmget.cc:67-108 */
{
  vector<float> raw;
  double umag = 0, dmag = 0;
  const int frame_count = data_frame_count();
  for (int f = 0; f < frame_count; f++)
    {
      for (int ch = 0; ch < C; ch++)
        for (int j = 0; j < 30; j++)
          {
            const int b = f * 30 + j;
            const size_t index = e[b].frame * C + ch;
            const size_t next = (index + C) < fft_out.size() ? index + C : index - C;
            const size_t prev = (int (index) - C) >= 0 ? index - C : index + C;
            const int u = e[b].up, d = e[b].down;
            umag += db_from_complex (fft_out[index][u]);
            umag -= (db_from_complex (fft_out[prev][u]) + db_from_complex (fft_out[next][u])) * 0.5;
            dmag += db_from_complex (fft_out[index][d]);
            dmag -= (db_from_complex (fft_out[prev][d]) + db_from_complex (fft_out[next][d])) * 0.5;
          }
      if ((f % P::frames_per_bit) == (P::frames_per_bit - 1))
        {
          raw.push_back (umag - dmag);
          umag = dmag = 0;
        }
    }
  return raw;
}

vector<float>
linear_decode (const uint8_t key[16], const vector<vector<cfloat>>& fft_out, int C)                   /* wmget.cc:110-152 */
{
  const auto pos = bit_pos (key);
  Rng rng (key, 0, data_up_down);
  vector<float> raw;
  double umag = 0, dmag = 0;
  for (size_t f = 0; f < data_frame_count(); f++)
    {
      for (int ch = 0; ch < C; ch++)
        {
          const size_t index = pos[f + sync_frame_count()] * C + ch;
          const size_t next = (index + C) < fft_out.size() ? index + C : index - C;
          const size_t prev = (int (index) - C) >= 0 ? index - C : index + C;
          int up[30], down[30];
          up_down (rng, data_up_down, f, up, down);
          for (int u : up)
            {
              umag += db_from_complex (fft_out[index][u]);
              umag -= 0.5 * (db_from_complex (fft_out[prev][u]) + db_from_complex (fft_out[next][u]));
            }
          for (int d : down)
            {
              dmag += db_from_complex (fft_out[index][d]);
              dmag -= 0.5 * (db_from_complex (fft_out[prev][d]) + db_from_complex (fft_out[next][d]));
            }
        }
      if ((f % P::frames_per_bit) == size_t (P::frames_per_bit - 1))
        {
          raw.push_back (umag - dmag);
          umag = dmag = 0;
        }
    }
  return raw;
}

vector<float>
normalize_soft_bits (const vector<float>& v)                                                          /* wmget.cc:40-65 */
{
  double mean = 0;
  for (float x : v)
    mean += fabs (x);
  mean /= v.size();
  vector<float> out;
  for (float x : v)
    out.push_back (0.5 * (x / mean + 1));
  return out;
}

struct Pattern
{
  double time = 0;
  Score  score { 0, 0, 0 };
  int    type = 0;                 /* 0 BLOCK 1 CLIP 2 ALL */
  float  decode_error = 0;
  double speed = 1;
  double rating = 0;
  vector<int> bits;
};

std::string
bits_str (const vector<int>& b)
{
  std::string s;
  for (size_t p = 0; p + 3 < b.size(); p += 4)
    s += "0123456789abcdef"[b[p] * 8 + b[p + 1] * 4 + b[p + 2] * 2 + b[p + 3]];
  return s;
}

struct Decoder
{
  const uint8_t *key;
  vector<MixEntry> mix;
  explicit Decoder (const uint8_t *k) : key (k), mix (mix_entries (k)) {}

  bool
  block_bits (const Wav& w, size_t index, vector<float>& out)
  {
    vector<vector<cfloat>> spect;
    if (!fft_range (w, index, block_frame_count(), spect))
      return false;
    const auto raw = P::mix ? mix_decode (mix, spect, w.C) : linear_decode (key, spect, w.C);
    out = randomize_bit_order (key, raw, false);
    return true;
  }
  void
  add_decoded (vector<Pattern>& res, int code_type, const vector<float>& bits, double time, Score score, int type)
  {
    Pattern p;
    p.bits = conv_decode_soft (code_type, normalize_soft_bits (bits), &p.decode_error);
    p.time = time;
    p.score = score;
    p.type = type;
    if (!p.bits.empty())
      res.push_back (p);
  }
  /* BlockDecoder::run (wmget.cc:502-706) */
  void
  block_decoder (const Wav& w, vector<Pattern>& res)
  {
    struct Raw { size_t index; double quality; vector<float> bits; int block_type; };
    vector<Raw> raws;
    const size_t block_len = block_frame_count() * P::frame_size;
    for (const auto& s : sync_search (key, w, false))
      {
        Raw r { s.index, s.quality, {}, s.block_type };
        if (!block_bits (w, s.index, r.bits))
          continue;
        raws.push_back (r);
        add_decoded (res, s.block_type, r.bits, double (s.index) / w.rate, s, 0);
      }
    for (size_t i = 0; i < raws.size(); i++)                               /* AB pairs */
      if (raws[i].block_type == 1)
        {
          int best_j = -1, best_dist = P::frame_size / 2;
          for (size_t j = 0; j < i; j++)
            if (raws[j].block_type == 0)
              {
                const int dist = std::abs (int (raws[i].index - raws[j].index) - int (block_len));
                if (dist < best_dist)
                  {
                    best_j = j;
                    best_dist = dist;
                  }
              }
          if (best_j >= 0)
            {
              vector<float> ab (raws[i].bits.size() * 2);
              for (size_t k = 0; k < raws[i].bits.size(); k++)
                {
                  ab[2 * k] = raws[best_j].bits[k];
                  ab[2 * k + 1] = raws[i].bits[k];
                }
              add_decoded (res, 2, ab, double (raws[i].index) / w.rate,
                           { raws[i].index, (raws[best_j].quality + raws[i].quality) / 2, 2 }, 0);
            }
        }
    vector<size_t> best_all;                                               /* "all" pattern */
    auto sync_sum = [&] (const vector<size_t>& v) { float s = 0; for (auto b : v) s += raws[b].quality; return s; };
    for (size_t i = 0; i < raws.size(); i++)
      {
        const size_t max_block_idx = lrint (raws.back().index / double (block_len) + 0.5);
        vector<size_t> all { i };
        size_t block_idx = 1;
        while (block_idx <= max_block_idx)
          {
            const size_t expect = raws[all.back()].index + block_idx * block_len;
            int best_j = -1, best_dist = block_idx * P::frame_size / 2;
            int expect_type = raws[all.back()].block_type;
            if (block_idx & 1)
              expect_type ^= 1;
            for (size_t j = all.back(); j < raws.size(); j++)
              {
                const int dist = std::abs (int (expect) - int (raws[j].index));
                if (dist < best_dist && raws[j].block_type == expect_type)
                  {
                    best_j = j;
                    best_dist = dist;
                  }
              }
            if (best_j >= 0)
              {
                all.push_back (best_j);
                block_idx = 1;
              }
            else
              block_idx++;
          }
        if (sync_sum (all) > sync_sum (best_all))
          best_all = all;
      }
    if (best_all.size() > 1)
      {
        vector<float> all_bits (code_size (2, P::payload_size));
        int norm[2] = { 0, 0 };
        Score score { 0, 0, 0 };
        for (auto bi : best_all)
          {
            score.quality += raws[bi].quality;
            const int ab = raws[bi].block_type == 1;
            for (size_t k = 0; k < raws[bi].bits.size(); k++)
              all_bits[2 * k + ab] += raws[bi].bits[k];
            norm[ab]++;
          }
        for (size_t k = 0; k < all_bits.size(); k += 2)
          {
            all_bits[k] /= std::max (norm[0], 1);
            all_bits[k + 1] /= std::max (norm[1], 1);
          }
        score.quality /= norm[0] + norm[1];
        add_decoded (res, 2, all_bits, 0.0, score, 2);
      }
  }
  /* ClipDecoder (wmget.cc:764-884) */
  void
  clip_block (const Wav& w, vector<Pattern>& res, bool at_end)
  {
    const size_t n = (block_frame_count() + 5) * P::frame_size * w.C;
    size_t first, last, pad_start = n, pad_end = n;
    if (!at_end)
      {
        first = 0;
        last = std::min (n, w.n_values);
        if (last < n)
          pad_start += n - last;
      }
    else
      {
        if (w.n_values <= n)
          return;
        first = w.n_values - n;
        last = w.n_values;
      }
    const double time_offset = double (first) / w.rate / w.C;
    vector<float> ext (pad_start + (last - first) + pad_end, 0.f);
    std::copy (w.s + first, w.s + last, ext.begin() + pad_start);
    const Wav lw { ext.data(), ext.size(), w.C, w.rate };
    const size_t count = block_frame_count();
    for (const auto& s : sync_search (key, lw, true))
      {
        vector<float> b1, b2;
        if (!block_bits (lw, s.index, b1) || !block_bits (lw, s.index + count * P::frame_size, b2))
          continue;
        vector<float> ab;
        for (size_t k = 0; k < b1.size(); k++)
          {
            ab.push_back (s.block_type == 0 ? b1[k] : b2[k]);
            ab.push_back (s.block_type == 0 ? b2[k] : b1[k]);
          }
        Score nopad = s;
        nopad.index = time_offset * w.rate;
        add_decoded (res, 2, ab, time_offset, nopad, 1);
      }
  }
  void
  decode_plain (const Wav& w, bool first_chunk, vector<Pattern>& res, double speed)
  {
    const size_t before = res.size();
    block_decoder (w, res);
    if (first_chunk && int (w.n_values / (P::frame_size * w.C)) < int (block_frame_count()) * 3.1)
      {
        clip_block (w, res, false);
        clip_block (w, res, true);
      }
    for (size_t i = before; i < res.size(); i++)
      res[i].speed = speed;
  }
  void decode (const Wav& w, bool first_chunk, vector<Pattern>& res);     /* wmget.cc:886-939, below (needs the speed detection) */
};


/* ---- speed detection (wmspeed.cc) and the VResampler call sequences (resample.cc:29-125) --------------------------
 * zita-resampler restated (zita_restated.h): PARITY UNPINNED against the real library. */
template<class R> void
process_resampler (R& rs, const float *in, size_t in_values, float *out, size_t out_values)      /* resample.cc:29-50 */
{
  rs.out_count = unsigned (out_values / rs.nchan());
  rs.out_data = out;
  rs.inp_count = rs.inpsize() / 2 - 1;               /* "avoid timeshift": k/2 - 1 null frames before the input */
  rs.inp_data = nullptr;
  rs.process();
  rs.inp_count = unsigned (in_values / rs.nchan());
  rs.inp_data = in;
  rs.process();
  rs.inp_count = rs.inpsize() / 2;                   /* k/2 null frames after it */
  rs.inp_data = nullptr;
  rs.process();
}

bool vresampler_setup_failed = false;

vector<float>
resample_ratio_truncate (const float *in, size_t n_values, int C, int rate, double ratio, double max_in_seconds)   /* resample.cc:96-119 */
{
  if (!(ratio > 0))                                  /* (the reference computes lrint (inf) on its way to the same exit) */
    {
      vresampler_setup_failed = true;
      return {};
    }
  size_t in_values = n_values;
  if (max_in_seconds > 0)
    in_values = std::min<size_t> (in_values, C * lrint (rate * max_in_seconds));
  vector<float> out (size_t (lrint (in_values / C * ratio)) * C);
  ZitaVResampler rs;
  if (rs.setup (ratio, C, 16) != 0)
    {
      vresampler_setup_failed = true;                /* the reference prints "failed to setup vresampler" and exits (resample.cc:110-114) */
      return {};
    }
  process_resampler (rs, in, in_values, out.data(), out.size());
  return out;
}

struct SpeedScanParams { double seconds; double step; int n_steps; int n_center_steps; };
struct SpeedScore { double speed = 0, quality = 0; };

struct SpeedClip { const float *s; size_t n_values; };
SpeedClip
get_speed_clip (double location, const float *s, size_t n_values, int C, int rate, double clip_seconds)   /* wmspeed.cc:33-52 */
{
  const size_t n_frames = n_values / C;
  const double end_sec = double (n_frames) / rate;
  double start_sec = location * (end_sec - clip_seconds);
  if (start_sec < 0)
    start_sec = 0;
  const size_t start_point = start_sec * rate;
  const size_t end_point = std::min<size_t> (start_point + clip_seconds * rate, n_frames);
  return { s + start_point * C, (end_point - start_point) * C };
}

const FFT& fft512() { static FFT f (512); return f; }

/* SpeedSync (wmspeed.cc:96-395): one centre speed; the magnitude matrix is [column = sync frame][row = time step] */
struct SpeedSync
{
  struct Bit { int bit; const FrameBit *fb; };
  vector<vector<FrameBit>> table;
  vector<Bit> bits;                                  /* all 510 sync frames of a block sorted by frame */
  vector<float> umag, dmag;                          /* [col * rows + row] */
  int rows = 0;
  SpeedClip clip;
  int C, rate;
  double center;
  SpeedSync (const uint8_t key[16], SpeedClip c, int C_, int rate_, double center_) :
    table (sync_bits_table (key, false)), clip (c), C (C_), rate (rate_), center (center_)
  {
    for (size_t b = 0; b < table.size(); b++)
      for (const auto& fb : table[b])
        bits.push_back ({ int (b), &fb });
    std::sort (bits.begin(), bits.end(), [] (const Bit& a, const Bit& b) { return a.fb->frame < b.fb->frame; });
  }
  void
  prepare_mags (const SpeedScanParams& sp)                                                       /* wmspeed.cc:204-268 */
  {
    /* the clip at half the mark rate, stretched by 1 / center */
    const vector<float> sub = resample_ratio_truncate (clip.s, clip.n_values, C, rate, center / 2, sp.seconds / center);
    const size_t sub_frames = sub.size() / C;
    const int N = P::frame_size / 2, hop = P::sync_search_step / 2;
    static const vector<float> win = normalized_window (N);
    rows = 0;
    for (size_t pos = 0; pos + N < sub_frames; pos += hop)
      rows++;
    const int cols = int (bits.size());
    umag.assign (size_t (rows) * cols, 0.f);
    dmag.assign (size_t (rows) * cols, 0.f);
    parallel_for (rows, [&] (size_t row) {
      const size_t pos = row * hop;
      float db[P::n_bands] = { 0 };
      float frame[512];
      cfloat spect[257];
      for (int ch = 0; ch < C; ch++)
        {
          for (int i = 0; i < N; i++)
            frame[i] = sub[ch + (pos + i) * C] * win[i];
          fft512().r2c (frame, spect);
          for (int i = P::min_band; i <= P::max_band; i++)
            db[i - P::min_band] += db_from_complex (spect[i]);
        }
      for (int col = 0; col < cols; col++)
        {
          float u = 0, d = 0;
          for (int i = 0; i < 30; i++)
            {
              u += db[bits[col].fb->up[i]];
              d += db[bits[col].fb->down[i]];
            }
          umag[size_t (col) * rows + row] = u;
          dmag[size_t (col) * rows + row] = d;
        }
    });
  }
  /* compare + compare_bits<0..2> (wmspeed.cc:270-395).  Q16 time steps; states are the candidate block starts
   * -pad_start .. -1 (in steps of sync_search_step at the centre speed) */
  SpeedScore
  compare (double relative_speed) const
  {
    constexpr int SHIFT = 16;
    const int steps_per_frame = P::frame_size / P::sync_search_step;
    const int frames_per_block = int (block_frame_count());
    const int pad_start = frames_per_block * steps_per_frame + steps_per_frame;
    const double relative_speed_inv = 1 / relative_speed;
    struct BitValue { float umag = 0, dmag = 0; int count = 0; };
    struct State { int offset; BitValue bv[P::sync_bits]; };
    vector<State> st;
    for (int offset = -pad_start; offset < 0; offset++)
      {
        State cs {};
        cs.offset = offset * ((1 << SHIFT) / relative_speed);
        st.push_back (cs);
      }
    for (int block = 0; block < 3; block++)
      {
        size_t begin = st.size(), end = st.size();
        for (size_t mi = 0; mi < bits.size(); mi++)
          {
            /* the reference keeps this in an int; for the last frames of block 2 at speeds < 0.82 the value passes
             * INT_MAX (undefined there; with gcc on x86-64 the conversion yields INT_MIN and the wrapped sums select no
             * state).  Those frames lie more than two block lengths after every state, beyond any matrix the 25 s / 50 s
             * scans build, so "no state" is also what the exact arithmetic gives: done in 64 bits here. */
            const int64_t frame_offset = int64_t (((block * frames_per_block + bits[mi].fb->frame) * steps_per_frame * relative_speed_inv + 0.5) * (1 << SHIFT));
            while (begin > 0 && int64_t (st[begin - 1].offset) + frame_offset >= 0)
              begin--;
            while (end > 0 && ((int64_t (st[end - 1].offset) + frame_offset) >> SHIFT) >= rows)
              end--;
            for (size_t i = begin; i < end; i++)
              {
                const int64_t index = (int64_t (st[i].offset) + frame_offset) >> SHIFT;
                BitValue& bv = st[i].bv[bits[mi].bit];
                const float u = umag[mi * rows + index], d = dmag[mi * rows + index];
                if (block & 1)
                  {
                    bv.umag += d;
                    bv.dmag += u;
                  }
                else
                  {
                    bv.umag += u;
                    bv.dmag += d;
                  }
                bv.count++;
              }
          }
      }
    SpeedScore best;
    for (const auto& cs : st)
      {
        double q = 0;
        int count = 0;
        for (int bit = 0; bit < P::sync_bits; bit++)
          {
            q += bit_quality (cs.bv[bit].umag, cs.bv[bit].dmag, bit) * cs.bv[bit].count;
            count += cs.bv[bit].count;
          }
        if (count)
          {
            q /= count;
            q = fabs (q / std::min (P::water_delta, 0.080) / 2.9);          /* normalize_sync_quality */
            if (q > best.quality)
              {
                best.quality = q;
                best.speed = relative_speed * center;
              }
          }
      }
    return best;
  }
};

/* SpeedSearch::get_jobs + run_search for one key (wmspeed.cc:461-492, 683-719) */
vector<SpeedScore>
speed_scan (const uint8_t key[16], const float *s, size_t n_values, int C, int rate, double clip_location,
            const SpeedScanParams& sp, const vector<double>& speeds)
{
  const SpeedClip clip = get_speed_clip (clip_location, s, n_values, C, rate, sp.seconds * 1.3);
  vector<SpeedScore> scores;
  for (double speed : speeds)
    for (int c = -sp.n_center_steps; c <= sp.n_center_steps; c++)
      {
        const double c_speed = speed * pow (sp.step, c * (sp.n_steps * 2 + 1));
        SpeedSync sync (key, clip, C, rate, c_speed);
        sync.prepare_mags (sp);
        vector<SpeedScore> part (2 * sp.n_steps + 1);
        parallel_for (part.size(), [&] (size_t i) {
          const int p = int (i) - sp.n_steps;
          part[i] = sync.compare (pow (sp.step, p) * c_speed / c_speed);
        });
        scores.insert (scores.end(), part.begin(), part.end());
      }
  return scores;
}

void
select_n_best_scores (vector<SpeedScore>& scores, size_t n)                                       /* wmspeed.cc:494-531 */
{
  std::sort (scores.begin(), scores.end(), [] (const SpeedScore& a, const SpeedScore& b) { return a.speed < b.speed; });
  auto quality = [&] (int pos) { return pos >= 0 && size_t (pos) < scores.size() ? scores[pos].quality : 0.0; };
  vector<SpeedScore> lmax;
  for (int x = 0; size_t (x) < scores.size(); x++)
    if (quality (x - 1) <= quality (x) && quality (x) >= quality (x + 1))
      {
        lmax.push_back (scores[x]);
        x++;
      }
  std::sort (lmax.begin(), lmax.end(), [] (const SpeedScore& a, const SpeedScore& b) { return a.quality > b.quality; });
  if (lmax.size() > n)
    lmax.resize (n);
  scores = lmax;
}

double
score_smooth_find_best (vector<SpeedScore> scores, double step, double distance)                 /* wmspeed.cc:397-428 */
{
  std::sort (scores.begin(), scores.end(), [] (const SpeedScore& a, const SpeedScore& b) { return a.speed < b.speed; });
  auto window_cos = [] (double x) { return fabs (x) > 1 ? 0.0 : 0.5 * cos (x * M_PI) + 0.5; };   /* wmcommon.hh:187-193 */
  double best_speed = 0, best_quality = 0;
  for (double speed = scores.front().speed; speed < scores.back().speed; speed += 0.000001)
    {
      double sum = 0, div = 0;
      for (const auto& sc : scores)
        {
          const double w = window_cos ((sc.speed - speed) / (step * distance));
          sum += sc.quality * w;
          div += w;
        }
      sum /= div;
      if (sum > best_quality)
        {
          best_speed = speed;
          best_quality = sum;
        }
    }
  return best_speed;
}

double
best_clip_location (const uint8_t key[16], const float *s, size_t n_values, int C, int rate, double seconds, int candidates)   /* wmspeed.cc:533-577 */
{
  Rng rng (key, 0, speed_clip);
  vector<float> x;
  for (size_t p = 0; p < n_values; p += rng() % 1000)
    x.push_back (s[p]);
  uint8_t hash[20];
  awm_sha1 (x.data(), x.size() * sizeof (float), hash);                  /* Random::seed_from_hash, random.cc:184-190 */
  uint64_t seed = 0;
  for (int i = 0; i < 8; i++)
    seed = (seed << 8) | hash[i];
  rng.seed (seed, speed_clip);
  std::uniform_real_distribution<double> dist;
  double location = 0, best_energy = 0;
  for (int c = 0; c < candidates; c++)
    {
      const double loc = dist (rng);
      const SpeedClip clip = get_speed_clip (loc, s, n_values, C, rate, seconds);
      double energy = 0;
      for (size_t i = 0; i < clip.n_values; i++)
        energy += clip.s[i] * clip.s[i];
      if (energy > best_energy)
        {
          best_energy = energy;
          location = loc;
        }
    }
  return location;
}

/* detect_speed for one key (wmspeed.cc:622-781); returns true if decoding at *speed should be tried */
bool
detect_speed (const uint8_t key[16], const float *s, size_t n_values, int C, int rate, bool patient, double *speed, double *quality)
{
  if (double (n_values / C) / rate < 0.25)
    return false;
  const SpeedScanParams scan1 = patient ? SpeedScanParams { 50, 1.00035, 11, 28 } : SpeedScanParams { 25, 1.0007, 5, 28 };
  const SpeedScanParams scan2 = patient ? SpeedScanParams { 50, 1.000175, 1, 0 } : SpeedScanParams { 50, 1.00035, 1, 0 };
  const SpeedScanParams scan3 { 50, 1.00005, 40, 0 };
  const size_t n_best = patient ? 15 : 5;
  const double loc = best_clip_location (key, s, n_values, C, rate, scan1.seconds, 5);
  vector<SpeedScore> scores = speed_scan (key, s, n_values, C, rate, loc, scan1, { 1.0 });
  select_n_best_scores (scores, n_best);
  vector<double> speeds;
  for (const auto& sc : scores)
    speeds.push_back (sc.speed);
  scores = speed_scan (key, s, n_values, C, rate, loc, scan2, speeds);
  select_n_best_scores (scores, 1);
  scores = speed_scan (key, s, n_values, C, rate, loc, scan3, { scores[0].speed });
  const double best_speed = score_smooth_find_best (scores, 1 - scan3.step, 20);
  double best_quality = 0;
  for (const auto& sc : scores)
    best_quality = std::max (best_quality, sc.quality);
  *speed = best_speed;
  *quality = best_quality;
  return best_quality > 0.4 && (best_speed < 0.9999 || best_speed > 1.0001);
}

struct SpeedMode { bool detect = false, patient = false; double try_speed = -1; };
SpeedMode speed_mode;

void
Decoder::decode (const Wav& w, bool first_chunk, vector<Pattern>& res)                           /* wmget.cc:886-939 */
{
  if (speed_mode.detect || speed_mode.patient || speed_mode.try_speed > 0)
    {
      double speed = speed_mode.try_speed, quality = 0;
      bool have = speed_mode.try_speed > 0;
      if (speed_mode.detect || speed_mode.patient)
        have = detect_speed (key, w.s, w.n_values, w.C, P::mark_sample_rate, speed_mode.patient, &speed, &quality);
      if (have)
        {
          const vector<float> stretched = resample_ratio_truncate (w.s, w.n_values, w.C, P::mark_sample_rate, speed, -1);
          decode_plain ({ stretched.data(), stretched.size(), w.C, int (P::mark_sample_rate * speed) }, first_chunk, res, speed);   /* wmget.cc:916 */
        }
    }
  decode_plain (w, first_chunk, res, 1);
}

bool
approx_match (const Pattern& a, const Pattern& b)                          /* wmget.cc:178-190 */
{
  const double time_delta = P::frame_size / double (P::mark_sample_rate);
  return (fabs (a.time - b.time) < time_delta || a.type == 2) && a.bits == b.bits && a.score.block_type == b.score.block_type
      && a.type == b.type && fabs (a.speed - b.speed) < 0.01;
}

vector<Pattern>
get_watermark (const uint8_t key[16], const float *s, size_t n_values, int C)                         /* wmget.cc:971-1013 */
{
  /* chunking of WavChunkLoader (wavchunkloader.cc:54-163) */
  const size_t max_size = size_t (lrint (P::chunk_size_min * 60 * P::mark_sample_rate)) * C;
  const double block_seconds = block_frame_count() * P::frame_size / double (P::mark_sample_rate);
  const size_t overlap = size_t (lrint (2 * block_seconds * 1.3 * P::mark_sample_rate)) * C;
  Decoder dec (key);
  vector<Pattern> all;
  size_t buf_start = 0, buf_len = 0;
  double time_offset = 0;
  bool first_chunk = true, last = false;
  while (!last)
    {
      if (buf_len)
        {
          time_offset += ((buf_len - overlap) / C) / double (P::mark_sample_rate);
          buf_start += buf_len - overlap;
          buf_len = overlap;
        }
      const size_t take = std::min (max_size - buf_len, n_values - (buf_start + buf_len));
      buf_len += take;
      if (buf_len < max_size)
        {
          last = true;
          if (!buf_len)
            break;
        }
      vector<Pattern> chunk;
      dec.decode ({ s + buf_start, buf_len, C }, first_chunk, chunk);
      for (auto& p : chunk)
        p.time += time_offset;
      std::stable_sort (chunk.begin(), chunk.end(), [] (const Pattern& a, const Pattern& b) { return a.time < b.time; });
      for (const auto& p : chunk)                                          /* ResultSet::merge, wmget.cc:288-316 */
        {
          bool is_new = true;
          for (const auto& m : all)
            if (approx_match (m, p))
              is_new = false;
          if (is_new)
            all.push_back (p);
        }
      first_chunk = false;
    }
  /* ResultSet::sort (wmget.cc:215-287) */
  std::map<std::string, float> rating;
  for (const auto& p : all)
    rating[bits_str (p.bits)] += p.score.quality * (p.type == 2 ? 2.f : 1.f);
  for (auto& p : all)
    p.rating = rating[bits_str (p.bits)];
  std::sort (all.begin(), all.end(), [] (const Pattern& a, const Pattern& b) {
    const int all1 = a.type == 2, all2 = b.type == 2;
    if (a.rating != b.rating) return a.rating > b.rating;
    if (all1 != all2) return all1 < all2;
    if (a.time != b.time) return a.time < b.time;
    if (a.score.block_type != b.score.block_type) return a.score.block_type < b.score.block_type;
    return bits_str (a.bits) < bits_str (b.bits);
  });
  return all;
}

int
fill_patterns (const vector<Pattern>& v, size_t max_out, orc_pattern *out)
{
  for (size_t i = 0; i < v.size() && i < max_out; i++)
    {
      orc_pattern& o = out[i];
      o.time = v[i].time;
      o.sync_index = v[i].score.index;
      o.sync_quality = v[i].score.quality;
      o.block_type = v[i].score.block_type;
      o.type = v[i].type;
      o.decode_error = v[i].decode_error;
      o.speed = v[i].speed;
      o.n_bits = std::min<int> (v[i].bits.size(), 128);
      for (int b = 0; b < o.n_bits; b++)
        o.bits[b] = v[i].bits[b];
    }
  return int (v.size());
}

} // namespace

extern "C" {


/* ---- sample rates other than 44100 Hz ------------------------------------------------------------------------------
 * zita-resampler restated: see zita_restated.h (PARITY UNPINNED: the library is absent here). */
/* BufferedResamplerImpl (resample.cc:128-231) fed with the whole stream: hl - 1 null frames first ("avoid timeshift"),
 * the input, and -- if `trailing` (WavChunkLoader at EOF, wavchunkloader.cc:212-216) -- hl null frames. */
extern "C++" {
template<class R> static vector<float>
zita_stream_run (R& rs, const float *in, size_t n_frames, int C, bool trailing)
{
  vector<float> out;
  vector<float> chunk (size_t (P::frame_size) * C);
  auto feed = [&] (const float *data, size_t frames) {
    size_t done = 0;
    while (done < frames)
      {
        rs.out_count = P::frame_size;
        rs.out_data = chunk.data();
        const unsigned given = unsigned (std::min<size_t> (frames - done, 1u << 30));
        rs.inp_count = given;
        rs.inp_data = data ? data + done * C : nullptr;
        rs.process();
        const size_t count = P::frame_size - rs.out_count;
        out.insert (out.end(), chunk.begin(), chunk.begin() + count * C);
        done += given - rs.inp_count;
      }
  };
  /* priming: inp_count = inpsize / 2 - 1 null frames, output discarded (none is produced) */
  rs.inp_count = rs.inpsize() / 2 - 1;
  rs.inp_data = nullptr;
  rs.out_count = 1000000;
  rs.out_data = nullptr;
  rs.process();
  feed (in, n_frames);
  if (trailing)
    feed (nullptr, rs.inpsize() / 2);
  return out;
}
} /* extern "C++" */

/* ResamplerImpl::create (resample.cc:233-270): the fixed-ratio Resampler if it accepts the rates, else VResampler */
static vector<float>
zita_stream (const float *in, size_t n_frames, int C, int rate_in, int rate_out, bool trailing)
{
  ZitaResampler rs;
  if (rs.setup (rate_in, rate_out, C, 16) == 0)
    return zita_stream_run (rs, in, n_frames, C, trailing);
  ZitaVResampler vrs;
  if (vrs.setup (double (rate_out) / rate_in, C, 16) == 0)
    return zita_stream_run (vrs, in, n_frames, C, trailing);
  return {};
}

/* add_stream_watermark with a WatermarkResampler (wmadd.cc:353-430, 520-589): input resampled to 44.1 kHz, watermark
 * generated there frame by frame, resampled back and added to the original; limiter blocks of one second at the
 * input rate.  The reference keeps feeding zero frames until as many frames are written as were read. */
vector<float>
add_watermark_rate (const uint8_t key[16], const float *in, size_t n_frames, int C, const vector<int>& payload, int rate)
{
  const size_t N = P::frame_size;
  const size_t extra = 16 * N;                                     /* zero frames after the input: more than anything can reach */
  vector<float> padded ((n_frames + extra) * C, 0.f);
  std::copy (in, in + n_frames * C, padded.begin());
  const vector<float> x44 = zita_stream (padded.data(), n_frames + extra, C, rate, P::mark_sample_rate, false);
  const size_t frames44 = x44.size() / C / N;                      /* whole frames that became available */
  const vector<float> wm44 = watermark_signal (key, x44.data(), frames44 * N, C, payload, frames44);
  const vector<float> wm = zita_stream (wm44.data(), wm44.size() / C, C, P::mark_sample_rate, rate, false);
  vector<float> mixed ((n_frames + 3 * size_t (rate)) * C, 0.f);
  for (size_t v = 0; v < n_frames * C; v++)
    mixed[v] = (v < wm.size() ? wm[v] : 0.f) + in[v];               /* samples[i] += orig_samples[i] */
  if (P::test_no_limiter)
    return vector<float> (mixed.begin(), mixed.begin() + n_frames * C);
  return limiter_process (mixed, n_frames, C, size_t (rate));
}

void
orc_set_params (double water_delta, int mix, int frames_per_bit, int test_no_limiter, double sync_threshold2, int n_best, double chunk_size_min)
{
  P::water_delta = water_delta;
  P::mix = mix;
  P::frames_per_bit = frames_per_bit;
  P::test_no_limiter = test_no_limiter;
  P::sync_threshold2 = sync_threshold2;
  P::n_best = n_best;
  P::chunk_size_min = chunk_size_min;
}
void orc_set_threads (int n) { P::threads = n; }

void
orc_random_u64 (const uint8_t key[16], uint64_t seed, int stream, size_t n, uint64_t *out)
{
  Rng rng (key, seed, stream);
  for (size_t i = 0; i < n; i++) out[i] = rng();
}
void
orc_random_double (const uint8_t key[16], uint64_t seed, int stream, size_t n, double *out)
{
  Rng rng (key, seed, stream);
  std::uniform_real_distribution<double> dist;       /* random.hh:94-98 */
  for (size_t i = 0; i < n; i++) out[i] = dist (rng);
}
void
orc_gen_noise (const uint8_t key[16], size_t n_values, float *out)         /* audiowmark.cc:399-417 */
{
  Rng rng (key, 0, data_up_down);
  std::uniform_real_distribution<double> dist;
  for (size_t i = 0; i < n_values; i++) out[i] = dist (rng) * 2 - 1;
}
void
orc_up_down (const uint8_t key[16], int stream, int f, int up[30], int down[30])
{
  Rng rng (key, 0, stream);
  up_down (rng, stream, f, up, down);
}
void
orc_bit_pos (const uint8_t key[16], int *pos)
{
  const auto p = bit_pos (key);
  std::copy (p.begin(), p.end(), pos);
}
size_t
orc_mix_entries (const uint8_t key[16], int *out)
{
  const auto e = mix_entries (key);
  for (size_t i = 0; i < e.size(); i++)
    {
      out[3 * i] = e[i].frame;
      out[3 * i + 1] = e[i].up;
      out[3 * i + 2] = e[i].down;
    }
  return e.size();
}
void orc_window (size_t n, float *out) { const auto w = normalized_window (n); std::copy (w.begin(), w.end(), out); }
void orc_synth_window (float *out) { const auto w = synth_window(); std::copy (w.begin(), w.end(), out); }
size_t
orc_bit_order (const uint8_t key[16], size_t n, unsigned *order)
{
  const auto o = bit_order (key, n);
  std::copy (o.begin(), o.end(), order);
  return n;
}
size_t
orc_conv_encode (int block_type, const int *bits, size_t n, int *out)
{
  const auto r = conv_encode (block_type, vector<int> (bits, bits + n));
  std::copy (r.begin(), r.end(), out);
  return r.size();
}
size_t
orc_conv_decode_soft (int block_type, const float *coded, size_t n, int *out, float *error_out)
{
  const auto r = conv_decode_soft (block_type, vector<float> (coded, coded + n), error_out);
  std::copy (r.begin(), r.end(), out);
  return r.size();
}
int
orc_frame_mod (const uint8_t key[16], const char *payload_hex, int ab, uint8_t *out)
{
  const auto payload = parse_payload (payload_hex);
  if (payload.empty())
    return -1;
  const auto fm = frame_mod_table (key, payload, ab);
  for (size_t f = 0; f < fm.size(); f++)
    std::copy (fm[f].begin(), fm[f].end(), out + f * (P::max_band + 1));
  return int (fm.size());
}
int
orc_sync_bits (const uint8_t key[16], int clip_mode, int *out)
{
  const auto t = sync_bits_table (key, clip_mode);
  size_t o = 0;
  for (const auto& bit : t)
    for (const auto& fb : bit)
      {
        out[o++] = fb.frame;
        for (int u : fb.up) out[o++] = u;
        for (int d : fb.down) out[o++] = d;
      }
  return int (t[0].size());
}
int
orc_fft_range (const float *samples, size_t n_values, int n_channels, size_t start_index, size_t frame_count, float *out)
{
  vector<vector<cfloat>> spect;
  if (!fft_range ({ samples, n_values, n_channels }, start_index, frame_count, spect))
    return 0;
  size_t o = 0;
  for (const auto& v : spect)
    for (const auto& c : v)
      {
        out[o++] = c.real();
        out[o++] = c.imag();
      }
  return int (spect.size());
}
void
orc_ifft (size_t n, const float *spect, float *out)
{
  FFT f (n);
  vector<cfloat> in (n / 2 + 1);
  for (size_t i = 0; i < in.size(); i++)
    in[i] = cfloat (spect[2 * i], spect[2 * i + 1]);
  f.c2r (in.data(), out);
}
int
orc_add (const uint8_t key[16], const float *samples, size_t n_frames, int n_channels, int sample_rate,
         const char *payload_hex, float *out, size_t *out_frames, double *)
{
  const auto payload = parse_payload (payload_hex);
  if (payload.empty())
    return 1;
  if (sample_rate != P::mark_sample_rate)
    {
      ZitaVResampler probe;                            /* rates neither Resampler nor VResampler takes (ratio < 1/16 or > 256) */
      if (probe.setup (double (P::mark_sample_rate) / sample_rate, n_channels, 16) || probe.setup (double (sample_rate) / P::mark_sample_rate, n_channels, 16))
        return 1;
    }
  const auto r = sample_rate == P::mark_sample_rate ? add_watermark (key, samples, n_frames, n_channels, payload)
                                                    : add_watermark_rate (key, samples, n_frames, n_channels, payload, sample_rate);
  std::copy (r.begin(), r.end(), out);
  if (out_frames)
    *out_frames = n_frames;
  return 0;
}
int
orc_sync_fft (const float *samples, size_t n_values, int n_channels, size_t index, size_t frame_count,
              const char *want_frames, size_t first, size_t last, float *db_out, char *have_out)
{
  SyncCtx sc { first, last };
  vector<float> db;
  vector<char> have;
  if (!sync_fft ({ samples, n_values, n_channels }, sc, index, frame_count, want_frames, db, have))
    return 0;
  std::copy (db.begin(), db.end(), db_out);
  std::copy (have.begin(), have.end(), have_out);
  return int (have.size());
}
double
orc_sync_decode (const uint8_t key[16], int clip_mode, size_t start_frame, const float *db, size_t, const char *have, size_t)
{
  return sync_decode (sync_bits_table (key, clip_mode), start_frame, db, have);
}
int
orc_sync_search (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int clip_mode,
                 size_t max_out, uint64_t *index, double *quality, int *block_type)
{
  const auto s = sync_search (key, { samples, n_values, n_channels }, clip_mode);
  for (size_t i = 0; i < s.size() && i < max_out; i++)
    {
      index[i] = s[i].index;
      quality[i] = s[i].quality;
      block_type[i] = s[i].block_type;
    }
  return int (s.size());
}
size_t
orc_search_approx (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int clip_mode,
                   size_t max_out, uint64_t *index, double *raw_quality, double *local_mean)
{
  vector<SearchScore> s;
  sync_search (key, { samples, n_values, n_channels }, clip_mode, &s);
  for (size_t i = 0; i < s.size() && i < max_out; i++)
    {
      index[i] = s[i].index;
      raw_quality[i] = s[i].raw;
      local_mean[i] = s[i].mean;
    }
  return s.size();
}
int
orc_mix_decode (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, size_t index, float *out)
{
  vector<vector<cfloat>> spect;
  if (!fft_range ({ samples, n_values, n_channels }, index, block_frame_count(), spect))
    return 0;
  const auto r = P::mix ? mix_decode (mix_entries (key), spect, n_channels) : linear_decode (key, spect, n_channels);
  std::copy (r.begin(), r.end(), out);
  return int (r.size());
}
int
orc_decode_chunk (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int first_chunk,
                  size_t max_out, orc_pattern *out)
{
  Decoder dec (key);
  vector<Pattern> res;
  dec.decode ({ samples, n_values, n_channels }, first_chunk, res);
  std::stable_sort (res.begin(), res.end(), [] (const Pattern& a, const Pattern& b) { return a.time < b.time; });
  return fill_patterns (res, max_out, out);
}
int
orc_get (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, size_t max_out, orc_pattern *out)
{
  return fill_patterns (get_watermark (key, samples, n_values, n_channels), max_out, out);
}

/* WavChunkLoader's view of a file with another sample rate: the 44.1 kHz stream it decodes (parity unpinned, see above).
 * Returns the number of output frames (0 if zita's fixed-ratio Resampler cannot do the conversion). */
size_t
orc_resample (const float *samples, size_t n_frames, int n_channels, int rate_in, int rate_out, size_t max_out_frames, float *out)
{
  const auto r = zita_stream (samples, n_frames, n_channels, rate_in, rate_out, true);
  const size_t frames = r.size() / n_channels;
  std::copy (r.begin(), r.begin() + std::min (frames, max_out_frames) * n_channels, out);
  return frames;
}

/* ---- speed detection (wmspeed.cc) and VResampler paths; zita-resampler restated: PARITY UNPINNED ------------------ */
void
orc_set_speed_params (int detect_speed, int patient, double try_speed)
{
  speed_mode.detect = detect_speed != 0;
  speed_mode.patient = patient != 0;
  speed_mode.try_speed = try_speed;
}
size_t
orc_resample_ratio (const float *samples, size_t n_frames, int n_channels, int rate, double ratio, double max_in_seconds,
                    size_t max_out_frames, float *out)
{
  const auto r = resample_ratio_truncate (samples, n_frames * n_channels, n_channels, rate, ratio, max_in_seconds);
  const size_t frames = r.size() / n_channels;
  std::copy (r.begin(), r.begin() + std::min (frames, max_out_frames) * n_channels, out);
  return frames;
}
double
orc_speed_clip_location (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate, double seconds, int candidates)
{
  return best_clip_location (key, samples, n_values, n_channels, rate, seconds, candidates);
}
int
orc_speed_mags (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate, double clip_location,
                double center, double seconds, size_t max_rows, float *out)
{
  SpeedSync sync (key, get_speed_clip (clip_location, samples, n_values, n_channels, rate, seconds * 1.3), n_channels, rate, center);
  sync.prepare_mags ({ seconds, 0, 0, 0 });
  const size_t cols = sync.bits.size();
  for (size_t r = 0; r < size_t (sync.rows) && r < max_rows; r++)
    for (size_t c = 0; c < cols; c++)
      {
        out[(r * cols + c) * 2] = sync.umag[c * sync.rows + r];
        out[(r * cols + c) * 2 + 1] = sync.dmag[c * sync.rows + r];
      }
  return sync.rows;
}
int
orc_speed_scan (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate, double clip_location,
                double seconds, double step, int n_steps, int n_center_steps, const double *speeds, int n_speeds,
                size_t max_out, double *out_speed, double *out_quality)
{
  auto scores = speed_scan (key, samples, n_values, n_channels, rate, clip_location, { seconds, step, n_steps, n_center_steps },
                            vector<double> (speeds, speeds + n_speeds));
  std::sort (scores.begin(), scores.end(), [] (const SpeedScore& a, const SpeedScore& b) { return a.speed < b.speed; });
  for (size_t i = 0; i < scores.size() && i < max_out; i++)
    {
      out_speed[i] = scores[i].speed;
      out_quality[i] = scores[i].quality;
    }
  return int (scores.size());
}
int
orc_speed_select_n_best (double *speed, double *quality, int count, int n)
{
  vector<SpeedScore> scores (count);
  for (int i = 0; i < count; i++)
    scores[i] = { speed[i], quality[i] };
  select_n_best_scores (scores, n);
  for (size_t i = 0; i < scores.size(); i++)
    {
      speed[i] = scores[i].speed;
      quality[i] = scores[i].quality;
    }
  return int (scores.size());
}
double
orc_speed_smooth_best (const double *speed, const double *quality, int count, double step, double distance)
{
  vector<SpeedScore> scores (count);
  for (int i = 0; i < count; i++)
    scores[i] = { speed[i], quality[i] };
  return score_smooth_find_best (scores, step, distance);
}
int
orc_detect_speed (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate, int patient,
                  double *speed_out, double *quality_out)
{
  double speed = 0, quality = 0;
  vresampler_setup_failed = false;
  const bool use = detect_speed (key, samples, n_values, n_channels, rate, patient != 0, &speed, &quality);
  if (vresampler_setup_failed)                       /* e.g. digital silence: every score is 0, the second pass asks for speed 0 */
    return -1;
  if (speed_out)
    *speed_out = speed;
  if (quality_out)
    *quality_out = quality;
  return use;
}

} /* extern "C" */
