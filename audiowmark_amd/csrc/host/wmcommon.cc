#include "wmcommon.hh"
#include "utils.hh"
#include <algorithm>
#include <cmath>

namespace awm {

namespace {
ParamValues g_params;                               // the process-wide settings (reference: the static members of Params)
thread_local ParamValues *tl_bound = nullptr;       // settings of the context whose entry point this thread is inside, if it has its own
}
ParamValues& global_params() { return g_params; }
ParamValues& params() { ParamValues *p = tl_bound; return p ? *p : g_params; }
ParamsBind::ParamsBind (ParamValues *p) : m_prev (tl_bound) { if (p) tl_bound = p; }
ParamsBind::~ParamsBind() { tl_bound = m_prev; }

size_t mark_data_frame_count() { return code_size (ConvBlockType::a, params().payload_size) * params().frames_per_bit; }
size_t mark_sync_frame_count() { return Params::sync_bits * Params::sync_frames_per_bit; }

void
UpDownGen::get (int f, UpDownArray& up, UpDownArray& down)
{
  std::vector<int> bands (Params::n_bands);
  for (int i = 0; i < Params::n_bands; i++)
    bands[i] = Params::min_band + i;
  m_random.seed (f, m_stream);            // per-frame seed
  m_random.shuffle (bands);
  for (size_t i = 0; i < Params::bands_per_frame; i++)
    {
      up[i]   = bands[i];
      down[i] = bands[Params::bands_per_frame + i];
    }
}

BitPosGen::BitPosGen (const Key& key)
{
  m_pos.resize (mark_block_frame_count());
  for (size_t i = 0; i < m_pos.size(); i++)
    m_pos[i] = i;
  Random random (key, 0, Random::Stream::frame_position);
  random.shuffle (m_pos);
}

std::vector<MixEntry>
gen_mix_entries (const Key& key)
{
  const int n_frames = mark_data_frame_count();
  std::vector<MixEntry> entries;
  entries.reserve (n_frames * Params::bands_per_frame);
  UpDownGen up_down_gen (key, Random::Stream::data_up_down);
  BitPosGen bit_pos_gen (key);
  for (int f = 0; f < n_frames; f++)
    {
      UpDownArray up, down;
      up_down_gen.get (f, up, down);
      for (size_t i = 0; i < up.size(); i++)
        entries.push_back ({ bit_pos_gen.data_frame (f), up[i], down[i] });
    }
  Random random (key, 0, Random::Stream::mix);
  random.shuffle (entries);
  return entries;
}

std::vector<unsigned>
bit_order (const Key& key, size_t n)
{
  std::vector<unsigned> order (n);
  for (size_t i = 0; i < n; i++)
    order[i] = i;
  Random random (key, 0, Random::Stream::bit_order);
  random.shuffle (order);
  return order;
}

std::vector<int>
parse_payload (const std::string& bits)
{
  auto bitvec = bit_str_to_vec (bits);
  if (bitvec.empty())
    {
      error ("audiowmark: cannot parse bits '%s'\n", bits.c_str());
      return {};
    }
  if (params().strict && bitvec.size() != params().payload_size)
    {
      error ("audiowmark: number of message bits must match payload size (%zd bits)\n", params().payload_size);
      return {};
    }
  if (bitvec.size() > params().payload_size)
    {
      error ("audiowmark: number of bits in message '%s' larger than payload size\n", bits.c_str());
      return {};
    }
  if (bitvec.size() < params().payload_size)     // repeat short messages up to the payload size
    {
      std::vector<int> expanded (params().payload_size);
      for (size_t i = 0; i < expanded.size(); i++)
        expanded[i] = bitvec[i % bitvec.size()];
      bitvec = expanded;
    }
  return bitvec;
}

std::vector<float>
gen_normalized_window (size_t n_values)
{
  // von Hann window evaluated in double, stored as float, then scaled by 2 / sum (float *= double)
  std::vector<float> window (n_values);
  const double half = n_values / 2.0;
  double weight = 0;
  for (size_t i = 0; i < n_values; i++)
    {
      const double x = (i - half) / half;
      const double w = std::fabs (x) > 1 ? 0 : 0.5 * std::cos (x * M_PI) + 0.5;
      window[i] = w;
      weight += w;
    }
  for (size_t i = 0; i < n_values; i++)
    window[i] *= 2.0 / weight;
  return window;
}

std::vector<float>
gen_synth_window()
{
  const size_t N = Params::frame_size;
  std::vector<float> window (3 * N);
  const double overlap = 0.1;
  for (size_t i = 0; i < window.size(); i++)
    {
      double pos = (double (i) - N) / N;          // -1 .. 2
      if (pos > 0.5)
        pos = 1 - pos;                            // symmetric around the centre frame
      double tri;
      if (pos < -overlap)
        tri = 0;
      else if (pos < overlap)
        tri = 0.5 + pos / (2 * overlap);
      else
        tri = 1;
      window[i] = (std::cos (tri * M_PI + M_PI) + 1) * 0.5;
    }
  return window;
}

std::vector<int8_t>
build_frame_mod_table (const Key& key, const std::vector<int>& payload_bits)
{
  enum : int8_t { KEEP = 0, UP = 1, DOWN = 2 };
  const size_t n_block = mark_block_frame_count();
  const int NB = Params::n_bands;
  std::vector<int8_t> table (2 * n_block * NB, KEEP);

  BitPosGen bit_pos_gen (key);
  auto set_bands = [&] (int8_t *row, int up_band, int down_band, int bit) {
    row[up_band - Params::min_band]   = bit ? UP : DOWN;
    row[down_band - Params::min_band] = bit ? DOWN : UP;
  };
  for (int ab = 0; ab < 2; ab++)
    {
      int8_t *block = &table[ab * n_block * NB];
      const ConvBlockType block_type = ab ? ConvBlockType::b : ConvBlockType::a;
      const std::vector<int> fec = randomize_bit_order (key, code_encode (block_type, payload_bits), /* encode */ true);

      // sync frames: always linear; A carries 010101, B carries 101010
      UpDownGen sync_gen (key, Random::Stream::sync_up_down);
      for (int f = 0; f < int (mark_sync_frame_count()); f++)
        {
          const int bit = (f / Params::sync_frames_per_bit + ab) & 1;
          UpDownArray up, down;
          sync_gen.get (f, up, down);
          int8_t *row = block + size_t (bit_pos_gen.sync_frame (f)) * NB;
          for (size_t i = 0; i < up.size(); i++)
            set_bands (row, up[i], down[i], bit);
        }
      // data frames
      const int n_data = mark_data_frame_count();
      if (params().mix)
        {
          const auto entries = gen_mix_entries (key);
          for (int f = 0; f < n_data; f++)
            for (size_t j = 0; j < Params::bands_per_frame; j++)
              {
                const MixEntry& e = entries[f * Params::bands_per_frame + j];
                set_bands (block + size_t (e.frame) * NB, e.up, e.down, fec[f / params().frames_per_bit]);
              }
        }
      else
        {
          UpDownGen data_gen (key, Random::Stream::data_up_down);
          for (int f = 0; f < n_data; f++)
            {
              UpDownArray up, down;
              data_gen.get (f, up, down);
              int8_t *row = block + size_t (bit_pos_gen.data_frame (f)) * NB;
              for (size_t i = 0; i < up.size(); i++)
                set_bands (row, up[i], down[i], fec[f / params().frames_per_bit]);
            }
        }
    }
  return table;
}

SyncTable
build_sync_table (const Key& key, bool clip_mode)
{
  const int block_frames = mark_block_frame_count();
  const int n_blocks = clip_mode ? 2 : 1;
  SyncTable t;
  t.rows_per_bit = Params::sync_frames_per_bit * n_blocks;

  UpDownGen sync_gen (key, Random::Stream::sync_up_down);
  BitPosGen bit_pos_gen (key);
  struct Row { int frame; std::array<uint8_t, 30> up, down; };
  for (int bit = 0; bit < Params::sync_bits; bit++)
    {
      std::vector<Row> rows;
      for (int f = 0; f < Params::sync_frames_per_bit; f++)
        {
          const int sf = f + bit * Params::sync_frames_per_bit;
          UpDownArray up, down;
          sync_gen.get (sf, up, down);
          for (int block = 0; block < n_blocks; block++)
            {
              Row r;
              r.frame = bit_pos_gen.sync_frame (sf) + block * block_frames;
              // the second block of a CLIP (AB) pattern carries the inverted sync sequence
              const UpDownArray& u = block == 0 ? up : down;
              const UpDownArray& d = block == 0 ? down : up;
              for (int i = 0; i < 30; i++)
                {
                  r.up[i]   = uint8_t (u[i] - Params::min_band);
                  r.down[i] = uint8_t (d[i] - Params::min_band);
                }
              std::sort (r.up.begin(), r.up.end());
              std::sort (r.down.begin(), r.down.end());
              rows.push_back (r);
            }
        }
      // frames are distinct, so ordering by frame is unambiguous (reference uses std::sort here)
      std::sort (rows.begin(), rows.end(), [] (const Row& a, const Row& b) { return a.frame < b.frame; });
      for (const Row& r : rows)
        {
          t.frame.push_back (r.frame);
          t.up.insert (t.up.end(), r.up.begin(), r.up.end());
          t.down.insert (t.down.end(), r.down.begin(), r.down.end());
        }
    }
  return t;
}

MixTable
build_mix_table (const Key& key)
{
  MixTable t;
  if (!params().mix)
    {
      // --linear (reference wmget.cc:110-152 linear_decode): data frame f uses its own 30 up / 30 down bands.  Written as the
      // entry list mix_decode consumes -- frame f's entries in band order -- the two decoders are the same computation:
      // umag and dmag are separate accumulators, so "all up terms, then all down terms" of a frame and channel is the same
      // sequence of additions per accumulator as the interleaved order.
      UpDownGen data_gen (key, Random::Stream::data_up_down);
      BitPosGen bit_pos_gen (key);
      for (int f = 0; f < int (mark_data_frame_count()); f++)
        {
          UpDownArray up, down;
          data_gen.get (f, up, down);
          for (size_t i = 0; i < up.size(); i++)
            {
              t.frame.push_back (int16_t (bit_pos_gen.data_frame (f)));
              t.up.push_back (uint8_t (up[i]));
              t.down.push_back (uint8_t (down[i]));
            }
        }
      return t;
    }
  const auto entries = gen_mix_entries (key);
  for (const auto& e : entries)
    {
      t.frame.push_back (int16_t (e.frame));
      t.up.push_back (uint8_t (e.up));
      t.down.push_back (uint8_t (e.down));
    }
  return t;
}

} // namespace awm
