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
  std::array<int, Params::n_bands> bands;
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
  // what depends on the key alone is drawn once for both block types (A and B differ in the coded bits and the sync sequence only)
  const int n_sync = int (mark_sync_frame_count()), n_data = mark_data_frame_count();
  std::vector<UpDownArray> sync_up (n_sync), sync_down (n_sync);
  {
    UpDownGen sync_gen (key, Random::Stream::sync_up_down);
    for (int f = 0; f < n_sync; f++)
      sync_gen.get (f, sync_up[f], sync_down[f]);
  }
  std::vector<MixEntry> entries;
  std::vector<UpDownArray> data_up, data_down;
  if (params().mix)
    entries = gen_mix_entries (key);
  else
    {
      data_up.resize (n_data);
      data_down.resize (n_data);
      UpDownGen data_gen (key, Random::Stream::data_up_down);
      for (int f = 0; f < n_data; f++)
        data_gen.get (f, data_up[f], data_down[f]);
    }
  const std::vector<unsigned> order = bit_order (key, code_size (ConvBlockType::a, params().payload_size));
  for (int ab = 0; ab < 2; ab++)
    {
      int8_t *block = &table[ab * n_block * NB];
      const ConvBlockType block_type = ab ? ConvBlockType::b : ConvBlockType::a;
      const std::vector<int> fec = apply_bit_order (order, code_encode (block_type, payload_bits), /* encode */ true);

      // sync frames: always linear; A carries 010101, B carries 101010
      for (int f = 0; f < n_sync; f++)
        {
          const int bit = (f / Params::sync_frames_per_bit + ab) & 1;
          int8_t *row = block + size_t (bit_pos_gen.sync_frame (f)) * NB;
          for (size_t i = 0; i < sync_up[f].size(); i++)
            set_bands (row, sync_up[f][i], sync_down[f][i], bit);
        }
      // data frames
      if (params().mix)
        {
          for (int f = 0; f < n_data; f++)
            for (size_t j = 0; j < Params::bands_per_frame; j++)
              {
                const MixEntry& e = entries[f * Params::bands_per_frame + j];
                set_bands (block + size_t (e.frame) * NB, e.up, e.down, fec[f / params().frames_per_bit]);
              }
        }
      else
        {
          for (int f = 0; f < n_data; f++)
            {
              int8_t *row = block + size_t (bit_pos_gen.data_frame (f)) * NB;
              for (size_t i = 0; i < data_up[f].size(); i++)
                set_bands (row, data_up[f][i], data_down[f][i], fec[f / params().frames_per_bit]);
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
          // band indices minus min_band, ascending (the reference sorts them, syncfinder.cc:52-53): the 30 bands of a list are
          // distinct members of 0..80, so marking and scanning does it
          auto ascending = [] (const UpDownArray& bands) {
            std::array<uint8_t, Params::n_bands> member {};
            for (int b : bands)
              member[b - Params::min_band] = 1;
            std::array<uint8_t, 30> out {};
            int n = 0;
            for (int b = 0; b < Params::n_bands && n < 30; b++)
              if (member[b])
                out[n++] = uint8_t (b);
            return out;
          };
          const std::array<uint8_t, 30> up_sorted = ascending (up), down_sorted = ascending (down);
          for (int block = 0; block < n_blocks; block++)
            {
              Row r;
              r.frame = bit_pos_gen.sync_frame (sf) + block * block_frames;
              // the second block of a CLIP (AB) pattern carries the inverted sync sequence
              r.up = block == 0 ? up_sorted : down_sorted;
              r.down = block == 0 ? down_sorted : up_sorted;
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
