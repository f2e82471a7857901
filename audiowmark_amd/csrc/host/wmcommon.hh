// Watermark parameters and key-derived integer tables shared by add and get.
// Names follow reference src/wmcommon.hh (Params :33-89, UpDownGen :92-123, BitPosGen
// :125-133, MixEntry/gen_mix_entries :146-153, randomize_bit_order :165-185); the packed
// device tables at the end are what the HIP kernels consume (include/awm_hip.h).
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>
#include "random.hh"
#include "convcode.hh"

namespace awm {

enum class Format { AUTO = 1, RAW, RF64, WAV_PIPE };

/* The settings of the reference's Params that can change at run time (wmcommon.hh:33-89, same names and defaults).  The reference
 * keeps them as static members, i.e. one set per process; here there is one process-wide set (global_params(): what the command
 * line writes, what awm_set_params changes) and, optionally, one per context (awm_ctx_set_params).  Code reads the set in force
 * through params(): the context's own set while a thread is inside one of that context's entry points (ParamsBind), else the
 * process-wide one.  Helper threads bind what their parent had in force. */
struct ParamValues
{
  int         frames_per_bit  = 2;
  double      water_delta     = 0.01;
  std::string json_output;
  bool        strict          = false;
  bool        mix             = true;
  bool        hard            = false;
  bool        snr             = false;
  size_t      payload_size    = 128;
  double      sync_threshold2 = 0.35;
  int         get_n_best      = 8;
  double      get_chunk_size  = 30;      // minutes
  bool        detect_speed    = false;   // --detect-speed (reference wmcommon.hh:49-52)
  bool        detect_speed_patient = false;
  double      try_speed       = -1;      // --try-speed: manual speed correction
  double      test_speed      = -1;      // --test-speed: expected speed, for the detect_speed report line
  int         test_cut        = 0;
  bool        test_no_sync    = false;
  bool        test_no_limiter = false;
  int         test_truncate   = 0;
  int         expect_matches  = -1;
  Format      input_format    = Format::AUTO;
  Format      output_format   = Format::AUTO;
};
ParamValues& global_params();
ParamValues& params();
class ParamsBind           // nullptr: keep what is in force
{
  ParamValues *m_prev;
public:
  explicit ParamsBind (ParamValues *p);
  ~ParamsBind();
  ParamsBind (const ParamsBind&) = delete;
  ParamsBind& operator= (const ParamsBind&) = delete;
};

// the constants (reference wmcommon.hh:36-68)
struct Params
{
  static constexpr size_t frame_size      = 1024;
  static constexpr size_t bands_per_frame = 30;
  static constexpr int    max_band        = 100;
  static constexpr int    min_band        = 20;
  static constexpr int    n_bands         = max_band - min_band + 1;   // 81

  static constexpr int sync_bits           = 6;
  static constexpr int sync_frames_per_bit = 85;
  static constexpr int sync_search_step    = 256;
  static constexpr int sync_search_fine    = 8;

  static constexpr size_t frames_pad_start = 250;
  static constexpr int    mark_sample_rate = 44100;

  static constexpr double limiter_block_size_ms = 1000;
  static constexpr double limiter_ceiling       = 0.99;
};

size_t mark_data_frame_count();
size_t mark_sync_frame_count();
inline size_t mark_block_frame_count() { return mark_sync_frame_count() + mark_data_frame_count(); }

typedef std::array<int, Params::bands_per_frame> UpDownArray;

class UpDownGen
{
  Random::Stream m_stream;
  Random         m_random;
public:
  UpDownGen (const Key& key, Random::Stream stream) : m_stream (stream), m_random (key, 0, stream) {}
  void get (int f, UpDownArray& up, UpDownArray& down);
};

class BitPosGen
{
  std::vector<int> m_pos;
public:
  explicit BitPosGen (const Key& key);
  int sync_frame (int f) const { return m_pos[f]; }
  int data_frame (int f) const { return m_pos[f + mark_sync_frame_count()]; }
};

struct MixEntry { int frame, up, down; };
std::vector<MixEntry> gen_mix_entries (const Key& key);

std::vector<unsigned> bit_order (const Key& key, size_t n);     // the permutation behind randomize_bit_order

template<class T> std::vector<T>
apply_bit_order (const std::vector<unsigned>& order, const std::vector<T>& bit_vec, bool encode)
{
  std::vector<T> out (bit_vec.size());
  for (size_t i = 0; i < bit_vec.size(); i++)
    {
      if (encode)
        out[i] = bit_vec[order[i]];
      else
        out[order[i]] = bit_vec[i];
    }
  return out;
}

template<class T> std::vector<T>
randomize_bit_order (const Key& key, const std::vector<T>& bit_vec, bool encode)
{
  const auto order = bit_order (key, bit_vec.size());
  std::vector<T> out (bit_vec.size());
  for (size_t i = 0; i < bit_vec.size(); i++)
    {
      if (encode)
        out[i] = bit_vec[order[i]];
      else
        out[order[i]] = bit_vec[i];
    }
  return out;
}

std::vector<int> parse_payload (const std::string& str);
std::vector<float> gen_normalized_window (size_t n_values);     // reference wmcommon.cc:68-89
std::vector<float> gen_synth_window();                          // reference wmadd.cc:177-206, 3 * frame_size values

/* ---- packed tables for the device ---- */

// frame_mod rows for one (key, payload): [2 (A,B)][2226][81] with 0 KEEP, 1 UP, 2 DOWN
// (reference wmadd.cc:86-162 init_frame_mod_vec / mark_sync / mark_data)
std::vector<int8_t> build_frame_mod_table (const Key& key, const std::vector<int>& payload_bits);

// sync table (reference syncfinder.cc:30-77): for bit 0..5, rows sorted by frame:
// frame (int32) and 30 up + 30 down band indices (band - 20, ascending).
struct SyncTable
{
  int rows_per_bit = 0;              // 85 (BLOCK) or 170 (CLIP)
  std::vector<int32_t> frame;        // [6][rows]
  std::vector<uint8_t> up;           // [6][rows][30]
  std::vector<uint8_t> down;         // [6][rows][30]
};
SyncTable build_sync_table (const Key& key, bool clip_mode);

// mix table (reference wmcommon.cc:179-202, consumed by wmget.cc:67-108): per entry frame / up / down
struct MixTable
{
  std::vector<int16_t> frame;        // [51480]
  std::vector<uint8_t> up, down;     // absolute band index 20..100
};
MixTable build_mix_table (const Key& key);

} // namespace awm
