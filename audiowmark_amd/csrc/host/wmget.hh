// Detect pipeline on resident PCM -- GPU counterpart of reference src/wmget.cc
// (BlockDecoder :492-735, ClipDecoder :764-884, decode :886-939, ResultSet :163-474,
// get_watermark chunk loop :971-1013 with the chunk arithmetic of wavchunkloader.cc:54-163).
#pragma once
#include "syncfinder.hh"

namespace awm {

class ResultSet
{
public:
  enum class Type { BLOCK, CLIP, ALL };
  struct Pattern
  {
    Key               key;
    double            time = 0;
    std::vector<int>  bit_vec;
    float             decode_error = 0;
    SyncFinder::Score sync_score { 0, 0, ConvBlockType::a };
    Type              type = Type::BLOCK;
    double            speed = 0;
    double            rating = 0;
    bool approx_match (const Pattern& p) const;
  };
  std::vector<Pattern> patterns;

  void add_pattern (const Key& key, double time, SyncFinder::Score sync_score, const std::vector<int>& bit_vec,
                    float decode_error, Type type, double speed);
  void apply_time_offset (double time_offset);
  void merge (ResultSet& other);
  void sort (const std::vector<Key>& key_list);
  void print() const;
  void print_json (size_t time_length, const std::string& json_file) const;
  int  print_match_count (const std::vector<int>& orig_bits) const;
  void set_debug_sync (const std::string& s) { m_debug_sync = s; }
  void print_debug_sync() const { printf ("%s", m_debug_sync.c_str()); }
private:
  std::string m_debug_sync;
  void rate_patterns (const Key& key);
};

// decode() of one chunk (reference wmget.cc:886-939): BlockDecoder + (first chunk) ClipDecoder
extern bool speed_print_results;   // print the "detect_speed" report line of decode() (reference wmget.cc:902: !orig_bits.empty())
int decode_chunk (awm_ctx *ctx, ResultSet& result_set, const std::vector<Key>& key_list, const DeviceWav& wav, bool first_chunk);

// chunk boundaries of WavChunkLoader for n_frames samples per channel at 44.1 kHz
struct ChunkRange { size_t first_frame, n_frames; double time_offset; };
std::vector<ChunkRange> plan_chunks (size_t n_frames, int n_channels);

// decode() for several chunks of one resident buffer (batched device work); result_sets[i] receives chunk i's patterns
int decode_chunks (awm_ctx *ctx, const std::vector<Key>& key_list, const DeviceWav& wav, const std::vector<ChunkRange>& chunks,
                   bool first_is_stream_start, std::vector<ResultSet>& result_sets);

// whole detect pass over resident PCM: chunks, merge, sort
int get_watermark_device (awm_ctx *ctx, const std::vector<Key>& key_list, const DeviceWav& wav, ResultSet& result_set);
// many independent inputs at once (one lane + host thread per input in flight); n_threads <= 0: as many as there are lanes
// clip_keys: one key per clip instead of the key list for all
int get_watermark_batch_device (awm_ctx *ctx, const std::vector<Key>& key_list, const std::vector<DeviceWav>& clips,
                                std::vector<ResultSet>& result_sets, int n_threads, const std::vector<Key> *clip_keys = nullptr);

// soft bits of whole blocks (fft_range + mix_decode), raw mix order
int block_soft_bits (awm_ctx *ctx, KeyTables *kt, const DeviceWav& wav, const std::vector<size_t>& index,
                     std::vector<std::vector<float>>& raw_bits, std::vector<char>& ok);
// batched soft Viterbi on the GPU
int viterbi_decode (awm_ctx *ctx, ConvBlockType block_type, const std::vector<std::vector<float>>& soft,
                    std::vector<std::vector<int>>& bits, std::vector<float>& errors);
std::vector<float> normalize_soft_bits (const std::vector<float>& soft_bits);

} // namespace awm
