// wmdecode.hh -- pieces of the block decoder shared by the single-GPU chunk loop (wmget.cc) and the multi-GPU protocol (wmshard.cc):
// the decode job list of BlockDecoder::run (reference wmget.cc:554-701) and its device side (K7b + K8).
#pragma once
#include "wmget.hh"

namespace awm {

struct PatternRawBits      // a decoded block: where its raw soft bits live on the device
{
  size_t        index;
  double        quality;
  int           slot;        // row of ctx->ws_soft
  ConvBlockType block_type;
};

struct PendingDecode       // one Viterbi job and what to do with its result
{
  ConvBlockType      code_type;
  int                mode;            // awmk::SoftJobDev::mode
  std::vector<std::pair<int, int>> src;      // (slot, half)
  int                norm0, norm1;
  double             time;
  SyncFinder::Score  score;
  ResultSet::Type    type;
  size_t             chunk = 0;       // which chunk's ResultSet receives the pattern
  int                order_off = 0;   // bit order table of the job's key: bit_order_inv_dev + order_off (a batch with one key per clip)
};

// Every pending decode goes to the GPU in one pass: K7b builds the normalised decoder inputs from the raw soft bits that
// are already on the device, K8 decodes A, B and AB blocks side by side; only payload bits come back.
// decode_launch queues all of it on the lane's stream without waiting, decode_finish collects the payloads.
struct DecodeJob
{
  std::vector<PendingDecode> pending;
  std::vector<size_t> which[3];
  size_t nb[3] = { 0, 0, 0 }, bits_off[3] = { 0, 0, 0 }, err_off[3] = { 0, 0, 0 };
  size_t bits_total = 0, err_total = 0, n_out = 0;
  bool   launched = false;
};

int block_soft_bits_dev (awm_ctx *ctx, WorkLane *lane, KeyTables *kt, const DeviceWav& wav, const std::vector<size_t>& index,
                         std::vector<int>& slot_of, std::vector<char>& ok, const long long *slice_range = nullptr, size_t slice_frames = 0);
// raw (optional): the rows the jobs' source slots index, instead of the lane's ws_soft
int decode_launch (awm_ctx *ctx, WorkLane *lane, KeyTables *kt, DecodeJob& job, const float *raw = nullptr);
// patterns_out (optional): the decoded patterns go there as (position in job.pending, bits, error) instead of into result_sets
struct DecodedPattern { size_t pending_index; std::vector<int> bits; float error; };
int decode_finish (WorkLane *lane, const Key& key, DecodeJob& job, const std::vector<ResultSet *>& result_sets, double speed,
                   std::vector<DecodedPattern> *patterns_out = nullptr);
/* AB pairing and "all" pattern of BlockDecoder::run (reference wmget.cc:554-701) for the blocks of one chunk */
void combine_blocks (const std::vector<PatternRawBits>& pattern_raw_vec, const DeviceWav& wav, size_t chunk, std::vector<PendingDecode>& pending);

int decode_chunks_blocks_only (awm_ctx *ctx, const std::vector<Key>& key_list, const DeviceWav& wav, const std::vector<ChunkRange>& chunks,
                               std::vector<ResultSet>& chunk_sets, std::string *debug_sync_first, int lane_base = 0);

/* decode()'s speed part (reference wmget.cc:886-927) for one chunk on one lane: speed search, stretched copy, block (and, for the first
 * chunk of a stream, clip) decoder on the copy; patterns go to result_set with their speed.  Nothing happens unless speed detection is on. */
int decode_speed_chunk (awm_ctx *ctx, WorkLane *lane, ResultSet& result_set, const std::vector<Key>& key_list, const DeviceWav& chunk_wav,
                        bool first_chunk, std::string *report);

} // namespace awm
