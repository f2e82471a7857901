#include "wmget.hh"
#include "wmdecode.hh"
#include "wmspeed.hh"
#include <atomic>
#include <future>
#include <chrono>
#include <memory>
#include <thread>
#include "utils.hh"
#include <algorithm>
#include <cmath>
#include <map>

namespace awm {
int g_key_tables_on_device = 1;         // awm_debug_set_key_tables_on_device (capi_kernels.cc): batches with one key per clip build their tables on the device
                                        // (`get`: 1 = a group's one group ahead of its lane, 2 = those of all keys first; 0 = host threads)

/* ---- soft bits ------------------------------------------------------------------------ */

/* reference wmget.cc:40-65 */
std::vector<float>
normalize_soft_bits (const std::vector<float>& soft_bits)
{
  std::vector<float> norm;
  norm.reserve (soft_bits.size());
  if (params().hard)
    {
      for (float v : soft_bits)
        norm.push_back (v > 0 ? 1.0 : 0.0);
      return norm;
    }
  double mean = 0;
  for (float v : soft_bits)
    mean += std::fabs (v);
  mean /= soft_bits.size();
  for (float v : soft_bits)
    norm.push_back (0.5 * (v / mean + 1));
  return norm;
}

/* FFTAnalyzer::fft_range (index, 2226 frames) + mix_decode for a batch of block starts; the soft bits stay on the device:
 * block i (if ok[i]) is slot[i] of ctx->ws_soft ([slots][858] floats) */
int
block_soft_bits_dev (awm_ctx *ctx, WorkLane *lane, KeyTables *kt, const DeviceWav& wav, const std::vector<size_t>& index,
                     std::vector<int>& slot_of, std::vector<char>& ok, const long long *slice_range, size_t slice_frames)
{
  // slice_range (a row of padded clips, kernels.hh launch_clip_pad): frames in the padding are not transformed, their dB values
  // are written directly (a frame of zeros transforms to exactly -96 dB per band)
  const size_t count = mark_block_frame_count();
  const int n_bits = mark_data_frame_count() / params().frames_per_bit;
  const int C = wav.n_channels;
  slot_of.assign (index.size(), -1);
  ok.assign (index.size(), 0);
  std::vector<long long> bases;
  for (size_t i = 0; i < index.size(); i++)
    if (wav.n_values() >= (index[i] + count * Params::frame_size) * C)    // fft_range bound, reference wmcommon.cc:128-130
      {
        ok[i] = 1;
        slot_of[i] = int (bases.size());
        bases.push_back ((long long) index[i]);
      }
  if (bases.empty())
    return 0;
  hipStream_t st = lane->stream;
  const long long ld = (count + 63) & ~size_t (63);
  const long long block_stride = (long long) C * Params::n_bands * ld;
  const size_t max_batch = std::max<size_t> (1, (size_t (2) << 30) / (block_stride * sizeof (float)));
  if (int rc = lane->ws_soft.reserve (bases.size() * n_bits * sizeof (float))) return rc;
  const size_t idx_bytes = bases.size() * (sizeof (long long) + sizeof (int));
  if (int rc = lane->ws_idx.reserve (idx_bytes)) return rc;
  if (int rc = lane->pin_blocks.reserve (idx_bytes)) return rc;
  std::copy (bases.begin(), bases.end(), lane->pin_blocks.as<long long>());
  int *slice_of = reinterpret_cast<int *> (lane->pin_blocks.as<long long>() + bases.size());
  for (size_t i = 0; i < bases.size(); i++)
    slice_of[i] = slice_frames ? int (size_t (bases[i]) / slice_frames) : 0;
  AWM_HIP_CHECK (hipMemcpyAsync (lane->ws_idx.ptr, lane->pin_blocks.ptr, idx_bytes, hipMemcpyHostToDevice, st));
  const int *d_slice_of = reinterpret_cast<const int *> (lane->ws_idx.as<long long>() + bases.size());
  for (size_t b0 = 0; b0 < bases.size(); b0 += max_batch)
    {
      const size_t nb = std::min (max_batch, bases.size() - b0);
      if (int rc = lane->ws_block_db.reserve (nb * block_stride * sizeof (float))) return rc;
      awmk::SyncDbArgs da {};
      da.pcm = wav.data;
      da.n_frames = wav.n_frames;
      da.n_channels = C;
      da.per_channel = 1;
      da.stream_base = lane->ws_idx.as<long long>() + b0;
      da.count0 = int (count);
      da.n_streams = (long long) nb;
      da.hop = Params::frame_size;
      da.out = lane->ws_block_db.as<float>();
      da.out_stream_stride = block_stride;
      da.ld = ld;
      da.have = nullptr;
      da.first = 0;
      da.last = (long long) wav.n_values();      // mix_decode uses plain run_fft: no silence skipping
      if (slice_range && slice_frames)
        {
          da.stream_range = slice_range;
          da.range_index = d_slice_of + b0;
          da.range_div = 1;
          da.silent_frames_are_zero = 1;
          da.skip_unread_silent_tiles = 1;           // (K7 below gets the same ranges: it does not read what is not written)
        }
      da.tile_frames = 32;
      {
        ProfScope ps (ctx, PROF_BLOCK_DB, ((slice_range && slice_frames) ? lane->prof_live_fraction : 1.0) * double (nb) * count * C * (4096.0 + 324.0), st);
        AWM_HIP_CHECK (awmk::launch_sync_db (st, ctx->tabs, da));
      }

      awmk::SoftBitsArgs sb {};
      sb.db = lane->ws_block_db.as<float>();
      sb.block_stride = block_stride;
      sb.ld = ld;
      sb.n_channels = C;
      sb.mix_frame = kt->mix_frame.as<int16_t>();
      sb.mix_up = kt->mix_up.as<uint8_t>();
      sb.mix_down = kt->mix_down.as<uint8_t>();
      sb.n_data_frames = mark_data_frame_count();
      sb.frames_per_bit = params().frames_per_bit;
      sb.block_frames = int (count);
      sb.n_blocks = (long long) nb;
      sb.out = lane->ws_soft.as<float>() + b0 * n_bits;
      if (kt->slices && slice_frames)
        sb.block_slice = d_slice_of + b0;              // one key per clip: the mix table of the block's slice
      if (slice_range && slice_frames)
        {
          // blocks of padded slices: the items in the padding's silence are known without a load (kernels.hh SoftBitsArgs)
          sb.block_base = lane->ws_idx.as<long long>() + b0;
          sb.stream_range = slice_range;
          sb.range_index = d_slice_of + b0;
        }
      {
        ProfScope ps (ctx, PROF_SOFT_BITS, ((slice_range && slice_frames) ? lane->prof_live_fraction : 1.0) * double (nb) * count * C * 324.0, st);
        AWM_HIP_CHECK (awmk::launch_soft_bits (st, sb));
      }
    }
  return 0;
}

/* the same with the soft bits copied to the host (awm_block_soft_bits_d) */
int
block_soft_bits (awm_ctx *ctx, KeyTables *kt, const DeviceWav& wav, const std::vector<size_t>& index,
                 std::vector<std::vector<float>>& raw_bits, std::vector<char>& ok)
{
  const int n_bits = mark_data_frame_count() / params().frames_per_bit;
  std::vector<int> slot_of;
  raw_bits.assign (index.size(), {});
  if (int rc = block_soft_bits_dev (ctx, ctx, kt, wav, index, slot_of, ok))
    return rc;
  size_t n_slots = 0;
  for (int sl : slot_of)
    n_slots = std::max (n_slots, size_t (sl + 1));
  if (!n_slots)
    return 0;
  std::vector<float> host (n_slots * n_bits);
  AWM_HIP_CHECK (hipMemcpyAsync (host.data(), ctx->ws_soft.ptr, host.size() * sizeof (float), hipMemcpyDeviceToHost, ctx->stream));
  AWM_HIP_CHECK (stream_wait (ctx->stream));
  for (size_t i = 0; i < index.size(); i++)
    if (slot_of[i] >= 0)
      raw_bits[i].assign (host.begin() + size_t (slot_of[i]) * n_bits, host.begin() + size_t (slot_of[i] + 1) * n_bits);
  return 0;
}

/* conv_decode_soft for up to three batches (A, B, AB blocks) in ONE launch */
/* The smallest decode error the arithmetic can produce is -1 / coded length (a block of NaN soft bits: the end state counts as
 * unreachable, convcode.cc:144-199); -2 is the one-launch kernel's mark for "a device-side wait gave up" (hip/viterbi.hip) -- the batch
 * is void and the lane's sync block has to be cleared before its next use */
int
viterbi_check_errors (WorkLane *lane, const float *errors, size_t n)
{
  for (size_t i = 0; i < n; i++)
    if (errors[i] <= -1.5f)
      {
        if (lane->ws_viterbi_sync.ptr)
          (void) hipMemsetAsync (lane->ws_viterbi_sync.ptr, 0, lane->ws_viterbi_sync.bytes, lane->stream);
        set_error (std::string ("viterbi: a device-side wait between the workgroups of a decode timed out; ") + awmk::viterbi_form_description()
                   + " -- awm_debug_set_viterbi_persistent (0) selects the launch chain");
        return AWM_ERR_HIP;
      }
  return 0;
}

int
viterbi_decode_all (awm_ctx *ctx, const std::vector<std::vector<float>> soft[3], std::vector<std::vector<int>> bits[3],
                    std::vector<float> errors[3])
{
  size_t n_steps = 0;
  for (int t = 0; t < 3; t++)
    {
      bits[t].assign (soft[t].size(), {});
      errors[t].assign (soft[t].size(), 0.f);
      const size_t rate = t == 2 ? 12 : 6;
      for (const auto& v : soft[t])
        {
          if (v.size() % rate || (n_steps && v.size() / rate != n_steps))
            {
              set_error ("viterbi_decode: ragged batch");
              return AWM_ERR_ARG;
            }
          n_steps = v.size() / rate;
        }
    }
  if (!n_steps)
    return 0;
  hipStream_t st = ctx->stream;
  const size_t n_out = n_steps - conv_order;
  const size_t max_batch = 512;            // decodes per launch and code type
  size_t done[3] = { 0, 0, 0 };
  while (done[0] < soft[0].size() || done[1] < soft[1].size() || done[2] < soft[2].size())
    {
      size_t nb[3], in_off[3], ws_off[3], bits_off[3], err_off[3];
      size_t in_total = 0, ws_total = 0, bits_total = 0, err_total = 0;
      for (int t = 0; t < 3; t++)
        {
          const size_t rate = t == 2 ? 12 : 6;
          nb[t] = std::min (max_batch, soft[t].size() - done[t]);
          in_off[t] = in_total;    in_total += nb[t] * n_steps * rate;
          ws_off[t] = ws_total;    ws_total += awmk::viterbi_workspace_bytes (n_steps * rate, rate, nb[t]);
          bits_off[t] = bits_total; bits_total += nb[t] * n_out;
          err_off[t] = err_total;  err_total += nb[t];
        }
      std::vector<float> flat (in_total);
      for (int t = 0; t < 3; t++)
        {
          const size_t len = n_steps * (t == 2 ? 12 : 6);
          for (size_t i = 0; i < nb[t]; i++)
            std::copy (soft[t][done[t] + i].begin(), soft[t][done[t] + i].end(), flat.begin() + in_off[t] + i * len);
        }
      if (int rc = ctx->ws_viterbi_in.reserve (std::max<size_t> (1, flat.size()) * sizeof (float))) return rc;
      if (int rc = ctx->ws_viterbi.reserve (std::max<size_t> (1, ws_total))) return rc;
      if (int rc = ctx->ws_viterbi_bits.reserve (std::max<size_t> (1, bits_total) * sizeof (int))) return rc;
      if (int rc = ctx->ws_viterbi_err.reserve (std::max<size_t> (1, err_total) * sizeof (float))) return rc;
      AWM_HIP_CHECK (hipMemcpyAsync (ctx->ws_viterbi_in.ptr, flat.data(), flat.size() * sizeof (float), hipMemcpyHostToDevice, st));
      const float *d_soft[3];
      unsigned char *d_ws[3];
      int *d_bits[3];
      float *d_err[3];
      long long n_blocks[3];
      double bytes = 0;
      for (int t = 0; t < 3; t++)
        {
          d_soft[t] = ctx->ws_viterbi_in.as<float>() + in_off[t];
          d_ws[t] = ctx->ws_viterbi.as<unsigned char>() + ws_off[t];
          d_bits[t] = ctx->ws_viterbi_bits.as<int>() + bits_off[t];
          d_err[t] = ctx->ws_viterbi_err.as<float>() + err_off[t];
          n_blocks[t] = (long long) nb[t];
          bytes += double (nb[t]) * n_steps * (t == 2 ? 12 : 6) * 4.0;
        }
      unsigned int *sync_ws = ctx->viterbi_sync (nb[0] + nb[1] + nb[2]);
      if (!sync_ws)
        return AWM_ERR_HIP;
      {
        ProfScope ps (ctx, PROF_VITERBI, bytes + 2.0 * ws_total);
        AWM_HIP_CHECK (awmk::launch_viterbi (st, d_soft, n_blocks, (long long) n_steps, d_ws, d_bits, d_err, sync_ws));
      }
      std::vector<int> hbits (bits_total);
      std::vector<float> herr (err_total);
      AWM_HIP_CHECK (hipMemcpyAsync (hbits.data(), ctx->ws_viterbi_bits.ptr, hbits.size() * sizeof (int), hipMemcpyDeviceToHost, st));
      AWM_HIP_CHECK (hipMemcpyAsync (herr.data(), ctx->ws_viterbi_err.ptr, herr.size() * sizeof (float), hipMemcpyDeviceToHost, st));
      AWM_HIP_CHECK (stream_wait (st));
      if (int rc = viterbi_check_errors (ctx, herr.data(), herr.size()))
        return rc;
      for (int t = 0; t < 3; t++)
        {
          for (size_t i = 0; i < nb[t]; i++)
            {
              bits[t][done[t] + i].assign (hbits.begin() + bits_off[t] + i * n_out, hbits.begin() + bits_off[t] + (i + 1) * n_out);
              errors[t][done[t] + i] = herr[err_off[t] + i];
            }
          done[t] += nb[t];
        }
    }
  return 0;
}

int
viterbi_decode (awm_ctx *ctx, ConvBlockType block_type, const std::vector<std::vector<float>>& soft,
                std::vector<std::vector<int>>& bits, std::vector<float>& errors)
{
  std::vector<std::vector<float>> in[3];
  std::vector<std::vector<int>> out_bits[3];
  std::vector<float> out_err[3];
  in[int (block_type)] = soft;
  if (int rc = viterbi_decode_all (ctx, in, out_bits, out_err))
    return rc;
  bits = out_bits[int (block_type)];
  errors = out_err[int (block_type)];
  return 0;
}

/* ---- BlockDecoder (reference wmget.cc:492-735) ---------------------------------------- */

int
decode_launch (awm_ctx *ctx, WorkLane *lane, KeyTables *kt, DecodeJob& job, const float *raw)
{
  job.launched = false;
  for (auto& w : job.which)
    w.clear();
  if (job.pending.empty())
    return 0;
  hipStream_t st = lane->stream;
  const int n_bits = mark_data_frame_count() / params().frames_per_bit;                 // 858
  const size_t n_steps = size_t (n_bits) / 6;                                          // trellis steps (payload + 15)
  job.n_out = n_steps - conv_order;
  for (size_t i = 0; i < job.pending.size(); i++)
    job.which[int (job.pending[i].code_type)].push_back (i);
  size_t in_off[3], ws_off[3];
  size_t in_total = 0, ws_total = 0, n_jobs = 0, n_src = 0;
  job.bits_total = job.err_total = 0;
  for (int t = 0; t < 3; t++)
    {
      const size_t rate = t == 2 ? 12 : 6;
      job.nb[t] = job.which[t].size();
      in_off[t] = in_total;    in_total += job.nb[t] * n_steps * rate;
      ws_off[t] = ws_total;    ws_total += awmk::viterbi_workspace_bytes (n_steps * rate, rate, job.nb[t]);
      job.bits_off[t] = job.bits_total; job.bits_total += job.nb[t] * job.n_out;
      job.err_off[t] = job.err_total;  job.err_total += job.nb[t];
      n_jobs += job.nb[t];
      for (size_t i : job.which[t])
        n_src += job.pending[i].src.size();
    }
  // job table + source list: one page-locked block, one copy
  const size_t jobs_bytes = (n_jobs * sizeof (awmk::SoftJobDev) + 15) & ~size_t (15);
  const size_t table_bytes = jobs_bytes + n_src * sizeof (int2);
  if (int rc = lane->pin_jobs.reserve (table_bytes)) return rc;
  if (int rc = lane->ws_jobs.reserve (table_bytes)) return rc;
  auto *jobs = lane->pin_jobs.as<awmk::SoftJobDev>();
  auto *srcs = reinterpret_cast<int2 *> (lane->pin_jobs.as<char>() + jobs_bytes);
  size_t j = 0, so = 0;
  for (int t = 0; t < 3; t++)
    {
      const size_t len = n_steps * (t == 2 ? 12 : 6);
      for (size_t i = 0; i < job.nb[t]; i++)
        {
          const PendingDecode& p = job.pending[job.which[t][i]];
          jobs[j].mode = p.mode;
          jobs[j].n_src = int (p.src.size());
          jobs[j].src_off = int (so);
          jobs[j].len = int (len);
          jobs[j].norm0 = p.norm0;
          jobs[j].norm1 = p.norm1;
          jobs[j].out_off = (long long) (in_off[t] + i * len);
          jobs[j].order_off = p.order_off;
          jobs[j].pad = 0;
          for (const auto& sp : p.src)
            srcs[so++] = make_int2 (sp.first, sp.second);
          j++;
        }
    }
  if (int rc = lane->ws_viterbi_in.reserve (std::max<size_t> (1, in_total) * sizeof (float))) return rc;
  if (int rc = lane->ws_viterbi.reserve (std::max<size_t> (1, ws_total))) return rc;
  if (int rc = lane->ws_viterbi_bits.reserve (std::max<size_t> (1, job.bits_total) * sizeof (int) + job.err_total * sizeof (float))) return rc;
  if (int rc = lane->pin_bits.reserve (job.bits_total * sizeof (int) + job.err_total * sizeof (float))) return rc;
  AWM_HIP_CHECK (hipMemcpyAsync (lane->ws_jobs.ptr, lane->pin_jobs.ptr, table_bytes, hipMemcpyHostToDevice, st));
  awmk::SoftPrepArgs pa {};
  pa.raw = raw ? raw : lane->ws_soft.as<float>();
  pa.n_bits = n_bits;
  pa.inv_order = kt->bit_order_inv_dev.as<int>();
  pa.jobs = lane->ws_jobs.as<awmk::SoftJobDev>();
  pa.src = reinterpret_cast<const int2 *> (lane->ws_jobs.as<char>() + jobs_bytes);
  pa.n_jobs = (long long) n_jobs;
  pa.hard = params().hard ? 1 : 0;
  pa.out = lane->ws_viterbi_in.as<float>();
  const float *d_soft[3];
  unsigned char *d_ws[3];
  int *d_bits[3];
  float *d_err[3];
  long long n_blocks[3];
  double bytes = 0;
  float *err_base = reinterpret_cast<float *> (lane->ws_viterbi_bits.as<int>() + job.bits_total);     // bits and errors: one block, one copy back
  for (int t = 0; t < 3; t++)
    {
      d_soft[t] = lane->ws_viterbi_in.as<float>() + in_off[t];
      d_ws[t] = lane->ws_viterbi.as<unsigned char>() + ws_off[t];
      d_bits[t] = lane->ws_viterbi_bits.as<int>() + job.bits_off[t];
      d_err[t] = err_base + job.err_off[t];
      n_blocks[t] = (long long) job.nb[t];
      bytes += double (job.nb[t]) * n_steps * (t == 2 ? 12 : 6) * 4.0;
    }
  unsigned int *sync_ws = lane->viterbi_sync (n_jobs);
  if (!sync_ws)
    return AWM_ERR_HIP;
  {
    ProfScope ps (ctx, PROF_VITERBI, 2.0 * bytes + 2.0 * ws_total, st);
    AWM_HIP_CHECK (awmk::launch_soft_prep (st, pa));
    AWM_HIP_CHECK (awmk::launch_viterbi (st, d_soft, n_blocks, (long long) n_steps, d_ws, d_bits, d_err, sync_ws));
  }
  AWM_HIP_CHECK (hipMemcpyAsync (lane->pin_bits.ptr, lane->ws_viterbi_bits.ptr, job.bits_total * sizeof (int) + job.err_total * sizeof (float),
                                 hipMemcpyDeviceToHost, st));
  job.launched = true;
  return 0;
}

int
decode_finish (WorkLane *lane, const Key& key, DecodeJob& job, const std::vector<ResultSet *>& result_sets, double speed,
               std::vector<DecodedPattern> *patterns_out)
{
  if (!job.launched)
    return 0;
  AWM_HIP_CHECK (stream_wait (lane->stream));
  job.launched = false;
  const int *hbits = lane->pin_bits.as<int>();
  const float *herr = reinterpret_cast<const float *> (hbits + job.bits_total);
  if (int rc = viterbi_check_errors (lane, herr, job.err_total))
    return rc;
  std::vector<std::vector<int>> bits (job.pending.size());
  std::vector<float> errors (job.pending.size(), 0.f);
  for (int t = 0; t < 3; t++)
    for (size_t i = 0; i < job.nb[t]; i++)
      {
        const size_t pi = job.which[t][i];
        bits[pi].assign (hbits + job.bits_off[t] + i * job.n_out, hbits + job.bits_off[t] + (i + 1) * job.n_out);
        errors[pi] = herr[job.err_off[t] + i];
      }
  // patterns are added in submission order (A/B block patterns first, then AB, then "all" -- like the reference's job order)
  for (size_t i = 0; i < job.pending.size(); i++)
    {
      const PendingDecode& p = job.pending[i];
      if (bits[i].empty())
        continue;
      if (patterns_out)
        patterns_out->push_back ({ i, bits[i], errors[i] });
      else
        result_sets[p.chunk]->add_pattern (key, p.time, p.score, bits[i], errors[i], p.type, speed);
    }
  return 0;
}

int
run_pending (awm_ctx *ctx, WorkLane *lane, KeyTables *kt, const Key& key, std::vector<PendingDecode>& pending,
             const std::vector<ResultSet *>& result_sets, double speed)
{
  DecodeJob job;
  job.pending = std::move (pending);
  if (int rc = decode_launch (ctx, lane, kt, job))
    return rc;
  return decode_finish (lane, key, job, result_sets, speed);
}

/* What BlockDecoder::run (reference wmget.cc:554-701) derives from the single blocks of one chunk, as decode jobs:
 *
 *   AB   every B block whose A partner -- an A block EARLIER in the list that starts one block length before it, to within half a
 *        frame -- exists; the nearest partner, the earliest one among equally near ones.
 *   ALL  the chain of blocks with the largest quality sum.  A chain grows from its last member: the block expected `gap` block
 *        lengths further on has the other type for an odd gap and the same type for an even one; a block found within gap half-frames
 *        of the expected start is appended (nearest, earliest on ties; searched from the last member's list position on) and the
 *        gap starts again at 1, otherwise the gap grows until it would pass the end of the list.  The sums are float sums in chain
 *        order and only a strictly larger sum replaces an earlier start's chain -- both are part of the result.
 *
 * The rules are the reference's; the formulation is this file's: one nearest-match helper serves both, chains are grown per start
 * block by a function of their own. */
namespace {

constexpr size_t NO_BLOCK = size_t (-1);

struct BlockList
{
  const std::vector<PatternRawBits>& blocks;
  long long block_len;

  /* list position in [lo, hi) of the block of `type` that starts nearest to `target`, strictly closer than `reach`; NO_BLOCK if none */
  size_t
  nearest (size_t lo, size_t hi, ConvBlockType type, long long target, long long reach) const
  {
    size_t found = NO_BLOCK;
    for (size_t pos = lo; pos < hi; pos++)
      {
        if (blocks[pos].block_type != type)
          continue;
        const long long off = std::llabs ((long long) blocks[pos].index - target);
        if (off < reach)
          {
            found = pos;
            reach = off;                     // (an equally near later block does not replace it)
          }
      }
    return found;
  }

  static ConvBlockType other (ConvBlockType t) { return t == ConvBlockType::a ? ConvBlockType::b : ConvBlockType::a; }

  std::vector<size_t>
  chain_from (size_t start) const
  {
    const long long half_frame = Params::frame_size / 2;
    const long long last_gap = lrint (blocks.back().index / double (block_len) + 0.5);
    std::vector<size_t> chain { start };
    for (long long gap = 1; gap <= last_gap; )
      {
        const PatternRawBits& tail = blocks[chain.back()];
        const ConvBlockType type = (gap & 1) ? other (tail.block_type) : tail.block_type;
        const size_t next = nearest (chain.back(), blocks.size(), type, (long long) tail.index + gap * block_len, gap * half_frame);
        if (next == NO_BLOCK)
          gap++;
        else
          {
            chain.push_back (next);
            gap = 1;
          }
      }
    return chain;
  }

  float
  quality_sum (const std::vector<size_t>& chain) const
  {
    float sum = 0;
    for (size_t pos : chain)
      sum += blocks[pos].quality;
    return sum;
  }
};

}  // namespace

void
combine_blocks (const std::vector<PatternRawBits>& raw_blocks, const DeviceWav& wav, size_t chunk, std::vector<PendingDecode>& pending)
{
  const BlockList list { raw_blocks, (long long) (mark_block_frame_count() * Params::frame_size) };
  for (size_t pos = 0; pos < raw_blocks.size(); pos++)
    {
      const PatternRawBits& b = raw_blocks[pos];
      if (b.block_type != ConvBlockType::b)
        continue;
      const size_t partner = list.nearest (0, pos, ConvBlockType::a, (long long) b.index - list.block_len, Params::frame_size / 2);
      if (partner == NO_BLOCK)
        continue;
      const PatternRawBits& a = raw_blocks[partner];
      const SyncFinder::Score score { b.index, (a.quality + b.quality) / 2, ConvBlockType::ab };
      pending.push_back ({ ConvBlockType::ab, 1, { { a.slot, 0 }, { b.slot, 1 } }, 0, 0, double (b.index) / wav.sample_rate, score,
                           ResultSet::Type::BLOCK, chunk });
    }
  std::vector<size_t> best;
  float best_sum = 0;
  for (size_t start = 0; start < raw_blocks.size(); start++)
    {
      std::vector<size_t> chain = list.chain_from (start);
      const float sum = list.quality_sum (chain);
      if (sum > best_sum)
        {
          best = std::move (chain);
          best_sum = sum;
        }
    }
  if (best.size() < 2)
    return;
  // all_bits[2 k + ab] = sum over the chain's blocks of that type (list order) / their number: done by K7b (mode 2)
  PendingDecode all { ConvBlockType::ab, 2, {}, 0, 0, 0.0, { 0, 0, ConvBlockType::a }, ResultSet::Type::ALL, chunk };
  for (size_t pos : best)
    {
      const PatternRawBits& blk = raw_blocks[pos];
      const int ab = blk.block_type == ConvBlockType::b;
      all.src.push_back ({ blk.slot, ab });
      (ab ? all.norm1 : all.norm0)++;
      all.score.quality += blk.quality;
    }
  all.score.quality /= all.norm0 + all.norm1;
  pending.push_back (std::move (all));
}

int g_merge_decodes = 0;         // (debug toggle, off: the decodes of all chunks of a `get` as one batch at the end -- see block_decoder_run)
extern "C" void awm_debug_set_merge_decodes (int on) { g_merge_decodes = on; }
/* Phase offset between the chunk lanes of a `get`.  Equal chunks that start together stay in step, so their latency-bound stages
 * (candidate round trip, refinement scan, soft bits, the Viterbi chain) coincide and nothing wide runs beside them; out of step, one
 * chunk's wide kernels fill the other's narrow ones.  1: chunk i + 1 starts when chunk i's FIRST kernel (the dB matrices of the
 * approximate search, ~0.36 ms for 30 minutes) is through; 2: when its scan is through as well (~0.9 ms).  Measured, alternating in
 * one process (tools/gpu_stagger.py, profiles/r04/chunk_stagger.txt): 60 min (3 chunks on 3 lanes) 5.18 / 5.10 / 5.33 ms per add + get
 * for 0 / 1 / 2 -- the big offset serialises wide kernels that do not fill the GPU alone; 8 h (18 chunks over 4 lanes) 40.9 / 39.7 / 39.1.
 * -1 (default): 1 for streams whose chunks all start at once, 2 for streams with more chunks than lanes.  Results do not depend on it. */
int g_chunk_stagger = -1;
extern "C" void awm_debug_set_chunk_stagger (int on) { g_chunk_stagger = on; }
/* (measurement knob) get with a speed search: 1 (default) the plain decode of the chunks runs beside the speed part | 0 after it */
int g_speed_overlap = 1;
extern "C" void awm_debug_set_speed_overlap (int on) { g_speed_overlap = on; }
namespace {

/* BlockDecoder::run (reference wmget.cc:502-706) for several chunks of one resident stream at once.  Every chunk
 * is searched and combined on its own exactly like the reference does; only the device work is batched across
 * chunks (soft bits in one pass, one Viterbi launch per code type). */
int
block_decoder_run (awm_ctx *ctx, WorkLane *home, bool spread, const std::vector<Key>& key_list, const DeviceWav& stream,
                   const std::vector<ChunkRange>& chunks, const std::vector<ResultSet *>& result_sets, double speed,
                   std::string *debug_sync_first_chunk, int lane_base = 0)
{
  const size_t count = mark_block_frame_count();
  std::vector<SyncFinder::Score> first_scores;
  /* The chunks of a stream are independent until their patterns are merged, so each one runs on its own lane (stream +
   * workspaces) and up to CHUNK_LANES of them are in flight: while one chunk waits for its candidate list, runs its
   * 150-workgroup refinement scan or its chain of Viterbi rounds, the wide kernels of the others fill the machine.
   * A single chunk (short files) runs on the context's own stream. */
  // `spread` = use the context's lanes 0 .. CHUNK_LANES - 1 (the caller owns the whole context); otherwise everything
  // stays on `home` (a batch of clips runs one clip per lane, each driven by its own host thread)
  const int n_lanes = !spread ? 1 : int (std::min<size_t> (chunks.size(), size_t (std::max (1, std::min (ctx->chunk_lanes, CHUNK_LANES)))));
  std::vector<WorkLane *> lanes;
  if (!spread)
    lanes.push_back (home);
  else
    for (int i = 0; i < std::max (n_lanes, 1); i++)
      {
        // lane_base > 0: a second decoder run of the same context beside another one (the multi-GPU protocol's local chunks beside
        // its shared ones, on a host thread of their own): its own set of lanes, and the caller has made sure the PCM is complete
        WorkLane *l = ctx->lane (lane_base + i);
        if (!l)
          {
            set_error ("cannot create a work lane (stream)");
            return AWM_ERR_HIP;
          }
        lanes.push_back (l);
      }
  // awm_add_get_watermark_d: the stream is the output of the `add` this call queued, with marks "final up to sample n" on the
  // context's stream -- a chunk starts behind the mark that covers its last sample.  The context's own stream (lane 0) is behind
  // the whole add anyway: it gets the LAST of the first round of chunks.
  // The file level `get` (host/wmfile.cc) arms LIVE marks: a loader thread is still bringing the stream in (its copies and sample decodes
  // on the copy stream, a mark behind every tile) while this function starts the chunks: every lane, the context's own included, waits for
  // the mark that covers its chunk -- on the host first, until the loader has recorded it.
  ReadyMarks *marks = (spread && lane_base == 0 && lanes.size() > 1 && ctx->ready.armed && ctx->ready.base == stream.data
                       && ctx->ready.n_frames == stream.n_frames && (ctx->ready.live || !ctx->ready.marks.empty())) ? &ctx->ready : nullptr;
  if (marks && !marks->live)
    std::rotate (lanes.begin(), lanes.begin() + 1, lanes.end());
  else if (lanes.size() > 1 && lane_base == 0)
    {
      // the PCM may still be in flight on the context's stream (e.g. add -> get): the other lanes wait for it
      if (!ctx->ev_sync)
        AWM_HIP_CHECK (hipEventCreateWithFlags (&ctx->ev_sync, hipEventDisableTiming));
      AWM_HIP_CHECK (hipEventRecord (ctx->ev_sync, ctx->stream));
      for (size_t i = 1; i < lanes.size(); i++)
        AWM_HIP_CHECK (hipStreamWaitEvent (lanes[i]->stream, ctx->ev_sync, 0));
    }
  auto wait_for_mark = [&] (WorkLane *lane, size_t last_sample_plus_one) -> int {
    if (!marks)
      return 0;
    if (marks->live)
      {
        std::unique_lock<std::mutex> lock (marks->mu);
        marks->cv.wait (lock, [&] { return marks->live_done || (!marks->marks.empty() && marks->marks.back().first >= last_sample_plus_one); });
        hipEvent_t ev = nullptr;
        for (const auto& m : marks->marks)
          if (m.first >= last_sample_plus_one)
            {
              ev = m.second;
              break;
            }
        if (!ev && !marks->marks.empty())
          ev = marks->marks.back().second;                     // (the stream ended short of what was announced: the caller will start over)
        lock.unlock();
        if (ev)
          AWM_HIP_CHECK (hipStreamWaitEvent (lane->stream, ev, 0));
        return 0;
      }
    if (lane->stream == ctx->stream)
      return 0;
    for (const auto& m : marks->marks)
      if (m.first >= last_sample_plus_one)
        {
          AWM_HIP_CHECK (hipStreamWaitEvent (lane->stream, m.second, 0));
          return 0;
        }
    AWM_HIP_CHECK (hipStreamWaitEvent (lane->stream, marks->marks.back().second, 0));
    return 0;
  };
  // an error return must not leave kernels in flight on lanes whose buffers the next call will reuse
  struct LaneDrain
  {
    const std::vector<WorkLane *>& lanes;
    bool ok = false;
    ~LaneDrain()
    {
      if (!ok)
        for (WorkLane *l : lanes)
          (void) hipStreamSynchronize (l->stream);
    }
  } drain { lanes };
  auto chunk_wav = [&] (size_t c) {
    DeviceWav cw = stream;
    cw.data = stream.data + chunks[c].first_frame * stream.n_channels;
    cw.n_frames = chunks[c].n_frames;
    return cw;
  };
  /* Every chunk walks through: approximate search -> (candidate list on the host) -> refinement -> (sync positions on the
   * host) -> block dB, soft bits, Viterbi -> (bits on the host), key after key (the approximate dB matrices of a chunk are
   * computed once and shared by its keys, reference syncfinder.cc:171-256).  Chunk c lives on lane c mod n; the host polls the
   * lanes and advances whichever chunk has its device results ready.
   * (Tried: starting a chunk's approximate search only when the previous chunk's is through, so that a chunk's decode chain --
   * ~40 small dependent launches -- runs beside the wide kernels of its successors instead of beside the other chains at the end.
   * 6.13 -> 6.45 ms for the 60 min step: the wide kernels do not fill the GPU alone -- 846 scan tiles are 3.3 rounds on 256
   * CUs --, running three chunks' kernels side by side is what fills it.) */
  struct ChunkState
  {
    size_t c;
    WorkLane *lane;
    size_t ki = 0;
    int stage = 0;                       // 0: to start (next key), 1: approximate search issued, 2: refinement issued, 3: decode issued, 4: done
    std::unique_ptr<SyncFinder> finder;  // keeps the non-silent range of the chunk between the stages
    SyncFinder::SearchJob job;
    DecodeJob decode;
  };
  std::vector<ChunkState> active;
  std::vector<char> lane_busy (lanes.size(), 0);
  size_t next_chunk = 0;
  /* Alternative behind awm_debug_set_merge_decodes (off): for a stream of up to CHUNK_LANES chunks the decodes of ALL chunks run as one
   * batch per key at the end.  A chunk's ~37 Viterbi jobs are one wave per SIMD, and a single wave gets an instruction out only every
   * ~5 cycles, so the 16 dependent launches of a batch cost the same for 37 jobs as for 150: stand-alone the decoder of a 60 min step
   * takes 0.49 instead of 0.92 ms.  The STEP is 3 % slower with it (5.33 - 5.41 against 5.17 - 5.22 ms, alternating in one process,
   * profiles/r03/variants.txt): per chunk, the decode chains of the first chunks run beside the wide kernels of the others and only the
   * last chunk's chain is exposed; as one batch the whole chain starts when the last chunk has its soft bits.  (8 h: per chunk 49 ms,
   * one batch at the end 54, batches of 128 jobs on a lane of their own 61 -- the host thread that feeds all lanes waits for the batch
   * before.)  A chunk copies the raw soft bits of its blocks into a buffer of the context (its own rows, on its own lane), appends its
   * jobs to the key's list and goes on; rows that do not fit the buffer are decoded at once, per chunk. */
  const int n_soft_bits = mark_data_frame_count() / params().frames_per_bit;
  struct Deferred
  {
    DecodeJob job;
    std::vector<hipEvent_t> rows_ready;
  };
  std::vector<Deferred> deferred (key_list.size());
  size_t merge_rows = 0, merge_capacity = 0;
  if (g_merge_decodes && chunks.size() > 1 && chunks.size() <= size_t (CHUNK_LANES))
    {
      for (const auto& c : chunks)
        merge_capacity += 2 * (c.n_frames / (count * Params::frame_size) + 4);
      merge_capacity *= key_list.size();
      if (ctx->ws_merge_soft.reserve (merge_capacity * n_soft_bits * sizeof (float)))
        merge_capacity = 0;
    }
  size_t merge_events_used = 0;
  std::vector<hipEvent_t> stagger_events (chunks.size(), nullptr);
  struct EventsFree { std::vector<hipEvent_t>& ev; ~EventsFree() { for (hipEvent_t e : ev) if (e) (void) hipEventDestroy (e); } } stagger_free { stagger_events };
  auto advance = [&] (ChunkState& cs) -> int {
    const Key& key = key_list[cs.ki];
    KeyTables *kt = ctx->get_key_tables (key);
    if (!kt)
      return AWM_ERR_HIP;
    if (!cs.finder)
      cs.finder = std::make_unique<SyncFinder> (ctx, cs.lane);
    SyncFinder& finder = *cs.finder;
    const size_t c = cs.c;
    switch (cs.stage)
      {
      case 0:
        {
          if (cs.ki == 0)
            if (int rc = wait_for_mark (cs.lane, chunks[c].first_frame + chunks[c].n_frames))
              return rc;
          const int stagger = g_chunk_stagger >= 0 ? g_chunk_stagger : (chunks.size() > lanes.size() ? 2 : 1);
          if (stagger && cs.ki == 0 && spread)
            {
              // start behind the previous chunk's dB kernel; leave a mark behind my own
              if (c > 0 && stagger_events[c - 1])
                AWM_HIP_CHECK (hipStreamWaitEvent (cs.lane->stream, stagger_events[c - 1], 0));
              if (c + 1 < chunks.size())
                {
                  AWM_HIP_CHECK (hipEventCreateWithFlags (&stagger_events[c], hipEventDisableTiming));
                  (stagger == 2 ? finder.after_scan_event : finder.after_db_event) = stagger_events[c];
                }
            }
          // (ki > 0: the dB matrices of this chunk are still in the lane's workspace from the previous key: nothing else writes there)
          if (int rc = finder.approx_launch (key, chunk_wav (c), SyncFinder::Mode::BLOCK, cs.job, /* prepared */ false, /* db_ready */ cs.ki > 0))
            return rc;
          finder.after_db_event = finder.after_scan_event = nullptr;
          cs.stage = 1;
          return 0;
        }
      case 1:
        if (int rc = finder.select_refine (cs.job))
          return rc;
        cs.stage = 2;
        return 0;
      case 2:
        {
          std::vector<SyncFinder::Score> scores;
          if (int rc = finder.search_finish (cs.job, scores))
            return rc;
          if (cs.ki == 0 && c == 0)
            first_scores = scores;
          // blocks of this chunk: fft_range refuses blocks that run past the end OF THE CHUNK (reference wmcommon.cc:128-130)
          std::vector<size_t> wanted;
          std::vector<const SyncFinder::Score *> wanted_score;
          for (const auto& sc : scores)
            if (chunks[c].n_frames >= sc.index + count * Params::frame_size)
              {
                wanted.push_back (chunks[c].first_frame + sc.index);
                wanted_score.push_back (&sc);
              }
          std::vector<int> slot_of;
          std::vector<char> ok;
          if (int rc = block_soft_bits_dev (ctx, cs.lane, kt, stream, wanted, slot_of, ok))
            return rc;
          std::vector<PatternRawBits> pattern_raw_vec;
          cs.decode = DecodeJob();
          auto& pending = cs.decode.pending;
          for (size_t w = 0; w < wanted.size(); w++)
            {
              if (!ok[w])
                continue;
              const SyncFinder::Score& sc = *wanted_score[w];
              PatternRawBits rb;
              rb.index = sc.index;
              rb.quality = sc.quality;
              rb.slot = slot_of[w];
              rb.block_type = sc.block_type;
              pending.push_back ({ rb.block_type, 0, { { rb.slot, 0 } }, 0, 0, double (rb.index) / stream.sample_rate,
                                   sc, ResultSet::Type::BLOCK, c });
              pattern_raw_vec.push_back (rb);
            }
          combine_blocks (pattern_raw_vec, stream, c, pending);
          size_t n_rows = 0;
          for (size_t w = 0; w < wanted.size(); w++)
            if (ok[w])
              n_rows = std::max (n_rows, size_t (slot_of[w]) + 1);
          if (merge_rows + n_rows <= merge_capacity && !pending.empty())
            {
              // rows -> the context's buffer (this lane's stream), jobs -> the key's list with their source rows renumbered
              hipStream_t st = cs.lane->stream;
              AWM_HIP_CHECK (hipMemcpyAsync (ctx->ws_merge_soft.as<float>() + merge_rows * n_soft_bits, cs.lane->ws_soft.ptr,
                                             n_rows * n_soft_bits * sizeof (float), hipMemcpyDeviceToDevice, st));
              if (merge_events_used == ctx->merge_events.size())
                {
                  hipEvent_t e = nullptr;
                  AWM_HIP_CHECK (hipEventCreateWithFlags (&e, hipEventDisableTiming));
                  ctx->merge_events.push_back (e);
                }
              hipEvent_t ev = ctx->merge_events[merge_events_used++];
              AWM_HIP_CHECK (hipEventRecord (ev, st));
              Deferred& d = deferred[cs.ki];
              d.rows_ready.push_back (ev);
              for (auto& p : pending)
                {
                  for (auto& sp : p.src)
                    sp.first += int (merge_rows);
                  d.job.pending.push_back (std::move (p));
                }
              merge_rows += n_rows;
              cs.decode = DecodeJob();
              cs.ki++;
              cs.stage = cs.ki < key_list.size() ? 0 : 4;
              return 0;
            }
          if (int rc = decode_launch (ctx, cs.lane, kt, cs.decode))
            return rc;
          cs.stage = 3;
          return 0;
        }
      default:
        if (int rc = decode_finish (cs.lane, key, cs.decode, result_sets, speed))
          return rc;
        cs.ki++;
        cs.stage = cs.ki < key_list.size() ? 0 : 4;
        return 0;
      }
  };
  while (next_chunk < chunks.size() || !active.empty())
    {
      bool progressed = false;
      // start chunks on free lanes, in order (with live marks: once the loader has brought the chunk's last sample in -- until then
      // the chunks that are running keep being served)
      auto chunk_is_loaded = [&] (size_t c) {
        if (!marks || !marks->live)
          return true;
        std::lock_guard<std::mutex> lock (marks->mu);
        return marks->live_done || (!marks->marks.empty() && marks->marks.back().first >= chunks[c].first_frame + chunks[c].n_frames);
      };
      while (next_chunk < chunks.size() && !lane_busy[next_chunk % lanes.size()] && !key_list.empty() && chunk_is_loaded (next_chunk))
        {
          ChunkState cs;
          cs.c = next_chunk;
          cs.lane = lanes[next_chunk % lanes.size()];
          lane_busy[next_chunk % lanes.size()] = 1;
          active.push_back (std::move (cs));
          if (int rc = advance (active.back()))
            return rc;
          next_chunk++;
          progressed = true;
        }
      if (key_list.empty())
        break;
      for (auto& cs : active)
        {
          // a stage that waits for the device is entered when the lane has run dry (its results are in page-locked memory then);
          // with a single chunk in flight there is nothing else to do: let the stage itself wait
          // (any answer other than "not ready" lets the stage run: its own wait reports a failed stream)
          if (cs.stage >= 1 && cs.stage <= 3 && (active.size() == 1 || hipStreamQuery (cs.lane->stream) != hipErrorNotReady))
            {
              if (int rc = advance (cs))
                return rc;
              if (cs.stage == 0)                   // next key of the same chunk
                if (int rc = advance (cs))
                  return rc;
              progressed = true;
            }
        }
      for (size_t i = 0; i < active.size(); )
        if (active[i].stage == 4)
          {
            lane_busy[active[i].c % lanes.size()] = 0;
            active.erase (active.begin() + i);
            progressed = true;
          }
        else
          i++;
      if (!progressed)
        {
          if (marks && marks->live && active.empty() && next_chunk < chunks.size())
            {
              // nothing in flight and the next chunk is not there yet: sleep until the loader says so
              std::unique_lock<std::mutex> lock (marks->mu);
              const size_t need = chunks[next_chunk].first_frame + chunks[next_chunk].n_frames;
              marks->cv.wait_for (lock, std::chrono::milliseconds (2),
                                  [&] { return marks->live_done || (!marks->marks.empty() && marks->marks.back().first >= need); });
            }
          else
            std::this_thread::yield();
        }
    }
  for (size_t ki = 0; ki < key_list.size(); ki++)
    {
      Deferred& d = deferred[ki];
      if (d.job.pending.empty())
        continue;
      KeyTables *kt = ctx->get_key_tables (key_list[ki]);
      if (!kt)
        return AWM_ERR_HIP;
      WorkLane *lane = lanes[0];                           // (every chunk is through: its workspaces are free)
      for (hipEvent_t ev : d.rows_ready)
        AWM_HIP_CHECK (hipStreamWaitEvent (lane->stream, ev, 0));
      if (int rc = decode_launch (ctx, lane, kt, d.job, ctx->ws_merge_soft.as<float>()))
        return rc;
      if (int rc = decode_finish (lane, key_list[ki], d.job, result_sets, speed))
        return rc;
    }
  drain.ok = true;
  if (debug_sync_first_chunk)
    {
      debug_sync_first_chunk->clear();
      if (key_list.size() == 1 && !chunks.empty())
        {
          const int expect0 = Params::frames_pad_start * Params::frame_size;
          const int expect_step = count * Params::frame_size;
          const int expect_end = int (chunks[0].n_frames / Params::frame_size) * Params::frame_size;
          int sync_match = 0;
          for (int expect_index = expect0; expect_index + expect_step < expect_end; expect_index += expect_step)
            for (const auto& s : first_scores)
              if (std::abs (int (s.index + params().test_cut) - expect_index) < int (Params::frame_size / 2))
                {
                  sync_match++;
                  break;
                }
          *debug_sync_first_chunk = string_printf ("sync_match %d %zd\n", sync_match, first_scores.size());
        }
    }
  return 0;
}

/* ---- ClipDecoder (reference wmget.cc:764-884) ----------------------------------------- */

int
clip_run_padded (awm_ctx *ctx, WorkLane *lane, const std::vector<Key>& key_list, const DeviceWav& wav, ResultSet& result_set,
                 double time_offset_sec, double speed)
{
  SyncFinder sync_finder (ctx, lane);
  const size_t count = mark_block_frame_count();
  for (size_t ki = 0; ki < key_list.size(); ki++)
    {
      const Key& key = key_list[ki];
      KeyTables *kt = ctx->get_key_tables (key);
      if (!kt)
        return AWM_ERR_HIP;
      std::vector<SyncFinder::Score> sync_scores;
      // (the dB matrices of the padded clip are computed for the first key and shared by the others: nothing between two searches
      // writes the lane's ws_db / ws_have)
      if (int rc = sync_finder.search (key, wav, SyncFinder::Mode::CLIP, sync_scores, /* db_ready */ ki > 0))
        return rc;
      std::vector<size_t> index;
      for (const auto& s : sync_scores)
        {
          index.push_back (s.index);
          index.push_back (s.index + count * Params::frame_size);
        }
      std::vector<int> slot_of;
      std::vector<char> ok;
      if (int rc = block_soft_bits_dev (ctx, lane, kt, wav, index, slot_of, ok))
        return rc;
      std::vector<PendingDecode> pending;
      for (size_t i = 0; i < sync_scores.size(); i++)
        {
          if (!ok[2 * i] || !ok[2 * i + 1])
            continue;
          // the block at the sync position carries the half its type says, the following block the other one
          const int first_half = sync_scores[i].block_type == ConvBlockType::a ? 0 : 1;
          SyncFinder::Score nopad = sync_scores[i];
          nopad.index = time_offset_sec * wav.sample_rate;
          pending.push_back ({ ConvBlockType::ab, 1, { { slot_of[2 * i], first_half }, { slot_of[2 * i + 1], 1 - first_half } }, 0, 0,
                               time_offset_sec, nopad, ResultSet::Type::CLIP, 0 });
        }
      if (int rc = run_pending (ctx, lane, kt, key, pending, { &result_set }, speed))
        return rc;
    }
  return 0;
}

enum class ClipPos { START, END };

int
clip_run_block (awm_ctx *ctx, WorkLane *lane, const std::vector<Key>& key_list, const DeviceWav& wav, ResultSet& result_set, ClipPos pos, double speed)
{
  const size_t C = wav.n_channels;
  const size_t n = (mark_block_frame_count() + 5) * Params::frame_size * C;      // in values
  size_t first_sample, last_sample, pad_start = n, pad_end = n;
  if (pos == ClipPos::START)
    {
      first_sample = 0;
      last_sample = std::min (n, wav.n_values());
      if (last_sample < n)
        pad_start += n - last_sample;            // data + padding must always cover one long block
    }
  else
    {
      if (wav.n_values() <= n)
        return 0;
      first_sample = wav.n_values() - n;
      last_sample = wav.n_values();
    }
  const double time_offset = double (first_sample) / wav.sample_rate / C;
  const size_t total = pad_start + (last_sample - first_sample) + pad_end;
  if (int rc = lane->ws_clip.reserve (total * sizeof (float)))
    return rc;
  float *ext = lane->ws_clip.as<float>();
  hipStream_t st = lane->stream;
  AWM_HIP_CHECK (hipMemsetAsync (ext, 0, pad_start * sizeof (float), st));
  AWM_HIP_CHECK (hipMemcpyAsync (ext + pad_start, wav.data + first_sample, (last_sample - first_sample) * sizeof (float), hipMemcpyDeviceToDevice, st));
  AWM_HIP_CHECK (hipMemsetAsync (ext + pad_start + (last_sample - first_sample), 0, pad_end * sizeof (float), st));
  DeviceWav l_wav;
  l_wav.data = ext;
  l_wav.n_frames = total / C;
  l_wav.n_channels = wav.n_channels;
  l_wav.sample_rate = wav.sample_rate;
  return clip_run_padded (ctx, lane, key_list, l_wav, result_set, time_offset, speed);
}

int
clip_decoder_run (awm_ctx *ctx, WorkLane *lane, bool spread, const std::vector<Key>& key_list, const DeviceWav& wav, ResultSet& result_set,
                  double speed)
{
  const int wav_frames = wav.n_values() / (Params::frame_size * wav.n_channels);
  if (wav_frames < int (mark_block_frame_count()) * 3.1)       // only short files
    {
      WorkLane *second = (spread && ctx->chunk_lanes > 1) ? ctx->lane (lane == ctx ? 1 : 0) : nullptr;
      if (!second)
        {
          if (int rc = clip_run_block (ctx, lane, key_list, wav, result_set, ClipPos::START, speed))
            return rc;
          return clip_run_block (ctx, lane, key_list, wav, result_set, ClipPos::END, speed);
        }
      // The START and the END search are independent chains of small kernels and host round trips (1 ms each for a
      // 30 s clip): the END one runs on a second lane, driven by a helper thread; its patterns are appended after
      // START's, the order the reference produces them in.
      if (!lane->ev_sync)
        AWM_HIP_CHECK (hipEventCreateWithFlags (&lane->ev_sync, hipEventDisableTiming));
      AWM_HIP_CHECK (hipEventRecord (lane->ev_sync, lane->stream));
      AWM_HIP_CHECK (hipStreamWaitEvent (second->stream, lane->ev_sync, 0));       // the PCM may still be in flight on `lane`
      ResultSet end_set;
      int end_rc = 0;
      std::string end_err;
      const int device = ctx->device;
      ParamValues *const pv = &params();           // helper threads run under the settings in force here
      std::thread helper ([&] {
        ParamsBind bind (pv);
        if (hipSetDevice (device) != hipSuccess)
          {
            end_rc = AWM_ERR_HIP;
            end_err = "hipSetDevice failed in the clip helper thread";
            return;
          }
        end_rc = clip_run_block (ctx, second, key_list, wav, end_set, ClipPos::END, speed);
        if (end_rc)
          end_err = last_error();
      });
      const int start_rc = clip_run_block (ctx, lane, key_list, wav, result_set, ClipPos::START, speed);
      helper.join();
      if (start_rc)
        return start_rc;
      if (end_rc)
        {
          set_error (end_err);
          return end_rc;
        }
      for (auto& p : end_set.patterns)
        result_set.patterns.push_back (std::move (p));
    }
  return 0;
}

} // namespace

/* The speed part of decode() (reference wmget.cc:886-927): "we always (unconditionally) try to decode the watermark on the
 * original wav data; if detected speed is somewhat different than 1.0, we also try to decode stretched data; we report all
 * normal and speed results we get".  Runs before the normal decoders of the chunk, like in the reference. */
static int
decode_speed (awm_ctx *ctx, WorkLane *home, bool spread, ResultSet& result_set, const std::vector<Key>& key_list, const DeviceWav& wav,
              bool first_chunk, std::string *report)
{
  if (!(params().detect_speed || params().detect_speed_patient || params().try_speed > 0))
    return 0;
  std::vector<DetectSpeedResult> speed_results;
  if (params().detect_speed || params().detect_speed_patient)
    {
      if (int rc = detect_speed (ctx, home, key_list, wav, report, speed_results))
        return rc;
    }
  else
    {
      for (const Key& key : key_list)
        speed_results.push_back ({ key, params().try_speed });
    }
  for (const auto& sr : speed_results)
    {
      // resample_ratio (wav, speed, mark_sample_rate * speed)
      size_t n_out = 0;
      DevBuffer& stretched = speed_stretch_buffer (home);
      if (int rc = resample_ratio_device (ctx, home, wav, sr.speed, -1, stretched, &n_out))
        return rc;
      DeviceWav sw;
      sw.data = stretched.as<float>();
      sw.n_frames = n_out;
      sw.n_channels = wav.n_channels;
      sw.sample_rate = int (Params::mark_sample_rate * sr.speed);
      if (int rc = block_decoder_run (ctx, home, spread, { sr.key }, sw, { ChunkRange { 0, sw.n_frames, 0.0 } }, { &result_set }, sr.speed, nullptr))
        return rc;
      if (first_chunk)
        if (int rc = clip_decoder_run (ctx, home, spread, { sr.key }, sw, result_set, sr.speed))
          return rc;
    }
  return 0;
}

bool speed_print_results = false;       // decode() passes !orig_bits.empty(): set by the command line front end for `cmp`

/* the speed part of ONE chunk on one lane (the multi-GPU protocol runs it on the rank that owns the chunk, wmshard.cc) */
int
decode_speed_chunk (awm_ctx *ctx, WorkLane *lane, ResultSet& result_set, const std::vector<Key>& key_list, const DeviceWav& chunk_wav,
                    bool first_chunk, std::string *report)
{
  return decode_speed (ctx, lane, false, result_set, key_list, chunk_wav, first_chunk, report);
}

int
decode_chunk (awm_ctx *ctx, ResultSet& result_set, const std::vector<Key>& key_list, const DeviceWav& wav, bool first_chunk)
{
  std::string report;
  const int speed_rc = decode_speed (ctx, ctx, true, result_set, key_list, wav, first_chunk, speed_print_results ? &report : nullptr);
  fputs (report.c_str(), stdout);
  if (speed_rc)
    return speed_rc;
  std::string debug_sync;
  if (int rc = block_decoder_run (ctx, ctx, true, key_list, wav, { ChunkRange { 0, wav.n_frames, 0.0 } }, { &result_set }, 1, &debug_sync))
    return rc;
  if (first_chunk)
    if (int rc = clip_decoder_run (ctx, ctx, true, key_list, wav, result_set, 1))
      return rc;
  result_set.set_debug_sync (debug_sync);
  return 0;
}

/* chunking of WavChunkLoader (reference wavchunkloader.cc:54-163), in samples per channel */
std::vector<ChunkRange>
plan_chunks (size_t n_frames, int n_channels)
{
  const size_t rate = Params::mark_sample_rate;
  const size_t max_size = size_t (lrint (params().get_chunk_size * 60 * rate));
  const double block_seconds = mark_block_frame_count() * Params::frame_size / double (rate);
  const size_t overlap = size_t (lrint (2 * block_seconds * 1.3 * rate));
  (void) n_channels;
  std::vector<ChunkRange> chunks;
  size_t buf_start = 0, buf_len = 0;
  double time_offset = 0;
  bool last = false;
  while (!last)
    {
      if (buf_len)
        {
          time_offset += double (buf_len - overlap) / double (rate);
          buf_start += buf_len - overlap;
          buf_len = overlap;
        }
      const size_t avail = n_frames - (buf_start + buf_len);
      const size_t take = std::min (max_size - buf_len, avail);
      buf_len += take;
      const bool eof = buf_len < max_size;         // the loader only notices EOF when a read comes back empty
      if (eof)
        {
          last = true;
          if (!buf_len)
            break;
        }
      chunks.push_back ({ buf_start, buf_len, time_offset });
    }
  return chunks;
}

static int
decode_chunks_on (awm_ctx *ctx, WorkLane *home, bool spread, const std::vector<Key>& key_list, const DeviceWav& wav,
                  const std::vector<ChunkRange>& chunks, bool first_is_stream_start, std::vector<ResultSet>& chunk_sets)
{
  chunk_sets.clear();
  chunk_sets.resize (chunks.size());
  std::vector<ResultSet *> ptrs;
  for (auto& cs : chunk_sets)
    ptrs.push_back (&cs);
  std::string debug_sync;
  /* With a speed search the PLAIN decode of the chunks (reference wmget.cc:929: after the speed part) depends on nothing the speed part
   * produces: it runs beside it, on lanes of its own behind the speed chunks' lanes and driven by a host thread of its own, into pattern
   * lists of its own that are appended to the chunks' lists afterwards (the reference's order: speed patterns, then block patterns).  The
   * speed part leaves a third of the GPU's time unused (host round trips between its passes). */
  std::vector<ResultSet> plain_sets;
  std::string plain_error;
  std::future<int> plain_run;                   // (declared after everything its task refers to)
  if (params().detect_speed || params().detect_speed_patient || params().try_speed > 0)
    {
      const size_t lanes_per_part = size_t (std::max (1, std::min (ctx->chunk_lanes, CHUNK_LANES)));
      if (spread && g_speed_overlap && !chunks.empty())
        {
          const int lane_base = int (std::min<size_t> (chunks.size(), lanes_per_part));          // behind the speed part's lanes
          const size_t n_plain = std::min<size_t> (chunks.size(), lanes_per_part);
          if (!ctx->ev_sync)
            AWM_HIP_CHECK (hipEventCreateWithFlags (&ctx->ev_sync, hipEventDisableTiming));
          AWM_HIP_CHECK (hipEventRecord (ctx->ev_sync, ctx->stream));                             // the PCM may still be in flight there
          for (size_t i = 0; i < n_plain; i++)
            {
              WorkLane *l = ctx->lane (lane_base + int (i));
              if (!l)
                {
                  set_error ("cannot create a work lane (stream)");
                  return AWM_ERR_HIP;
                }
              AWM_HIP_CHECK (hipStreamWaitEvent (l->stream, ctx->ev_sync, 0));
            }
          plain_sets.resize (chunks.size());
          ParamValues *const pv = &params();
          plain_run = std::async (std::launch::async, [&, pv, lane_base] {
            ParamsBind bind (pv);
            if (hipSetDevice (ctx->device) != hipSuccess)
              return int (AWM_ERR_HIP);
            std::vector<ResultSet *> plain_ptrs;
            for (auto& ps : plain_sets)
              plain_ptrs.push_back (&ps);
            const int rc = block_decoder_run (ctx, ctx, true, key_list, wav, chunks, plain_ptrs, 1, &debug_sync, lane_base);
            if (rc)
              plain_error = last_error();
            return rc;
          });
        }
      /* The speed part of decode() for every chunk (reference wmget.cc:886-927) -- speed search, stretched copy, block and clip
       * decoder on the stretched copy -- is a long chain with a dozen host round trips (three search passes, each waiting for
       * its scores) that keeps the GPU busy for two thirds of its duration.  The chunks are independent: each one runs the
       * chain on its own lane, driven by its own host thread (buffers of a search are per lane, the tables shared). */
      const size_t n_par = !spread ? 1 : std::min<size_t> (chunks.size(), size_t (std::max (1, std::min (ctx->chunk_lanes, CHUNK_LANES))));
      std::vector<WorkLane *> lanes;
      for (size_t i = 0; i < n_par; i++)
        {
          WorkLane *l = spread ? ctx->lane (int (i)) : home;
          if (!l)
            {
              set_error ("cannot create a work lane (stream)");
              return AWM_ERR_HIP;
            }
          lanes.push_back (l);
        }
      if (lanes.size() > 1)
        {
          // the PCM may still be in flight on the context's stream
          if (!ctx->ev_sync)
            AWM_HIP_CHECK (hipEventCreateWithFlags (&ctx->ev_sync, hipEventDisableTiming));
          AWM_HIP_CHECK (hipEventRecord (ctx->ev_sync, ctx->stream));
          for (size_t i = 1; i < lanes.size(); i++)
            AWM_HIP_CHECK (hipStreamWaitEvent (lanes[i]->stream, ctx->ev_sync, 0));
        }
      std::vector<std::string> reports (chunks.size());
      std::vector<int> rcs (chunks.size(), 0);
      std::vector<std::string> messages (chunks.size());
      auto run_chunk = [&] (size_t c, WorkLane *lane) {
        DeviceWav cw = wav;
        cw.data = wav.data + chunks[c].first_frame * wav.n_channels;
        cw.n_frames = chunks[c].n_frames;
        rcs[c] = decode_speed (ctx, lane, false, chunk_sets[c], key_list, cw, c == 0 && first_is_stream_start,
                               speed_print_results ? &reports[c] : nullptr);
        if (rcs[c])
          messages[c] = last_error();
      };
      if (lanes.size() == 1)
        for (size_t c = 0; c < chunks.size() && !(c && rcs[c - 1]); c++)
          run_chunk (c, lanes[0]);
      else
        {
          // chunk c on lane c mod n, whatever the threads' timing: the lanes' workspaces reach their final sizes in the first call (with
          // "whoever is free takes the next chunk" the pairing changed from call to call and buffers kept growing over several calls)
          std::vector<std::thread> workers;
          const int device = ctx->device;
          ParamValues *const pv = &params();
          for (size_t li = 0; li < lanes.size(); li++)
            workers.emplace_back ([&, li] {
              ParamsBind bind (pv);
              if (hipSetDevice (device) != hipSuccess)
                return;
              for (size_t c = li; c < chunks.size(); c += lanes.size())
                {
                  run_chunk (c, lanes[li]);
                  if (rcs[c])
                    break;
                }
            });
          for (auto& w : workers)
            w.join();
        }
      for (size_t c = 0; c < chunks.size(); c++)
        {
          fputs (reports[c].c_str(), stdout);          // in chunk order, as the reference prints them
          if (rcs[c])
            {
              set_error (messages[c]);
              return rcs[c];
            }
        }
    }
  if (plain_run.valid())
    {
      if (int rc = plain_run.get())
        {
          set_error (plain_error);
          return rc;
        }
      for (size_t c = 0; c < chunks.size(); c++)
        chunk_sets[c].patterns.insert (chunk_sets[c].patterns.end(), plain_sets[c].patterns.begin(), plain_sets[c].patterns.end());
    }
  else if (int rc = block_decoder_run (ctx, home, spread, key_list, wav, chunks, ptrs, 1, &debug_sync))
    return rc;
  if (!chunks.empty() && first_is_stream_start)
    {
      // ClipDecoder only looks at the first chunk of the stream (reference wmget.cc:932-936)
      DeviceWav cw = wav;
      cw.data = wav.data + chunks[0].first_frame * wav.n_channels;
      cw.n_frames = chunks[0].n_frames;
      if (int rc = clip_decoder_run (ctx, home, spread, key_list, cw, chunk_sets[0], 1))
        return rc;
      chunk_sets[0].set_debug_sync (debug_sync);
    }
  return 0;
}

/* the BlockDecoder alone for chunks of one resident buffer (the multi-GPU protocol decodes the chunks that lie completely inside a
 * rank's span this way: plain path, chunks on concurrent lanes) */
int
decode_chunks_blocks_only (awm_ctx *ctx, const std::vector<Key>& key_list, const DeviceWav& wav, const std::vector<ChunkRange>& chunks,
                           std::vector<ResultSet>& chunk_sets, std::string *debug_sync_first, int lane_base)
{
  chunk_sets.clear();
  chunk_sets.resize (chunks.size());
  std::vector<ResultSet *> ptrs;
  for (auto& cs : chunk_sets)
    ptrs.push_back (&cs);
  return block_decoder_run (ctx, ctx, true, key_list, wav, chunks, ptrs, 1, debug_sync_first, lane_base);
}

int
decode_chunks (awm_ctx *ctx, const std::vector<Key>& key_list, const DeviceWav& wav, const std::vector<ChunkRange>& chunks,
               bool first_is_stream_start, std::vector<ResultSet>& chunk_sets)
{
  return decode_chunks_on (ctx, ctx, true, key_list, wav, chunks, first_is_stream_start, chunk_sets);
}

static int
get_watermark_on (awm_ctx *ctx, WorkLane *home, bool spread, const std::vector<Key>& key_list, const DeviceWav& wav, ResultSet& result_set)
{
  const auto chunks = plan_chunks (wav.n_frames, wav.n_channels);
  std::vector<ResultSet> chunk_sets;
  if (int rc = decode_chunks_on (ctx, home, spread, key_list, wav, chunks, true, chunk_sets))
    return rc;
  for (size_t c = 0; c < chunks.size(); c++)
    {
      chunk_sets[c].apply_time_offset (chunks[c].time_offset);
      result_set.merge (chunk_sets[c]);
    }
  result_set.sort (key_list);
  return 0;
}

int
get_watermark_device (awm_ctx *ctx, const std::vector<Key>& key_list, const DeviceWav& wav, ResultSet& result_set)
{
  return get_watermark_on (ctx, ctx, true, key_list, wav, result_set);
}

/* Short clips (less than one block + 2 frames: 51.7 s) go through exactly one thing in `get`: the ClipDecoder's START pass
 * (the block decoder finds no room for a block, the END pass needs more than one long block; reference wmget.cc:769-884,
 * 886-939).  Alone, such a clip is ~45 small launches and copies and 3 host round trips around a few hundred microseconds of GPU
 * work.  A batch of them is therefore processed in GROUPS: the padded copies of a group lie side by side in one buffer (equal
 * slices), and every stage -- pad + silence scan, dB matrices, approximate scan, local mean, peak selection, refinement, block dB +
 * soft bits, Viterbi -- is ONE launch for the whole group (SyncFinder::group_*), with one wait per stage.  Two to four host threads work
 * on their own lanes, so that the waits of one overlap the kernels of the others. */
constexpr int CLIP_GROUP = 64;          // clips per group
static std::atomic<long long> g_clip_key_us[3];
extern "C" void awm_debug_clip_key_timing (double out[3]) { for (int i = 0; i < 3; i++) out[i] = double (g_clip_key_us[i].exchange (0)); }
/* host threads (= lanes) of a batch: 0 = as shipped -- four (get of 1024 clips with one key: 203 / 199 / 195 / 193 ms with 2 / 3 / 4 / 6
 * threads; with a key per clip and the tables from the device, K16g: 211 / 204 / 201 with 2 / 3 / 4), but two for a batch with a key per
 * clip whose tables come from host threads: there every thread also builds and packs its groups' key tables (64 table threads
 * each), and a third and fourth make the groups wait for their tables longer than they shorten the device's idle time (206 / 205 /
 * 208 / 211 ms with 2 / 3 / 4 / 6; tools/gpu_clip_keys_ab.py) */
static int g_staged_threads = 0;
extern "C" void awm_debug_set_staged_threads (int n) { g_staged_threads = n < 0 ? 0 : (n > 8 ? 8 : n); }

static bool
clip_is_short (const DeviceWav& w)
{
  return w.n_frames > 0 && w.n_frames < (mark_block_frame_count() + 2) * Params::frame_size;
}

/* The key tables of a group of clips with ONE KEY PER CLIP: built on host threads (2226 up / down draws, three shuffles per key:
 * ~3 ms of one core), packed into one page-locked block, one copy to the device; `kt` then describes the group (KeyTables::slices). */
namespace {
int g_table_threads = 64;
size_t align256 (size_t n) { return (n + 255) & ~size_t (255); }

std::vector<ClipKeyHost>
build_group_hosts (const std::vector<Key>& keys)
{
  std::vector<ClipKeyHost> hosts (keys.size());
  const size_t n_threads = std::max<size_t> (1, std::min<size_t> ({ keys.size(), size_t (g_table_threads), size_t (std::max (1u, std::thread::hardware_concurrency())) }));
  std::atomic<size_t> next { 0 };
  ParamValues *const pv = &params();
  auto work = [&] {
    ParamsBind bind (pv);
    for (size_t i = next.fetch_add (1); i < keys.size(); i = next.fetch_add (1))
      hosts[i] = build_clip_key_host (keys[i]);
  };
  std::vector<std::thread> threads;
  for (size_t t = 1; t < n_threads; t++)
    threads.emplace_back (work);
  work();
  for (auto& t : threads)
    t.join();
  return hosts;
}

}
/* (measurement, host only) wall time in ms of the key tables of one group of n_keys clips, built as `get` builds them; threads <= 0: as shipped */
extern "C" double
awm_debug_time_group_key_tables (int n_keys, int threads)
{
  std::vector<Key> keys (std::max (n_keys, 1));
  for (size_t i = 0; i < keys.size(); i++)
    keys[i].set_test_key (1000 + i);
  const int before = g_table_threads;
  if (threads > 0)
    g_table_threads = threads;
  const auto t0 = std::chrono::steady_clock::now();
  const auto hosts = build_group_hosts (keys);
  const double ms = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - t0).count();
  g_table_threads = before;
  return hosts.size() == keys.size() ? ms : -1;
}
namespace {

int
upload_group_tables (WorkLane *lane, const std::vector<ClipKeyHost>& hosts, KeyTables& kt)
{
  const size_t gn = hosts.size();
  const size_t n_chain = hosts[0].chains.size(), NW = hosts[0].want.size(), n_pos = hosts[0].pos.size(), n_mix = hosts[0].mix.frame.size(),
               n_ord = hosts[0].inv_order.size();
  const size_t off_chain = 0, off_perm = align256 (off_chain + gn * n_chain * sizeof (unsigned)), off_pos = align256 (off_perm + gn * NW * sizeof (int)),
               off_mf = align256 (off_pos + gn * n_pos), off_mu = align256 (off_mf + gn * n_mix * sizeof (int16_t)), off_md = align256 (off_mu + gn * n_mix),
               off_ord = align256 (off_md + gn * n_mix), off_rf = align256 (off_ord + gn * n_ord * sizeof (int)),
               n_rf = hosts[0].row_frames.size(), bytes = align256 (off_rf + gn * n_rf * sizeof (int));
  if (int rc = lane->pin_keytab.reserve (bytes)) return rc;
  if (int rc = lane->ws_keytab.reserve (bytes)) return rc;
  char *h = lane->pin_keytab.as<char>();
  for (size_t i = 0; i < gn; i++)
    {
      const ClipKeyHost& k = hosts[i];
      if (k.chains.size() != n_chain || k.want.size() != NW || k.mix.frame.size() != n_mix)
        {
          set_error ("clip batch: key tables of different shapes");
          return AWM_ERR_GENERIC;
        }
      std::copy (k.chains.begin(), k.chains.end(), reinterpret_cast<unsigned *> (h + off_chain) + i * n_chain);
      std::copy (k.perm.begin(), k.perm.end(), reinterpret_cast<int *> (h + off_perm) + i * NW);
      std::copy (k.pos.begin(), k.pos.end(), reinterpret_cast<unsigned char *> (h + off_pos) + i * n_pos);
      std::copy (k.mix.frame.begin(), k.mix.frame.end(), reinterpret_cast<int16_t *> (h + off_mf) + i * n_mix);
      std::copy (k.mix.up.begin(), k.mix.up.end(), reinterpret_cast<uint8_t *> (h + off_mu) + i * n_mix);
      std::copy (k.mix.down.begin(), k.mix.down.end(), reinterpret_cast<uint8_t *> (h + off_md) + i * n_mix);
      std::copy (k.inv_order.begin(), k.inv_order.end(), reinterpret_cast<int *> (h + off_ord) + i * n_ord);
      std::copy (k.row_frames.begin(), k.row_frames.end(), reinterpret_cast<int *> (h + off_rf) + i * n_rf);
    }
  AWM_HIP_CHECK (hipMemcpyAsync (lane->ws_keytab.ptr, h, bytes, hipMemcpyHostToDevice, lane->stream));
  char *d = lane->ws_keytab.as<char>();
  auto view = [] (DevBuffer& b, void *ptr, size_t n) { b.ptr = ptr; b.bytes = n; };     // (non-owning: never released through kt)
  kt = KeyTables();
  kt.slices = int (gn);
  kt.mix = params().mix;
  kt.sync[1].host.rows_per_bit = int (n_chain / (12 * 8));
  view (kt.sync[1].chains_approx, d + off_chain, gn * n_chain * sizeof (unsigned));
  view (kt.sync[1].refine_perm, d + off_perm, gn * NW * sizeof (int));
  view (kt.sync[1].refine_pos, d + off_pos, gn * n_pos);
  view (kt.mix_frame, d + off_mf, gn * n_mix * sizeof (int16_t));
  view (kt.mix_up, d + off_mu, gn * n_mix);
  view (kt.mix_down, d + off_md, gn * n_mix);
  view (kt.bit_order_inv_dev, d + off_ord, gn * n_ord * sizeof (int));
  view (kt.sync[1].row_frames, d + off_rf, gn * n_rf * sizeof (int));
  for (const ClipKeyHost& k : hosts)
    kt.slice_want.push_back (k.want);
  return 0;
}

/* The same group tables built ON THE DEVICE (K16g, hip/keytab.hip): one workgroup per key on a stream of its own, one group ahead of the
 * lane -- two table areas in turn.  What the host contributes per key is the AES key schedule (176 bytes); what comes back is the want
 * list (4 KB per key: the host places the refinement's rows with it).  A key's tables cost 1.3 ms of a host core, and a process on the
 * GPU box may use 16 cores (cgroup cpu.max): 1024 keys are 1.3 s of host time per call beside 0.2 s of device work -- the lanes
 * waited 12 ms per call for their tables with two host threads and longer with more (tools/gpu_clip_keys_ab.py). */
struct DeviceGroupTables
{
  static constexpr size_t N_CHAIN = size_t (12) * awmk::CLIP_KEY_ROWS * 8, NW = awmk::CLIP_KEY_WANT, N_POS = NW * Params::n_bands,
                          N_MIX = awmk::CLIP_KEY_MIX, N_ORD = awmk::CLIP_KEY_CODED, G = CLIP_GROUP, LAUNCH = 256;
  awm_ctx    *ctx = nullptr;
  WorkLane   *lane = nullptr;            // owner of the buffers (ws_keytab, ws_keytab_aux, ws_keytab_scratch, pin_keytab)
  hipStream_t table_stream = nullptr;
  hipEvent_t  ev[2] = { nullptr, nullptr };
  size_t      cap = G;                   // keys per table area: a group (two areas in turn), or ALL keys of a call (one area)
  int         areas = 2;
  size_t      off_chain = 0, off_perm = 0, off_pos = 0, off_mf = 0, off_mu = 0, off_md = 0, off_ord = 0, off_rf = 0, off_want = 0, half_bytes = 0, pin_half = 0;
  size_t aux_bytes() const { return 256 + cap * 176; }
  size_t want_bytes() const { return cap * NW * sizeof (int); }

  static bool
  possible()
  {
    return g_key_tables_on_device && params().mix && mark_block_frame_count() == 2226 && mark_sync_frame_count() == 510
        && mark_data_frame_count() * Params::bands_per_frame == N_MIX && code_size (ConvBlockType::a, params().payload_size) == N_ORD;
  }
  ~DeviceGroupTables()
  {
    if (table_stream && (ev[0] || ev[1]))
      (void) hipStreamSynchronize (table_stream);                       // (an early return: nothing of ours is left in flight)
    for (hipEvent_t e : ev)
      if (e)
        (void) hipEventDestroy (e);
  }
  /* table areas for `keys_per_area` keys each in the buffers of lane l, filled on table_stream */
  int
  init (awm_ctx *c, WorkLane *l, hipStream_t tables_on, size_t keys_per_area = G, int n_areas = 2)
  {
    ctx = c;
    lane = l;
    table_stream = tables_on;
    cap = keys_per_area;
    areas = n_areas;
    off_chain = 0;
    off_perm = align256 (off_chain + cap * N_CHAIN * sizeof (unsigned));
    off_pos = align256 (off_perm + cap * NW * sizeof (int));
    off_mf = align256 (off_pos + cap * N_POS);
    off_mu = align256 (off_mf + cap * N_MIX * sizeof (int16_t));
    off_md = align256 (off_mu + cap * N_MIX);
    off_ord = align256 (off_md + cap * N_MIX);
    off_rf = align256 (off_ord + cap * N_ORD * sizeof (int));
    off_want = align256 (off_rf + cap * NW * sizeof (int));
    half_bytes = align256 (off_want + want_bytes());
    pin_half = align256 (aux_bytes()) + align256 (want_bytes());
    if (int rc = lane->ws_keytab.reserve (areas * half_bytes)) return rc;
    if (int rc = lane->ws_keytab_aux.reserve (areas * align256 (aux_bytes()))) return rc;
    if (int rc = lane->ws_keytab_scratch.reserve (std::min (cap, LAUNCH) * awmk::key_table_scratch_bytes())) return rc;
    if (int rc = lane->pin_keytab.reserve (areas * pin_half)) return rc;
    for (hipEvent_t& e : ev)
      AWM_HIP_CHECK (hipEventCreateWithFlags (&e, hipEventDisableTiming));
    // (the lane's earlier work may still read these buffers; and the clips' producers are behind the lane's stream)
    if (table_stream != lane->stream)
      {
        AWM_HIP_CHECK (hipEventRecord (ev[0], lane->stream));
        AWM_HIP_CHECK (hipStreamWaitEvent (table_stream, ev[0], 0));
      }
    return 0;
  }
  /* queue the tables of `keys` (<= cap) into area `half`; the area's previous group must be through on the lane */
  int
  launch (const std::vector<Key>& keys, int half)
  {
    unsigned char *aux = lane->pin_keytab.as<unsigned char>() + size_t (half) * pin_half;
    std::memcpy (aux, Aes128::sbox(), 256);
    for (size_t i = 0; i < keys.size(); i++)
      {
        Aes128 aes;
        aes.set_key (keys[i].aes_key());
        std::memcpy (aux + 256 + 176 * i, aes.round_keys(), 176);
      }
    unsigned char *d_aux = lane->ws_keytab_aux.as<unsigned char>() + size_t (half) * align256 (aux_bytes());
    AWM_HIP_CHECK (hipMemcpyAsync (d_aux, aux, 256 + 176 * keys.size(), hipMemcpyHostToDevice, table_stream));
    char *d = lane->ws_keytab.as<char>() + size_t (half) * half_bytes;
    // (one workgroup = one compute unit per key: LAUNCH keys per launch, the launches of a stream one after the other share the scratch)
    for (size_t k0 = 0; k0 < keys.size(); k0 += LAUNCH)
      {
        const size_t kn = std::min (LAUNCH, keys.size() - k0);
        awmk::KeyTableArgs ka {};
        ka.sbox = d_aux;
        ka.round_keys = d_aux + 256 + 176 * k0;
        ka.scratch = lane->ws_keytab_scratch.as<unsigned char>();
        ka.scratch_slots = int (std::min (cap, LAUNCH));
        ka.n_keys = (long long) kn;
        awmk::ClipKeyTableOut o {};
        o.chains = reinterpret_cast<unsigned int *> (d + off_chain) + k0 * N_CHAIN;
        o.perm = reinterpret_cast<int *> (d + off_perm) + k0 * NW;
        o.pos = reinterpret_cast<unsigned char *> (d + off_pos) + k0 * N_POS;
        o.mix_frame = reinterpret_cast<short *> (d + off_mf) + k0 * N_MIX;
        o.mix_up = reinterpret_cast<unsigned char *> (d + off_mu) + k0 * N_MIX;
        o.mix_down = reinterpret_cast<unsigned char *> (d + off_md) + k0 * N_MIX;
        o.inv_order = reinterpret_cast<int *> (d + off_ord) + k0 * N_ORD;
        o.row_frames = reinterpret_cast<int *> (d + off_rf) + k0 * NW;
        o.want = reinterpret_cast<int *> (d + off_want) + k0 * NW;
        ProfScope ps (ctx, PROF_KEYTAB, double (kn) * double (half_bytes) / double (cap), table_stream);
        AWM_HIP_CHECK (awmk::launch_clip_key_tables (table_stream, ka, o));
      }
    AWM_HIP_CHECK (hipMemcpyAsync (aux + align256 (aux_bytes()), d + off_want, keys.size() * NW * sizeof (int), hipMemcpyDeviceToHost, table_stream));
    AWM_HIP_CHECK (hipEventRecord (ev[half], table_stream));
    return 0;
  }
  /* stream `st` waits for area `half`; kt describes its keys [slot0, slot0 + gn) (want lists: want_ready) */
  int
  use (int half, size_t gn, KeyTables& kt, hipStream_t st = nullptr, size_t slot0 = 0) const
  {
    AWM_HIP_CHECK (hipStreamWaitEvent (st ? st : lane->stream, ev[half], 0));
    char *d = lane->ws_keytab.as<char>() + size_t (half) * half_bytes;
    auto view = [] (DevBuffer& b, void *ptr, size_t n) { b.ptr = ptr; b.bytes = n; };     // (non-owning: never released through kt)
    kt = KeyTables();
    kt.slices = int (gn);
    kt.mix = params().mix;
    kt.sync[1].host.rows_per_bit = awmk::CLIP_KEY_ROWS;
    view (kt.sync[1].chains_approx, d + off_chain + slot0 * N_CHAIN * sizeof (unsigned), gn * N_CHAIN * sizeof (unsigned));
    view (kt.sync[1].refine_perm, d + off_perm + slot0 * NW * sizeof (int), gn * NW * sizeof (int));
    view (kt.sync[1].refine_pos, d + off_pos + slot0 * N_POS, gn * N_POS);
    view (kt.mix_frame, d + off_mf + slot0 * N_MIX * sizeof (int16_t), gn * N_MIX * sizeof (int16_t));
    view (kt.mix_up, d + off_mu + slot0 * N_MIX, gn * N_MIX);
    view (kt.mix_down, d + off_md + slot0 * N_MIX, gn * N_MIX);
    view (kt.bit_order_inv_dev, d + off_ord + slot0 * N_ORD * sizeof (int), gn * N_ORD * sizeof (int));
    view (kt.sync[1].row_frames, d + off_rf + slot0 * NW * sizeof (int), gn * NW * sizeof (int));
    kt.slice_want_flat = reinterpret_cast<const int *> (lane->pin_keytab.as<unsigned char>() + size_t (half) * pin_half + align256 (aux_bytes())) + slot0 * NW;
    kt.slice_want_n = int (NW);
    return 0;
  }
  /* before the host reads the want lists of area `half` */
  int
  want_ready (int half) const
  {
    AWM_HIP_CHECK (hipEventSynchronize (ev[half]));
    return 0;
  }
};
}

/* (test) the tables of n_keys keys from the device against the host's, group by group as `get` builds them: mismatch_out[0 .. 8] = number
 * of differing elements in chains, row_frames, want, perm, pos, mix_frame, mix_up, mix_down, inv_order */
int
clip_key_tables_check (awm_ctx *ctx, const uint8_t *keys, size_t n_keys, long long mismatch_out[9])
{
  if (!keys || !mismatch_out || !DeviceGroupTables::possible())
    {
      set_error ("awm_debug_clip_key_tables_check_d: bad argument, or parameters the device tables do not cover");
      return AWM_ERR_ARG;
    }
  std::fill (mismatch_out, mismatch_out + 9, 0);
  using D = DeviceGroupTables;
  WorkLane *lane = ctx->lane (0), *table_lane = ctx->lane (1);
  if (!table_lane)
    return AWM_ERR_HIP;
  AWM_HIP_CHECK (stream_wait (lane->stream));
  D dev;
  if (int rc = dev.init (ctx, lane, table_lane->stream)) return rc;
  std::vector<char> back (dev.half_bytes);
  auto count = [] (const auto *a, const auto *b, size_t n) { long long bad = 0; for (size_t i = 0; i < n; i++) bad += a[i] != b[i]; return bad; };
  size_t group = 0;
  for (size_t g0 = 0; g0 < n_keys; g0 += D::G, group++)
    {
      const size_t gn = std::min (D::G, n_keys - g0);
      std::vector<Key> group_keys (gn);
      for (size_t i = 0; i < gn; i++)
        group_keys[i].set_raw (keys + 16 * (g0 + i));
      const int half = int (group & 1);
      if (int rc = dev.launch (group_keys, half)) return rc;
      KeyTables kt;
      if (int rc = dev.use (half, gn, kt)) return rc;
      if (int rc = dev.want_ready (half)) return rc;
      AWM_HIP_CHECK (hipMemcpyAsync (back.data(), lane->ws_keytab.as<char>() + size_t (half) * dev.half_bytes, dev.half_bytes, hipMemcpyDeviceToHost, lane->stream));
      AWM_HIP_CHECK (stream_wait (lane->stream));
      const char *b = back.data();
      for (size_t i = 0; i < gn; i++)
        {
          const ClipKeyHost h = build_clip_key_host (group_keys[i]);
          if (h.chains.size() != D::N_CHAIN || h.want.size() != D::NW || h.pos.size() != D::N_POS || h.mix.frame.size() != D::N_MIX || h.inv_order.size() != D::N_ORD)
            {
              set_error ("clip key tables of unexpected shape");
              return AWM_ERR_GENERIC;
            }
          mismatch_out[0] += count (h.chains.data(), reinterpret_cast<const unsigned *> (b + dev.off_chain) + i * D::N_CHAIN, D::N_CHAIN);
          mismatch_out[1] += count (h.row_frames.data(), reinterpret_cast<const int *> (b + dev.off_rf) + i * D::NW, D::NW);
          mismatch_out[2] += count (h.want.data(), kt.want_of_slice (int (i)), D::NW);
          mismatch_out[2] += count (h.want.data(), reinterpret_cast<const int *> (b + dev.off_want) + i * D::NW, D::NW);
          mismatch_out[3] += count (h.perm.data(), reinterpret_cast<const int *> (b + dev.off_perm) + i * D::NW, D::NW);
          mismatch_out[4] += count (h.pos.data(), reinterpret_cast<const unsigned char *> (b + dev.off_pos) + i * D::N_POS, D::N_POS);
          mismatch_out[5] += count (h.mix.frame.data(), reinterpret_cast<const int16_t *> (b + dev.off_mf) + i * D::N_MIX, D::N_MIX);
          mismatch_out[6] += count (h.mix.up.data(), reinterpret_cast<const uint8_t *> (b + dev.off_mu) + i * D::N_MIX, D::N_MIX);
          mismatch_out[7] += count (h.mix.down.data(), reinterpret_cast<const uint8_t *> (b + dev.off_md) + i * D::N_MIX, D::N_MIX);
          mismatch_out[8] += count (h.inv_order.data(), reinterpret_cast<const int *> (b + dev.off_ord) + i * D::N_ORD, D::N_ORD);
        }
    }
  return 0;
}

/* frames of zeros written on either side of a clip in its padded slice (kernels.hh launch_clip_pad: the consumers read at most
 * 1024 + 8 * 64 + read-ahead frames beyond the clip) */
constexpr int CLIP_PAD_MARGIN = 2048;
/* (measurement knob) another margin, in frames; one slice or more = whole slices are written, as before this margin existed */
static int g_clip_pad_margin = CLIP_PAD_MARGIN;
extern "C" void awm_debug_set_clip_pad_margin (int frames) { g_clip_pad_margin = frames > 0 ? frames : CLIP_PAD_MARGIN; }
/* (test knob) every clip of a group takes the sequential selection */
static int g_group_force_fallback = 0;
extern "C" void awm_debug_set_group_fallback (int on) { g_group_force_fallback = on; }

/* clip_keys (may be null): one key per clip -- clip i is searched and decoded with (*clip_keys)[i] alone (key_list is not used then) */
static int
clip_batch_staged (awm_ctx *ctx, WorkLane *lane, const std::vector<Key>& key_list, const std::vector<DeviceWav>& clips, const std::vector<size_t>& which,
                   std::vector<ResultSet>& result_sets, const std::vector<Key> *clip_keys = nullptr, WorkLane *table_lane = nullptr,
                   const DeviceGroupTables *all_tables = nullptr, const std::vector<int> *slot_of_clip = nullptr)
{
  const size_t count = mark_block_frame_count();
  const int n_bits_a = int (mark_data_frame_count() / params().frames_per_bit);
  auto group_keys = [&] (size_t g0, size_t gn) {
    std::vector<Key> keys;
    for (size_t i = 0; i < gn; i++)
      keys.push_back ((*clip_keys)[which[g0 + i]]);
    return keys;
  };
  // a group: up to CLIP_GROUP clips with the same number of channels (the slices of a group are equal) -- and, with the tables of all
  // keys built first, with consecutive table slots (a group cut short by a change of channels would otherwise let the next one reach
  // across two strides of the lane's share of the slots)
  auto group_size = [&] (size_t g0) {
    size_t gn = 0;
    while (g0 + gn < which.size() && gn < size_t (CLIP_GROUP) && clips[which[g0 + gn]].n_channels == clips[which[g0]].n_channels
           && (!(clip_keys && all_tables) || (*slot_of_clip)[which[g0 + gn]] == (*slot_of_clip)[which[g0]] + int (gn)))
      gn++;
    return gn;
  };
  // the key tables of the NEXT group are built on host threads while the device works on this one.  (Declared after everything the
  // task could refer to, and the task gets its keys BY VALUE: on an early error return the future's destructor waits for the task.)
  std::future<std::vector<ClipKeyHost>> next_hosts;
  hipStream_t st = lane->stream;
  struct LaneDrain
  {
    hipStream_t st;
    bool ok = false;
    ~LaneDrain() { if (!ok) (void) hipStreamSynchronize (st); }
  } drain { st };
  std::vector<ResultSet> chunk_sets (which.size());
  // one key per clip: the groups' tables from the device (K16g), one group ahead on the table lane's stream -- or from host threads
  DeviceGroupTables dev_tables;
  // (all_tables: the tables of ALL keys of the call were queued before the lanes started, clip c's at slot (*slot_of_clip)[c])
  const bool tables_on_device = clip_keys && !all_tables && table_lane && !which.empty() && DeviceGroupTables::possible();
  if (tables_on_device)
    {
      AWM_HIP_CHECK (stream_wait (st));                      // (the lane's table buffers may be re-allocated: nothing of an earlier call reads them)
      if (int rc = dev_tables.init (ctx, lane, table_lane->stream)) return rc;
      if (int rc = dev_tables.launch (group_keys (0, group_size (0)), 0)) return rc;
    }
  size_t group_index = 0;
  for (size_t g0 = 0; g0 < which.size(); group_index++)
    {
      const int C = clips[which[g0]].n_channels;
      const size_t gn = group_size (g0);
      const size_t n = (count + 5) * Params::frame_size * C;                       // in values
      const size_t slice_values = 3 * n;                                            // pad_start + len + n with pad_start = n + (n - len)
      const size_t slice_frames = slice_values / C;
      {
        // (profiling only) frames of the group's slices that carry samples: a clip's own frames + one at each edge
        double live = 0;
        for (size_t i = 0; i < gn; i++)
          live += double (clips[which[g0 + i]].n_frames) / Params::frame_size + 2;
        lane->prof_live_fraction = std::min (1.0, live / (double (gn) * double (slice_frames) / Params::frame_size));
      }
      // stage 0: padded copies (reference wmget.cc:830-867, START position: data + padding cover one long block) and the
      // non-silent range of every slice
      if (int rc = lane->ws_clip.reserve (gn * slice_values * sizeof (float))) return rc;
      const size_t desc_bytes = gn * sizeof (awmk::ClipSrc);
      if (int rc = lane->pin_group.reserve (desc_bytes)) return rc;
      if (int rc = lane->ws_group.reserve (desc_bytes + gn * 2 * sizeof (long long))) return rc;
      AWM_HIP_CHECK (stream_wait (st));                                             // (the previous group's descriptors are consumed)
      auto *desc = lane->pin_group.as<awmk::ClipSrc>();
      for (size_t i = 0; i < gn; i++)
        {
          const DeviceWav& wav = clips[which[g0 + i]];
          desc[i] = { wav.data, (long long) wav.n_values(), (long long) (n + (n - wav.n_values())) };
        }
      auto *d_desc = lane->ws_group.as<awmk::ClipSrc>();
      auto *d_range = reinterpret_cast<long long *> (lane->ws_group.as<char>() + desc_bytes);
      AWM_HIP_CHECK (hipMemcpyAsync (d_desc, desc, desc_bytes, hipMemcpyHostToDevice, st));
      // (only CLIP_PAD_MARGIN frames of zeros on either side of a clip are written: nothing further out is read -- kernels.hh)
      const long long margin_values = (long long) std::max (g_clip_pad_margin, CLIP_PAD_MARGIN) * C;
      AWM_HIP_CHECK (awmk::launch_clip_pad (st, d_desc, int (gn), lane->ws_clip.as<float>(), (long long) slice_values, margin_values, d_range));
      DeviceWav group;
      group.data = lane->ws_clip.as<float>();
      group.n_frames = gn * slice_frames;
      group.n_channels = C;
      group.sample_rate = clips[which[g0]].sample_rate;
      std::vector<ResultSet *> ptrs;
      for (auto& cs : chunk_sets)
        ptrs.push_back (&cs);
      KeyTables group_kt;
      std::vector<Key> keys_of_group;
      const int half = all_tables ? 0 : int (group_index & 1);
      if (clip_keys && all_tables)
        {
          keys_of_group = group_keys (g0, gn);
          const int slot0 = (*slot_of_clip)[which[g0]];
          for (size_t i = 0; i < gn; i++)
            if ((*slot_of_clip)[which[g0 + i]] != slot0 + int (i))
              {
                set_error ("clip batch: the tables of a group's keys are not consecutive");
                return AWM_ERR_GENERIC;
              }
          if (int rc = all_tables->use (0, gn, group_kt, st, size_t (slot0))) return rc;
        }
      else if (tables_on_device)
        {
          keys_of_group = group_keys (g0, gn);
          // (the other area's group -- the previous one -- is through: its results have been waited for)
          if (g0 + gn < which.size())
            if (int rc = dev_tables.launch (group_keys (g0 + gn, group_size (g0 + gn)), half ^ 1)) return rc;
          if (int rc = dev_tables.use (half, gn, group_kt)) return rc;
        }
      else if (clip_keys)
        {
          keys_of_group = group_keys (g0, gn);
          const auto tk0 = std::chrono::steady_clock::now();
          std::vector<ClipKeyHost> hosts = next_hosts.valid() ? next_hosts.get() : build_group_hosts (keys_of_group);
          const auto tk1 = std::chrono::steady_clock::now();
          if (g0 + gn < which.size())
            {
              const size_t n0 = g0 + gn;
              ParamValues *const pv = &params();
              std::vector<Key> next_keys = group_keys (n0, group_size (n0));
              next_hosts = std::async (std::launch::async, [pv, keys = std::move (next_keys)] { ParamsBind bind (pv); return build_group_hosts (keys); });
            }
          if (int rc = upload_group_tables (lane, hosts, group_kt))
            return rc;
          const auto tk2 = std::chrono::steady_clock::now();
          g_clip_key_us[0] += std::chrono::duration_cast<std::chrono::microseconds> (tk1 - tk0).count();
          g_clip_key_us[1] += std::chrono::duration_cast<std::chrono::microseconds> (tk2 - tk1).count();
          g_clip_key_us[2] += 1;
        }
      bool db_ready = false;                 // the group's dB matrices are shared by the keys
      const std::vector<Key> one_pass { Key() };
      for (const Key& key : (clip_keys ? one_pass : key_list))
        {
          KeyTables *kt = clip_keys ? &group_kt : ctx->get_key_tables (key);
          if (!kt)
            return AWM_ERR_HIP;
          SyncFinder finder (ctx, lane);
          SyncFinder::GroupJob gj;
          std::vector<std::vector<SyncFinder::Score>> scores;
          if (int rc = finder.group_approx_launch (kt, group, int (gn), d_range, gj, db_ready)) return rc;
          db_ready = gj.n_scores > 0;
          if (tables_on_device)
            if (int rc = dev_tables.want_ready (half)) return rc;
          if (clip_keys && all_tables)
            if (int rc = all_tables->want_ready (0)) return rc;
          if (int rc = finder.group_select_refine (gj)) return rc;
          if (int rc = finder.group_finish (gj, scores)) return rc;
          // soft bits and Viterbi decodes of ALL clips of the group in one batch
          std::vector<size_t> index;
          struct Cand { size_t clip; SyncFinder::Score score; };
          std::vector<Cand> cands;
          for (size_t i = 0; i < gn; i++)
            {
              if (gj.fallback[i] || g_group_force_fallback)
                {
                  // the rare sequential selection: this clip's slice on its own (same result set, same order of keys)
                  DeviceWav slice = group;
                  slice.data = group.data + i * slice_values;
                  slice.n_frames = slice_frames;
                  // (that search scans and transforms the whole slice: the zeros the padded copy left out are written now)
                  const long long lo = std::max<long long> (0, (desc[i].pad_start - margin_values) / 4 * 4);
                  const long long hi = std::min<long long> ((long long) slice_values, (desc[i].pad_start + desc[i].n_values + margin_values + 3) / 4 * 4);
                  float *slice_w = lane->ws_clip.as<float>() + i * slice_values;
                  if (lo > 0)
                    AWM_HIP_CHECK (hipMemsetAsync (slice_w, 0, size_t (lo) * sizeof (float), st));
                  if (hi < (long long) slice_values)
                    AWM_HIP_CHECK (hipMemsetAsync (slice_w + hi, 0, size_t ((long long) slice_values - hi) * sizeof (float), st));
                  if (int rc = clip_run_padded (ctx, lane, { clip_keys ? keys_of_group[i] : key }, slice, chunk_sets[g0 + i], 0.0, 1))
                    return rc;
                  db_ready = false;              // (that search used the lane's dB workspace)
                  continue;
                }
              for (const auto& sc : scores[i])
                {
                  // both halves of the long block must lie inside the clip's own slice (fft_range bound, reference wmcommon.cc:128-130)
                  if (sc.index + 2 * count * Params::frame_size > slice_frames)
                    continue;
                  index.push_back (i * slice_frames + sc.index);
                  index.push_back (i * slice_frames + sc.index + count * Params::frame_size);
                  cands.push_back ({ i, sc });
                }
            }
          std::vector<int> slot_of;
          std::vector<char> ok;
          if (int rc = block_soft_bits_dev (ctx, lane, kt, group, index, slot_of, ok, d_range, slice_frames))
            return rc;
          DecodeJob decode;
          for (size_t k = 0; k < cands.size(); k++)
            {
              if (!ok[2 * k] || !ok[2 * k + 1])
                continue;
              const int first_half = cands[k].score.block_type == ConvBlockType::a ? 0 : 1;
              SyncFinder::Score nopad = cands[k].score;
              nopad.index = 0;                                                     // time offset of the START position is 0
              decode.pending.push_back ({ ConvBlockType::ab, 1, { { slot_of[2 * k], first_half }, { slot_of[2 * k + 1], 1 - first_half } }, 0, 0,
                                          0.0, nopad, ResultSet::Type::CLIP, g0 + cands[k].clip,
                                          clip_keys ? int (cands[k].clip) * n_bits_a : 0 });
            }
          if (int rc = decode_launch (ctx, lane, kt, decode))
            return rc;
          if (clip_keys)
            {
              // every pattern under the key of its clip
              std::vector<DecodedPattern> decoded;
              if (int rc = decode_finish (lane, key, decode, ptrs, 1, &decoded))
                return rc;
              for (const DecodedPattern& d : decoded)
                {
                  const PendingDecode& p = decode.pending[d.pending_index];
                  ptrs[p.chunk]->add_pattern (keys_of_group[p.chunk - g0], p.time, p.score, d.bits, d.error, p.type, 1);
                }
            }
          else if (int rc = decode_finish (lane, key, decode, ptrs, 1))
            return rc;
        }
      g0 += gn;
    }
  drain.ok = true;
  for (size_t j = 0; j < which.size(); j++)
    {
      ResultSet& rs = result_sets[which[j]];                        // as get_watermark_device: one chunk at offset 0, merge, sort
      chunk_sets[j].apply_time_offset (0.0);
      rs.merge (chunk_sets[j]);
      rs.sort (clip_keys ? std::vector<Key> { (*clip_keys)[which[j]] } : key_list);
    }
  return 0;
}

/* get_watermark for many independent inputs (BASELINE config 5: a batch of short clips).  A clip's `get` is a chain of
 * small, latency bound steps (two padded CLIP searches, two one-CU Viterbi decodes, ~10 host round trips: 2 ms for a
 * 30 s clip that keeps the GPU busy for a fraction of that), so the clips are spread over the context's lanes, each
 * lane driven by its own host thread that pulls the next clip when it is done.  Results are those of
 * get_watermark_device per clip. */
int
get_watermark_batch_device (awm_ctx *ctx, const std::vector<Key>& key_list, const std::vector<DeviceWav>& clips,
                            std::vector<ResultSet>& result_sets, int n_threads, const std::vector<Key> *clip_keys)
{
  // clip_keys: one key per clip (clip i is decoded with (*clip_keys)[i] alone); else every clip with the whole key_list
  auto keys_of = [&] (size_t i) { return clip_keys ? std::vector<Key> { (*clip_keys)[i] } : key_list; };
  result_sets.clear();
  result_sets.resize (clips.size());
  if (clips.empty())
    return 0;
  // short clips: staged over the lanes from this thread; everything else: one clip per lane and host thread
  std::vector<size_t> staged, threaded;
  if (params().detect_speed || params().detect_speed_patient || params().try_speed > 0)
    {
      // the speed search and the stretched copy of a clip live in per-context buffers: one clip after the other
      for (size_t i = 0; i < clips.size(); i++)
        if (int rc = get_watermark_on (ctx, ctx, true, keys_of (i), clips[i], result_sets[i]))
          return rc;
      return 0;
    }
  for (size_t i = 0; i < clips.size(); i++)
    (clip_is_short (clips[i]) ? staged : threaded).push_back (i);
  if (!staged.empty())
    {
      if (!ctx->ev_sync)
        AWM_HIP_CHECK (hipEventCreateWithFlags (&ctx->ev_sync, hipEventDisableTiming));
      AWM_HIP_CHECK (hipEventRecord (ctx->ev_sync, ctx->stream));                   // the clips may still be in flight there
      const int n_staged_threads = std::max (1, std::min<int> (g_staged_threads ? g_staged_threads : (clip_keys && !DeviceGroupTables::possible() ? 2 : 4), int ((staged.size() + CLIP_GROUP - 1) / CLIP_GROUP)));
      if (!clip_keys)
        for (const Key& key : key_list)
          if (!ctx->get_key_tables (key))               // built once, before the workers start
            return AWM_ERR_HIP;
      std::vector<WorkLane *> staged_lanes;
      for (int t = 0; t < n_staged_threads; t++)
        {
          WorkLane *l = ctx->lane (t);
          if (!l)
            {
              set_error ("cannot create a work lane (stream)");
              return AWM_ERR_HIP;
            }
          if (t)
            AWM_HIP_CHECK (hipStreamWaitEvent (l->stream, ctx->ev_sync, 0));
          staged_lanes.push_back (l);
        }
      // one key per clip: every thread builds its groups' tables on a stream beside its lane's (lanes 8 ..: `add`'s batches use 0 .. 8)
      std::vector<WorkLane *> table_lanes (n_staged_threads, nullptr);
      if (clip_keys && DeviceGroupTables::possible())
        for (int t = 0; t < n_staged_threads; t++)
          if (!(table_lanes[t] = ctx->lane (std::min (8 + t, MAX_LANES - 1))))
            {
              set_error ("cannot create a work lane (stream)");
              return AWM_ERR_HIP;
            }
      // ... or (awm_debug_set_key_tables_on_device (2)), for up to 4096 keys (1.5 GB of tables), the tables of ALL keys first, on the context's
      // stream in front of the lanes' work.  Measured: `get` of 1024 clips 201.8 ms against 202.1 with a group's tables one group ahead
      // (193.2 with one key for all clips) -- what a key per clip costs `get` is not K16g beside the groups but every clip reading tables
      // of its own; the default stays the group ahead (two table areas of 64 keys per lane instead of all keys' tables at once).
      DeviceGroupTables all_tables;
      std::vector<int> slot_of_clip;
      const bool all_first = clip_keys && g_key_tables_on_device == 2 && DeviceGroupTables::possible() && staged.size() <= 4096;
      if (all_first)
        {
          std::vector<Key> table_keys;
          slot_of_clip.assign (clips.size(), -1);
          for (size_t i = 0; i < staged.size(); i++)
            {
              slot_of_clip[staged[i]] = int (i);
              table_keys.push_back ((*clip_keys)[staged[i]]);
            }
          AWM_HIP_CHECK (stream_wait (ctx->stream));            // (the table buffers may be re-allocated: nothing of an earlier call reads them)
          if (int rc = all_tables.init (ctx, ctx, ctx->stream, staged.size(), 1)) return rc;
          if (int rc = all_tables.launch (table_keys, 0)) return rc;
        }
      const DeviceGroupTables *all_ptr = all_first ? &all_tables : nullptr;
      // whole groups per thread, dealt round robin
      std::vector<std::vector<size_t>> share (n_staged_threads);
      for (size_t i = 0; i < staged.size(); i++)
        share[(i / CLIP_GROUP) % n_staged_threads].push_back (staged[i]);
      std::vector<int> rcs (n_staged_threads, 0);
      std::vector<std::string> messages (n_staged_threads);          // the error text is per thread
      std::vector<std::thread> workers;
      ParamValues *const pv = &params();
      for (int t = 1; t < n_staged_threads; t++)
        workers.emplace_back ([&, t] {
          ParamsBind bind (pv);
          (void) hipSetDevice (ctx->device);
          rcs[t] = clip_batch_staged (ctx, staged_lanes[t], key_list, clips, share[t], result_sets, clip_keys, table_lanes[t], all_ptr, &slot_of_clip);
          if (rcs[t])
            messages[t] = last_error();
        });
      rcs[0] = clip_batch_staged (ctx, staged_lanes[0], key_list, clips, share[0], result_sets, clip_keys, table_lanes[0], all_ptr, &slot_of_clip);
      for (auto& w : workers)
        w.join();
      for (int t = 0; t < n_staged_threads; t++)
        if (rcs[t])
          {
            if (t)
              set_error (messages[t]);
            return rcs[t];
          }
    }
  if (threaded.empty())
    return 0;
  n_threads = std::max (1, std::min<int> ({ n_threads > 0 ? n_threads : MAX_LANES, MAX_LANES, int (threaded.size()) }));
  std::vector<WorkLane *> lanes;
  for (int i = 0; i < n_threads; i++)
    {
      WorkLane *l = ctx->lane (i);
      if (!l)
        {
          set_error ("cannot create a work lane (stream)");
          return AWM_ERR_HIP;
        }
      lanes.push_back (l);
    }
  if (!clip_keys)
    for (const Key& key : key_list)
      if (!ctx->get_key_tables (key))               // built once, before the workers start
        return AWM_ERR_HIP;
  // the clips may still be in flight on the context's stream
  if (!ctx->ev_sync)
    AWM_HIP_CHECK (hipEventCreateWithFlags (&ctx->ev_sync, hipEventDisableTiming));
  AWM_HIP_CHECK (hipEventRecord (ctx->ev_sync, ctx->stream));
  for (size_t i = 1; i < lanes.size(); i++)
    AWM_HIP_CHECK (hipStreamWaitEvent (lanes[i]->stream, ctx->ev_sync, 0));
  std::atomic<size_t> next { 0 };
  std::vector<int> rc (lanes.size(), 0);
  std::vector<std::string> err (lanes.size());
  ParamValues *const pv = &params();
  auto worker = [&] (size_t li) {
    ParamsBind bind (pv);
    const bool was_blocking = wait_blocking();
    wait_blocking() = lanes.size() > 1;
    struct Restore { bool v; ~Restore() { wait_blocking() = v; } } restore { was_blocking };
    if (hipSetDevice (ctx->device) != hipSuccess)
      {
        rc[li] = AWM_ERR_HIP;
        err[li] = "hipSetDevice failed in a batch worker";
        return;
      }
    for (;;)
      {
        const size_t t = next.fetch_add (1);
        if (t >= threaded.size())
          break;
        const size_t i = threaded[t];
        if (int r = get_watermark_on (ctx, lanes[li], false, keys_of (i), clips[i], result_sets[i]))
          {
            rc[li] = r;
            err[li] = last_error();
            next.store (threaded.size());          // stop the other workers
            break;
          }
      }
  };
  std::vector<std::thread> threads;
  for (size_t li = 1; li < lanes.size(); li++)
    threads.emplace_back (worker, li);
  worker (0);
  for (auto& t : threads)
    t.join();
  for (size_t li = 0; li < lanes.size(); li++)
    if (rc[li])
      {
        set_error (err[li]);
        return rc[li];
      }
  return 0;
}

} // namespace awm
