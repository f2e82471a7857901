// wmshard.cc -- one stream over several GPUs: the per-rank side of awm_sharded_add_d / awm_sharded_get_d (include/awm_hip.h).
//
// `get`.  The reference decodes a long stream chunk by chunk (wavchunkloader.cc:75-84; 30 minutes with 134 s of overlap) and every
// chunk on its own: local mean, peak selection, n_best and the A / B / "all" combination see exactly one chunk (syncfinder.cc:171-558,
// wmget.cc:502-706).  Those decisions stay per chunk here -- results are identical to the single GPU path -- but the WORK of a chunk
// is split by position among the ranks whose spans it covers:
//   scores     sync_decode of a candidate start frame reads the dB rows of ONE block behind it; a rank scores the start frames that
//              lie in its span from its own samples plus one block + 2 frames of its successor's ("overlap stitch": 18 MB stereo)
//   selection  local mean / local maxima / mask / threshold / n_best need the chunk's whole score list: the participants exchange
//              their score segments (32 B per frame of audio: the "score gather") and run the selection redundantly
//   refinement and soft bits of a candidate: by the rank that scored it (same samples)
//   decoding   final sync selection (needs all refined scores: a few hundred bytes), AB pairs and the "all" chain (need the soft bits
//              of blocks on both sides of a span edge: 3.4 KB per block) are built redundantly; the Viterbi jobs are dealt round robin
//   merge      rank 0 collects the patterns and merges the chunks like ResultSet does (wmget.cc:288-316)
// Work per rank is proportional to the length of its span: balanced to within one block, whatever the chunk grid.
//
// `add`: frame spans; one frame of halo each way and a max-reduction of the limiter's per-second maxima.
//
// All data between ranks moves through the caller's awm_comm; sizes are derived from the span list on both sides.
#include "context.hh"
#include "syncfinder.hh"
#include "wmget.hh"
#include "wmdecode.hh"
#include "utils.hh"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <future>
#include <map>
#include <set>

namespace awm {

Key capi_key (const uint8_t key[16]);
// the fused add kernel on a span (capi_kernels.cc): frame_mod_dev = the cached device table of (key, payload)
int add_mix_device (awm_ctx *ctx, const float *pcm_in_d, float *out_d, size_t n_frames, int n_channels, const int8_t *frame_mod_dev,
                    double water_delta, size_t first_frame, const float *halo_before_d, const float *halo_after_d, float *block_max_d,
                    size_t first_block, size_t n_blocks);

namespace {

constexpr size_t FRAME = Params::frame_size;

struct ShardPart
{
  int    chunk, rank;
  size_t first_sf, n_sf;      // candidate start frames [first_sf, first_sf + n_sf) of the chunk
  bool   tail;                // reads the rank's stitched tail buffer (its samples reach into the successor's span)
};

struct ShardPlan
{
  int                     world = 0;
  std::vector<size_t>     start;        // [world + 1] global sample position of every span
  std::vector<ChunkRange> chunks;
  std::vector<long long>  S;            // candidate start frames per chunk (<= 0: the block decoder has no room)
  std::vector<ShardPart>  parts;        // by chunk, then by position
  std::vector<size_t>     tail_lo, tail_hi;   // [world] global range of every rank's tail buffer (lo == hi: none)
  size_t total() const { return start.back(); }
  size_t block_frames = 0;

  // samples a start frame needs, chunk relative: one frame before it (the refinement looks 256 samples back) up to one block + 2
  // frames behind it (sync_fft_parallel transforms one frame more than the last candidate uses, syncfinder.cc:632)
  size_t need_lo (size_t sf) const { return (sf ? sf - 1 : 0) * FRAME; }
  size_t need_hi (size_t sf, size_t chunk_frames) const { return std::min ((sf + block_frames + 2) * FRAME, chunk_frames); }
  int owner (size_t global_pos) const
  {
    for (int r = 0; r < world; r++)
      if (global_pos >= start[r] && global_pos < start[r + 1])
        return r;
    return -1;
  }
};

ShardPlan
make_plan (const uint64_t *span_frames, int world)
{
  ShardPlan p;
  p.world = world;
  p.block_frames = mark_block_frame_count();
  p.start.assign (world + 1, 0);
  for (int r = 0; r < world; r++)
    p.start[r + 1] = p.start[r] + size_t (span_frames[r]);
  p.chunks = plan_chunks (p.total(), 1);
  p.tail_lo.assign (world, 0);
  p.tail_hi.assign (world, 0);
  for (size_t c = 0; c < p.chunks.size(); c++)
    {
      const ChunkRange& ch = p.chunks[c];
      const long long S = (long long) (ch.n_frames / FRAME) - 1 - (long long) p.block_frames;
      p.S.push_back (S);
      if (S <= 0)
        continue;
      // start frame sf belongs to the rank that holds the first sample it needs (position of the frame before it)
      auto first_sf_at = [&] (size_t global) -> size_t {          // smallest sf whose position is >= global
        if (global <= ch.first_frame)
          return 0;
        return (global - ch.first_frame + FRAME - 1) / FRAME + 1;
      };
      for (int r = 0; r < world; r++)
        {
          if (p.start[r] == p.start[r + 1])
            continue;
          const size_t lo = std::min (first_sf_at (p.start[r]), size_t (S));
          const size_t hi = r + 1 == world || p.start[r + 1] >= p.total() ? size_t (S) : std::min (first_sf_at (p.start[r + 1]), size_t (S));
          if (lo >= hi)
            continue;
          // interior: everything the start frame needs lies in the rank's own span
          size_t mid = lo;
          const size_t end_r = p.start[r + 1];
          if (ch.first_frame + ch.n_frames <= end_r)
            mid = hi;                                               // the chunk ends inside the span
          else if (end_r > ch.first_frame && (end_r - ch.first_frame) / FRAME >= p.block_frames + 2)
            mid = std::min (hi, std::max (lo, (end_r - ch.first_frame) / FRAME - (p.block_frames + 2) + 1));
          if (mid > lo)
            p.parts.push_back ({ int (c), r, lo, mid - lo, false });
          if (hi > mid)
            {
              p.parts.push_back ({ int (c), r, mid, hi - mid, true });
              const size_t t_lo = ch.first_frame + p.need_lo (mid), t_hi = ch.first_frame + p.need_hi (hi - 1, ch.n_frames);
              if (p.tail_lo[r] == p.tail_hi[r])
                {
                  p.tail_lo[r] = t_lo;
                  p.tail_hi[r] = t_hi;
                }
              else
                {
                  p.tail_lo[r] = std::min (p.tail_lo[r], t_lo);
                  p.tail_hi[r] = std::max (p.tail_hi[r], t_hi);
                }
            }
        }
    }
  return p;
}

struct Transfer { int src, dst; size_t lo, hi; };      // global sample range

/* the pieces of every rank's tail buffer, by the rank that holds them (src == dst: the rank's own samples) */
std::vector<Transfer>
tail_transfers (const ShardPlan& p)
{
  std::vector<Transfer> out;
  for (int dst = 0; dst < p.world; dst++)
    for (int src = 0; src < p.world; src++)
      {
        const size_t a = std::max (p.tail_lo[dst], p.start[src]), b = std::min (p.tail_hi[dst], p.start[src + 1]);
        if (a < b)
          out.push_back ({ src, dst, a, b });
      }
  return out;
}

int
comm_fail (const char *what)
{
  set_error (std::string ("awm_comm callback failed: ") + what);
  return AWM_ERR_GENERIC;
}

struct Msg { const void *send = nullptr; void *recv = nullptr; size_t bytes = 0; int peer = 0; };
int
run_exchange (const awm_comm *comm, bool device, const std::vector<Msg>& sends, const std::vector<Msg>& recvs, const char *what)
{
  std::vector<const void *> sp;
  std::vector<void *> rp;
  std::vector<size_t> sb, rb;
  std::vector<int> st, rf;
  for (const Msg& m : sends) { sp.push_back (m.send); sb.push_back (m.bytes); st.push_back (m.peer); }
  for (const Msg& m : recvs) { rp.push_back (m.recv); rb.push_back (m.bytes); rf.push_back (m.peer); }
  if (comm->world == 1 && sp.empty() && rp.empty())
    return 0;                                   // (a single rank never has anything to exchange: spare the transport the call)
  auto fn = device ? comm->exchange_d : comm->exchange_h;
  if (fn (comm->user, int (sp.size()), sp.data(), sb.data(), st.data(), int (rp.size()), rp.data(), rb.data(), rf.data()))
    return comm_fail (what);
  return 0;
}

struct ScoreRec { uint32_t cand; uint32_t pad; uint64_t index; double raw_quality, local_mean; };    // a refined candidate
struct PatternRec { int32_t chunk, job; awm_pattern pat; };

PatternRec
make_record (int chunk, size_t job, const PendingDecode& p, const DecodedPattern& d)
{
  PatternRec rec {};
  rec.chunk = chunk;
  rec.job = int32_t (job);
  rec.pat.time = p.time;
  rec.pat.sync_index = p.score.index;
  rec.pat.sync_quality = p.score.quality;
  rec.pat.block_type = int (p.score.block_type);
  rec.pat.type = int (p.type);
  rec.pat.decode_error = d.error;
  rec.pat.speed = 1;
  rec.pat.n_bits = std::min<int> (int (d.bits.size()), 128);
  for (int b = 0; b < rec.pat.n_bits; b++)
    rec.pat.bits[b] = d.bits[b];
  return rec;
}

} // namespace

int
sharded_plan_c (const uint64_t *span_frames, int world, size_t max_out, int *chunk, int *rank, uint64_t *first_sf, uint64_t *n_sf)
{
  const ShardPlan plan = make_plan (span_frames, world);
  // one entry per (chunk, rank): interior and tail part of a rank are adjacent
  size_t n = 0;
  for (size_t c = 0; c < plan.chunks.size(); c++)
    for (int r = 0; r < world; r++)
      {
        size_t lo = 0, cnt = 0;
        for (const ShardPart& pt : plan.parts)
          if (pt.chunk == int (c) && pt.rank == r)
            {
              if (!cnt)
                lo = pt.first_sf;
              cnt += pt.n_sf;
            }
        if (n < max_out)
          {
            chunk[n] = int (c);
            rank[n] = r;
            first_sf[n] = lo;
            n_sf[n] = cnt;
          }
        n++;
      }
  return int (n);
}

/* ---- add ---------------------------------------------------------------------------------------------------------------- */

int
sharded_add (awm_ctx *ctx, const Key& key, const std::string& payload_hex, const float *pcm_in, float *out, int C,
             const uint64_t *span_frames, const awm_comm *comm)
{
  const int rank = comm->rank, world = comm->world;
  std::vector<size_t> start (world + 1, 0);
  for (int r = 0; r < world; r++)
    {
      start[r + 1] = start[r] + size_t (span_frames[r]);
      if (span_frames[r] % FRAME && std::any_of (span_frames + r + 1, span_frames + world, [] (uint64_t n) { return n != 0; }))
        {
          set_error ("awm_sharded_add_d: every span but the last non-empty one must be a whole number of 1024-sample frames");
          return AWM_ERR_ARG;
        }
    }
  const size_t n = size_t (span_frames[rank]), total = start[world];
  FrameModTable *fm = ctx->get_frame_mod (key, payload_hex);
  if (!fm)
    return AWM_ERR_ARG;
  hipStream_t st = ctx->stream;
  const int BS = Params::mark_sample_rate * int (Params::limiter_block_size_ms) / 1000;
  const bool limiter = !params().test_no_limiter;
  // edge frames: my first frame goes to the nearest non-empty rank before me, my last one to the nearest one after me
  int prev = -1, next = -1;
  for (int r = rank - 1; r >= 0 && prev < 0; r--)
    if (span_frames[r])
      prev = r;
  for (int r = rank + 1; r < world && next < 0; r++)
    if (span_frames[r])
      next = r;
  const size_t edge_bytes = FRAME * C * sizeof (float);
  if (int rc = ctx->ws_shard_edge.reserve (4 * edge_bytes)) return rc;
  float *e_first = ctx->ws_shard_edge.as<float>(), *e_last = e_first + FRAME * C, *h_before = e_last + FRAME * C, *h_after = h_before + FRAME * C;
  std::vector<Msg> sends, recvs;
  if (n)
    {
      AWM_HIP_CHECK (hipMemsetAsync (e_first, 0, 2 * edge_bytes, st));
      AWM_HIP_CHECK (hipMemcpyAsync (e_first, pcm_in, std::min (n, FRAME) * C * sizeof (float), hipMemcpyDeviceToDevice, st));
      const size_t last_start = ((n - 1) / FRAME) * FRAME;           // a ragged last frame is zero padded (it can only end the stream)
      AWM_HIP_CHECK (hipMemcpyAsync (e_last, pcm_in + last_start * C, (n - last_start) * C * sizeof (float), hipMemcpyDeviceToDevice, st));
      AWM_HIP_CHECK (stream_wait (st));
      if (prev >= 0) { sends.push_back ({ e_first, nullptr, edge_bytes, prev }); recvs.push_back ({ nullptr, h_before, edge_bytes, prev }); }
      if (next >= 0) { sends.push_back ({ e_last, nullptr, edge_bytes, next }); recvs.push_back ({ nullptr, h_after, edge_bytes, next }); }
    }
  if (int rc = run_exchange (comm, true, sends, recvs, "exchange_d (edge frames)"))
    return rc;
  const size_t n_blocks = total / BS + 2;
  unsigned int *block_max = nullptr;
  if (limiter)
    {
      if (int rc = ctx->ws_block_max.reserve (n_blocks * sizeof (float))) return rc;
      block_max = ctx->ws_block_max.as<unsigned int>();
      if (int rc = awm_add_init_block_max_d (ctx, ctx->ws_block_max.as<float>(), n_blocks)) return rc;
    }
  if (n)
    if (int rc = add_mix_device (ctx, pcm_in, out, n, C, fm->dev.as<int8_t>(), params().water_delta, start[rank] / FRAME,
                                 prev >= 0 ? h_before : nullptr, next >= 0 ? h_after : nullptr, reinterpret_cast<float *> (block_max), 0, n_blocks))
      return rc;
  if (limiter)
    {
      AWM_HIP_CHECK (stream_wait (st));
      if (comm->all_reduce_max_u32_d (comm->user, block_max, n_blocks))            // seconds that straddle span edges
        return comm_fail ("all_reduce_max_u32_d");
      if (n)
        if (int rc = awm_add_limit_d (ctx, out, n, C, start[rank], reinterpret_cast<const float *> (block_max), 0, n_blocks))
          return rc;
    }
  return 0;
}

/* ---- get ---------------------------------------------------------------------------------------------------------------- */

// the "sync_match" line of the last sharded get that this thread ran as rank 0 (the C ABI returns patterns only; the file level `cmp`
// prints the line)
std::string&
last_shard_debug_sync()
{
  static thread_local std::string s;
  return s;
}

namespace {

// what a rank keeps per chunk it takes part in
struct ChunkWork
{
  int       c = 0;
  size_t    N = 0;                           // samples per channel of the chunk
  long long S = 0;
  std::vector<int> ranks;                    // participants, position order
  std::vector<size_t> rank_first, rank_n;    // their start frame ranges
  int       me = -1;                         // my position among them
  const ShardPart *interior = nullptr, *tail = nullptr;
  size_t    q_off = 0;                       // my block [4][rank_n[me]] in ws_shard_q (doubles)
  std::vector<size_t> recv_off;              // the other participants' blocks
  WorkLane *lane = nullptr;
  std::unique_ptr<SyncFinder> finder;
  SyncFinder::SearchJob select_job;
  std::vector<SyncFinder::SearchScore> candidates;
  std::vector<int> cand_owner;               // position in `ranks`
  std::vector<char> cand_tail;
  SyncFinder::SearchJob refine_job[2];       // interior / tail
  std::vector<uint32_t> refine_cand[2];      // candidate numbers of the two jobs
  std::vector<ScoreRec> refined_mine, refined_all;
  std::vector<SyncFinder::Score> scores;     // final sync positions of the chunk
  std::vector<int> score_owner;
  std::vector<char> score_tail;
  std::vector<float> soft_mine, soft_all;    // [blocks][858]
  std::vector<size_t> wanted;                // positions in `scores` of the blocks that fit the chunk
  DecodeJob decode;
  std::vector<size_t> job_number;            // job number (position in the chunk's complete pending list) of decode.pending[i]
};

DeviceWav
virtual_chunk_wav (const float *view, size_t view_first, size_t chunk_frames, int C)
{
  // the chunk as a DeviceWav whose sample 0 lies view_first samples BEFORE `view`: only samples inside the view are ever touched
  // (the plan sizes the views; refine / block ranges are checked against them before every launch)
  DeviceWav w;
  w.data = view - view_first * C;
  w.n_frames = chunk_frames;
  w.n_channels = C;
  w.sample_rate = Params::mark_sample_rate;
  return w;
}

} // namespace

int
sharded_get (awm_ctx *ctx, const Key& key, const float *pcm, int C, const uint64_t *span_frames, const awm_comm *comm, ResultSet& result)
{
  const int rank = comm->rank, world = comm->world;
  if (params().test_no_sync)
    {
      set_error ("awm_sharded_get_d: --test-no-sync is not sharded");
      return AWM_ERR_ARG;
    }
  const bool speed_on = params().detect_speed || params().detect_speed_patient || params().try_speed > 0;
  const ShardPlan plan = make_plan (span_frames, world);
  const size_t total = plan.total(), block = plan.block_frames;
  const size_t my_lo = plan.start[rank], my_hi = plan.start[rank + 1];
  const std::vector<Key> key_list { key };
  KeyTables *kt = ctx->get_key_tables (key);
  if (!kt)
    return AWM_ERR_HIP;
  hipStream_t st0 = ctx->stream;

  /* Short streams (the ClipDecoder's domain: less than 3.1 blocks, wmget.cc:769-884) are not worth splitting: rank 0 gets the
   * samples and decodes alone.  The same when only the FIRST CHUNK is that short (--chunk-size below ~2.7 minutes): the reference
   * runs the ClipDecoder on it (wmget.cc:886-1013), which the position-split phases below do not -- rare enough to pay one gather
   * for "results identical to awm_get_watermark_d" to hold for every chunk size. */
  const bool first_chunk_is_a_clip = !plan.chunks.empty() && plan.chunks[0].n_frames / FRAME < size_t (block * 3.1);
  if (total / FRAME < size_t (block * 3.1) || first_chunk_is_a_clip)
    {
      std::vector<Msg> sends, recvs;
      if (int rc = ctx->ws_shard_tail.reserve (std::max<size_t> (1, rank == 0 ? total * C * sizeof (float) : 0))) return rc;
      float *all = ctx->ws_shard_tail.as<float>();
      if (rank == 0)
        {
          if (my_hi > my_lo)
            AWM_HIP_CHECK (hipMemcpyAsync (all, pcm, (my_hi - my_lo) * C * sizeof (float), hipMemcpyDeviceToDevice, st0));
          for (int r = 1; r < world; r++)
            if (span_frames[r])
              recvs.push_back ({ nullptr, all + plan.start[r] * C, size_t (span_frames[r]) * C * sizeof (float), r });
        }
      else if (my_hi > my_lo)
        sends.push_back ({ pcm, nullptr, (my_hi - my_lo) * C * sizeof (float), 0 });
      AWM_HIP_CHECK (stream_wait (st0));
      if (int rc = run_exchange (comm, true, sends, recvs, "exchange_d (short stream)"))
        return rc;
      if (rank != 0)
        return 0;
      DeviceWav w;
      w.data = all;
      w.n_frames = total;
      w.n_channels = C;
      return get_watermark_device (ctx, key_list, w, result);
    }

  /* ---- phase 1: the overlap stitch -- every rank's tail buffer = the end of its own span + what follows it on the next rank(s) */
  const size_t t_lo = plan.tail_lo[rank], t_hi = plan.tail_hi[rank];
  if (int rc = ctx->ws_shard_tail.reserve (std::max<size_t> (1, (t_hi - t_lo) * C * sizeof (float)))) return rc;
  float *tail_buf = ctx->ws_shard_tail.as<float>();
  const std::vector<Transfer> transfers = tail_transfers (plan);
  for (const Transfer& t : transfers)                       // my own part of my tail buffer
    if (t.src == rank && t.dst == rank)
      AWM_HIP_CHECK (hipMemcpyAsync (tail_buf + (t.lo - t_lo) * C, pcm + (t.lo - my_lo) * C, (t.hi - t.lo) * C * sizeof (float), hipMemcpyDeviceToDevice, st0));
  AWM_HIP_CHECK (stream_wait (st0));                        // (the samples may still be in flight on the context's stream, e.g. add -> get)

  /* ---- speed detection (decode()'s speed part, reference wmget.cc:886-927; wmspeed.cc:622-781 works on a whole chunk: clip selection
   * over the chunk, then the chunk stretched back and decoded) is sharded BY CHUNK: a chunk belongs to the rank that holds most of it,
   * which fetches the rest (nothing, when the spans follow the chunk grid; at most half a chunk per span edge otherwise) and runs the
   * part on a host thread and lanes of its own, beside the position-split protocol of the plain decoders below.  Its patterns join
   * the chunk's list IN FRONT of the plain ones, as in the reference (the order decides which of two equivalent patterns survives
   * the merge and how the ratings add up). */
  struct SpeedChunk { size_t c; DevBuffer buf; const float *data; ResultSet set; std::string report; };
  std::vector<SpeedChunk> speed_chunks;
  struct SpeedFree { std::vector<SpeedChunk>& v; ~SpeedFree() { for (SpeedChunk& s : v) s.buf.release(); } } speed_free { speed_chunks };
  std::string speed_error;
  std::future<int> speed_run;                              // (declared last: on every return path the task ends before the buffers go)
  if (speed_on)
    {
      std::vector<Msg> sends, recvs;
      for (size_t c = 0; c < plan.chunks.size(); c++)
        {
          const size_t c_lo = plan.chunks[c].first_frame, c_hi = c_lo + plan.chunks[c].n_frames;
          int owner = -1;
          size_t best = 0;
          for (int r = 0; r < world; r++)
            {
              const size_t lo = std::max (c_lo, plan.start[r]), hi = std::min (c_hi, plan.start[r + 1]);
              if (hi > lo && hi - lo > best)
                {
                  best = hi - lo;
                  owner = r;
                }
            }
          if (owner < 0)
            continue;
          const size_t lo = std::max (c_lo, my_lo), hi = std::min (c_hi, my_hi);      // my part of the chunk
          if (owner != rank)
            {
              if (hi > lo)
                sends.push_back ({ pcm + (lo - my_lo) * C, nullptr, (hi - lo) * C * sizeof (float), owner });
              continue;
            }
          speed_chunks.emplace_back();
          SpeedChunk& sc = speed_chunks.back();
          sc.c = c;
          if (lo == c_lo && hi == c_hi)
            {
              sc.data = pcm + (c_lo - my_lo) * C;               // all of it is here: a view
              continue;
            }
          if (int rc = sc.buf.reserve ((c_hi - c_lo) * C * sizeof (float))) return rc;
          sc.data = sc.buf.as<float>();
          AWM_HIP_CHECK (hipMemcpyAsync (sc.buf.as<float>() + (lo - c_lo) * C, pcm + (lo - my_lo) * C, (hi - lo) * C * sizeof (float), hipMemcpyDeviceToDevice, st0));
          for (int r = 0; r < world; r++)
            {
              const size_t rl = std::max (c_lo, plan.start[r]), rh = std::min (c_hi, plan.start[r + 1]);
              if (r != rank && rh > rl)
                recvs.push_back ({ nullptr, sc.buf.as<float>() + (rl - c_lo) * C, (rh - rl) * C * sizeof (float), r });
            }
        }
      AWM_HIP_CHECK (stream_wait (st0));
      if (int rc = run_exchange (comm, true, sends, recvs, "exchange_d (chunks for the speed search)"))
        return rc;
      AWM_HIP_CHECK (stream_wait (st0));                   // (the fetched samples are read on other lanes)
      if (!speed_chunks.empty())
        {
          ParamValues *const pv = &params();
          const int lane_base = 2 * std::max (1, std::min (ctx->chunk_lanes, CHUNK_LANES));       // behind the shared and the local chunks' lanes
          const bool want_report = speed_print_results;
          speed_run = std::async (std::launch::async, [&, pv, lane_base, want_report] {
            ParamsBind bind (pv);
            if (hipSetDevice (ctx->device) != hipSuccess)
              return int (AWM_ERR_HIP);
            WorkLane *lane = ctx->lane (lane_base);
            if (!lane)
              {
                speed_error = "cannot create a work lane (stream)";
                return int (AWM_ERR_HIP);
              }
            for (SpeedChunk& sc : speed_chunks)
              {
                DeviceWav cw;
                cw.data = sc.data;
                cw.n_frames = plan.chunks[sc.c].n_frames;
                cw.n_channels = C;
                cw.sample_rate = Params::mark_sample_rate;
                if (int rc = decode_speed_chunk (ctx, lane, sc.set, key_list, cw, sc.c == 0, want_report ? &sc.report : nullptr))
                  {
                    speed_error = last_error();
                    return rc;
                  }
              }
            return 0;
          });
        }
    }

  /* ---- my chunks ---- */
  std::vector<size_t> local_chunks;
  std::vector<ChunkWork> work;
  // (state of the local chunks' decoder thread; declared before the future so that it outlives the task on every return path)
  std::vector<ChunkRange> local_ranges;
  DeviceWav local_wav;
  std::vector<ResultSet> local_sets;
  std::string local_dbg, local_error;
  std::future<int> local_run;
  size_t q_total = 0;
  for (size_t c = 0; c < plan.chunks.size(); c++)
    {
      ChunkWork w;
      w.c = int (c);
      w.N = plan.chunks[c].n_frames;
      w.S = plan.S[c];
      for (const ShardPart& pt : plan.parts)
        if (pt.chunk == int (c))
          {
            if (w.ranks.empty() || w.ranks.back() != pt.rank)
              {
                w.ranks.push_back (pt.rank);
                w.rank_first.push_back (pt.first_sf);
                w.rank_n.push_back (0);
              }
            w.rank_n.back() += pt.n_sf;
            if (pt.rank == rank)
              (pt.tail ? w.tail : w.interior) = &pt;
          }
      for (size_t i = 0; i < w.ranks.size(); i++)
        if (w.ranks[i] == rank)
          w.me = int (i);
      if (w.me < 0)
        continue;
      // a chunk that lies completely inside my span is mine alone and needs nothing from anybody: it takes the plain path below
      // (chunks on concurrent lanes, every stage of one chunk overlapping the others' -- the phases here end when ALL chunks of a rank
      // are through them)
      if (w.ranks.size() == 1 && plan.chunks[c].first_frame >= my_lo && plan.chunks[c].first_frame + plan.chunks[c].n_frames <= my_hi)
        {
          local_chunks.push_back (c);
          continue;
        }
      w.recv_off.assign (w.ranks.size(), 0);
      for (size_t i = 0; i < w.ranks.size(); i++)
        {
          (int (i) == w.me ? w.q_off : w.recv_off[i]) = q_total;
          q_total += 4 * w.rank_n[i];
        }
      work.push_back (std::move (w));
    }
  if (int rc = ctx->ws_shard_q.reserve (std::max<size_t> (1, q_total * sizeof (double)))) return rc;
  double *q_all = ctx->ws_shard_q.as<double>();
  const int n_lanes = std::max (1, std::min (ctx->chunk_lanes, CHUNK_LANES));
  std::vector<WorkLane *> lanes;
  for (int i = 0; i < n_lanes; i++)
    {
      WorkLane *l = ctx->lane (i);
      if (!l)
        {
          set_error ("cannot create a work lane (stream)");
          return AWM_ERR_HIP;
        }
      lanes.push_back (l);
    }
  auto sync_lanes = [&] () -> int {
    for (WorkLane *l : lanes)
      AWM_HIP_CHECK (stream_wait (l->stream));
    return 0;
  };
  // awm_comm's contract for exchange_d: the callback's writes are visible to the CONTEXT's stream when it returns (a transport may
  // complete in stream order there).  What was received is consumed on the lanes' streams, so they are ordered behind the context's.
  auto lanes_after_exchange = [&] () -> int {
    if (!ctx->ev_sync)
      AWM_HIP_CHECK (hipEventCreateWithFlags (&ctx->ev_sync, hipEventDisableTiming));
    AWM_HIP_CHECK (hipEventRecord (ctx->ev_sync, ctx->stream));
    for (WorkLane *l : lanes)
      if (l->stream != ctx->stream)
        AWM_HIP_CHECK (hipStreamWaitEvent (l->stream, ctx->ev_sync, 0));
    return 0;
  };
  struct Drain { std::vector<WorkLane *>& lanes; ~Drain() { for (WorkLane *l : lanes) (void) hipStreamSynchronize (l->stream); } } drain { lanes };
  for (size_t i = 0; i < work.size(); i++)
    {
      work[i].lane = lanes[i % lanes.size()];
      work[i].finder = std::make_unique<SyncFinder> (ctx, work[i].lane);
    }
  // view of a part's buffer: pointer to the chunk relative sample view_first
  auto part_view = [&] (const ChunkWork& w, bool tail, size_t& view_first) -> const float * {
    const size_t first = plan.chunks[w.c].first_frame;
    const size_t buf_lo = tail ? t_lo : my_lo;
    const float *buf = tail ? tail_buf : pcm;
    const size_t g = std::max (first, buf_lo);
    view_first = g - first;
    return buf + (g - buf_lo) * C;
  };
  auto view_hi = [&] (const ChunkWork& w, bool tail) -> size_t {                // chunk relative end of the buffer
    const size_t first = plan.chunks[w.c].first_frame;
    return std::min (tail ? t_hi : my_hi, first + w.N) - first;
  };

  /* ---- phase 2: scores of my start frames (K4 + K5w on the part's samples), into my block [4][n] of the chunk.  The INTERIOR parts
   * need nothing from anybody: their kernels are queued before the stitch, so that the device works while the 18 MB per boundary
   * travel; the TAIL parts follow when the tail buffer is complete. */
  auto score_parts = [&] (bool tail_parts) -> int {
  for (ChunkWork& w : work)
    for (const ShardPart *pt : { tail_parts ? w.tail : w.interior })
      {
        if (!pt)
          continue;
        size_t vf = 0;
        const float *view = part_view (w, pt->tail, vf);
        const bool last = pt->first_sf + pt->n_sf == size_t (w.S);
        DeviceWav sub;
        sub.n_channels = C;
        sub.data = view + (pt->first_sf * FRAME - vf) * C;
        // whole frames: n_sf + block + 1 of them give n_sf scores (syncfinder.cc:632); a ragged tail of the chunk is never read
        sub.n_frames = last ? (w.N / FRAME - pt->first_sf) * FRAME : (pt->n_sf + block + 1) * FRAME;
        if (pt->first_sf * FRAME < vf || pt->first_sf * FRAME + sub.n_frames > view_hi (w, pt->tail))
          {
            set_error ("awm_sharded_get_d: internal error (part outside its buffer)");
            return AWM_ERR_GENERIC;
          }
        if (int rc = w.finder->prepare (sub, SyncFinder::Mode::BLOCK)) return rc;
        long long n_scores = 0;
        if (int rc = w.finder->approx_device (kt, sub, SyncFinder::Mode::BLOCK, n_scores, false, /* scores_only */ true)) return rc;
        if (n_scores != 4 * (long long) pt->n_sf)
          {
            set_error ("awm_sharded_get_d: internal error (score count of a part)");
            return AWM_ERR_GENERIC;
          }
        const size_t mine_n = w.rank_n[w.me], col = pt->first_sf - w.rank_first[w.me];
        const size_t q_stride = (pt->n_sf + 63) & ~size_t (63);
        AWM_HIP_CHECK (hipMemcpy2DAsync (q_all + w.q_off + col, mine_n * sizeof (double), w.lane->ws_q.ptr, q_stride * sizeof (double),
                                         pt->n_sf * sizeof (double), 4, hipMemcpyDeviceToDevice, w.lane->stream));
      }
  return 0;
  };
  /* ---- the chunks that lie completely inside my span need nothing from anybody: the plain BlockDecoder, on a host thread and a set of
   * lanes of their own, BESIDE the shared chunks' phases (whose waits -- for the other ranks, for the lanes at the end of a phase --
   * they fill; until round 4 they ran after phase 8 and a rank's shared and local work were serialised) */
  if (!local_chunks.empty())
    {
      for (size_t c : local_chunks)
        local_ranges.push_back ({ plan.chunks[c].first_frame - my_lo, plan.chunks[c].n_frames, 0.0 });
      local_wav.data = pcm;
      local_wav.n_frames = my_hi - my_lo;
      local_wav.n_channels = C;
      ParamValues *const pv = &params();
      const int lane_base = int (lanes.size());
      local_run = std::async (std::launch::async, [&, pv, lane_base] {
        ParamsBind bind (pv);
        (void) hipSetDevice (ctx->device);
        const int rc = decode_chunks_blocks_only (ctx, key_list, local_wav, local_ranges, local_sets, &local_dbg, lane_base);
        if (rc)
          local_error = last_error();
        return rc;
      });
    }
  if (int rc = score_parts (false)) return rc;
  {
    std::vector<Msg> sends, recvs;
    for (const Transfer& t : transfers)
      {
        const size_t bytes = (t.hi - t.lo) * C * sizeof (float);
        if (t.src == rank && t.dst == rank)
          continue;
        else if (t.src == rank)
          sends.push_back ({ pcm + (t.lo - my_lo) * C, nullptr, bytes, t.dst });
        else if (t.dst == rank)
          recvs.push_back ({ nullptr, tail_buf + (t.lo - t_lo) * C, bytes, t.src });
      }
    if (int rc = run_exchange (comm, true, sends, recvs, "exchange_d (overlap stitch)"))
      return rc;
    if (int rc = lanes_after_exchange()) return rc;
  }
  if (int rc = score_parts (true)) return rc;
  if (int rc = sync_lanes()) return rc;

  /* ---- phase 3: the score gather among the participants of every chunk */
  {
    std::vector<Msg> sends, recvs;
    for (ChunkWork& w : work)
      for (size_t i = 0; i < w.ranks.size(); i++)
        if (int (i) != w.me)
          {
            sends.push_back ({ q_all + w.q_off, nullptr, 4 * w.rank_n[w.me] * sizeof (double), w.ranks[i] });
            recvs.push_back ({ nullptr, q_all + w.recv_off[i], 4 * w.rank_n[i] * sizeof (double), w.ranks[i] });
          }
    if (int rc = run_exchange (comm, true, sends, recvs, "exchange_d (score gather)"))
      return rc;
    if (int rc = lanes_after_exchange()) return rc;
  }

  /* ---- phase 4: selection on the complete list (every participant, same result), then refinement of MY candidates */
  const double threshold = params().sync_threshold2 * 0.75;
  // (a lane's buffers -- score lists, peak lists, refinement rows and their page-locked staging -- serve one chunk at a time: rounds of
  // one chunk per lane)
  for (size_t w0 = 0; w0 < work.size(); w0 += lanes.size())
    {
      const size_t w1 = std::min (work.size(), w0 + lanes.size());
      for (size_t wi_ = w0; wi_ < w1; wi_++)
        {
          ChunkWork& w = work[wi_];
          const size_t q_stride = (size_t (w.S) + 63) & ~size_t (63);
          if (int rc = w.lane->ws_q.reserve (4 * q_stride * sizeof (double))) return rc;
          for (size_t i = 0; i < w.ranks.size(); i++)
            {
              const double *src = q_all + (int (i) == w.me ? w.q_off : w.recv_off[i]);
              AWM_HIP_CHECK (hipMemcpy2DAsync (w.lane->ws_q.as<double>() + (w.rank_first[i] - w.rank_first[0]), q_stride * sizeof (double), src,
                                               w.rank_n[i] * sizeof (double), w.rank_n[i] * sizeof (double), 4, hipMemcpyDeviceToDevice, w.lane->stream));
            }
          if (int rc = w.finder->scores_loaded (w.S)) return rc;
          if (int rc = w.finder->select_launch (4 * w.S, threshold, false)) return rc;
        }
      for (size_t wi_ = w0; wi_ < w1; wi_++)
        {
          ChunkWork& w = work[wi_];
          if (int rc = w.finder->select_finish (4 * w.S, threshold, w.candidates, false)) return rc;
          // whose candidate is it?  (start frame -> participant; interior or tail part for mine)
          for (size_t k = 0; k < w.candidates.size(); k++)
            {
              const size_t sf = w.candidates[k].index / FRAME;
              int owner = -1;
              for (size_t i = 0; i < w.ranks.size(); i++)
                if (sf >= w.rank_first[i] && sf < w.rank_first[i] + w.rank_n[i])
                  owner = int (i);
              w.cand_owner.push_back (owner);
              const bool tail = owner == w.me && w.tail && sf >= w.tail->first_sf;
              w.cand_tail.push_back (tail);
              if (owner == w.me)
                {
                  w.refine_job[tail].candidates.push_back (w.candidates[k]);
                  w.refine_cand[tail].push_back (uint32_t (k));
                }
            }
          for (int tail = 0; tail < 2; tail++)
            {
              if (w.refine_job[tail].candidates.empty())
                continue;
              size_t vf = 0;
              const float *view = part_view (w, tail, vf);
              const size_t vh = view_hi (w, tail);
              for (const auto& cand : w.refine_job[tail].candidates)
                {
                  const size_t lo = cand.index > size_t (Params::sync_search_step) ? cand.index - Params::sync_search_step : 0;
                  const size_t hi = std::min (cand.index + Params::sync_search_step + (block + 1) * FRAME, w.N);
                  if (lo < vf || hi > vh)
                    {
                      set_error ("awm_sharded_get_d: internal error (refinement outside the part's buffer)");
                      return AWM_ERR_GENERIC;
                    }
                }
              w.refine_job[tail].slot = tail;
              const DeviceWav vw = virtual_chunk_wav (view, vf, w.N, C);
              if (int rc = w.finder->prepare (vw, SyncFinder::Mode::BLOCK)) return rc;
              if (int rc = w.finder->refine_launch (kt, vw, SyncFinder::Mode::BLOCK, w.refine_job[tail])) return rc;
            }
        }
      for (size_t wi_ = w0; wi_ < w1; wi_++)
        for (int tail = 0; tail < 2; tail++)
          {
            ChunkWork& w = work[wi_];
            if (w.refine_job[tail].candidates.empty())
              continue;
            if (int rc = w.finder->refine_collect (w.refine_job[tail])) return rc;
            const auto& refined = w.refine_job[tail].refined;               // candidate order
            for (size_t k = 0; k < refined.size(); k++)
              w.refined_mine.push_back ({ w.refine_cand[tail][k], 0, uint64_t (refined[k].index), refined[k].raw_quality, refined[k].local_mean });
          }
    }

  /* ---- phase 5: the refined scores travel to the other participants (a few dozen records per chunk) */
  {
    std::vector<Msg> sends, recvs;
    std::vector<std::vector<ScoreRec>> inbox;
    inbox.reserve (work.size() * world);
    for (ChunkWork& w : work)
      for (size_t i = 0; i < w.ranks.size(); i++)
        if (int (i) != w.me)
          {
            const size_t theirs = size_t (std::count (w.cand_owner.begin(), w.cand_owner.end(), int (i)));
            if (!w.refined_mine.empty())
              sends.push_back ({ w.refined_mine.data(), nullptr, w.refined_mine.size() * sizeof (ScoreRec), w.ranks[i] });
            if (theirs)
              {
                inbox.emplace_back (theirs);
                recvs.push_back ({ nullptr, inbox.back().data(), theirs * sizeof (ScoreRec), w.ranks[i] });
              }
          }
    if (int rc = run_exchange (comm, false, sends, recvs, "exchange_h (refined scores)"))
      return rc;
    size_t ib = 0;
    for (ChunkWork& w : work)
      {
        w.refined_all = w.refined_mine;
        for (size_t i = 0; i < w.ranks.size(); i++)
          if (int (i) != w.me && std::count (w.cand_owner.begin(), w.cand_owner.end(), int (i)))
            {
              w.refined_all.insert (w.refined_all.end(), inbox[ib].begin(), inbox[ib].end());
              ib++;
            }
      }
  }

  /* ---- phase 6: final sync positions of every chunk (SyncFinder::search_finish on the complete refined list, in candidate
   * order), soft bits of MY blocks */
  const int n_bits = int (mark_data_frame_count() / params().frames_per_bit);
  // (device -> host staging of the soft bits: the lane's page-locked buffer, one region per launch; sized for every block of the
  // lane's chunks up front so that it never moves while copies are in flight)
  struct SoftCopy { ChunkWork *w; std::vector<size_t> blocks; std::vector<int> slot_of; const float *host; };
  std::vector<SoftCopy> soft_copies;
  std::map<WorkLane *, size_t> pin_used;
  for (ChunkWork& w : work)
    {
      std::sort (w.refined_all.begin(), w.refined_all.end(), [] (const ScoreRec& a, const ScoreRec& b) { return a.cand < b.cand; });
      if (w.refined_all.size() != w.candidates.size())
        {
          set_error ("awm_sharded_get_d: internal error (refined scores missing)");
          return AWM_ERR_GENERIC;
        }
      std::vector<SyncFinder::SearchScore> refined;
      for (const ScoreRec& r : w.refined_all)
        refined.push_back ({ size_t (r.index), r.raw_quality, r.local_mean });
      // (which candidate a final score came from: by (index, quality) -- equal twins have the same owner or are interchangeable)
      SyncFinder::finish_scores (refined, w.scores);
      for (const auto& sc : w.scores)
        {
          int owner = -1;
          bool tail = false;
          for (size_t k = 0; k < w.refined_all.size(); k++)
            {
              const ScoreRec& r = w.refined_all[k];
              if (size_t (r.index) == sc.index && std::fabs (r.raw_quality - r.local_mean) == sc.quality)
                {
                  owner = w.cand_owner[r.cand];
                  tail = w.cand_tail[r.cand];
                  break;
                }
            }
          w.score_owner.push_back (owner);
          w.score_tail.push_back (tail);
        }
      // blocks that fit the chunk (fft_range bound, wmcommon.cc:128-130), in score order; mine by part
      for (size_t k = 0; k < w.scores.size(); k++)
        if (w.N >= w.scores[k].index + block * FRAME)
          w.wanted.push_back (k);
      pin_used[w.lane] += w.wanted.size() * n_bits * sizeof (float);
    }
  for (auto& pu : pin_used)
    {
      if (int rc = pu.first->pin_shard.reserve (std::max<size_t> (1, pu.second))) return rc;
      pu.second = 0;
    }
  std::set<WorkLane *> lanes_in_use;
  for (ChunkWork& w : work)
    {
      for (int tail = 0; tail < 2; tail++)
        {
          SoftCopy sc { &w, {}, {}, nullptr };
          size_t& used = pin_used[w.lane];
          std::vector<size_t> index;
          for (size_t wi = 0; wi < w.wanted.size(); wi++)
            {
              const size_t k = w.wanted[wi];
              if (w.score_owner[k] == w.me && bool (w.score_tail[k]) == bool (tail))
                {
                  sc.blocks.push_back (wi);
                  index.push_back (w.scores[k].index);
                }
            }
          if (index.empty())
            continue;
          size_t vf = 0;
          const float *view = part_view (w, tail, vf);
          const size_t vh = view_hi (w, tail);
          for (size_t idx : index)
            if (idx < vf || idx + block * FRAME > vh)
              {
                set_error ("awm_sharded_get_d: internal error (block outside the part's buffer)");
                return AWM_ERR_GENERIC;
              }
          std::vector<char> ok;
          // (block_soft_bits_dev stages the block list through the lane's page-locked buffer and its soft bits through the lane's
          // ws_soft: ANY second call on the same lane -- the tail part after the interior part, or the next chunk of a rank that has more
          // shared chunks than lanes -- must not refill them before the previous call's copies have run)
          if (!lanes_in_use.insert (w.lane).second)
            AWM_HIP_CHECK (stream_wait (w.lane->stream));
          if (int rc = block_soft_bits_dev (ctx, w.lane, kt, virtual_chunk_wav (view, vf, w.N, C), index, sc.slot_of, ok))
            return rc;
          float *host = reinterpret_cast<float *> (w.lane->pin_shard.as<char>() + used);
          used += index.size() * n_bits * sizeof (float);
          sc.host = host;
          AWM_HIP_CHECK (hipMemcpyAsync (host, w.lane->ws_soft.ptr, index.size() * n_bits * sizeof (float), hipMemcpyDeviceToHost, w.lane->stream));
          soft_copies.push_back (std::move (sc));
        }
    }
  if (int rc = sync_lanes()) return rc;
  for (ChunkWork& w : work)
    w.soft_all.assign (w.wanted.size() * n_bits, 0.f);
  for (SoftCopy& sc : soft_copies)
    for (size_t j = 0; j < sc.blocks.size(); j++)
      std::copy (sc.host + size_t (sc.slot_of[j]) * n_bits, sc.host + size_t (sc.slot_of[j] + 1) * n_bits,
                 sc.w->soft_all.begin() + sc.blocks[j] * n_bits);

  // the "sync_match" line of `cmp` (BlockDecoder::run, wmget.cc:708-733): sync positions of the first chunk against where the blocks
  // of an uncut file start -- every participant of chunk 0 knows them; rank 0 is one unless its span is empty
  std::string debug_sync;
  for (ChunkWork& w : work)
    if (w.c == 0)
      {
        const int expect0 = Params::frames_pad_start * Params::frame_size, expect_step = int (block * FRAME);
        const int expect_end = int (w.N / FRAME) * int (FRAME);
        int sync_match = 0;
        for (int expect_index = expect0; expect_index + expect_step < expect_end; expect_index += expect_step)
          for (const auto& sc : w.scores)
            if (std::abs (int (sc.index + params().test_cut) - expect_index) < int (FRAME / 2))
              {
                sync_match++;
                break;
              }
        debug_sync = string_printf ("sync_match %d %zd\n", sync_match, w.scores.size());
      }

  /* ---- phase 7: the blocks' soft bits go to the other participants (AB pairs and the "all" chain reach across span edges) */
  {
    std::vector<Msg> sends, recvs;
    std::vector<std::vector<float>> outbox, inbox;
    struct In { ChunkWork *w; int from; };
    std::vector<In> in_of;
    outbox.reserve (work.size());
    inbox.reserve (work.size() * world);
    for (ChunkWork& w : work)
      {
        if (w.ranks.size() < 2)
          continue;
        outbox.emplace_back();
        std::vector<float>& mine = outbox.back();
        for (size_t wi = 0; wi < w.wanted.size(); wi++)
          if (w.score_owner[w.wanted[wi]] == w.me)
            mine.insert (mine.end(), w.soft_all.begin() + wi * n_bits, w.soft_all.begin() + (wi + 1) * n_bits);
        for (size_t i = 0; i < w.ranks.size(); i++)
          if (int (i) != w.me)
            {
              if (!mine.empty())
                sends.push_back ({ mine.data(), nullptr, mine.size() * sizeof (float), w.ranks[i] });
              size_t theirs = 0;
              for (size_t wi = 0; wi < w.wanted.size(); wi++)
                theirs += w.score_owner[w.wanted[wi]] == int (i);
              if (theirs)
                {
                  inbox.emplace_back (theirs * n_bits);
                  in_of.push_back ({ &w, int (i) });
                  recvs.push_back ({ nullptr, inbox.back().data(), theirs * n_bits * sizeof (float), w.ranks[i] });
                }
            }
      }
    if (int rc = run_exchange (comm, false, sends, recvs, "exchange_h (soft bits)"))
      return rc;
    for (size_t b = 0; b < inbox.size(); b++)
      {
        ChunkWork& w = *in_of[b].w;
        size_t pos = 0;
        for (size_t wi = 0; wi < w.wanted.size(); wi++)
          if (w.score_owner[w.wanted[wi]] == in_of[b].from)
            {
              std::copy (inbox[b].begin() + pos * n_bits, inbox[b].begin() + (pos + 1) * n_bits, w.soft_all.begin() + wi * n_bits);
              pos++;
            }
      }
  }

  /* ---- phase 8: the chunk's decode jobs (single blocks, AB pairs, "all": BlockDecoder::run, wmget.cc:554-701), dealt round robin
   * among the participants.  A lane decodes one chunk at a time (its decoder buffers): rounds of one chunk per lane. */
  std::vector<PatternRec> my_patterns;
  auto decode_start = [&] (ChunkWork& w) -> int {
    DeviceWav tw;
    tw.sample_rate = Params::mark_sample_rate;
    std::vector<PatternRawBits> raw_vec;
    std::vector<PendingDecode> pending;
    for (size_t wi = 0; wi < w.wanted.size(); wi++)
      {
        const SyncFinder::Score& sc = w.scores[w.wanted[wi]];
        raw_vec.push_back ({ sc.index, sc.quality, int (wi), sc.block_type });
        pending.push_back ({ sc.block_type, 0, { { int (wi), 0 } }, 0, 0, double (sc.index) / tw.sample_rate, sc, ResultSet::Type::BLOCK, 0 });
      }
    combine_blocks (raw_vec, tw, 0, pending);
    for (size_t j = 0; j < pending.size(); j++)
      if (int (j % w.ranks.size()) == w.me)
        {
          w.decode.pending.push_back (pending[j]);
          w.job_number.push_back (j);
        }
    if (w.decode.pending.empty())
      return 0;
    if (int rc = w.lane->ws_soft.reserve (std::max<size_t> (1, w.soft_all.size() * sizeof (float)))) return rc;
    if (int rc = w.lane->pin_shard_up.reserve (std::max<size_t> (1, w.soft_all.size() * sizeof (float)))) return rc;
    std::copy (w.soft_all.begin(), w.soft_all.end(), w.lane->pin_shard_up.as<float>());
    AWM_HIP_CHECK (hipMemcpyAsync (w.lane->ws_soft.ptr, w.lane->pin_shard_up.ptr, w.soft_all.size() * sizeof (float), hipMemcpyHostToDevice, w.lane->stream));
    return decode_launch (ctx, w.lane, kt, w.decode);
  };
  auto decode_collect = [&] (ChunkWork& w) -> int {
    std::vector<DecodedPattern> decoded;
    if (int rc = decode_finish (w.lane, key, w.decode, {}, 1, &decoded)) return rc;
    for (const DecodedPattern& d : decoded)
      my_patterns.push_back (make_record (w.c, w.job_number[d.pending_index], w.decode.pending[d.pending_index], d));
    return 0;
  };
  for (size_t w0 = 0; w0 < work.size(); w0 += lanes.size())
    {
      const size_t w1 = std::min (work.size(), w0 + lanes.size());
      for (size_t i = w0; i < w1; i++)
        if (int rc = decode_start (work[i])) return rc;
      for (size_t i = w0; i < w1; i++)
        if (int rc = decode_collect (work[i])) return rc;
    }

  /* ---- the speed part of my chunks: its patterns come first in their chunk's list (job numbers below every plain one) */
  std::string my_reports;                                  // "chunk <c>\n<report text>" records, merged by rank 0 in chunk order
  if (speed_run.valid())
    {
      if (int rc = speed_run.get())
        {
          set_error (speed_error);
          return rc;
        }
      for (SpeedChunk& sc : speed_chunks)
        {
          for (size_t j = 0; j < sc.set.patterns.size(); j++)
            {
              const ResultSet::Pattern& p = sc.set.patterns[j];
              PatternRec rec {};
              rec.chunk = int32_t (sc.c);
              rec.job = int32_t (j) - (1 << 24);
              rec.pat.time = p.time;
              rec.pat.sync_index = p.sync_score.index;
              rec.pat.sync_quality = p.sync_score.quality;
              rec.pat.block_type = int (p.sync_score.block_type);
              rec.pat.type = int (p.type);
              rec.pat.decode_error = p.decode_error;
              rec.pat.speed = p.speed;
              rec.pat.n_bits = std::min<int> (int (p.bit_vec.size()), 128);
              for (int b = 0; b < rec.pat.n_bits; b++)
                rec.pat.bits[b] = p.bit_vec[b];
              my_patterns.push_back (rec);
            }
          if (!sc.report.empty())
            my_reports += string_printf ("%zu\n", sc.c) + sc.report + '\0';
        }
    }

  /* ---- the chunks inside my span: collect what the decoder thread found */
  if (local_run.valid())
    {
      if (int rc = local_run.get())
        {
          set_error (local_error);
          return rc;
        }
      if (local_chunks[0] == 0)
        debug_sync = local_dbg;
      for (size_t i = 0; i < local_sets.size(); i++)
        for (size_t j = 0; j < local_sets[i].patterns.size(); j++)
          {
            const ResultSet::Pattern& p = local_sets[i].patterns[j];
            PatternRec rec {};
            rec.chunk = int32_t (local_chunks[i]);
            rec.job = int32_t (j);                               // (submission order of the chunk's patterns)
            rec.pat.time = p.time;
            rec.pat.sync_index = p.sync_score.index;
            rec.pat.sync_quality = p.sync_score.quality;
            rec.pat.block_type = int (p.sync_score.block_type);
            rec.pat.type = int (p.type);
            rec.pat.decode_error = p.decode_error;
            rec.pat.speed = p.speed;
            rec.pat.n_bits = std::min<int> (int (p.bit_vec.size()), 128);
            for (int b = 0; b < rec.pat.n_bits; b++)
              rec.pat.bits[b] = p.bit_vec[b];
            my_patterns.push_back (rec);
          }
    }

  /* ---- phase 9: rank 0 collects the patterns and merges the chunks (ResultSet, wmget.cc:288-316) */
  std::vector<uint64_t> counts (world, 0);
  {
    std::vector<Msg> sends, recvs;
    uint64_t mine = my_patterns.size();
    if (rank == 0)
      {
        counts[0] = mine;
        for (int r = 1; r < world; r++)
          recvs.push_back ({ nullptr, &counts[r], sizeof (uint64_t), r });
      }
    else
      sends.push_back ({ &mine, nullptr, sizeof (uint64_t), 0 });
    if (int rc = run_exchange (comm, false, sends, recvs, "exchange_h (pattern counts)"))
      return rc;
  }
  std::vector<std::vector<PatternRec>> from (world);
  {
    std::vector<Msg> sends, recvs;
    if (rank == 0)
      for (int r = 1; r < world; r++)
        {
          from[r].resize (counts[r]);
          if (counts[r])
            recvs.push_back ({ nullptr, from[r].data(), counts[r] * sizeof (PatternRec), r });
        }
    else if (!my_patterns.empty())
      sends.push_back ({ my_patterns.data(), nullptr, my_patterns.size() * sizeof (PatternRec), 0 });
    if (int rc = run_exchange (comm, false, sends, recvs, "exchange_h (patterns)"))
      return rc;
  }
  if (speed_on)
    {
      // the `detect_speed` report lines of `cmp` (decode() prints one per chunk and key, wmget.cc:902): to rank 0, printed in chunk order
      std::vector<uint64_t> sizes (world, 0);
      std::vector<std::string> texts (world);
      {
        std::vector<Msg> sends, recvs;
        uint64_t mine = my_reports.size();
        if (rank == 0)
          for (int r = 1; r < world; r++)
            recvs.push_back ({ nullptr, &sizes[r], sizeof (uint64_t), r });
        else
          sends.push_back ({ &mine, nullptr, sizeof (uint64_t), 0 });
        if (int rc = run_exchange (comm, false, sends, recvs, "exchange_h (report sizes)"))
          return rc;
      }
      {
        std::vector<Msg> sends, recvs;
        if (rank == 0)
          for (int r = 1; r < world; r++)
            {
              texts[r].resize (sizes[r]);
              if (sizes[r])
                recvs.push_back ({ nullptr, &texts[r][0], size_t (sizes[r]), r });
            }
        else if (!my_reports.empty())
          sends.push_back ({ my_reports.data(), nullptr, my_reports.size(), 0 });
        if (int rc = run_exchange (comm, false, sends, recvs, "exchange_h (reports)"))
          return rc;
      }
      if (rank == 0)
        {
          texts[0] = my_reports;
          std::map<size_t, std::string> by_chunk;
          for (const std::string& t : texts)
            for (size_t pos = 0; pos < t.size(); )
              {
                const size_t end = t.find ('\0', pos), nl = t.find ('\n', pos);
                if (end == std::string::npos || nl == std::string::npos || nl > end)
                  break;
                by_chunk[size_t (std::strtoull (t.substr (pos, nl - pos).c_str(), nullptr, 10))] += t.substr (nl + 1, end - nl - 1);
                pos = end + 1;
              }
          for (const auto& kv : by_chunk)
            fputs (kv.second.c_str(), stdout);
        }
    }
  if (rank != 0)
    return 0;
  from[0] = std::move (my_patterns);
  std::vector<std::vector<PatternRec>> per_chunk (plan.chunks.size());
  for (auto& list : from)
    for (const PatternRec& rec : list)
      if (rec.chunk >= 0 && size_t (rec.chunk) < per_chunk.size())
        per_chunk[rec.chunk].push_back (rec);
  for (size_t c = 0; c < per_chunk.size(); c++)
    {
      auto& recs = per_chunk[c];
      std::sort (recs.begin(), recs.end(), [] (const PatternRec& a, const PatternRec& b) { return a.job < b.job; });     // submission order
      ResultSet chunk;
      for (const PatternRec& rec : recs)
        {
          const awm_pattern& p = rec.pat;
          SyncFinder::Score score { size_t (p.sync_index), p.sync_quality, ConvBlockType (p.block_type) };
          chunk.add_pattern (key, p.time, score, std::vector<int> (p.bits, p.bits + p.n_bits), p.decode_error, ResultSet::Type (p.type), p.speed);
        }
      chunk.apply_time_offset (plan.chunks[c].time_offset);
      result.merge (chunk);
    }
  result.sort (key_list);
  result.set_debug_sync (debug_sync);
  last_shard_debug_sync() = debug_sync;
  return 0;
}

} // namespace awm
