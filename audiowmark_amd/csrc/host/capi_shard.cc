// C ABI of the multi-GPU protocol (include/awm_hip.h: awm_sharded_*, awm_multi_*): the per-rank entry points with the caller's
// transport, and the single-process driver that runs one host thread per context with hipMemcpyPeer as transport.
#include "context.hh"
#include "wmget.hh"
#include "utils.hh"
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

namespace awm {
Key capi_key (const uint8_t key[16]);
int sharded_plan_c (const uint64_t *span_frames, int world, size_t max_out, int *chunk, int *rank, uint64_t *first_sf, uint64_t *n_sf);
int sharded_add (awm_ctx *ctx, const Key& key, const std::string& payload_hex, const float *pcm_in, float *out, int C,
                 const uint64_t *span_frames, const awm_comm *comm);
int sharded_get (awm_ctx *ctx, const Key& key, const float *pcm, int C, const uint64_t *span_frames, const awm_comm *comm, ResultSet& result);
}

using namespace awm;

namespace {

int
enter (awm_ctx *ctx, const awm_comm *comm, const uint64_t *span_frames, int n_channels)
{
  if (!ctx || !comm || !span_frames || n_channels < 1 || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world
      || !comm->exchange_d || !comm->exchange_h || !comm->all_reduce_max_u32_d)
    {
      set_error ("awm_sharded_*: bad argument");
      return AWM_ERR_ARG;
    }
  if (hipSetDevice (ctx->device) != hipSuccess)
    {
      set_error ("hipSetDevice failed");
      return AWM_ERR_HIP;
    }
  if (params().frames_per_bit < 1 || params().frames_per_bit > 8 || params().payload_size != 128)      // (as check_ctx, capi_kernels.cc)
    {
      set_error ("unsupported watermark parameters (frames_per_bit outside 1 .. 8 or a short payload)");
      return AWM_ERR_ARG;
    }
  return 0;
}

void
fill_c_pattern (const ResultSet::Pattern& p, awm_pattern& o)
{
  o.time = p.time;
  o.sync_index = p.sync_score.index;
  o.sync_quality = p.sync_score.quality;
  o.block_type = int (p.sync_score.block_type);
  o.type = int (p.type);
  o.decode_error = p.decode_error;
  o.speed = p.speed;
  o.n_bits = std::min<int> (int (p.bit_vec.size()), 128);
  for (int b = 0; b < o.n_bits; b++)
    o.bits[b] = p.bit_vec[b];
}

/* ---- transport between the threads of one process ---------------------------------------------------------------------- */

struct LocalWorld
{
  int world;
  std::mutex mutex;
  std::condition_variable cond;
  int arrived = 0;
  unsigned long generation = 0;
  bool failed = false;
  struct Post { std::vector<const void *> ptr; std::vector<size_t> bytes; std::vector<int> to; bool device = false; };
  std::vector<Post> posts;                       // per rank: the sends of the current round
  std::vector<std::vector<uint32_t>> reduce;     // per rank: host copy of its maxima
  std::vector<int> device_of;
  explicit LocalWorld (int n) : world (n), posts (n), reduce (n), device_of (n, 0), active (n) {}
  int active;                                    // ranks still inside their entry point
  void
  barrier()
  {
    std::unique_lock<std::mutex> lock (mutex);
    const unsigned long gen = generation;
    arrived++;
    if (arrived >= active)
      {
        arrived = 0;
        generation++;
        cond.notify_all();
      }
    else
      cond.wait (lock, [&] { return generation != gen; });
  }
  void
  leave (bool with_error)                        // a rank is through (or gave up): the others must not wait for it
  {
    std::lock_guard<std::mutex> lock (mutex);
    failed = failed || with_error;
    active--;
    if (active > 0 && arrived >= active)
      {
        arrived = 0;
        generation++;
      }
    cond.notify_all();
  }
};

struct LocalRank { LocalWorld *w; int rank; };

int
local_exchange (void *user, bool device, int n_send, const void *const *send, const size_t *send_bytes, const int *send_to,
                int n_recv, void *const *recv, const size_t *recv_bytes, const int *recv_from)
{
  auto *me = static_cast<LocalRank *> (user);
  LocalWorld& w = *me->w;
  LocalWorld::Post& post = w.posts[me->rank];
  post.ptr.assign (send, send + n_send);
  post.bytes.assign (send_bytes, send_bytes + n_send);
  post.to.assign (send_to, send_to + n_send);
  w.barrier();                                   // every rank's sends are posted (and their data is complete)
  bool ok = !w.failed;                           // (a rank that gave up has left stale posts behind)
  std::vector<size_t> next (w.world, 0);         // per sender: how many of its messages to me were consumed
  for (int i = 0; i < n_recv && ok; i++)
    {
      const int src = recv_from[i];
      const LocalWorld::Post& sp = w.posts[src];
      size_t k = next[src];
      while (k < sp.to.size() && sp.to[k] != me->rank)
        k++;
      if (k >= sp.to.size() || sp.bytes[k] != recv_bytes[i])
        {
          ok = false;                            // the two sides disagree about the plan
          break;
        }
      next[src] = k + 1;
      if (!recv_bytes[i])
        continue;
      if (!device)
        std::memcpy (recv[i], sp.ptr[k], recv_bytes[i]);
      else if (w.device_of[src] == w.device_of[me->rank])
        ok = hipMemcpy (recv[i], sp.ptr[k], recv_bytes[i], hipMemcpyDeviceToDevice) == hipSuccess;
      else
        ok = hipMemcpyPeer (recv[i], w.device_of[me->rank], sp.ptr[k], w.device_of[src], recv_bytes[i]) == hipSuccess;
    }
  // device-to-device copies may return before they are done, and the lanes (non-blocking streams) do not wait for the null stream
  if (device && n_recv)
    ok = ok && hipDeviceSynchronize() == hipSuccess;
  if (!ok)
    {
      std::lock_guard<std::mutex> lock (w.mutex);
      w.failed = true;
    }
  w.barrier();                                   // all copies done: the senders may reuse their buffers
  return w.failed ? 1 : 0;
}

int
local_exchange_d (void *user, int n_send, const void *const *send, const size_t *send_bytes, const int *send_to,
                  int n_recv, void *const *recv, const size_t *recv_bytes, const int *recv_from)
{
  return local_exchange (user, true, n_send, send, send_bytes, send_to, n_recv, recv, recv_bytes, recv_from);
}

int
local_exchange_h (void *user, int n_send, const void *const *send, const size_t *send_bytes, const int *send_to,
                  int n_recv, void *const *recv, const size_t *recv_bytes, const int *recv_from)
{
  return local_exchange (user, false, n_send, send, send_bytes, send_to, n_recv, recv, recv_bytes, recv_from);
}

int
local_all_reduce_max (void *user, uint32_t *data, size_t n)
{
  auto *me = static_cast<LocalRank *> (user);
  LocalWorld& w = *me->w;
  std::vector<uint32_t>& mine = w.reduce[me->rank];
  mine.resize (n);
  bool ok = hipMemcpy (mine.data(), data, n * sizeof (uint32_t), hipMemcpyDeviceToHost) == hipSuccess;
  w.barrier();
  std::vector<uint32_t> all (n, 0);
  for (int r = 0; r < w.world && ok; r++)
    {
      ok = w.reduce[r].size() == n;
      for (size_t i = 0; i < n && ok; i++)
        all[i] = std::max (all[i], w.reduce[r][i]);
    }
  ok = ok && hipMemcpy (data, all.data(), n * sizeof (uint32_t), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok)
    {
      std::lock_guard<std::mutex> lock (w.mutex);
      w.failed = true;
    }
  w.barrier();
  return w.failed ? 1 : 0;
}

/* one host thread per context; fn (rank, comm) is the per-rank entry point */
template<class F> int
run_ranks (awm_ctx *const *ctxs, int n_ctx, F fn)
{
  LocalWorld world (n_ctx);
  for (int r = 0; r < n_ctx; r++)
    world.device_of[r] = ctxs[r]->device;
  // peer access between the devices involved (hipMemcpyPeer works without it, through host memory; with it the copy is direct)
  for (int a = 0; a < n_ctx; a++)
    for (int b = 0; b < n_ctx; b++)
      if (world.device_of[a] != world.device_of[b])
        {
          int can = 0;
          if (hipDeviceCanAccessPeer (&can, world.device_of[a], world.device_of[b]) == hipSuccess && can
              && hipSetDevice (world.device_of[a]) == hipSuccess)
            (void) hipDeviceEnablePeerAccess (world.device_of[b], 0);        // (already enabled: an error we do not care about)
        }
  (void) hipGetLastError();
  std::vector<int> rcs (n_ctx, 0);
  std::vector<std::string> messages (n_ctx);
  std::vector<LocalRank> ranks (n_ctx);
  std::vector<awm_comm> comms (n_ctx);
  ParamValues *const pv = &params();
  auto body = [&] (int r) {
    ParamsBind bind (pv);
    ranks[r] = { &world, r };
    comms[r] = { &ranks[r], r, n_ctx, local_exchange_d, local_exchange_h, local_all_reduce_max };
    rcs[r] = fn (r, &comms[r]);
    if (rcs[r])
      messages[r] = last_error();
    world.leave (rcs[r] != 0);                   // (after an error every callback of the others fails instead of waiting for this rank)
  };
  std::vector<std::thread> threads;
  for (int r = 1; r < n_ctx; r++)
    threads.emplace_back (body, r);
  body (0);
  for (auto& t : threads)
    t.join();
  for (int r = 0; r < n_ctx; r++)
    if (rcs[r])
      {
        set_error (messages[r]);
        return rcs[r];
      }
  return 0;
}

} // namespace

extern "C" {

int
awm_ctx_set_helpers (awm_ctx *ctx, awm_ctx *const *helpers, int n_helpers)
{
  if (!ctx || n_helpers < 0 || (n_helpers && !helpers))
    {
      set_error ("awm_ctx_set_helpers: bad argument");
      return AWM_ERR_ARG;
    }
  ctx->helpers.assign (helpers, helpers + n_helpers);
  if (std::any_of (ctx->helpers.begin(), ctx->helpers.end(), [ctx] (awm_ctx *h) { return !h || h == ctx; }))
    {
      ctx->helpers.clear();
      set_error ("awm_ctx_set_helpers: a helper must be another context");
      return AWM_ERR_ARG;
    }
  return 0;
}

int
awm_sharded_plan (const uint64_t *span_frames, int world, size_t max_out, int *chunk, int *rank, uint64_t *first_sf, uint64_t *n_sf)
{
  if (!span_frames || world < 1)
    return AWM_ERR_ARG;
  return sharded_plan_c (span_frames, world, max_out, chunk, rank, first_sf, n_sf);
}

/* the entry points proper, with whatever parameter set is in force on the calling thread (awm_sharded_*: the context's own one;
 * awm_multi_*: ONE set for all ranks -- helpers that carried other settings than the main context would build different plans and
 * wait for messages that never come) */
static int
sharded_add_entry (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, const float *pcm_in_d, float *out_d, int n_channels,
                   const uint64_t *span_frames, const awm_comm *comm)
{
  if (int rc = enter (ctx, comm, span_frames, n_channels))
    return rc;
  if (!payload_hex || (span_frames[comm->rank] && (!pcm_in_d || !out_d)))
    {
      set_error ("awm_sharded_add_d: bad argument");
      return AWM_ERR_ARG;
    }
  return sharded_add (ctx, capi_key (key), payload_hex, pcm_in_d, out_d, n_channels, span_frames, comm);
}

int
awm_sharded_add_d (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, const float *pcm_in_d, float *out_d, int n_channels,
                   const uint64_t *span_frames, const awm_comm *comm)
{
  ParamsBind bind (ctx ? ctx->own_params.get() : nullptr);
  return sharded_add_entry (ctx, key, payload_hex, pcm_in_d, out_d, n_channels, span_frames, comm);
}

static int
sharded_get_entry (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, int n_channels, const uint64_t *span_frames,
                   const awm_comm *comm, size_t max_out, awm_pattern *out)
{
  if (int rc = enter (ctx, comm, span_frames, n_channels))
    return rc;
  if ((span_frames[comm->rank] && !pcm_d) || (max_out && !out))
    {
      set_error ("awm_sharded_get_d: bad argument");
      return AWM_ERR_ARG;
    }
  ResultSet rs;
  if (int rc = sharded_get (ctx, capi_key (key), pcm_d, n_channels, span_frames, comm, rs))
    return rc;
  for (size_t i = 0; i < rs.patterns.size() && i < max_out; i++)
    fill_c_pattern (rs.patterns[i], out[i]);
  return int (rs.patterns.size());
}

int
awm_sharded_get_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, int n_channels, const uint64_t *span_frames,
                   const awm_comm *comm, size_t max_out, awm_pattern *out)
{
  ParamsBind bind (ctx ? ctx->own_params.get() : nullptr);
  return sharded_get_entry (ctx, key, pcm_d, n_channels, span_frames, comm, max_out, out);
}

int
awm_multi_add_d (awm_ctx *const *ctxs, int n_ctx, const uint8_t key[16], const char *payload_hex, const float *const *pcm_in_d,
                 float *const *out_d, int n_channels, const uint64_t *span_frames)
{
  if (!ctxs || n_ctx < 1 || !pcm_in_d || !out_d || !span_frames || std::any_of (ctxs, ctxs + n_ctx, [] (awm_ctx *c) { return !c; }))
    {
      set_error ("awm_multi_add_d: bad argument");
      return AWM_ERR_ARG;
    }
  ParamsBind bind (ctxs[0]->own_params.get());             // the main context's settings for every rank (run_ranks hands them to its threads)
  return run_ranks (ctxs, n_ctx, [&] (int r, const awm_comm *comm) {
    return sharded_add_entry (ctxs[r], key, payload_hex, pcm_in_d[r], out_d[r], n_channels, span_frames, comm);
  });
}

int
awm_multi_get_d (awm_ctx *const *ctxs, int n_ctx, const uint8_t key[16], const float *const *pcm_d, int n_channels,
                 const uint64_t *span_frames, size_t max_out, awm_pattern *out)
{
  if (!ctxs || n_ctx < 1 || !pcm_d || !span_frames || std::any_of (ctxs, ctxs + n_ctx, [] (awm_ctx *c) { return !c; }))
    {
      set_error ("awm_multi_get_d: bad argument");
      return AWM_ERR_ARG;
    }
  int count = 0;
  ParamsBind bind (ctxs[0]->own_params.get());             // (see awm_multi_add_d)
  const int rc = run_ranks (ctxs, n_ctx, [&] (int r, const awm_comm *comm) {
    const int n = sharded_get_entry (ctxs[r], key, pcm_d[r], n_channels, span_frames, comm, r == 0 ? max_out : 0, r == 0 ? out : nullptr);
    if (n < 0)
      return n;
    if (r == 0)
      count = n;
    return 0;
  });
  return rc ? rc : count;
}

} // extern "C"

/* ---- batches of clips over several contexts: replicas, one host thread per context */
namespace {
template<class F> int
run_clip_shares (awm_ctx *const *ctxs, int n_ctx, const int *ctx_of_clip, size_t n_clips, const char *what, F fn)
{
  if (!ctxs || n_ctx < 1 || (n_clips && !ctx_of_clip) || std::any_of (ctxs, ctxs + n_ctx, [] (awm_ctx *c) { return !c; }))
    {
      set_error (std::string (what) + ": bad argument");
      return AWM_ERR_ARG;
    }
  std::vector<std::vector<size_t>> share (n_ctx);
  for (size_t i = 0; i < n_clips; i++)
    {
      if (ctx_of_clip[i] < 0 || ctx_of_clip[i] >= n_ctx)
        {
          set_error (std::string (what) + ": ctx_of_clip out of range");
          return AWM_ERR_ARG;
        }
      share[ctx_of_clip[i]].push_back (i);
    }
  ParamsBind bind (ctxs[0]->own_params.get());
  ParamValues *const pv = &params();
  std::vector<int> rcs (n_ctx, 0);
  std::vector<std::string> messages (n_ctx);
  auto body = [&] (int r) {
    ParamsBind thread_bind (pv);
    if (share[r].empty())
      return;
    rcs[r] = hipSetDevice (ctxs[r]->device) == hipSuccess ? fn (r, share[r]) : AWM_ERR_HIP;
    if (rcs[r])
      messages[r] = last_error();
  };
  std::vector<std::thread> threads;
  for (int r = 1; r < n_ctx; r++)
    threads.emplace_back (body, r);
  body (0);
  for (auto& t : threads)
    t.join();
  for (int r = 0; r < n_ctx; r++)
    if (rcs[r])
      {
        set_error (messages[r]);
        return rcs[r];
      }
  return 0;
}

/* the keys of a share, 16 bytes each (one_key repeated when there is no list) */
std::vector<uint8_t>
share_keys (const std::vector<size_t>& idx, const uint8_t *keys, const uint8_t *one_key)
{
  std::vector<uint8_t> out (idx.size() * 16);
  for (size_t j = 0; j < idx.size(); j++)
    std::memcpy (out.data() + 16 * j, keys ? keys + 16 * idx[j] : one_key, 16);
  return out;
}
} // namespace

extern "C" {

int
awm_multi_add_watermark_batch_d (awm_ctx *const *ctxs, int n_ctx, const int *ctx_of_clip, const uint8_t *keys, const uint8_t *one_key,
                                 const char *payload_hex, size_t n_clips, const float *const *pcm_in_d, float *const *out_d,
                                 const size_t *n_frames, int n_channels)
{
  if ((!keys && !one_key) || (n_clips && (!pcm_in_d || !out_d || !n_frames)))
    {
      set_error ("awm_multi_add_watermark_batch_d: bad argument");
      return AWM_ERR_ARG;
    }
  return run_clip_shares (ctxs, n_ctx, ctx_of_clip, n_clips, "awm_multi_add_watermark_batch_d", [&] (int r, const std::vector<size_t>& idx) {
    std::vector<const float *> in;
    std::vector<float *> out;
    std::vector<size_t> len;
    for (size_t i : idx)
      {
        in.push_back (pcm_in_d[i]);
        out.push_back (out_d[i]);
        len.push_back (n_frames[i]);
      }
    // (the single-context entry points bind the context's own settings; a helper without its own set follows the calling thread's)
    if (keys)
      return awm_add_watermark_batch_keys_d (ctxs[r], share_keys (idx, keys, one_key).data(), payload_hex, idx.size(), in.data(), out.data(), len.data(), n_channels);
    return awm_add_watermark_batch_d (ctxs[r], one_key, payload_hex, idx.size(), in.data(), out.data(), len.data(), n_channels);
  });
}

int
awm_multi_get_watermark_batch_d (awm_ctx *const *ctxs, int n_ctx, const int *ctx_of_clip, const uint8_t *keys, const uint8_t *one_key,
                                 size_t n_clips, const float *const *pcm_d, const size_t *n_frames, int n_channels,
                                 size_t max_out_per_clip, awm_pattern *out, int *n_out)
{
  if ((!keys && !one_key) || (n_clips && (!pcm_d || !n_frames || !n_out || (max_out_per_clip && !out))))
    {
      set_error ("awm_multi_get_watermark_batch_d: bad argument");
      return AWM_ERR_ARG;
    }
  return run_clip_shares (ctxs, n_ctx, ctx_of_clip, n_clips, "awm_multi_get_watermark_batch_d", [&] (int r, const std::vector<size_t>& idx) {
    std::vector<const float *> in;
    std::vector<size_t> len;
    for (size_t i : idx)
      {
        in.push_back (pcm_d[i]);
        len.push_back (n_frames[i]);
      }
    std::vector<awm_pattern> pats (idx.size() * max_out_per_clip);
    std::vector<int> counts (idx.size(), 0);
    const int rc = keys ? awm_get_watermark_batch_keys_d (ctxs[r], share_keys (idx, keys, one_key).data(), idx.size(), in.data(), len.data(), n_channels, 0,
                                                          max_out_per_clip, pats.data(), counts.data())
                        : awm_get_watermark_batch_d (ctxs[r], one_key, idx.size(), in.data(), len.data(), n_channels, 0, max_out_per_clip, pats.data(),
                                                     counts.data());
    if (rc)
      return rc;
    for (size_t j = 0; j < idx.size(); j++)
      {
        n_out[idx[j]] = counts[j];
        std::copy (pats.begin() + j * max_out_per_clip, pats.begin() + j * max_out_per_clip + std::min<size_t> (size_t (std::max (counts[j], 0)), max_out_per_clip),
                   out + idx[j] * max_out_per_clip);
      }
    return 0;
  });
}

} // extern "C"
