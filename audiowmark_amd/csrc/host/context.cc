#include "context.hh"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include "utils.hh"
#include <cmath>
#include <cstring>

namespace awm {

static thread_local std::string g_last_error;
void set_error (const std::string& msg) { g_last_error = msg; }
const std::string& last_error() { return g_last_error; }
std::string hip_error_string (hipError_t e) { return std::string (hipGetErrorName (e)) + " (" + hipGetErrorString (e) + ")"; }

// allocation census (awm_debug_alloc_stats): how many device / page-locked allocations the calls of this process made and how long
// the runtime took for them -- what a FIRST call pays before its workspaces exist
static std::atomic<long>   g_dev_allocs { 0 }, g_pin_allocs { 0 };
static std::atomic<double> g_dev_alloc_ms { 0.0 }, g_pin_alloc_ms { 0.0 };
static void
note_alloc (std::atomic<long>& count, std::atomic<double>& ms, std::chrono::steady_clock::time_point t0)
{
  const double dt = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - t0).count();
  count.fetch_add (1, std::memory_order_relaxed);
  double cur = ms.load (std::memory_order_relaxed);
  while (!ms.compare_exchange_weak (cur, cur + dt, std::memory_order_relaxed))
    ;
}

int
DevBuffer::reserve (size_t want)
{
  if (want <= bytes)
    return 0;
  if (want > MAX_BYTES)
    {
      set_error ("device buffer of " + std::to_string (want) + " bytes requested (size overflow?)");
      return AWM_ERR_ARG;
    }
  release();
  // round up generously: workspaces are reused across calls, HBM is plentiful
  size_t cap = size_t (1) << 20;
  while (cap < want)
    cap += cap / 2 > (size_t (1) << 30) ? (size_t (1) << 30) : cap / 2 + 1;
  cap = (cap + 255) & ~size_t (255);
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc (&ptr, cap);
  note_alloc (g_dev_allocs, g_dev_alloc_ms, t0);
  if (e != hipSuccess)
    {
      ptr = nullptr;
      set_error ("hipMalloc of " + std::to_string (cap) + " bytes failed: " + hip_error_string (e));
      return AWM_ERR_HIP;
    }
  bytes = cap;
  return 0;
}

int
PinnedBuffer::reserve (size_t want)
{
  if (want <= bytes)
    return 0;
  release();
  size_t cap = size_t (64) << 10;
  while (cap < want)
    cap *= 2;
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipHostMalloc (&ptr, cap, hipHostMallocDefault);
  note_alloc (g_pin_allocs, g_pin_alloc_ms, t0);
  if (e != hipSuccess)
    {
      ptr = nullptr;
      set_error ("hipHostMalloc of " + std::to_string (cap) + " bytes failed: " + hip_error_string (e));
      return AWM_ERR_HIP;
    }
  bytes = cap;
  return 0;
}

void
PinnedBuffer::release()
{
  if (ptr)
    (void) hipHostFree (ptr);
  ptr = nullptr;
  bytes = 0;
}

void
DevBuffer::release()
{
  if (ptr)
    (void) hipFree (ptr);
  ptr = nullptr;
  bytes = 0;
}

} // namespace awm

extern "C" void
awm_debug_alloc_stats (long *dev_allocs, double *dev_ms, long *pinned_allocs, double *pinned_ms)
{
  if (dev_allocs) *dev_allocs = awm::g_dev_allocs.load();
  if (dev_ms) *dev_ms = awm::g_dev_alloc_ms.load();
  if (pinned_allocs) *pinned_allocs = awm::g_pin_allocs.load();
  if (pinned_ms) *pinned_ms = awm::g_pin_alloc_ms.load();
}

hipEvent_t
awm::ReadyMarks::next_event()
{
  if (used == pool.size())
    {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags (&e, hipEventDisableTiming) != hipSuccess)
        return nullptr;
      pool.push_back (e);
    }
  return pool[used++];
}

void
awm::ReadyMarks::release()
{
  disarm();
  for (hipEvent_t e : pool)
    (void) hipEventDestroy (e);
  pool.clear();
}

bool
awm::FileStaging::ensure_events()
{
  if (have_events)
    return true;
  for (int i = 0; i < RING; i++)
    for (hipEvent_t *e : { &in_copied[i], &in_used[i], &out_encoded[i], &out_copied[i] })
      if (!*e && hipEventCreateWithFlags (e, hipEventDisableTiming) != hipSuccess)
        return false;
  have_events = true;
  return true;
}

void
awm::FileStaging::release()
{
  for (int i = 0; i < RING; i++)
    {
      for (hipEvent_t *e : { &in_copied[i], &in_used[i], &out_encoded[i], &out_copied[i] })
        if (*e)
          {
            (void) hipEventDestroy (*e);
            *e = nullptr;
          }
      in_host[i].release();
      out_host[i].release();
      in_dev[i].release();
      out_dev[i].release();
    }
  pcm.release();
  have_events = false;
}

void
awm::WorkLane::release_lane()
{
  for (DevBuffer *b : { &ws_db, &ws_block_db, &ws_have, &ws_q, &ws_raw, &ws_mean, &ws_misc, &ws_refine, &ws_refine_have, &ws_soft, &ws_viterbi,
                        &ws_viterbi_in, &ws_viterbi_bits, &ws_viterbi_err, &ws_viterbi_sync, &ws_block_max, &ws_clip, &ws_idx, &ws_limit_tab, &ws_jobs, &ws_group, &ws_keytab, &ws_keytab_aux, &ws_keytab_scratch, &ws_shard_edge, &ws_shard_tail, &ws_shard_q })
    b->release();
  for (PinnedBuffer *b : { &pin_refine_in[0], &pin_refine_in[1], &pin_refine_q[0], &pin_refine_q[1], &pin_peaks, &pin_blocks, &pin_jobs, &pin_bits, &pin_small, &pin_group, &pin_shard, &pin_shard_up, &pin_keytab })
    b->release();
  for (hipEvent_t& ev : ev_refine)
    if (ev)
      {
        (void) hipEventDestroy (ev);
        ev = nullptr;
      }
  if (ev_sync)
    (void) hipEventDestroy (ev_sync);
  ev_sync = nullptr;
  awm::speed_scratch_free (this);
}

unsigned int *
awm::WorkLane::viterbi_sync (size_t n_decodes)
{
  // The one-launch Viterbi kernel leaves its counters at zero (hip/viterbi.hip), so the block is cleared only when it is (re)allocated
  // -- on the lane's own stream, in front of the launch that uses it first.
  // (growth is detected by CAPACITY: a freed block may come back from hipMalloc at the same address, larger)
  const size_t before = ws_viterbi_sync.bytes;
  if (ws_viterbi_sync.reserve (awmk::viterbi_sync_bytes ((long long) n_decodes)))
    return nullptr;
  if (ws_viterbi_sync.bytes != before && hipMemsetAsync (ws_viterbi_sync.ptr, 0, ws_viterbi_sync.bytes, stream) != hipSuccess)
    {
      ws_viterbi_sync.release();
      return nullptr;
    }
  return ws_viterbi_sync.as<unsigned int>();
}

awm::WorkLane *
awm_ctx::lane (int i)
{
  if (i <= 0)
    return this;
  if (i >= awm::MAX_LANES)
    return nullptr;
  auto& l = extra_lanes[i - 1];
  if (!l)
    {
      auto nl = std::make_unique<awm::WorkLane>();
      if (hipStreamCreateWithFlags (&nl->stream, hipStreamNonBlocking) != hipSuccess)
        return nullptr;
      nl->own_stream = true;
      l = std::move (nl);
    }
  return l.get();
}

using namespace awm;

int
awm::upload_sync (DevBuffer& buf, const void *src, size_t bytes, hipStream_t st)
{
  if (int rc = buf.reserve (bytes ? bytes : 1))
    return rc;
  // through a page-locked staging buffer of our own: the runtime's path for PAGEABLE sources costs ~190 MB of resident host
  // memory at its first use (tools/rss_probe)
  static std::mutex staging_mutex;
  static PinnedBuffer staging;
  std::lock_guard<std::mutex> lock (staging_mutex);
  if (int rc = staging.reserve (bytes ? bytes : 1))
    return rc;
  std::memcpy (staging.ptr, src, bytes);
  AWM_HIP_CHECK (hipMemcpyAsync (buf.ptr, staging.ptr, bytes, hipMemcpyHostToDevice, st));
  AWM_HIP_CHECK (hipStreamSynchronize (st));
  return 0;
}

std::vector<float>
awm::zita_table (double frel, unsigned h, unsigned n)
{
  auto sinc = [] (double x) { x = std::fabs (x); if (x < 1e-6) return 1.0; x *= M_PI; return std::sin (x) / x; };
  auto wind = [] (double x) { x = std::fabs (x); if (x >= 1.0) return 0.0; x *= M_PI; return 0.384 + 0.500 * std::cos (x) + 0.116 * std::cos (2 * x); };
  std::vector<float> ctab (size_t (h) * (n + 1));
  float *p = ctab.data();
  for (unsigned j = 0; j <= n; j++)
    {
      double t = double (j) / double (n);
      for (unsigned i = 0; i < h; i++)
        {
          p[h - i - 1] = float (frel * sinc (t * frel) * wind (t / h));
          t += 1;
        }
      p += h;
    }
  return ctab;
}

static int
upload (DevBuffer& buf, const void *src, size_t bytes, hipStream_t st)
{
  return upload_sync (buf, src, bytes, st);
}

static std::vector<int>
pack_sync_table (const SyncTable& t, const std::vector<int>& want_pos /* empty: use frame */)
{
  const int R = t.rows_per_bit;
  std::vector<int> packed (size_t (6) * R * 64, 0);
  for (int bit = 0; bit < 6; bit++)
    for (int r = 0; r < R; r++)
      {
        int *row = &packed[(size_t (bit) * R + r) * 64];
        const size_t src = size_t (bit) * R + r;
        for (int i = 0; i < 30; i++)
          {
            row[i] = t.up[src * 30 + i];
            row[30 + i] = t.down[src * 30 + i];
          }
        row[60] = want_pos.empty() ? t.frame[src] : want_pos[t.frame[src]];
        // [61]: row index of the NEXT row of this bit (or a sentinel), so that a scan that stops at a frame limit
        // knows the next frame without another dependent load (K5w)
        row[61] = r + 1 < R ? (want_pos.empty() ? t.frame[src + 1] : want_pos[t.frame[src + 1]]) : 0x7fffffff;
      }
  return packed;
}

namespace awm {
ClipKeyHost
build_clip_key_host (const Key& key)
{
  ClipKeyHost h;
  const SyncTable st = build_sync_table (key, true);
  const int total = mark_block_frame_count() * 2, R = st.rows_per_bit;
  std::vector<int> want_pos (total, -1);
  std::vector<char> want (total, 0);
  for (int f : st.frame)
    want[f] = 1;
  for (int f = 0; f < total; f++)
    if (want[f])
      {
        want_pos[f] = int (h.want.size());
        h.want.push_back (f);
      }
  const auto pa = pack_sync_table (st, {});
  h.chains.resize (size_t (12) * R * 8);
  awmk::pack_scan_chains (pa.data(), R, h.chains.data());
  h.row_frames.assign (st.frame.begin(), st.frame.end());
  const int NW = int (h.want.size());
  h.perm.assign (NW, 0);
  h.pos.assign (size_t (NW) * Params::n_bands, 255);
  for (int bit = 0; bit < 6; bit++)
    for (int r = 0; r < R; r++)
      {
        const size_t src = size_t (bit) * R + r;
        const int w = want_pos[st.frame[src]];
        h.perm[w] = int (src);
        for (int i = 0; i < 30; i++)
          {
            h.pos[size_t (w) * Params::n_bands + st.up[src * 30 + i]] = (unsigned char) i;
            h.pos[size_t (w) * Params::n_bands + st.down[src * 30 + i]] = (unsigned char) (30 + i);
          }
      }
  h.mix = build_mix_table (key);
  const auto order = bit_order (key, code_size (ConvBlockType::a, params().payload_size));
  h.inv_order.resize (order.size());
  for (size_t i = 0; i < order.size(); i++)
    h.inv_order[order[i]] = int (i);
  return h;
}
}

static void
release_key_tables (KeyTables& kt)
{
  for (auto& s : kt.sync)
    {
      s.packed_approx.release();
      s.packed_refine.release();
      s.chains_approx.release();
      s.want_list_dev.release();
      s.refine_perm.release();
      s.refine_pos.release();
    }
  kt.mix_frame.release();
  kt.mix_up.release();
  kt.mix_down.release();
  kt.bit_order_inv_dev.release();
}

KeyTables *
awm_ctx::get_key_tables (const Key& key)
{
  std::lock_guard<std::mutex> lock (table_mutex);
  std::vector<unsigned char> kb (key.aes_key(), key.aes_key() + Key::SIZE);
  for (auto& kt : key_tables)
    if (kt->key == kb && kt->mix == params().mix && kt->frames_per_bit == params().frames_per_bit)
      {
        kt->last_use = ++table_clock;
        return kt.get();
      }
  if (key_tables.size() >= MAX_CACHED_TABLES)
    {
      // bounded: a `get` over many keys must not keep every key's tables in HBM.  The victim is the least recently used
      // of MAX_CACHED_TABLES entries (far more than the lanes that can be decoding at once); drain the device before freeing it.
      auto victim = std::min_element (key_tables.begin(), key_tables.end(),
                                      [] (const auto& a, const auto& b) { return a->last_use < b->last_use; });
      (void) hipDeviceSynchronize();
      release_key_tables (**victim);
      key_tables.erase (victim);
    }
  auto kt = std::make_unique<KeyTables>();
  kt->last_use = ++table_clock;
  kt->key = kb;
  kt->mix = params().mix;
  kt->frames_per_bit = params().frames_per_bit;
  for (int clip = 0; clip < 2; clip++)
    {
      auto& s = kt->sync[clip];
      s.host = build_sync_table (key, clip);
      const int total = mark_block_frame_count() * (clip ? 2 : 1);
      std::vector<int> want_pos (total, -1);
      std::vector<char> want (total, 0);
      for (int f : s.host.frame)
        want[f] = 1;
      for (int f = 0; f < total; f++)
        if (want[f])
          {
            want_pos[f] = s.want_list.size();
            s.want_list.push_back (f);
          }
      auto pa = pack_sync_table (s.host, {});
      auto pr = pack_sync_table (s.host, want_pos);
      if (upload (s.packed_approx, pa.data(), pa.size() * sizeof (int), stream)) return nullptr;
      if (upload (s.packed_refine, pr.data(), pr.size() * sizeof (int), stream)) return nullptr;
      std::vector<unsigned> pc (size_t (12) * s.host.rows_per_bit * 8);
      awmk::pack_scan_chains (pa.data(), s.host.rows_per_bit, pc.data());
      if (upload (s.chains_approx, pc.data(), pc.size() * sizeof (unsigned), stream)) return nullptr;
      if (upload (s.want_list_dev, s.want_list.data(), s.want_list.size() * sizeof (int), stream)) return nullptr;
      {
        const int R = s.host.rows_per_bit, NW = int (s.want_list.size());
        std::vector<int> perm (NW, 0);
        std::vector<unsigned char> pos (size_t (NW) * Params::n_bands, 255);
        for (int bit = 0; bit < 6; bit++)
          for (int r = 0; r < R; r++)
            {
              const size_t src = size_t (bit) * R + r;
              const int w = want_pos[s.host.frame[src]];
              perm[w] = int (src);
              for (int i = 0; i < 30; i++)
                {
                  pos[size_t (w) * Params::n_bands + s.host.up[src * 30 + i]] = (unsigned char) i;
                  pos[size_t (w) * Params::n_bands + s.host.down[src * 30 + i]] = (unsigned char) (30 + i);
                }
            }
        if (upload (s.refine_perm, perm.data(), perm.size() * sizeof (int), stream)) return nullptr;
        if (upload (s.refine_pos, pos.data(), pos.size(), stream)) return nullptr;
      }
    }
  kt->mix_host = build_mix_table (key);
  if (upload (kt->mix_frame, kt->mix_host.frame.data(), kt->mix_host.frame.size() * sizeof (int16_t), stream)) return nullptr;
  if (upload (kt->mix_up, kt->mix_host.up.data(), kt->mix_host.up.size(), stream)) return nullptr;
  if (upload (kt->mix_down, kt->mix_host.down.data(), kt->mix_host.down.size(), stream)) return nullptr;
  kt->bit_order_a = bit_order (key, code_size (ConvBlockType::a, params().payload_size));
  {
    std::vector<int> inv (kt->bit_order_a.size());
    for (size_t i = 0; i < inv.size(); i++)
      inv[kt->bit_order_a[i]] = int (i);            // apply_bit_order (decode): out[order[i]] = in[i]
    if (upload (kt->bit_order_inv_dev, inv.data(), inv.size() * sizeof (int), stream)) return nullptr;
  }
  key_tables.push_back (std::move (kt));
  return key_tables.back().get();
}

/* zita-resampler 1.x, Resampler::setup (fs_inp, fs_out, nchan, hlen) and Resampler_table::Resampler_table (fr, hl, np),
 * restated from the library's published algorithm (the library itself is not part of the reference tree): parity unpinned. */
ResampleTable *
awm_ctx::get_resample_table (int rate_in, int rate_out)
{
  std::lock_guard<std::mutex> lock (table_mutex);
  for (auto& t : resample_tables)
    if (t->rate_in == rate_in && t->rate_out == rate_out)
      return t.get();
  if (rate_in <= 0 || rate_out <= 0)
    return nullptr;
  const int hlen = 16;
  double frel = 1.0 - 2.6 / hlen;
  const double r = double (rate_out) / double (rate_in);
  unsigned a = unsigned (rate_out), b = unsigned (rate_in);
  while (b)
    {
      const unsigned tmp = a % b;
      a = b;
      b = tmp;
    }
  const unsigned n = unsigned (rate_out) / a, s = unsigned (rate_in) / a;
  if (!(16 * r >= 1 && n <= 1000))
    return nullptr;                      // zita's Resampler::setup refuses: ResamplerImpl::create falls back to VResampler (capi_kernels.cc)
  unsigned h = hlen;
  if (r < 1)
    {
      frel *= r;
      h = unsigned (std::ceil (h / r));
    }
  const std::vector<float> ctab = zita_table (frel, h, n);
  auto rt = std::make_unique<ResampleTable>();
  rt->rate_in = rate_in;
  rt->rate_out = rate_out;
  rt->hl = int (h);
  rt->np = int (n);
  rt->step = int (s);
  if (upload (rt->ctab, ctab.data(), ctab.size() * sizeof (float), stream))
    return nullptr;
  resample_tables.push_back (std::move (rt));
  return resample_tables.back().get();
}

hipStream_t
awm_ctx::get_copy_stream()
{
  std::lock_guard<std::mutex> lock (table_mutex);
  if (!copy_stream && hipStreamCreateWithFlags (&copy_stream, hipStreamNonBlocking) != hipSuccess)
    copy_stream = nullptr;
  return copy_stream;
}

FrameModTable *
awm_ctx::get_frame_mod (const Key& key, const std::string& payload_hex)
{
  std::lock_guard<std::mutex> lock (table_mutex);
  std::vector<unsigned char> kb (key.aes_key(), key.aes_key() + Key::SIZE);
  for (auto& t : frame_mod_tables)
    if (t->key == kb && t->payload == payload_hex && t->mix == params().mix && t->frames_per_bit == params().frames_per_bit)
      {
        t->last_use = ++table_clock;
        return t.get();
      }
  auto bits = parse_payload (payload_hex);
  if (bits.empty())
    {
      set_error ("cannot parse payload '" + payload_hex + "'");
      return nullptr;
    }
  auto table = build_frame_mod_table (key, bits);
  auto t = std::make_unique<FrameModTable>();
  t->key = kb;
  t->payload = payload_hex;
  t->mix = params().mix;
  t->frames_per_bit = params().frames_per_bit;
  t->last_use = ++table_clock;
  if (upload (t->dev, table.data(), table.size(), stream))
    return nullptr;
  if (frame_mod_tables.size() >= MAX_CACHED_TABLES)
    {
      // least recently used entry; a kernel that was given this table may still be queued on some lane: drain the device first
      auto victim = std::min_element (frame_mod_tables.begin(), frame_mod_tables.end(),
                                      [] (const auto& a, const auto& b) { return a->last_use < b->last_use; });
      (void) hipDeviceSynchronize();
      (*victim)->dev.release();
      frame_mod_tables.erase (victim);
    }
  frame_mod_tables.push_back (std::move (t));
  return frame_mod_tables.back().get();
}

void
awm_ctx::prof_collect()
{
  std::lock_guard<std::mutex> lock (prof_mutex);
  for (auto& p : prof_pending)
    {
      float ms = 0;
      if (hipEventSynchronize (p.stop) == hipSuccess && hipEventElapsedTime (&ms, p.start, p.stop) == hipSuccess)
        prof_ms[p.id] += ms;
      (void) hipEventDestroy (p.start);
      (void) hipEventDestroy (p.stop);
    }
  prof_pending.clear();
}

static const char *prof_names[awm::PROF_COUNT] = {
  "add_mix_kernel", "limiter_kernel", "sync_db_kernel(approx)", "sync_scan_kernel(approx)", "local_mean_kernel",
  "sync_db_kernel(refine)", "sync_scan_kernel(refine)", "sync_db_kernel(block)", "soft_bits_kernel", "viterbi_kernel",
  "stft_full_kernel",
  "resample_kernel", "resample_var_kernel", "speed_mags_kernel", "speed_compare_kernel", "frame_mod_table_kernel"
};

extern "C" {

int
awm_prof_enable (awm_ctx *ctx, int on)
{
  if (!ctx) return AWM_ERR_ARG;
  ctx->prof_collect();
  ctx->prof_enabled = on != 0;
  return 0;
}

int
awm_prof_reset (awm_ctx *ctx)
{
  if (!ctx) return AWM_ERR_ARG;
  ctx->prof_collect();
  for (int i = 0; i < awm::PROF_COUNT; i++)
    {
      ctx->prof_ms[i] = 0;
      ctx->prof_launches[i] = 0;
      ctx->prof_bytes[i] = 0;
    }
  return 0;
}

int awm_prof_count (void) { return awm::PROF_COUNT; }
const char *awm_prof_name (int id) { return id >= 0 && id < awm::PROF_COUNT ? prof_names[id] : ""; }

int
awm_prof_read (awm_ctx *ctx, int id, double *ms, long *launches, double *algorithmic_bytes)
{
  if (!ctx || id < 0 || id >= awm::PROF_COUNT) return AWM_ERR_ARG;
  ctx->prof_collect();
  if (ms) *ms = ctx->prof_ms[id];
  if (launches) *launches = ctx->prof_launches[id];
  if (algorithmic_bytes) *algorithmic_bytes = ctx->prof_bytes[id];
  return 0;
}

const char *awm_last_error (void) { return awm::last_error().c_str(); }
const char *awm_version (void) { return "audiowmark_amd 0.1 (gfx950)"; }

static int ctx_create (int device, bool own_stream, hipStream_t given, awm_ctx **ctx_out);

int
awm_ctx_create (int device, awm_ctx **ctx_out)
{
  return ctx_create (device, true, nullptr, ctx_out);
}

int
awm_ctx_create_on_stream (int device, void *hip_stream, awm_ctx **ctx_out)
{
  return ctx_create (device, false, (hipStream_t) hip_stream, ctx_out);
}

static int
ctx_create (int device, bool own_stream, hipStream_t given, awm_ctx **ctx_out)
{
  if (!ctx_out)
    return AWM_ERR_ARG;
  *ctx_out = nullptr;
  // Lanes are HIP streams; the runtime maps streams to GPU_MAX_HW_QUEUES hardware queues (default 4) and streams on one queue
  // serialise.  The library does not touch its host's environment: a process that wants all lanes on queues of their own (the
  // clip batch mode: 0.64 -> 0.535 ms per clip over 8 queues) exports GPU_MAX_HW_QUEUES=16 itself before its first HIP call
  // (awm_hip.h, "runtime settings"; bench.py does).
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount (&n_dev);
  if (e != hipSuccess || n_dev <= 0)
    {
      set_error ("no HIP device available (" + hip_error_string (e) + "); this library has no CPU fallback");
      return AWM_ERR_NO_DEVICE;
    }
  if (device < 0 || device >= n_dev)
    {
      set_error ("device index out of range");
      return AWM_ERR_ARG;
    }
  AWM_HIP_CHECK (hipSetDevice (device));
  {
    // the code object holds gfx950 code only: say so here instead of failing at the first launch.  (The architecture name is the
    // test; what a runtime release reports as sharedMemPerBlock is not -- some report the 64 KB default instead of the 160 KB a
    // gfx950 workgroup may ask for -- so a small figure there is only worth a warning.)
    hipDeviceProp_t prop;
    AWM_HIP_CHECK (hipGetDeviceProperties (&prop, device));
    if (std::strncmp (prop.gcnArchName, "gfx950", 6) != 0)
      {
        set_error (std::string ("device ") + std::to_string (device) + " is " + prop.gcnArchName + "; this library is built for gfx950 (MI355X) only");
        return AWM_ERR_NO_DEVICE;
      }
    static std::atomic<bool> warned { false };
    if (prop.sharedMemPerBlock < size_t (160) * 1024 && !warned.exchange (true))
      warning ("audiowmark: device %d reports %zu bytes of LDS per workgroup (a gfx950 has 160 KB: K5w and the key table kernels use them)\n",
               device, size_t (prop.sharedMemPerBlock));
  }
  auto ctx = std::make_unique<awm_ctx>();
  ctx->device = device;
  if (own_stream)
    AWM_HIP_CHECK (hipStreamCreateWithFlags (&ctx->stream, hipStreamNonBlocking));
  else
    ctx->stream = given;
  ctx->own_stream = own_stream;

  // constant tables, evaluated in double and rounded once
  std::vector<float> blob;
  auto push_c = [&] (double re, double im) { blob.push_back (float (re)); blob.push_back (float (im)); };
  const size_t off_tw512 = blob.size();
  for (int k = 0; k < 512; k++)
    push_c (std::cos (-2 * M_PI * k / 512), std::sin (-2 * M_PI * k / 512));
  const size_t off_tw1024 = blob.size();
  for (int k = 0; k <= 512; k++)
    push_c (std::cos (-2 * M_PI * k / 1024), std::sin (-2 * M_PI * k / 1024));
  while (blob.size() % 4) blob.push_back (0);
  const size_t off_win = blob.size();
  auto win = gen_normalized_window (Params::frame_size);
  blob.insert (blob.end(), win.begin(), win.end());
  const size_t off_synth = blob.size();
  auto synth = gen_synth_window();
  blob.insert (blob.end(), synth.begin(), synth.end());
  // sliding DFT rotations for bins 19..102 (refinement), double precision
  std::vector<double> slide;
  for (int k = 19; k <= 102; k++)
    {
      for (int j = 0; j < 8; j++)
        {
          slide.push_back (std::cos (-2 * M_PI * k * j / 1024));
          slide.push_back (std::sin (-2 * M_PI * k * j / 1024));
        }
      slide.push_back (std::cos (2 * M_PI * k * 8 / 1024));
      slide.push_back (std::sin (2 * M_PI * k * 8 / 1024));
    }
  const size_t off_tw512d = slide.size();
  for (int k = 0; k < 512; k++)
    {
      slide.push_back (std::cos (-2 * M_PI * k / 512));
      slide.push_back (std::sin (-2 * M_PI * k / 512));
    }
  // ... and the update term's factors with the rotation folded in, as floats (two per double slot)
  const size_t off_slide32 = slide.size();
  {
    std::vector<float> f;
    for (int k = 19; k <= 102; k++)
      for (int j = 0; j < 8; j++)
        {
          f.push_back (float (std::cos (2 * M_PI * k * (8 - j) / 1024)));
          f.push_back (float (std::sin (2 * M_PI * k * (8 - j) / 1024)));
        }
    slide.resize (off_slide32 + f.size() / 2);
    std::memcpy (slide.data() + off_slide32, f.data(), f.size() * sizeof (float));
  }
  if (int rc = upload (ctx->tab_slide, slide.data(), slide.size() * sizeof (double), ctx->stream))
    return rc;
  ctx->tabs.slide = ctx->tab_slide.as<double2>();
  ctx->tabs.tw512d = reinterpret_cast<const double2 *> (ctx->tab_slide.as<double>() + off_tw512d);
  ctx->tabs.slide32 = reinterpret_cast<const float2 *> (ctx->tab_slide.as<double>() + off_slide32);
  if (int rc = upload (ctx->tab_mem, blob.data(), blob.size() * sizeof (float), ctx->stream))
    return rc;
  const float *base = ctx->tab_mem.as<float>();
  ctx->tabs.tw512 = reinterpret_cast<const float2 *> (base + off_tw512);
  ctx->tabs.tw1024 = reinterpret_cast<const float2 *> (base + off_tw1024);
  ctx->tabs.window = base + off_win;
  ctx->tabs.synth = base + off_synth;
  // (also waits for the uploads above; ~0.5 ms for the first context of the process, which latches the result)
  if (awmk::probe_dependent_launch_us (ctx->stream) < 0 && hipStreamSynchronize (ctx->stream) != hipSuccess)
    {
      set_error ("awm_ctx_create: the device does not complete work on the context's stream");
      return AWM_ERR_HIP;
    }
  *ctx_out = ctx.release();
  return 0;
}

int
awm_ctx_set_chunk_lanes (awm_ctx *ctx, int n_lanes)
{
  if (!ctx || n_lanes < 1)
    return AWM_ERR_ARG;
  ctx->chunk_lanes = std::min (n_lanes, awm::CHUNK_LANES);
  return 0;
}

/* The opposite of awm_ctx_trim: pay NOW what the first call of a fresh context would pay -- HIP streams (the runtime takes 4 - 10 ms to
 * create one, profiles/r05/stream_create_probe.txt): the chunk lanes of `get`, with detect_speed the second set of lanes the plain decode
 * runs on beside the speed search, and the copy stream of the file level calls.  A service creates its contexts at start-up and calls this
 * once; the first request then runs like the second. */
int
awm_ctx_warm_up (awm_ctx *ctx, int detect_speed, int file_level)
{
  if (!ctx)
    return AWM_ERR_ARG;
  AWM_HIP_CHECK (hipSetDevice (ctx->device));
  const int lanes = std::max (1, std::min (ctx->chunk_lanes, awm::CHUNK_LANES)) * (detect_speed ? 2 : 1);
  for (int i = 0; i < lanes && i < awm::MAX_LANES; i++)
    if (!ctx->lane (i))
      {
        awm::set_error ("cannot create a work lane (stream)");
        return AWM_ERR_HIP;
      }
  if (file_level && !ctx->get_copy_stream())
    {
      awm::set_error ("cannot create the copy stream");
      return AWM_ERR_HIP;
    }
  return 0;
}

/* Give back what the context keeps between calls for speed: the workspaces of its lanes, the file level staging rings (page-locked
 * tiles + their device twins) and the whole-stream PCM buffer of the file level `get` -- about 1.3 GB per hour of the longest file
 * seen so far.  Key tables, the constant tables and the streams stay; the next call allocates what it needs again. */
int
awm_ctx_trim (awm_ctx *ctx)
{
  if (!ctx)
    return AWM_ERR_ARG;
  AWM_HIP_CHECK (hipSetDevice (ctx->device));
  AWM_HIP_CHECK (hipStreamSynchronize (ctx->stream));
  for (auto& l : ctx->extra_lanes)
    if (l)
      {
        if (l->stream)
          AWM_HIP_CHECK (hipStreamSynchronize (l->stream));
        l->release_lane();
      }
  if (ctx->copy_stream)
    AWM_HIP_CHECK (hipStreamSynchronize (ctx->copy_stream));
  ctx->release_lane();
  ctx->file_staging.release();
  awm::speed_workspace_free (ctx);
  ctx->ws_add_batch.release();
  ctx->pin_add_batch.release();
  ctx->ws_merge_soft.release();
  ctx->ws_rate_a.release();
  ctx->ws_rate_b.release();
  ctx->ws_rate_c.release();
  for (awm_ctx *h : ctx->helpers)
    if (h && h != ctx)
      if (int rc = awm_ctx_trim (h)) return rc;
  return 0;
}

void
awm_ctx_destroy (awm_ctx *ctx)
{
  if (!ctx)
    return;
  (void) hipSetDevice (ctx->device);
  (void) hipStreamSynchronize (ctx->stream);
  for (auto& kt : ctx->key_tables)
    release_key_tables (*kt);
  for (auto& t : ctx->frame_mod_tables)
    t->dev.release();
  for (auto& t : ctx->resample_tables)
    t->ctab.release();
  awm::speed_workspace_free (ctx);
  ctx->ws_snr.release();
  ctx->ws_add_batch.release();
  ctx->pin_add_batch.release();
  if (ctx->ev_add_batch)
    (void) hipEventDestroy (ctx->ev_add_batch);
  ctx->ev_add_batch = nullptr;
  ctx->ws_merge_soft.release();
  for (hipEvent_t e : ctx->merge_events)
    if (e)
      (void) hipEventDestroy (e);
  ctx->merge_events.clear();
  ctx->ws_rate_a.release();
  ctx->ws_rate_b.release();
  ctx->ws_rate_c.release();
  ctx->tab_mem.release();
  ctx->tab_slide.release();
  for (auto& l : ctx->extra_lanes)
    if (l)
      {
        if (l->stream)
          (void) hipStreamSynchronize (l->stream);
        l->release_lane();
        if (l->own_stream && l->stream)
          (void) hipStreamDestroy (l->stream);
      }
  ctx->release_lane();
  if (ctx->copy_stream)
    (void) hipStreamSynchronize (ctx->copy_stream);
  ctx->file_staging.release();
  ctx->ready.release();
  if (ctx->copy_stream)
    (void) hipStreamDestroy (ctx->copy_stream);
  if (ctx->own_stream && ctx->stream)
    (void) hipStreamDestroy (ctx->stream);
  delete ctx;
}

int awm_ctx_device (const awm_ctx *ctx) { return ctx ? ctx->device : -1; }

int
awm_ctx_synchronize (awm_ctx *ctx)
{
  if (!ctx)
    return AWM_ERR_ARG;
  AWM_HIP_CHECK (hipStreamSynchronize (ctx->stream));
  return 0;
}

void *awm_ctx_stream (awm_ctx *ctx) { return ctx ? (void *) ctx->stream : nullptr; }

int
awm_ctx_set_stream (awm_ctx *ctx, void *hip_stream)
{
  if (!ctx)
    return AWM_ERR_ARG;
  AWM_HIP_CHECK (hipStreamSynchronize (ctx->stream));
  if (ctx->own_stream && ctx->stream)
    (void) hipStreamDestroy (ctx->stream);
  ctx->stream = (hipStream_t) hip_stream;
  ctx->own_stream = false;
  return 0;
}

} // extern "C"
