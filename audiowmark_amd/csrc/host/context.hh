// awm_ctx: per-GPU state of the watermark path -- HIP stream, constant tables in HBM,
// per-key device tables (cached) and grow-only workspaces sized for 288 GB parts.
#pragma once
#include <condition_variable>
#include <mutex>
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>
#include "../hip/kernels.hh"
#include "wmcommon.hh"

namespace awm { void set_error (const std::string& msg); const std::string& last_error(); std::string hip_error_string (hipError_t e); }

#define AWM_HIP_CHECK(expr) \
  do { hipError_t e__ = (expr); if (e__ != hipSuccess) { awm::set_error (std::string (#expr) + ": " + awm::hip_error_string (e__)); return AWM_ERR_HIP; } } while (0)

namespace awm {

// Wait for the stream (or an event) by polling: a blocking hipStreamSynchronize wakes the thread up tens of
// microseconds late, which is paid at every point where the host has to look at a device result before it can
// issue the next kernels.  Falls back to the blocking call after a few milliseconds.
// Between two queries the thread stays off the runtime's locks for about a microsecond, so that another host thread
// driving another lane can get its calls through.
inline void
cpu_relax()
{
  for (int k = 0; k < 48; k++)
    __builtin_ia32_pause();
}
// (several host threads polling at once fight over the runtime's locks: batch workers set this and block instead)
inline bool& wait_blocking() { static thread_local bool b = false; return b; }
inline hipError_t
stream_wait (hipStream_t st)
{
  if (wait_blocking())
    return hipStreamSynchronize (st);
  for (int i = 0; i < 20000; i++)
    {
      const hipError_t e = hipStreamQuery (st);
      if (e != hipErrorNotReady)
        return e;
      cpu_relax();
    }
  return hipStreamSynchronize (st);
}
inline hipError_t
event_wait (hipEvent_t ev)
{
  if (wait_blocking())
    return hipEventSynchronize (ev);
  for (int i = 0; i < 20000; i++)
    {
      const hipError_t e = hipEventQuery (ev);
      if (e != hipErrorNotReady)
        return e;
      cpu_relax();
    }
  return hipEventSynchronize (ev);
}

// grow-only device buffer
struct DevBuffer
{
  void  *ptr = nullptr;
  size_t bytes = 0;
  static constexpr size_t MAX_BYTES = size_t (1) << 40;     // no device holds more: larger requests are arithmetic gone wrong
  int reserve (size_t want);     // 0 ok
  void release();
  template<class T> T *as() const { return static_cast<T *> (ptr); }
};

// grow-only page-locked host buffer: staging area for copies that must not block the host (hipMemcpyAsync from or
// to pageable memory is staged by the runtime and stalls the calling thread)
struct PinnedBuffer
{
  void  *ptr = nullptr;
  size_t bytes = 0;
  int reserve (size_t want);
  void release();
  template<class T> T *as() const { return static_cast<T *> (ptr); }
};

// device copies of the key-derived tables
struct KeyTables
{
  std::vector<unsigned char> key;
  unsigned long last_use = 0;
  bool mix = true;                 // params().mix the data tables were built for (--linear: per-frame up / down bands)
  int  frames_per_bit = 2;         // ... and params().frames_per_bit (the block's geometry: 510 sync + 858 frames_per_bit data frames)
  // sync tables, BLOCK and CLIP flavour
  struct Sync
  {
    SyncTable host;
    DevBuffer packed_approx;       // [6][rows][64] row = frame
    DevBuffer packed_refine;       // [6][rows][64] row = position in the want list
    DevBuffer chains_approx;       // [12][rows][8] words: byte-packed per-chain rows of packed_approx for K5w (scan.hip)
    DevBuffer row_frames;          // group tables only (KeyTables::slices): [slice][6][rows] frame of every row
    std::vector<int> want_list;    // sorted sync frames (510 or 1020)
    DevBuffer want_list_dev;
    DevBuffer refine_perm;         // [want rows] int: row w of the want list -> bit * rows_per_bit + j (K4s gathered layout)
    DevBuffer refine_pos;          // [want rows][81] uint8: band -> 0..29 (up), 30..59 (down), 255 (unused)
  } sync[2];
  MixTable  mix_host;
  DevBuffer mix_frame, mix_up, mix_down;
  std::vector<unsigned> bit_order_a;        // randomize_bit_order permutation for 858 bits
  DevBuffer bit_order_inv_dev;              // int [858]: restored[k] = raw[inv[k]]
  // A batch of clips with ONE KEY PER CLIP (wmget.cc clip_batch_staged): this object then describes the tables of a whole group --
  // every device table above holds `slices` tables back to back (slice i = clip i of the group; only sync[1], the mix tables and
  // the bit order are filled), slice_want[i] = the want list of slice i.  0: one key, the normal case.
  // Tables built on the device (K16g): the want lists come back in one page-locked block, slice_want_flat[i * n + w].
  int slices = 0;
  std::vector<std::vector<int>> slice_want;
  const int *slice_want_flat = nullptr;
  int        slice_want_n = 0;
  const int *want_of_slice (int i) const { return slice_want_flat ? slice_want_flat + size_t (i) * slice_want_n : slice_want[i].data(); }
  int        want_rows_of_slices() const { return slice_want_flat ? slice_want_n : int (slice_want[0].size()); }
};

// host side tables of ONE key for the clip batch path with a key per clip (CLIP mode sync tables in the kernels' formats, mix table,
// bit order): what get_key_tables uploads per key, without device buffers -- the batch path packs a whole group into one upload
struct ClipKeyHost
{
  std::vector<unsigned>      chains;      // [12][170][8]   K5w
  std::vector<int>           row_frames;  // [6][170]       K5w (CLIP mode: rows by frame)
  std::vector<int>           want;        // [1020] sorted sync frames of the long block
  std::vector<int>           perm;        // [1020]         K4s gathered layout
  std::vector<unsigned char> pos;         // [1020][81]
  MixTable                   mix;
  std::vector<int>           inv_order;   // [858]
};
ClipKeyHost build_clip_key_host (const Key& key);

// polyphase table of zita-resampler's fixed-ratio Resampler for one (input rate, output rate) pair (hlen 16)
struct ResampleTable
{
  int       rate_in = 0, rate_out = 0;
  int       hl = 0, np = 0, step = 0;
  DevBuffer ctab;                  // (np + 1) * hl floats
};

// coefficient table of zita-resampler (Resampler_table): (np + 1) * hl floats
std::vector<float> zita_table (double frel, unsigned hl, unsigned np);
// grow `buf` to `bytes` and copy host memory into it on `st`; returns when the copy is done (the source may be a temporary)
int upload_sync (DevBuffer& buf, const void *src, size_t bytes, hipStream_t st);

struct SpeedWorkspace;            // wmspeed.hh
struct SpeedScratch;
struct WorkLane;
void speed_scratch_free (WorkLane *lane);
} struct awm_ctx; namespace awm {
void speed_workspace_free (awm_ctx *ctx);

// Staging of the file level calls (host/wmfile.cc), kept by the context between calls like every other workspace: rings of
// page-locked tiles and their device-side twins for both directions, the events that order them, and the float32 PCM of a whole
// stream for `get` (hipHostMalloc / hipMalloc + the frees of these cost 20 - 80 ms per call when done call by call).
struct FileStaging
{
  static constexpr int RING = 4;
  PinnedBuffer in_host[RING], out_host[RING];
  DevBuffer    in_dev[RING], out_dev[RING];
  DevBuffer    pcm;
  // awm_add_get_watermark_file ("watermark, then verify"): the output stage also leaves what it writes -- the samples as the file holds
  // them, i.e. after the sample format's quantisation -- in `pcm` as float32, so that `get` never reads the file back
  bool         keep = false;
  size_t       kept_values = 0;
  int          kept_channels = 0, kept_rate = 0;
  hipEvent_t   in_copied[RING] = {}, in_used[RING] = {}, out_encoded[RING] = {}, out_copied[RING] = {};
  bool         have_events = false;
  bool         ensure_events();
  void         release();
};

// awm_add_get_watermark_d ("watermark, then verify"): `add` leaves marks on the context's stream -- the limiter has passed sample
// `upto` of the output -- so that `get` of the same buffer can start a chunk behind ITS mark instead of behind the whole add.  Only
// the fused entry point arms them: two separate calls cannot know what the caller queued on the stream in between.
struct ReadyMarks
{
  const float *base = nullptr;           // the output buffer the marks belong to
  size_t       n_frames = 0;
  bool         armed = false;
  std::vector<std::pair<size_t, hipEvent_t>> marks;    // (samples per channel that are final, event) in stream order
  std::vector<hipEvent_t> pool;          // events kept between calls
  size_t       used = 0;
  hipEvent_t   next_event();             // nullptr on failure
  // LIVE marks (the file level `get`: a loader thread is still bringing the stream in while the chunks start): marks appear over
  // time, under `mu`; a chunk that needs samples nobody has marked yet waits on `cv` -- until the mark is there or the loader is done
  bool         live = false, live_done = false;
  std::mutex   mu;
  std::condition_variable cv;
  void         disarm() { armed = false; live = live_done = false; marks.clear(); used = 0; base = nullptr; n_frames = 0; }
  void         release();
};

struct FrameModTable
{
  std::vector<unsigned char> key;
  std::string payload;
  bool        mix = true;
  int         frames_per_bit = 2;
  unsigned long last_use = 0;
  DevBuffer   dev;                 // [2 * block frames][81] int8 (2226 frames with two frames per bit)
};

} // namespace awm

namespace awm {
// per-kernel timing with HIP events on the context's stream (awm_prof_* in awm_hip.h)
enum ProfId { PROF_ADD_MIX, PROF_LIMITER, PROF_SYNC_DB, PROF_SYNC_SCAN, PROF_LOCAL_MEAN, PROF_REFINE_DB, PROF_REFINE_SCAN,
              PROF_BLOCK_DB, PROF_SOFT_BITS, PROF_VITERBI, PROF_STFT,
              PROF_RESAMPLE, PROF_RESAMPLE_VAR, PROF_SPEED_MAGS, PROF_SPEED_COMPARE, PROF_KEYTAB, PROF_COUNT };
struct ProfPending { int id; hipEvent_t start, stop; };
}

namespace awm {
// A stream with its own workspaces and staging buffers.  The context itself is lane 0 (its stream is the one the
// caller sees); `get` runs the chunks of a stream on up to MAX_LANES lanes concurrently, so that the latency bound
// parts of one chunk (candidate round trip, refinement scan, Viterbi) overlap with the wide kernels of the others.
struct WorkLane
{
  hipStream_t    stream = nullptr;
  bool           own_stream = false;
  // workspaces
  DevBuffer ws_db, ws_block_db, ws_have, ws_q, ws_raw, ws_mean, ws_misc, ws_refine, ws_refine_have, ws_soft,
            ws_viterbi, ws_viterbi_in, ws_viterbi_bits, ws_viterbi_err, ws_viterbi_sync, ws_block_max, ws_clip, ws_idx, ws_limit_tab, ws_jobs, ws_group, ws_keytab, ws_keytab_aux, ws_keytab_scratch;
  DevBuffer ws_shard_edge, ws_shard_tail, ws_shard_q;      // multi-GPU protocol (wmshard.cc): edge frames, stitched tail buffer, score blocks
  // host staging (two refinement slots: see SyncFinder::SearchJob)
  PinnedBuffer pin_refine_in[2], pin_refine_q[2], pin_peaks, pin_blocks, pin_jobs, pin_bits, pin_small, pin_group, pin_shard, pin_shard_up, pin_keytab;
  hipEvent_t   ev_refine[2] = { nullptr, nullptr };
  hipEvent_t   ev_sync = nullptr;        // cross-lane ordering (input ready / lane done)
  SpeedScratch *speed_scratch = nullptr; // buffers of a speed search on this lane (wmspeed.cc), created on first use
  // Groups of padded clips (wmget.cc clip_batch_staged): the share of a padded slice's frames that carry samples.  The kernels skip the
  // silent frames of the padding (syncfinder.cc:578-590), so the ALGORITHMIC bytes the profiling scopes report for a group are scaled by it.
  double       prof_live_fraction = 1.0;
  unsigned int *viterbi_sync (size_t n_decodes);          // the one-launch Viterbi kernel's sync block (zero between launches); nullptr on failure
  void release_lane();
};
constexpr int MAX_LANES = 16;       // lanes a context can own (batch of clips: one clip per lane)
constexpr int CHUNK_LANES = 4;      // lanes the chunks of ONE stream are spread over
}

struct awm_ctx : awm::WorkLane
{
  int            device = -1;
  awmk::DevTables tabs {};
  awm::DevBuffer tab_mem, tab_slide;

  std::vector<std::unique_ptr<awm::KeyTables>>     key_tables;
  std::vector<std::unique_ptr<awm::FrameModTable>> frame_mod_tables;
  std::vector<std::unique_ptr<awm::ResampleTable>> resample_tables;
  awm::ResampleTable *get_resample_table (int rate_in, int rate_out);      // nullptr: ratio not supported by the fixed-ratio resampler
  awm::DevBuffer ws_rate_a, ws_rate_b, ws_rate_c;                          // resampled input / watermark signals of the other-rate add path
  std::unique_ptr<awm::WorkLane> extra_lanes[awm::MAX_LANES - 1];
  awm::WorkLane *lane (int i);           // 0 = the context itself; others are created on first use (nullptr on failure)
  awm::FileStaging file_staging;         // rings + whole-stream buffer of the file level calls (grow-only, like the workspaces)
  awm::ReadyMarks ready;                 // add -> get hand-over of awm_add_get_watermark_d
  hipStream_t    copy_stream = nullptr;  // H2D / D2H staging of the file level paths (created on first use)
  hipStream_t    get_copy_stream();
  std::vector<awm_ctx *> helpers;        // other GPUs the file level `get` may spread a long stream over (awm_ctx_set_helpers; not owned)
  std::unique_ptr<awm::ParamValues> own_params;   // settings of this context (awm_ctx_set_params); null: the process-wide ones
  awm::DevBuffer ws_merge_soft;          // raw soft bits of the chunks of one `get` whose decodes run as ONE batch at the end (wmget.cc: block_decoder_run)
  std::vector<hipEvent_t> merge_events;  // one per chunk in flight: its rows are in ws_merge_soft
  awm::DevBuffer    ws_add_batch;        // batches of clips in one launch per stage (capi_kernels.cc add_clips_batched): the clips' kernel arguments,
  awm::PinnedBuffer pin_add_batch;       // their page-locked staging,
  hipEvent_t        ev_add_batch = nullptr;   // and "the staging has been copied" (the next batch may overwrite it)
  awm::DevBuffer ws_snr;                 // `add --snr`: { power of the watermark signal, power of the input } accumulated by every mix while snr_on
  bool           snr_on = false;
  int            chunk_lanes = awm::CHUNK_LANES;   // lanes the chunks of one stream may be spread over (awm_ctx_set_chunk_lanes)

  awm::SpeedWorkspace *speed = nullptr;  // tables and buffers of the speed detection (wmspeed.cc), created on first use
  std::mutex     speed_mutex;            // one speed search at a time per context

  std::mutex     table_mutex;            // key / frame_mod table caches (lanes may be driven by different host threads)
  unsigned long  table_clock = 0;        // LRU stamps of both caches
  static constexpr size_t MAX_CACHED_TABLES = 64;
  std::mutex     prof_mutex;
  // profiling
  bool   prof_enabled = false;
  std::vector<awm::ProfPending> prof_pending;
  double prof_ms[awm::PROF_COUNT] = { 0 };
  long   prof_launches[awm::PROF_COUNT] = { 0 };
  double prof_bytes[awm::PROF_COUNT] = { 0 };
  void   prof_collect();

  awm::KeyTables     *get_key_tables (const awm::Key& key);
  awm::FrameModTable *get_frame_mod (const awm::Key& key, const std::string& payload_hex);
};

#include "../../../include/awm_hip.h"

namespace awm {
// brackets the launches issued during its lifetime with two events (only when profiling is on)
struct ProfScope
{
  awm_ctx *ctx;
  int      id;
  hipEvent_t start = nullptr;
  hipStream_t st;
  ProfScope (awm_ctx *c, int i, double algorithmic_bytes, hipStream_t stream = nullptr) : ctx (c), id (i), st (stream ? stream : c->stream)
  {
    if (!ctx->prof_enabled)
      return;
    {
      std::lock_guard<std::mutex> lock (ctx->prof_mutex);
      ctx->prof_bytes[id] += algorithmic_bytes;
      ctx->prof_launches[id]++;
    }
    if (hipEventCreate (&start) != hipSuccess) { start = nullptr; return; }
    (void) hipEventRecord (start, st);
  }
  ~ProfScope()
  {
    if (!start)
      return;
    hipEvent_t stop = nullptr;
    if (hipEventCreate (&stop) != hipSuccess) { (void) hipEventDestroy (start); return; }
    (void) hipEventRecord (stop, st);
    std::lock_guard<std::mutex> lock (ctx->prof_mutex);
    ctx->prof_pending.push_back ({ id, start, stop });
  }
};
}
