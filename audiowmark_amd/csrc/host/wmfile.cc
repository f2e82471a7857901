#include "wmfile.hh"
#include "context.hh"
#include "wmget.hh"
#include "utils.hh"
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <unistd.h>

namespace awm {

// The file level functions return 0 / 1 like the reference's (the value becomes the exit code of the command line); the C ABI
// wants to tell argument, I/O and GPU failures apart: every failure path notes its kind for the calling thread.
static thread_local int tl_fail_kind = 0;
int file_fail_kind() { return tl_fail_kind ? tl_fail_kind : AWM_ERR_GENERIC; }
void file_fail_reset() { tl_fail_kind = 0; }
static int fail (int kind) { tl_fail_kind = kind; return 1; }


namespace {

/* File <-> HBM staging with BOUNDED host memory: the stream crosses PCIe tile by tile through small rings of page-locked buffers,
 * in its own sample format where the stream can hand out its bytes (raw / WAV): conversion is RawConverter's arithmetic on the
 * device (awm_pcm_decode_d / awm_pcm_encode_d).
 *
 * The HOST side of that is memory copies between the page cache and the rings, and one thread moves ~5 - 10 GB/s -- an hour of
 * 16 bit stereo is 635 MB each way, a few milliseconds of GPU work.  So the copies run on a pool of workers:
 *   input   regular files (AudioInputStream::raw_region): the tiles are read AHEAD of the consumer, every tile split over the
 *           workers (pread); pipes and streams without byte access: one reader thread, still ahead of the consumer;
 *   output  ONE writer thread in stream order.  Creating the page cache pages of one file is serial in the kernel whatever the number of
 *           writers (a fresh 640 MB tmpfs file: 85 - 105 ms with one thread's write(), 109 - 258 ms with 4 - 16 threads' pwrite -- the inode
 *           lock --, 180 - 196 ms with 4 - 16 threads copying into a shared mapping -- the file's page tree --, 14 - 32 ms when the same
 *           threads write SEPARATE files: tools/io_probe.cc -> profiles/r05/io_probe.txt), so the ordered writer is at the floor.  The
 *           worker path for regular output files of known length (AudioOutputStream::raw_region: the file grown to its final size and
 *           mapped once, tiles copied by the workers) exists behind awm_debug_set_io_flags, writes the same bytes and is not faster.      */
constexpr size_t STAGE_FRAMES = size_t (1) << 22;       // frames per staging tile of the whole-stream loaders (16 MiB of 16 bit stereo)
constexpr int    IN_RING = FileStaging::RING, OUT_RING = FileStaging::RING;   // page-locked tiles per direction (kept by the context)
constexpr size_t IO_PART_MIN = size_t (1) << 20;        // a worker's share of a tile is at least this many bytes

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23                          // (Linux 5.14; older headers)
#endif
enum { IO_REGIONS = 1, IO_MAP_OUTPUT = 2, IO_POPULATE = 4, IO_REGIONS_OUT = 8 };          // awm_debug_set_io_flags
// where the wall time of the last file level add / load of this thread went (awm_debug_file_timing): milliseconds the calling thread
// spent { setting up (streams, rings, tables), waiting for input tiles, waiting for a free output slot, queueing GPU work, in the
// final wait for the GPU, in the final wait for the writers, tearing down, handing output tiles on (incl. the wait for a slot) }
static thread_local double tl_file_ms[8] = { 0 };
struct Lap
{
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void to (int i) { const auto n = std::chrono::steady_clock::now(); tl_file_ms[i] += std::chrono::duration<double, std::milli> (n - t).count(); t = n; }
};
static std::atomic<int> g_io_threads { 0 };             // 0: default
static std::atomic<int> g_io_flags { IO_REGIONS };      // default: input through the workers, output by the ordered writer thread (see above)
int
io_threads()
{
  const int n = g_io_threads.load (std::memory_order_relaxed);
  if (n > 0)
    return std::min (n, 64);
  // (reads of a tile scale with the workers up to ~16 on the box this was tuned on: 640 MB from tmpfs in 70 / 18 / 9 / 7 ms with
  // 1 / 4 / 8 / 16 threads, tools/io_probe.cc; the workers sleep while the GPU and the writer are busy)
  const unsigned hw = std::thread::hardware_concurrency();
  return int (std::max (2u, std::min (16u, hw ? hw : 4u)));
}

/* worker threads for the host side copies; one pool per process, started at the first file level call and resized when the setting
 * changes BETWEEN calls (awm_set_io_threads is not meant to be called while another thread is inside a file level call) */
class IoPool
{
  std::vector<std::thread> m_threads;
  std::mutex               m_mutex;
  std::condition_variable  m_cond;
  std::deque<std::function<void()>> m_jobs;
  bool                     m_quit = false;
  void
  run()
  {
    for (;;)
      {
        std::function<void()> job;
        {
          std::unique_lock<std::mutex> lock (m_mutex);
          m_cond.wait (lock, [&] { return m_quit || !m_jobs.empty(); });
          if (m_jobs.empty())
            return;
          job = std::move (m_jobs.front());
          m_jobs.pop_front();
        }
        job();
      }
  }
  void
  stop()
  {
    {
      std::lock_guard<std::mutex> lock (m_mutex);
      m_quit = true;
      m_cond.notify_all();
    }
    for (auto& t : m_threads)
      if (t.joinable())
        t.join();
    m_threads.clear();
    m_quit = false;
  }
public:
  ~IoPool() { stop(); }
  static IoPool&
  get()
  {
    static IoPool pool;
    static std::mutex create_mutex;
    std::lock_guard<std::mutex> lock (create_mutex);
    const size_t want = size_t (io_threads());
    if (pool.m_threads.size() != want)
      {
        pool.stop();                                       // (queued jobs are finished first: workers leave only on an empty queue)
        for (size_t i = 0; i < want; i++)
          pool.m_threads.emplace_back ([p = &pool] { p->run(); });
      }
    return pool;
  }
  size_t size() const { return m_threads.size(); }
  void
  submit (std::function<void()> job)
  {
    std::lock_guard<std::mutex> lock (m_mutex);
    m_jobs.push_back (std::move (job));
    m_cond.notify_one();
  }
};

bool
device_codec_supported (const RawFormat& f)
{
  if (f.encoding == Encoding::FLOAT)
    return f.bit_depth == 32 || f.bit_depth == 64;
  return f.bit_depth == 8 || f.bit_depth == 16 || f.bit_depth == 24 || f.bit_depth == 32;
}

int encoding_id (Encoding e) { return e == Encoding::SIGNED ? 0 : (e == Encoding::UNSIGNED ? 1 : 2); }

/* up to max_frames frames from the stream into `dst` (sample bytes if `raw`, else float32), looping over short reads */
Error
read_chunk (AudioInputStream *in, bool raw, size_t unit_bytes, unsigned char *dst, size_t max_frames, size_t& got)
{
  got = 0;
  if (raw)
    while (got < max_frames)
      {
        size_t n = 0;
        Error err = in->read_raw (dst + got * unit_bytes, max_frames - got, n);
        if (err)
          return err;
        if (!n)
          break;
        got += n;
      }
  else
    {
      std::vector<float> part;
      while (got < max_frames)
        {
          Error err = in->read_frames (part, std::min<size_t> (max_frames - got, 1 << 20));
          if (err)
            return err;
          if (part.empty())
            break;
          std::copy (part.begin(), part.end(), reinterpret_cast<float *> (dst) + got * (unit_bytes / sizeof (float)));
          got += part.size() / (unit_bytes / sizeof (float));
        }
    }
  return Error::Code::NONE;
}

/* Input tiles, read AHEAD of the consumer into a ring of page-locked buffers.
 *   next()     hands out tile 0, 1, 2, ... in order (blocks until it is in memory); frames < tile_frames: the stream ends there
 *   recycle()  the consumer has queued its last use of the slot (an H2D copy) and recorded `done` behind it: the slot is refilled
 *              with the tile IN_RING further on as soon as `done` has happened
 * One copy stream per context serves both directions: every additional HIP stream costs ~190 MB of resident host memory
 * on this runtime (tools/rss_probe), and a tile crosses PCIe in well under a millisecond. */
class TileReader
{
  struct Slot
  {
    unsigned char *host = nullptr;           // the context's page-locked tile (FileStaging)
    // state of the tile the slot currently holds (guarded by m_mutex)
    long long    tile = -1;                  // which tile has been scheduled into the slot
    int          parts_left = 0;
    std::vector<size_t> part_got;            // bytes per part (region mode: a short part = the file ends inside it)
    bool         failed = false;
    int          err_no = 0;
    size_t       frames = 0;                 // sequential mode: frames read
    Error        error = Error::Code::NONE;
  };
  awm_ctx          *m_ctx;
  AudioInputStream *m_in;
  bool              m_raw;
  size_t            m_unit, m_tile_frames;
  Slot              m_slots[IN_RING];
  std::mutex              m_mutex;
  std::condition_variable m_cond;
  long long         m_next = 0;              // next tile next() hands out
  bool              m_ended = false;         // a short tile has been handed out
  // region mode
  bool              m_region = false;
  int               m_fd = -1;
  uint64_t          m_offset = 0;
  size_t            m_total_frames = 0, m_consumed = 0;
  long long         m_n_tiles = 0;
  // sequential mode (pipes, streams without byte access): one reader thread
  std::thread       m_thread;
  std::deque<std::pair<int, hipEvent_t>> m_free;      // slots the reader may fill, with the event to wait for first (nullptr: none)
  bool              m_quit = false;

  void
  schedule_region (int slot, long long tile, hipEvent_t after)
  {
    Slot& s = m_slots[slot];
    const size_t first = size_t (tile) * m_tile_frames;
    const size_t frames = std::min (m_tile_frames, m_total_frames - first);
    const size_t bytes = frames * m_unit;
    IoPool& pool = IoPool::get();
    const size_t n_parts = std::max<size_t> (1, std::min (pool.size(), bytes / IO_PART_MIN));
    const size_t part = ((bytes + n_parts - 1) / n_parts + 4095) & ~size_t (4095);
    {
      std::lock_guard<std::mutex> lock (m_mutex);
      s.tile = tile;
      s.parts_left = int (n_parts);
      s.part_got.assign (n_parts, 0);
      s.failed = false;
    }
    const int device = m_ctx->device;
    auto read_part = [this, &s, part, bytes, first] (size_t i) {
      const size_t lo = std::min (bytes, part * i), hi = std::min (bytes, part * (i + 1));
      unsigned char *dst = s.host + lo;
      size_t n = 0;
      bool bad = false;
      int err = 0;
      while (lo + n < hi)
        {
          const ssize_t r = pread (m_fd, dst + n, hi - lo - n, off_t (m_offset + first * m_unit + lo + n));
          if (r < 0 && errno == EINTR)
            continue;
          if (r <= 0)
            {
              bad = r < 0;
              err = r < 0 ? errno : 0;
              break;
            }
          n += size_t (r);
        }
      std::lock_guard<std::mutex> lock (m_mutex);
      s.part_got[i] = n;
      if (bad && !s.failed)
        {
          s.failed = true;
          s.err_no = err;
        }
      if (--s.parts_left == 0)
        m_cond.notify_all();
    };
    // the first job waits for the slot's previous contents to have crossed PCIe, then fans the parts out
    pool.submit ([=, &pool] {
      if (after)
        {
          (void) hipSetDevice (device);
          (void) hipEventSynchronize (after);
        }
      for (size_t i = 1; i < n_parts; i++)
        pool.submit ([=] { read_part (i); });
      read_part (0);
    });
  }
  void
  run_sequential()
  {
    (void) hipSetDevice (m_ctx->device);
    for (long long tile = 0; ; tile++)
      {
        std::pair<int, hipEvent_t> f;
        {
          std::unique_lock<std::mutex> lock (m_mutex);
          m_cond.wait (lock, [&] { return m_quit || !m_free.empty(); });
          if (m_quit)
            return;
          f = m_free.front();
          m_free.pop_front();
        }
        if (f.second)
          (void) hipEventSynchronize (f.second);
        Slot& s = m_slots[f.first];
        size_t got = 0;
        Error err = read_chunk (m_in, m_raw, m_unit, s.host, m_tile_frames, got);
        std::lock_guard<std::mutex> lock (m_mutex);
        s.tile = tile;
        s.frames = got;
        s.error = err;
        s.parts_left = 0;
        m_cond.notify_all();
        if (err || got < m_tile_frames)
          return;
      }
  }
public:
  bool ok = false;
  hipStream_t copy = nullptr;
  TileReader (awm_ctx *ctx, AudioInputStream *in, bool raw, size_t unit_bytes, size_t tile_frames, bool need_dev)
    : m_ctx (ctx), m_in (in), m_raw (raw), m_unit (unit_bytes), m_tile_frames (tile_frames)
  {
    copy = ctx->get_copy_stream();
    FileStaging& fs = ctx->file_staging;
    ok = copy != nullptr && fs.ensure_events();
    for (int i = 0; i < IN_RING && ok; i++)
      {
        ok = fs.in_host[i].reserve (tile_frames * unit_bytes) == 0 && (!need_dev || fs.in_dev[i].reserve (tile_frames * unit_bytes) == 0);
        m_slots[i].host = fs.in_host[i].as<unsigned char>();
      }
    if (!ok)
      return;
    m_region = raw && (g_io_flags.load (std::memory_order_relaxed) & IO_REGIONS) != 0 && in->raw_region (m_fd, m_offset, m_total_frames);
    if (m_region)
      {
        m_n_tiles = (long long) ((m_total_frames + tile_frames - 1) / tile_frames);
        for (int i = 0; i < IN_RING && i < m_n_tiles; i++)
          schedule_region (i, i, nullptr);
      }
    else
      {
        for (int i = 0; i < IN_RING; i++)
          m_free.emplace_back (i, nullptr);
        for (auto& s : m_slots)
          s.parts_left = 1;                  // "not there yet"
        m_thread = std::thread ([this] { run_sequential(); });
      }
  }
  ~TileReader()
  {
    {
      std::unique_lock<std::mutex> lock (m_mutex);
      m_quit = true;
      m_cond.notify_all();
      if (m_region)                          // reads in flight still write into the ring
        m_cond.wait (lock, [&] { for (auto& s : m_slots) if (s.tile >= 0 && s.parts_left) return false; return true; });
    }
    if (m_thread.joinable())
      m_thread.join();
    if (m_region && m_consumed)
      m_in->raw_region_consume (m_consumed);
    if (copy)
      (void) hipStreamSynchronize (copy);               // (the rings stay with the context; nothing of this call may still use them:
    (void) hipStreamSynchronize (m_ctx->stream);        //  on an error return sample decodes of this call may still be queued)
  }
  size_t announced_frames() const { return m_region ? m_total_frames : m_in->n_frames(); }
  unsigned char *host (int slot) { return m_slots[slot].host; }
  void          *dev (int slot) { return m_ctx->file_staging.in_dev[slot].ptr; }
  hipEvent_t     ev_copied (int slot) { return m_ctx->file_staging.in_copied[slot]; }
  hipEvent_t     ev_used (int slot) { return m_ctx->file_staging.in_used[slot]; }
  /* the next tile: its slot and its frames (0 at / after the end of the stream) */
  Error
  next (int& slot, size_t& frames)
  {
    frames = 0;
    slot = int (m_next % IN_RING);
    if (m_ended || (m_region && m_next >= m_n_tiles))
      return Error::Code::NONE;
    Slot& s = m_slots[slot];
    std::unique_lock<std::mutex> lock (m_mutex);
    m_cond.wait (lock, [&] { return s.tile == m_next && s.parts_left == 0; });
    if (m_region)
      {
        if (s.failed)
          return Error (string_printf ("error reading sample data: %s", strerror (s.err_no)));
        const size_t first = size_t (m_next) * m_tile_frames;
        const size_t want = std::min (m_tile_frames, m_total_frames - first) * m_unit;
        const size_t n_parts = s.part_got.size();
        const size_t part = ((want + n_parts - 1) / n_parts + 4095) & ~size_t (4095);
        size_t bytes = 0;
        for (size_t i = 0; i < n_parts; i++)
          {
            bytes += s.part_got[i];
            if (s.part_got[i] < std::min (want, part * (i + 1)) - std::min (want, part * i))
              break;                                             // the file ends inside this part: later parts lie beyond it
          }
        frames = bytes / m_unit;
        m_consumed += frames;
      }
    else
      {
        if (s.error)
          return s.error;
        frames = s.frames;
      }
    if (frames < m_tile_frames)
      m_ended = true;
    m_next++;
    return Error::Code::NONE;
  }
  void
  recycle (int slot, hipEvent_t done)
  {
    if (m_ended)
      return;
    if (m_region)
      {
        const long long tile = m_slots[slot].tile + IN_RING;
        if (tile < m_n_tiles)
          schedule_region (slot, tile, done);
      }
    else
      {
        std::lock_guard<std::mutex> lock (m_mutex);
        m_slots[slot].parts_left = 1;
        m_free.emplace_back (slot, done);
        m_cond.notify_all();
      }
  }
};

/* Whole stream -> float32 PCM in HBM (288 GB hold days of audio; what is bounded is the HOST side: the ring of staging tiles). */
/* live != nullptr: the caller decodes the stream's chunks WHILE this function (on a thread of its own) brings the stream in: the buffer
 * has its final size (announced frames + 1) and never moves, copies AND sample decodes run on the copy stream (the compute stream
 * belongs to the decoder), and behind every tile a mark "the stream is final up to frame n" with an event on the copy stream is
 * published under live->mu (ReadyMarks, context.hh).  The last act, on every path out, is live_done. */
Error
load_stream_to_device (awm_ctx *ctx, AudioInputStream *in_stream, DevBuffer& d_pcm, size_t& n_values, ReadyMarks *live = nullptr)
{
  struct Done
  {
    ReadyMarks *live;
    ~Done()
    {
      if (live)
        {
          std::lock_guard<std::mutex> lock (live->mu);
          live->live_done = true;
          live->cv.notify_all();
        }
    }
  } done { live };
  const int C = in_stream->n_channels();
  n_values = 0;
  RawFormat fmt;
  const bool raw = in_stream->raw_access (fmt) && device_codec_supported (fmt);
  const size_t unit = raw ? size_t (C) * (fmt.bit_depth / 8) : size_t (C) * sizeof (float);
  TileReader rd (ctx, in_stream, raw, unit, STAGE_FRAMES, raw);
  if (!rd.ok)
    return Error ("out of memory for input staging");
  // The announced length only sizes the first allocation (the loop below grows the buffer when more arrives).  A length whose
  // float32 size would not fit a device buffer -- a crafted ds64 / RIFF size on a pipe, where it cannot be checked against the
  // file -- is treated like an unknown one: cap_frames * C * 4 must never wrap.
  size_t announced = rd.announced_frames();
  if (live && announced == AudioInputStream::N_FRAMES_UNKNOWN)
    announced = live->n_frames;                  // (headerless PCM read by ONE thread, awm_debug_set_io_flags (0): the caller took the length from the file)
  const size_t max_frames = DevBuffer::MAX_BYTES / (size_t (C) * sizeof (float));
  size_t cap_frames = (announced != AudioInputStream::N_FRAMES_UNKNOWN && announced < max_frames) ? announced + 1 : STAGE_FRAMES * 4;
  if (d_pcm.reserve (cap_frames * C * sizeof (float)))
    return Error (awm_last_error());
  size_t frames = 0;
  for (size_t k = 0; ; k++)
    {
      int b = 0;
      size_t got = 0;
      Error err = rd.next (b, got);
      if (err)
        return err;
      if (!got)
        break;
      if (frames + got > cap_frames && live)
        return Error ("input stream is longer than announced");      // (the caller starts over without the overlap)
      if (frames + got > cap_frames)
        {
          // stream of unknown (or understated) length: move to a buffer twice the size
          DevBuffer bigger;
          cap_frames = std::max (cap_frames * 2, frames + got);
          if (cap_frames >= max_frames)
            return Error ("input stream is too long for device memory");
          if (hipStreamSynchronize (rd.copy) != hipSuccess || hipStreamSynchronize (ctx->stream) != hipSuccess
              || bigger.reserve (cap_frames * C * sizeof (float))
              || hipMemcpy (bigger.ptr, d_pcm.ptr, frames * C * sizeof (float), hipMemcpyDeviceToDevice) != hipSuccess)
            return Error ("out of device memory while loading the stream");
          d_pcm.release();
          d_pcm = bigger;
        }
      float *dst = d_pcm.as<float>() + frames * C;
      bool ok;
      if (raw && live)
        {
          // everything on the copy stream, in order: copy, sample decode (the staging buffer is free again behind it)
          const awmk::PcmFormatDev f { fmt.bit_depth / 8, encoding_id (fmt.encoding), fmt.endian == RawFormat::BIG, 0 };
          ok = hipMemcpyAsync (rd.dev (b), rd.host (b), got * unit, hipMemcpyHostToDevice, rd.copy) == hipSuccess
            && hipEventRecord (rd.ev_copied (b), rd.copy) == hipSuccess
            && awmk::launch_pcm_decode (rd.copy, static_cast<const unsigned char *> (rd.dev (b)), dst, (long long) (got * C), f) == hipSuccess;
        }
      else if (raw)
        {
          // (dev[b] was last read by the decode of the tile IN_RING earlier: the copy stream waits for that decode)
          ok = (k < size_t (IN_RING) || hipStreamWaitEvent (rd.copy, rd.ev_used (b), 0) == hipSuccess)
            && hipMemcpyAsync (rd.dev (b), rd.host (b), got * unit, hipMemcpyHostToDevice, rd.copy) == hipSuccess
            && hipEventRecord (rd.ev_copied (b), rd.copy) == hipSuccess
            && hipStreamWaitEvent (ctx->stream, rd.ev_copied (b), 0) == hipSuccess
            && awm_pcm_decode_d (ctx, rd.dev (b), got * C, fmt.bit_depth, encoding_id (fmt.encoding), fmt.endian == RawFormat::BIG, dst) == 0
            && hipEventRecord (rd.ev_used (b), ctx->stream) == hipSuccess;
        }
      else
        ok = hipMemcpyAsync (dst, rd.host (b), got * unit, hipMemcpyHostToDevice, rd.copy) == hipSuccess
          && hipEventRecord (rd.ev_copied (b), rd.copy) == hipSuccess;
      if (!ok)
        return Error (std::string ("GPU staging failed: ") + awm_last_error());
      rd.recycle (b, rd.ev_copied (b));
      frames += got;
      if (live)
        {
          hipEvent_t ev = live->next_event();                 // (only this thread takes events while the marks are live)
          if (!ev || hipEventRecord (ev, rd.copy) != hipSuccess)
            return Error ("GPU staging failed: cannot record an event");
          std::lock_guard<std::mutex> lock (live->mu);
          live->marks.push_back ({ frames, ev });
          live->cv.notify_all();
        }
      if (got < STAGE_FRAMES)
        break;
    }
  if (hipStreamSynchronize (rd.copy) != hipSuccess || (!live && hipStreamSynchronize (ctx->stream) != hipSuccess))
    return Error ("GPU transfer failed");
  n_values = frames * C;
  return Error::Code::NONE;
}

/* Completed output tiles leave the ring on other host threads, so that file writes overlap file reads and GPU work.
 *   wait_slot (b)   until the bytes of slot b have been written (the buffer may be reused)
 *   submit (...)    the D2H copy of slot b has been queued and `ready` recorded behind it
 * Streams in order (pipes, stdout, float samples through write_frames, outputs of unknown length): one writer thread.
 * Regular files of KNOWN length: the file is grown to its final size and mapped ONCE; the workers of the pool copy every tile into its
 * range of the mapping and drop the range's page table entries afterwards (MADV_DONTNEED on a shared mapping: the pages stay in the page
 * cache, the resident set stays bounded by the ring).  Nothing on the calling thread per tile but the queueing: a per-tile ftruncate /
 * mmap / munmap there costs 2 - 3 ms each while the workers fault pages in (inode and address-space locks: measured 90 - 130 ms of
 * the calling thread for an hour of audio, profiles/r05/io_sweep.json). */
class ChunkWriter
{
  struct Job { const unsigned char *bytes; size_t frames; hipEvent_t ready; int slot; bool raw; };
  AudioOutputStream *m_out;
  int                m_channels, m_device;
  std::thread        m_thread;
  std::mutex         m_mutex;
  std::condition_variable m_cond;
  std::deque<Job>    m_jobs;
  std::vector<int>   m_busy;          // per slot: handed to the writer(s) and not yet written
  std::vector<int>   m_parts;         // region mode, per slot: parts of the tile still being copied
  bool               m_quit = false, m_finished = false;
  Error              m_error = Error::Code::NONE;
  // region mode
  bool               m_region = false;
  int                m_fd = -1;
  uint64_t           m_offset = 0;     // file offset of the next tile's first byte
  uint64_t           m_region_start = 0, m_region_bytes = 0;
  size_t             m_unit = 0, m_frames_placed = 0;
  unsigned char     *m_map = nullptr;  // the sample bytes' range of the file, from the start of the page m_region_start lies in
  size_t             m_map_delta = 0, m_map_len = 0;
  void
  note_error (const Error& err)
  {
    if (err && !m_error)
      m_error = err;
  }
  void
  run()
  {
    (void) hipSetDevice (m_device);
    for (;;)
      {
        Job job;
        {
          std::unique_lock<std::mutex> lock (m_mutex);
          m_cond.wait (lock, [&] { return m_quit || !m_jobs.empty(); });
          if (m_jobs.empty())
            return;
          job = m_jobs.front();
          m_jobs.pop_front();
        }
        Error err = Error::Code::NONE;
        if (hipEventSynchronize (job.ready) != hipSuccess)
          err = Error ("GPU transfer failed");
        else if (job.raw)
          err = m_out->write_raw (job.bytes, job.frames);
        else
          {
            const float *f = reinterpret_cast<const float *> (job.bytes);
            err = m_out->write_frames (std::vector<float> (f, f + job.frames * m_channels));
          }
        std::lock_guard<std::mutex> lock (m_mutex);
        note_error (err);
        m_busy[job.slot] = 0;
        m_cond.notify_all();
      }
  }
  /* bytes [lo, hi) of a tile that starts at file offset `at`: through the mapping, or (no mapping) with pwrite */
  Error
  place (uint64_t at, const unsigned char *src, size_t lo, size_t hi, bool populate)
  {
    if (m_map)
      {
        unsigned char *dst = m_map + m_map_delta + (at - m_region_start);
        if (populate)
          {
            // (one call instead of a trap per page; not available on every kernel / file system: ignored then)
            const uintptr_t a = reinterpret_cast<uintptr_t> (dst + lo) & ~uintptr_t (4095);
            (void) madvise (reinterpret_cast<void *> (a), reinterpret_cast<uintptr_t> (dst + hi) - a, MADV_POPULATE_WRITE);
          }
        std::memcpy (dst + lo, src + lo, hi - lo);
        return Error::Code::NONE;
      }
    size_t n = lo;
    while (n < hi)
      {
        const ssize_t r = pwrite (m_fd, src + n, hi - n, off_t (at + n));
        if (r < 0 && errno == EINTR)
          continue;
        if (r <= 0)
          return Error (string_printf ("write sample data failed (%s)", strerror (r < 0 ? errno : ENOSPC)));
        n += size_t (r);
      }
    return Error::Code::NONE;
  }
  void
  submit_region (const unsigned char *bytes, size_t frames, hipEvent_t ready, int slot)
  {
    const size_t n = frames * m_unit;
    const uint64_t at = m_offset;
    m_offset += n;
    m_frames_placed += frames;
    IoPool& pool = IoPool::get();
    const size_t n_parts = std::max<size_t> (1, std::min (pool.size(), n / IO_PART_MIN));
    const size_t part = ((n + n_parts - 1) / n_parts + 4095) & ~size_t (4095);
    {
      std::lock_guard<std::mutex> lock (m_mutex);
      m_busy[slot] = 1;
      m_parts[slot] = int (n_parts);
      if (at + n > m_region_start + m_region_bytes)        // (more frames than announced: cannot happen through add_tiles / store)
        note_error (Error ("output stream is longer than announced"));
    }
    const int device = m_device;
    const bool populate = (g_io_flags.load (std::memory_order_relaxed) & IO_POPULATE) != 0;
    const bool inside = at + n <= m_region_start + m_region_bytes;
    auto write_part = [=] (size_t i) {
      const size_t lo = std::min (n, part * i), hi = std::min (n, part * (i + 1));
      Error err = hi > lo && inside ? place (at, bytes, lo, hi, populate) : Error (Error::Code::NONE);
      bool last;
      {
        std::lock_guard<std::mutex> lock (m_mutex);
        note_error (err);
        last = --m_parts[slot] == 0;
      }
      if (last)
        {
          if (m_map && inside)
            {
              // the tile is in the page cache: drop its page table entries (whole pages inside the tile; the two edge pages are
              // shared with the neighbouring tiles and go with the final munmap)
              unsigned char *dst = m_map + m_map_delta + (at - m_region_start);
              const uintptr_t a = (reinterpret_cast<uintptr_t> (dst) + 4095) & ~uintptr_t (4095), b = reinterpret_cast<uintptr_t> (dst + n) & ~uintptr_t (4095);
              if (b > a)
                (void) madvise (reinterpret_cast<void *> (a), b - a, MADV_DONTNEED);
            }
          std::lock_guard<std::mutex> lock (m_mutex);
          m_busy[slot] = 0;
          m_cond.notify_all();
        }
    };
    pool.submit ([=, &pool] {
      (void) hipSetDevice (device);
      if (hipEventSynchronize (ready) != hipSuccess)
        {
          std::lock_guard<std::mutex> lock (m_mutex);
          note_error (Error ("GPU transfer failed"));
        }
      for (size_t i = 1; i < n_parts; i++)
        pool.submit ([=] { write_part (i); });
      write_part (0);
    });
  }
public:
  /* total_frames: how many frames the caller is going to hand over (AudioInputStream::N_FRAMES_UNKNOWN: unknown -> stream order) */
  ChunkWriter (AudioOutputStream *out, int n_slots, int device, bool raw, size_t unit_bytes, size_t total_frames)
    : m_out (out), m_channels (out->n_channels()), m_device (device), m_busy (n_slots, 0), m_parts (n_slots, 0), m_unit (unit_bytes)
  {
    const int flags = g_io_flags.load (std::memory_order_relaxed);
    m_region = raw && (flags & IO_REGIONS_OUT) != 0 && total_frames != AudioInputStream::N_FRAMES_UNKNOWN && total_frames > 0
            && out->raw_region (m_fd, m_offset);
    if (m_region)
      {
        m_region_start = m_offset;
        m_region_bytes = uint64_t (total_frames) * unit_bytes;
        // the file at its final length (sparse until the tiles arrive).  A mapping of a sparse range cannot report "no space
        // left" -- it raises SIGBUS -- so the mapping is used only if the file system has room for the whole output with a
        // margin now; otherwise (and where mmap is refused) the parts go through pwrite, which reports errors.
        struct stat st;
        const uint64_t have = fstat (m_fd, &st) == 0 ? uint64_t (st.st_size) : 0;
        const bool grown = have >= m_region_start + m_region_bytes || ftruncate (m_fd, off_t (m_region_start + m_region_bytes)) == 0;
        struct statvfs vfs;
        const bool room = fstatvfs (m_fd, &vfs) == 0 && uint64_t (vfs.f_bavail) * vfs.f_frsize > m_region_bytes + (uint64_t (256) << 20);
        if (grown && room && (flags & IO_MAP_OUTPUT))
          {
            m_map_delta = size_t (m_region_start & 4095);
            m_map_len = size_t (m_region_bytes) + m_map_delta;
            void *m = mmap (nullptr, m_map_len, PROT_READ | PROT_WRITE, MAP_SHARED, m_fd, off_t (m_region_start - m_map_delta));
            if (m != MAP_FAILED)
              m_map = static_cast<unsigned char *> (m);
          }
      }
    else
      m_thread = std::thread ([this] { run(); });
  }
  ~ChunkWriter() { finish(); }
  void
  wait_slot (int slot)                // until the bytes of this slot have been written (the buffer may be reused)
  {
    std::unique_lock<std::mutex> lock (m_mutex);
    m_cond.wait (lock, [&] { return !m_busy[slot]; });
  }
  void
  submit (const unsigned char *bytes, size_t frames, hipEvent_t ready, int slot, bool raw)
  {
    if (m_region)
      {
        submit_region (bytes, frames, ready, slot);
        return;
      }
    std::lock_guard<std::mutex> lock (m_mutex);
    m_busy[slot] = 1;
    m_jobs.push_back ({ bytes, frames, ready, slot, raw });
    m_cond.notify_all();
  }
  Error
  finish()
  {
    if (m_finished)
      return m_error;
    m_finished = true;
    if (m_region)
      {
        {
          std::unique_lock<std::mutex> lock (m_mutex);
          m_cond.wait (lock, [&] { for (int b : m_busy) if (b) return false; return true; });
        }
        if (m_map)
          (void) munmap (m_map, m_map_len);
        m_map = nullptr;
        if (m_frames_placed * m_unit < m_region_bytes)       // fewer frames than announced (a truncated input): the file ends with them
          if (ftruncate (m_fd, off_t (m_region_start + m_frames_placed * m_unit)) != 0)
            note_error (Error (string_printf ("write sample data failed (%s)", strerror (errno))));
        Error err = m_out->raw_region_written (m_frames_placed);
        note_error (err);
        return m_error;
      }
    {
      std::lock_guard<std::mutex> lock (m_mutex);
      m_quit = true;
      m_cond.notify_all();
    }
    if (m_thread.joinable())
      m_thread.join();
    return m_error;
  }
};

/* float32 PCM in HBM -> output stream: encode, copy and write tile by tile (mirror image of load_stream_to_device).
 * `OutputStage` is also what the tile loop of `add` uses for its finished tiles. */
struct OutputStage
{
  awm_ctx *ctx;
  AudioOutputStream *out;
  int C;
  RawFormat fmt;
  bool direct16 = false, raw = false;
  size_t unit = 0, chunk_frames;
  static constexpr int SLOTS = OUT_RING;
  hipStream_t copy = nullptr;
  hipEvent_t *ev_encoded = nullptr, *ev_copied = nullptr;
  PinnedBuffer *host = nullptr;
  DevBuffer    *dev = nullptr;
  std::unique_ptr<ChunkWriter> writer;
  size_t k = 0;
  bool ok = false;
  OutputStage (awm_ctx *c, AudioOutputStream *o, size_t frames_per_chunk, size_t total_frames)
    : ctx (c), out (o), C (o->n_channels()), chunk_frames (frames_per_chunk)
  {
    raw = out->raw_access (fmt, direct16) && device_codec_supported (fmt);
    unit = raw ? size_t (C) * (fmt.bit_depth / 8) : size_t (C) * sizeof (float);
    copy = ctx->get_copy_stream();
    FileStaging& fs = ctx->file_staging;
    ok = copy != nullptr && fs.ensure_events();
    ev_encoded = fs.out_encoded;
    ev_copied = fs.out_copied;
    host = fs.out_host;
    dev = fs.out_dev;
    for (int i = 0; i < SLOTS && ok; i++)
      ok = host[i].reserve (chunk_frames * unit) == 0 && (!raw || dev[i].reserve (chunk_frames * unit) == 0);
    if (fs.keep)
      {
        fs.kept_channels = C;
        fs.kept_rate = o->sample_rate();
        if (total_frames != AudioInputStream::N_FRAMES_UNKNOWN && ok)
          ok = fs.pcm.reserve (std::max<size_t> (1, (total_frames + 1) * C * sizeof (float))) == 0;
      }
    if (ok)
      writer = std::make_unique<ChunkWriter> (out, SLOTS, ctx->device, raw, unit, total_frames);
  }
  ~OutputStage()
  {
    writer.reset();
    if (copy)
      (void) hipStreamSynchronize (copy);                 // (the ring stays with the context: on an error return encodes of this call
    (void) hipStreamSynchronize (ctx->stream);            //  may still be queued on the compute stream)
  }
  /* n_frames <= chunk_frames of finished float32 samples at d_pcm (produced on ctx->stream) */
  bool
  put (const float *d_pcm, size_t n_frames)
  {
    if (!n_frames)
      return true;
    const int b = int (k++ % SLOTS);
    {
      Lap lap;
      writer->wait_slot (b);                     // host[b] written out; its copy and encode are long done
      lap.to (2);
    }
    bool good;
    FileStaging& fs = ctx->file_staging;
    if (fs.keep)
      {
        // (grow-only; a stream longer than announced moves to a buffer twice the size)
        const size_t need = (fs.kept_values + n_frames * C) * sizeof (float);
        if (need > fs.pcm.bytes)
          {
            DevBuffer bigger;
            if (bigger.reserve (std::max (need, 2 * fs.pcm.bytes))
                || (fs.kept_values && hipMemcpyAsync (bigger.ptr, fs.pcm.ptr, fs.kept_values * sizeof (float), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                || hipStreamSynchronize (ctx->stream) != hipSuccess)
              return false;
            fs.pcm.release();
            fs.pcm = bigger;
          }
      }
    if (raw)
      good = awm_pcm_encode_d (ctx, d_pcm, n_frames * C, fmt.bit_depth, encoding_id (fmt.encoding), fmt.endian == RawFormat::BIG, direct16, dev[b].ptr) == 0
          && (!fs.keep || awm_pcm_decode_d (ctx, dev[b].ptr, n_frames * C, fmt.bit_depth, encoding_id (fmt.encoding), fmt.endian == RawFormat::BIG,
                                            fs.pcm.as<float>() + fs.kept_values) == 0)
          && hipEventRecord (ev_encoded[b], ctx->stream) == hipSuccess
          && hipStreamWaitEvent (copy, ev_encoded[b], 0) == hipSuccess
          && hipMemcpyAsync (host[b].ptr, dev[b].ptr, n_frames * unit, hipMemcpyDeviceToHost, copy) == hipSuccess;
    else
      good = (!fs.keep || hipMemcpyAsync (fs.pcm.as<float>() + fs.kept_values, d_pcm, n_frames * C * sizeof (float), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess)
          && hipEventRecord (ev_encoded[b], ctx->stream) == hipSuccess
          && hipStreamWaitEvent (copy, ev_encoded[b], 0) == hipSuccess
          && hipMemcpyAsync (host[b].ptr, d_pcm, n_frames * unit, hipMemcpyDeviceToHost, copy) == hipSuccess;
    if (fs.keep && good)
      fs.kept_values += n_frames * C;
    good = good && hipEventRecord (ev_copied[b], copy) == hipSuccess
        // the producer may overwrite d_pcm / dev[b] only after the copy: later work on the compute stream waits for it
        && hipStreamWaitEvent (ctx->stream, ev_copied[b], 0) == hipSuccess;
    if (good)
      writer->submit (host[b].as<unsigned char>(), n_frames, ev_copied[b], b, raw);
    return good;
  }
  Error finish() { return writer->finish(); }
};

Error
store_device_to_stream (awm_ctx *ctx, AudioOutputStream *out_stream, const float *d_pcm, size_t n_values)
{
  if (!n_values)
    return Error::Code::NONE;
  const int C = out_stream->n_channels();
  const size_t n_frames = n_values / C;
  OutputStage stage (ctx, out_stream, STAGE_FRAMES, n_frames);
  if (!stage.ok)
    return Error ("out of memory for output staging");
  for (size_t pos = 0; pos < n_frames; pos += STAGE_FRAMES)
    if (!stage.put (d_pcm + pos * C, std::min (STAGE_FRAMES, n_frames - pos)))
      return Error (std::string ("GPU staging failed: ") + awm_last_error());
  return stage.finish();
}

} // namespace

extern "C" void awm_set_io_threads (int n) { g_io_threads.store (n < 0 ? 0 : n, std::memory_order_relaxed); }
extern "C" void awm_debug_set_io_flags (int flags) { g_io_flags.store (flags, std::memory_order_relaxed); }
extern "C" void awm_debug_file_timing (double ms_out[8]) { for (int i = 0; i < 8; i++) ms_out[i] = tl_file_ms[i]; }

namespace {

/* "Data Blocks" counter of WatermarkGen (reference wmadd.cc:311-313, 346-351): depends only on how many frames the
 * streaming loop of the reference pushes through the generator, i.e. on the latency of synth + limiter */
int
count_data_blocks (size_t n_frames, int sample_rate, bool limiter, size_t zero_frames)
{
  const size_t N = Params::frame_size, block = mark_block_frame_count();
  const size_t lim_block = size_t (sample_rate) * size_t (Params::limiter_block_size_ms) / 1000;
  size_t total_in = 0, total_out = 0, frame_number = 2 * block - Params::frames_pad_start, data_blocks = 0, lim_buffer = 0;
  size_t zero_in = zero_frames, zero_out = zero_frames;
  bool first_frame = true;
  auto limiter_out = [&] (size_t out) {                     // Limiter::process / Limiter::skip (limiter.cc:51-88): what comes out
    lim_buffer += out;
    const size_t buffered_blocks = lim_buffer / lim_block;
    out = buffered_blocks < 2 ? 0 : (buffered_blocks - 1) * lim_block;
    lim_buffer -= out;
    return out;
  };
  if (zero_in >= N)
    {
      // whole frames of zeros are skipped, not run (wmadd.cc:501-517): the frame counter moves, the block counter does not
      const size_t skip_frames = zero_in - zero_in % N;
      total_in += skip_frames;
      frame_number += skip_frames / N;
      size_t out = skip_frames - N;                        // WatermarkSynth::skip: the first frame's latency (wmadd.cc:253-263)
      first_frame = false;
      out = limiter_out (out);
      zero_out -= out;
      total_out += out;
      zero_in -= skip_frames;
    }
  while (true)
    {
      const size_t got = zero_in + std::min (N - zero_in, n_frames + zero_frames - total_in - zero_in);
      zero_in = 0;
      total_in += got;
      if (got < N && total_in == total_out)
        break;
      frame_number++;
      if (frame_number % block == 0)
        data_blocks++;
      size_t out = first_frame ? 0 : N;                    // WatermarkSynth emits nothing for the first frame
      first_frame = false;
      if (limiter)
        out = limiter_out (out);
      out = std::min (out, total_in - total_out);
      const size_t cut = std::min (out, zero_out);
      zero_out -= cut;
      total_out += out;
    }
  return std::max (int (data_blocks) - 1, 0);
}

} // namespace

namespace {

/* SNR report of `add --snr` (reference wmadd.cc:553-563, 591-592): the watermark is measured BEFORE the limiter, on the device,
 * by every mix of the context between begin and end (awm_ctx_snr_begin / awm_ctx_snr_end) */
struct SnrMeter
{
  awm_ctx *ctx;
  bool     on = false;
  explicit SnrMeter (awm_ctx *c) : ctx (c) {}
  ~SnrMeter() { if (on) (void) awm_ctx_snr_end (ctx, nullptr, nullptr); }
  bool begin() { on = awm_ctx_snr_begin (ctx) == 0; return on; }
  bool
  report()
  {
    double signal_power = 0, delta_power = 0;
    on = false;
    if (awm_ctx_snr_end (ctx, &signal_power, &delta_power))
      return false;
    info ("SNR:          %f dB\n", 10 * log10 (signal_power / delta_power));
    return true;
  }
};

/* `add` at the watermark rate as a tile loop (awm_add_stream, include/awm_hip.h): bounded memory on BOTH sides -- two
 * staging chunks of input and three of output on the host, three input and three mix tiles in HBM, whatever the length of
 * the stream (also for pipes of unknown length).  Per tile: file read -> H2D (copy stream) -> sample decode -> fused STFT /
 * band edit / inverse / overlap-add / mix of the PREVIOUS tile -> limiter + sample encode of the tile before that -> D2H
 * (copy stream) -> file write (writer thread); the stages of neighbouring tiles overlap. */
int
add_tiles (awm_ctx *ctx, const Key& key, AudioInputStream *in_stream, AudioOutputStream *out_stream, const std::string& payload_hex,
           size_t zero_frames, size_t& n_frames)
{
  constexpr size_t TILE_FRAMES1024 = 4096;                 // 95 s of audio, 32 MiB of float32 stereo
  const size_t tile = TILE_FRAMES1024 * Params::frame_size;
  const int C = in_stream->n_channels();
  n_frames = 0;
  for (double& v : tl_file_ms)
    v = 0;
  Lap lap;
  struct TeardownLap { Lap& l; ~TeardownLap() { l.to (6); } };
  awm_add_stream *add = nullptr;
  if (awm_add_stream_create_at (ctx, key.aes_key(), payload_hex.c_str(), C, TILE_FRAMES1024, zero_frames, &add))
    {
      error ("audiowmark: GPU watermarking failed: %s\n", awm_last_error());
      return fail (AWM_ERR_HIP);
    }
  TeardownLap teardown_lap { lap };                       // (declared before the guard and the rings: runs after their destructors)
  struct Guard { awm_add_stream *s; ~Guard() { awm_add_stream_destroy (s); } } guard { add };
  RawFormat fmt;
  const bool raw = in_stream->raw_access (fmt) && device_codec_supported (fmt);
  const size_t unit = raw ? size_t (C) * (fmt.bit_depth / 8) : size_t (C) * sizeof (float);
  TileReader rd (ctx, in_stream, raw, unit, tile, raw);
  // (the output is as long as the input: known for files and announced lengths, unknown for pipes -- those are written in order)
  OutputStage stage (ctx, out_stream, tile, rd.ok ? rd.announced_frames() : AudioInputStream::N_FRAMES_UNKNOWN);
  if (!rd.ok || !stage.ok)
    {
      error ("audiowmark: out of memory for the staging buffers\n");
      return fail (AWM_ERR_HIP);
    }
  SnrMeter snr (ctx);
  if (params().snr && !snr.begin())
    {
      error ("audiowmark: GPU watermarking failed: %s\n", awm_last_error());
      return fail (AWM_ERR_HIP);
    }
  bool eof = false;
  lap.to (0);
  for (size_t k = 0; !eof; k++)
    {
      int b = 0;
      size_t got = 0;
      Error err = rd.next (b, got);                        // (read ahead by the I/O workers / the reader thread)
      lap.to (1);
      if (err)
        {
          error ("audiowmark: input stream read failed: %s\n", err.message());
          return fail (AWM_ERR_IO);
        }
      eof = got < tile;
      float *slot = awm_add_stream_input (add);
      bool ok = true;
      if (got && raw)
        ok = (k < size_t (IN_RING) || hipStreamWaitEvent (rd.copy, rd.ev_used (b), 0) == hipSuccess)
          && hipMemcpyAsync (rd.dev (b), rd.host (b), got * unit, hipMemcpyHostToDevice, rd.copy) == hipSuccess
          && hipEventRecord (rd.ev_copied (b), rd.copy) == hipSuccess
          && hipStreamWaitEvent (ctx->stream, rd.ev_copied (b), 0) == hipSuccess
          && awm_pcm_decode_d (ctx, rd.dev (b), got * C, fmt.bit_depth, encoding_id (fmt.encoding), fmt.endian == RawFormat::BIG, slot) == 0
          && hipEventRecord (rd.ev_used (b), ctx->stream) == hipSuccess;
      else if (got)
        // the slot was last read by kernels queued on the compute stream: the copy must not overtake them
        ok = hipEventRecord (rd.ev_used (b), ctx->stream) == hipSuccess
          && hipStreamWaitEvent (rd.copy, rd.ev_used (b), 0) == hipSuccess
          && hipMemcpyAsync (slot, rd.host (b), got * unit, hipMemcpyHostToDevice, rd.copy) == hipSuccess
          && hipEventRecord (rd.ev_copied (b), rd.copy) == hipSuccess
          && hipStreamWaitEvent (ctx->stream, rd.ev_copied (b), 0) == hipSuccess;
      if (ok && got)
        rd.recycle (b, rd.ev_copied (b));
      const float *done[3];
      size_t done_frames[3];
      int n_done = ok ? awm_add_stream_push (add, got, eof, done, done_frames) : -1;
      if (n_done < 0)
        {
          error ("audiowmark: GPU watermarking failed: %s\n", awm_last_error());
          return fail (AWM_ERR_HIP);
        }
      lap.to (3);
      for (int i = 0; i < n_done; i++)
        {
          // (with zero_frames the last tile carries what hung over the one before it: up to 1023 frames more than a ring slot takes)
          for (size_t at = 0; at < done_frames[i]; at += tile)
            if (!stage.put (done[i] + at * C, std::min (tile, done_frames[i] - at)))
              {
                error ("audiowmark: GPU staging failed: %s\n", awm_last_error());
                return fail (AWM_ERR_HIP);
              }
        }
      lap.to (7);                                          // (output tiles handed on: includes [2], the wait for a free slot)
      n_frames += got;
    }
  if (hipStreamSynchronize (ctx->stream) != hipSuccess)
    {
      error ("audiowmark: GPU watermarking failed\n");
      return fail (AWM_ERR_HIP);
    }
  lap.to (4);
  Error err = stage.finish();
  lap.to (5);
  if (err)
    {
      error ("audiowmark output write failed: %s\n", err.message());
      return fail (AWM_ERR_IO);
    }
  if (params().snr && n_frames && !snr.report())
    {
      error ("audiowmark: GPU watermarking failed: %s\n", awm_last_error());
      return fail (AWM_ERR_HIP);
    }
  return 0;
}

/* `add` at another sample rate: the WatermarkResampler path works on the whole stream in HBM (awm_add_watermark_d); the
 * host side is still bounded (chunked staging both ways) */
int
add_whole (awm_ctx *ctx, const Key& key, AudioInputStream *in_stream, AudioOutputStream *out_stream, const std::string& payload_hex,
           size_t zero_frames, size_t& n_frames)
{
  const int C = in_stream->n_channels();
  DevBuffer d_in, d_out;
  struct Guard { DevBuffer& a; DevBuffer& b; ~Guard() { a.release(); b.release(); } } guard { d_in, d_out };
  size_t n_values = 0;
  Error err = load_stream_to_device (ctx, in_stream, d_in, n_values);
  if (err)
    {
      error ("audiowmark: input stream read failed: %s\n", err.message());
      return fail (AWM_ERR_IO);
    }
  n_frames = n_values / C;
  if (!n_values)
    return 0;
  if (zero_frames)
    {
      // the resamplers' skip (resample.cc:150-168) leaves them in the state zeros would have: the stream behind zero_frames zeros
      const size_t zero_values = zero_frames * C;
      if (zero_frames > (DevBuffer::MAX_BYTES / sizeof (float) - n_values) / C)
        {
          error ("audiowmark: zero_frames is too large for device memory at this sample rate\n");
          return fail (AWM_ERR_ARG);
        }
      DevBuffer shifted;
      if (shifted.reserve ((zero_values + n_values) * sizeof (float))
          || hipMemsetAsync (shifted.ptr, 0, zero_values * sizeof (float), ctx->stream) != hipSuccess
          || hipMemcpyAsync (shifted.as<float>() + zero_values, d_in.ptr, n_values * sizeof (float), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess
          || hipStreamSynchronize (ctx->stream) != hipSuccess)
        {
          shifted.release();
          error ("audiowmark: GPU watermarking failed: %s\n", awm_last_error());
          return fail (AWM_ERR_HIP);
        }
      d_in.release();
      d_in = shifted;
    }
  const size_t in_values = n_values;
  n_values += zero_frames * C;
  SnrMeter snr (ctx);
  if (params().snr && !snr.begin())
    {
      error ("audiowmark: GPU watermarking failed: %s\n", awm_last_error());
      return fail (AWM_ERR_HIP);
    }
  if (d_out.reserve (n_values * sizeof (float))
      || awm_add_watermark_d (ctx, key.aes_key(), payload_hex.c_str(), d_in.as<float>(), d_out.as<float>(), n_values / C, C, in_stream->sample_rate()) != 0)
    {
      error ("audiowmark: GPU watermarking failed: %s\n", awm_last_error());
      return fail (AWM_ERR_HIP);
    }
  if (params().snr && !snr.report())
    {
      error ("audiowmark: GPU watermarking failed: %s\n", awm_last_error());
      return fail (AWM_ERR_HIP);
    }
  err = store_device_to_stream (ctx, out_stream, d_out.as<float>() + zero_frames * C, in_values);
  if (err)
    {
      error ("audiowmark output write failed: %s\n", err.message());
      return fail (AWM_ERR_IO);
    }
  return 0;
}

} // namespace

int
add_stream_watermark (awm_ctx *ctx, const Key& key, AudioInputStream *in_stream, AudioOutputStream *out_stream,
                      const std::string& bits, size_t zero_frames)
{
  auto bitvec = parse_payload (bits);
  if (bitvec.empty())
    return fail (AWM_ERR_ARG);
  if (in_stream->sample_rate() != out_stream->sample_rate())
    {
      error ("audiowmark: input sample rate (%d) and output sample rate (%d) don't match\n", in_stream->sample_rate(), out_stream->sample_rate());
      return fail (AWM_ERR_ARG);
    }
  if (in_stream->n_channels() != out_stream->n_channels())
    {
      error ("audiowmark: input channels (%d) and output channels (%d) don't match\n", in_stream->n_channels(), out_stream->n_channels());
      return fail (AWM_ERR_ARG);
    }
  if (in_stream->sample_rate() != Params::mark_sample_rate
      && (!awm_resample_frames (ctx, 1024, in_stream->sample_rate(), Params::mark_sample_rate)
          || !awm_resample_frames (ctx, 1024, Params::mark_sample_rate, in_stream->sample_rate())))
    {
      // the reference falls back to zita's VResampler for such ratios (resample.cc:233-270)
      error ("audiowmark: resampling from old_rate=%d to new_rate=%d not implemented\n", in_stream->sample_rate(), Params::mark_sample_rate);
      return fail (AWM_ERR_ARG);
    }
  info ("Message:      %s\n", bit_vec_to_str (bitvec).c_str());
  info ("Strength:     %.6g\n\n", params().water_delta * 1000);
  if (in_stream->n_frames() == AudioInputStream::N_FRAMES_UNKNOWN)
    info ("Time:         unknown\n");
  else
    {
      const size_t orig_seconds = in_stream->n_frames() / in_stream->sample_rate();
      info ("Time:         %zd:%02zd\n", orig_seconds / 60, orig_seconds % 60);
    }
  info ("Sample Rate:  %d\n", in_stream->sample_rate());
  info ("Channels:     %d\n", in_stream->n_channels());

  const int C = in_stream->n_channels();
  size_t n_frames = 0;
  int rc = in_stream->sample_rate() == Params::mark_sample_rate
         ? add_tiles (ctx, key, in_stream, out_stream, bit_vec_to_str (bitvec), zero_frames, n_frames)
         : add_whole (ctx, key, in_stream, out_stream, bit_vec_to_str (bitvec), zero_frames, n_frames);
  if (rc)
    return rc;
  (void) C;
  info ("Data Blocks:  %d\n", count_data_blocks (n_frames, in_stream->sample_rate(), !params().test_no_limiter, zero_frames));
  if (in_stream->n_frames() != AudioInputStream::N_FRAMES_UNKNOWN && n_frames != in_stream->n_frames())
    {
      auto msg = string_printf ("unexpected EOF; input frames (%zd) != output frames (%zd)", in_stream->n_frames() + zero_frames, n_frames + zero_frames);
      if (params().strict)
        {
          error ("audiowmark: error: %s\n", msg.c_str());
          return fail (AWM_ERR_IO);
        }
      warning ("audiowmark: warning: %s\n", msg.c_str());
    }
  Error err = out_stream->close();
  if (err)
    {
      error ("audiowmark: closing output stream failed: %s\n", err.message());
      return fail (AWM_ERR_IO);
    }
  return 0;
}

static void
info_format (const std::string& label, const RawFormat& format)
{
  const char *e = format.encoding == Encoding::SIGNED ? "signed" : format.encoding == Encoding::UNSIGNED ? "unsigned" : "float";
  info ("%-13s %d Hz, %d Channels, %d Bit (%s %s-endian)\n", (label + ":").c_str(), format.sample_rate, format.n_channels,
        format.bit_depth, e, format.endian == RawFormat::LITTLE ? "little" : "big");
}

int
add_watermark (awm_ctx *ctx, const Key& key, const std::string& infile, const std::string& outfile, const std::string& bits)
{
  return add_watermark_at (ctx, key, infile, outfile, bits, 0);
}

/* add_watermark for a file that is the continuation of a stream `zero_frames` samples in (what hls.cc:279 does with a segment) */
int
add_watermark_at (awm_ctx *ctx, const Key& key, const std::string& infile, const std::string& outfile, const std::string& bits,
                  size_t zero_frames)
{
  Error err;
  auto in_stream = AudioInputStream::create (infile, err);
  if (err)
    {
      error ("audiowmark: error opening %s: %s\n", infile.c_str(), err.message());
      return fail (AWM_ERR_IO);
    }
  int out_bit_depth = in_stream->bit_depth();
  Encoding out_encoding = in_stream->encoding();
  if (in_stream->bit_depth() < 16)
    {
      out_bit_depth = 16;
      out_encoding = Encoding::SIGNED;
    }
  auto out_stream = AudioOutputStream::create (outfile, in_stream->n_channels(), in_stream->sample_rate(), out_bit_depth, out_encoding,
                                               in_stream->n_frames(), err);
  if (err)
    {
      error ("audiowmark: error writing to %s: %s\n", outfile.c_str(), err.message());
      return fail (AWM_ERR_IO);
    }
  info ("Input:        %s\n", infile.c_str());
  if (params().input_format == Format::RAW)
    info_format ("Raw Input", StreamParams::raw_input_format);
  info ("Output:       %s\n", outfile.c_str());
  if (params().output_format == Format::RAW)
    info_format ("Raw Output", StreamParams::raw_output_format);
  return add_stream_watermark (ctx, key, in_stream.get(), out_stream.get(), bits, zero_frames);
}

/* add_watermark (key, infile, outfile, bits) followed by get_watermark (key, outfile) -- "watermark, then verify that the payload decodes"
 * -- with the input read once and the output never read back: the output stage decodes what it has just encoded for the file (the
 * samples as the file holds them) into the context's stream buffer, and `get` runs on that (reference wmadd.cc:620-657, wmget.cc:971-1013). */
int
add_get_watermark (awm_ctx *ctx, const Key& key, const std::string& infile, const std::string& outfile, const std::string& bits,
                   ResultSet& result_set)
{
  FileStaging& fs = ctx->file_staging;
  fs.keep = true;
  fs.kept_values = 0;
  fs.kept_channels = fs.kept_rate = 0;
  struct Off { FileStaging& f; ~Off() { f.keep = false; } } off { fs };
  if (int rc = add_watermark_at (ctx, key, infile, outfile, bits, 0))
    return rc;
  fs.keep = false;
  if (!fs.kept_channels || !fs.kept_rate)
    {
      result_set.sort ({ key });
      return 0;
    }
  size_t n_values = 0;
  return get_watermark_loaded (ctx, { key }, fs.kept_values, fs.kept_channels, fs.kept_rate, false, result_set, n_values);
}

std::string& last_shard_debug_sync();      // wmshard.cc

/* A long stream over the context and its helpers (other GPUs, awm_ctx_set_helpers): equal frame spans, the helpers' spans copied
 * device to device, awm_multi_get_d.  `spread` = false: not applicable (no helpers, several keys, speed detection, or too short to
 * be worth it) -- the caller decodes on the context alone. */
static int
get_watermark_multi (awm_ctx *ctx, const std::vector<Key>& key_list, const DeviceWav& wav, ResultSet& result_set, bool& spread)
{
  spread = false;
  const size_t n_ctx = ctx->helpers.size() + 1;
  const size_t min_span = 4 * mark_block_frame_count() * Params::frame_size;          // at least four blocks per GPU
  if (n_ctx < 2 || key_list.size() != 1 || params().test_no_sync || wav.n_frames < n_ctx * min_span)
    return 0;
  const int C = wav.n_channels;
  std::vector<awm_ctx *> ctxs { ctx };
  ctxs.insert (ctxs.end(), ctx->helpers.begin(), ctx->helpers.end());
  std::vector<uint64_t> span (n_ctx);
  std::vector<const float *> ptr (n_ctx);
  std::vector<DevBuffer> bufs (n_ctx);
  const size_t per = wav.n_frames / n_ctx / Params::frame_size * Params::frame_size;
  size_t pos = 0;
  int rc = 0;
  AWM_HIP_CHECK (hipStreamSynchronize (ctx->stream));
  for (size_t i = 0; i < n_ctx && !rc; i++)
    {
      span[i] = i + 1 < n_ctx ? per : wav.n_frames - pos;
      if (i == 0)
        ptr[i] = wav.data;
      else
        {
          const size_t bytes = size_t (span[i]) * C * sizeof (float);
          if (hipSetDevice (ctxs[i]->device) != hipSuccess || bufs[i].reserve (std::max<size_t> (1, bytes))
              || hipMemcpyPeer (bufs[i].ptr, ctxs[i]->device, wav.data + pos * C, ctx->device, bytes) != hipSuccess)
            {
              set_error ("cannot place a span of the stream on device " + std::to_string (ctxs[i]->device));
              rc = AWM_ERR_HIP;
            }
          ptr[i] = bufs[i].as<float>();
        }
      pos += span[i];
    }
  for (size_t i = 1; i < n_ctx && !rc; i++)            // (peer copies may return before they are done)
    if (hipSetDevice (ctxs[i]->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
      rc = AWM_ERR_HIP;
  (void) hipSetDevice (ctx->device);
  std::vector<awm_pattern> pats (rc ? 0 : 4096);
  int n = 0;
  if (!rc)
    {
      n = awm_multi_get_d (ctxs.data(), int (n_ctx), key_list[0].aes_key(), ptr.data(), C, span.data(), pats.size(), pats.data());
      if (n > int (pats.size()))
        {
          pats.resize (n);
          n = awm_multi_get_d (ctxs.data(), int (n_ctx), key_list[0].aes_key(), ptr.data(), C, span.data(), pats.size(), pats.data());
        }
      if (n < 0)
        rc = n;
    }
  for (size_t i = 1; i < n_ctx; i++)
    {
      (void) hipSetDevice (ctxs[i]->device);
      bufs[i].release();
    }
  (void) hipSetDevice (ctx->device);
  if (rc)
    return rc;
  for (int i = 0; i < n; i++)
    {
      const awm_pattern& p = pats[i];
      SyncFinder::Score score { size_t (p.sync_index), p.sync_quality, ConvBlockType (p.block_type) };
      result_set.add_pattern (key_list[0], p.time, score, std::vector<int> (p.bits, p.bits + p.n_bits), p.decode_error,
                              ResultSet::Type (p.type), p.speed);
    }
  result_set.sort (key_list);
  result_set.set_debug_sync (last_shard_debug_sync());
  spread = true;
  return 0;
}

/* 1 (default): the chunks of a file level `get` start while the rest of the stream is still crossing PCIe (below) | 0: the whole stream
 * first, as in rounds 1 - 5.  Measured alternating on one file in the page cache (tools/gpu_get_overlap_ab.py, profiles/r06/get_overlap_ab.txt):
 * 60 min of s16 stereo 15.8 - 17.5 against 17.1 - 17.7 ms, 8 h 111 - 113 against 134 - 135 ms (two boxes); pattern lists identical.  (Inside bench.py's add -> get loop the `get`
 * that follows a fresh `add` is 8 - 10 ms slower either way: the output file's pages were created a moment ago.) */
static int g_get_overlap = 1;
extern "C" void awm_debug_set_get_overlap (int on) { g_get_overlap = on; }

/* body of get_watermark (reference wmget.cc:971-1013): stream -> HBM (bounded host memory), loader resampling, chunk loop */
int
get_watermark_stream (awm_ctx *ctx, const std::vector<Key>& key_list, AudioInputStream *in_stream, bool print_speed, ResultSet& result_set,
                      size_t& n_values_out, const std::string& what)
{
  const int C = in_stream->n_channels();
  // the stream's float32 PCM lives in a buffer the context keeps between calls (grow-only, like the workspaces: a hipMalloc of
  // 1.3 GB per hour of audio costs 1 - 70 ms, its hipFree as much again)
  DevBuffer& d_in = ctx->file_staging.pcm;
  size_t n_values = 0;
  /* A stream of announced length at the watermark rate with two chunks or more: the chunks start WHILE the stream is still crossing
   * PCIe.  The chunk plan follows from the length (wavchunkloader.cc:75-84); a loader thread brings the tiles in and leaves a mark
   * behind each; chunk k is queued on its lane the moment the mark that covers its last sample is there (block_decoder_run, live
   * marks).  If the stream turns out shorter or longer than announced, the result is thrown away and the plain order takes over. */
  size_t announced = in_stream->n_frames();
  {
    // (headerless PCM does not announce a length -- audiostream.hh, like the reference's RawInputStream -- but a regular file has one)
    int fd = -1;
    uint64_t offset = 0;
    size_t frames = 0;
    if (announced == AudioInputStream::N_FRAMES_UNKNOWN && in_stream->raw_region (fd, offset, frames))
      announced = frames;
  }
  const bool speed = params().detect_speed || params().detect_speed_patient || params().try_speed > 0;
  const size_t max_frames = DevBuffer::MAX_BYTES / (size_t (C) * sizeof (float));
  if (g_get_overlap && announced != AudioInputStream::N_FRAMES_UNKNOWN && announced + 1 < max_frames && in_stream->sample_rate() == Params::mark_sample_rate
      && !params().test_truncate && !speed && ctx->helpers.empty() && !key_list.empty() && ctx->chunk_lanes > 1
      && plan_chunks (announced, C).size() >= 2 && d_in.reserve ((announced + 1) * C * sizeof (float)) == 0)
    {
      ReadyMarks& rm = ctx->ready;
      rm.disarm();
      rm.base = d_in.as<float>();
      rm.n_frames = announced;
      rm.live = true;
      rm.armed = true;
      Error load_err;
      size_t loaded_values = 0;
      ParamValues *const pv = &params();
      std::thread loader ([&, pv] {
        ParamsBind bind (pv);
        if (hipSetDevice (ctx->device) != hipSuccess)
          {
            load_err = Error ("cannot select the device");
            std::lock_guard<std::mutex> lock (rm.mu);
            rm.live_done = true;
            rm.cv.notify_all();
            return;
          }
        load_err = load_stream_to_device (ctx, in_stream, d_in, loaded_values, &rm);
      });
      DeviceWav wav;
      wav.data = d_in.as<float>();
      wav.n_frames = announced;
      wav.n_channels = C;
      wav.sample_rate = Params::mark_sample_rate;
      speed_print_results = print_speed;
      ResultSet early;
      const int rc = get_watermark_device (ctx, key_list, wav, early);
      loader.join();
      rm.disarm();
      if (load_err)
        {
          error ("audiowmark: error loading %s: %s\n", what.c_str(), load_err.message());
          return fail (AWM_ERR_IO);
        }
      if (loaded_values == announced * C)
        {
          if (rc)
            {
              error ("audiowmark: GPU detection failed: %s\n", awm_last_error());
              return fail (AWM_ERR_HIP);
            }
          result_set = early;
          n_values_out = loaded_values;
          return 0;
        }
      // (the file ends before the length its header announces: what the chunks saw beyond the end was not the stream)
      if (hipStreamSynchronize (ctx->stream) != hipSuccess)
        return fail (AWM_ERR_HIP);
      return get_watermark_loaded (ctx, key_list, loaded_values, C, in_stream->sample_rate(), print_speed, result_set, n_values_out);
    }
  Error err = load_stream_to_device (ctx, in_stream, d_in, n_values);
  if (err)
    {
      error ("audiowmark: error loading %s: %s\n", what.c_str(), err.message());
      return fail (AWM_ERR_IO);
    }
  return get_watermark_loaded (ctx, key_list, n_values, C, in_stream->sample_rate(), print_speed, result_set, n_values_out);
}

/* ... from the stream's float32 PCM in the context's buffer (file_staging.pcm: n_values samples at `rate`) on */
int
get_watermark_loaded (awm_ctx *ctx, const std::vector<Key>& key_list, size_t n_values, int C, int rate, bool print_speed, ResultSet& result_set,
                      size_t& n_values_out)
{
  DevBuffer& d_in = ctx->file_staging.pcm;
  if (rate != Params::mark_sample_rate)
    {
      // WavChunkLoader resamples the whole stream to the watermark rate before anything else (wavchunkloader.cc:70-71, 200-216)
      const size_t in_frames = n_values / C;
      const size_t out_frames = awm_resample_frames (ctx, in_frames, rate, Params::mark_sample_rate);
      DevBuffer d_res;
      if ((in_frames && !out_frames) || d_res.reserve (std::max<size_t> (1, out_frames * C * sizeof (float)))
          || awm_resample_d (ctx, d_in.as<float>(), in_frames, C, rate, Params::mark_sample_rate, d_res.as<float>(), out_frames))
        {
          error ("audiowmark: resampling from old_rate=%d to new_rate=%d not implemented\n", rate, Params::mark_sample_rate);
          d_res.release();
          return fail (AWM_ERR_ARG);
        }
      (void) hipStreamSynchronize (ctx->stream);
      d_in.release();
      d_in = d_res;
      n_values = out_frames * C;
    }
  if (params().test_truncate)
    n_values = std::min (n_values, size_t (Params::mark_sample_rate) * C * params().test_truncate);
  const size_t n_frames = n_values / C;
  n_values_out = n_values;
  if (n_frames)
    {
      DeviceWav wav;
      wav.data = d_in.as<float>();
      wav.n_frames = n_frames;
      wav.n_channels = C;
      wav.sample_rate = Params::mark_sample_rate;
      speed_print_results = print_speed;                // decode (..., orig_bits, ...): the detect_speed report line of `cmp`
      int rc = AWM_ERR_GENERIC;
      bool spread = false;
      if (int r = get_watermark_multi (ctx, key_list, wav, result_set, spread))
        rc = r;
      else if (spread)
        rc = 0;
      else
        rc = get_watermark_device (ctx, key_list, wav, result_set);
      if (rc)
        {
          error ("audiowmark: GPU detection failed: %s\n", awm_last_error());
          return fail (AWM_ERR_HIP);
        }
    }
  else
    result_set.sort (key_list);
  return 0;
}

int
get_watermark (awm_ctx *ctx, const std::vector<Key>& key_list, const std::string& infile, const std::string& orig_pattern)
{
  std::vector<int> orig_bitvec;
  if (!orig_pattern.empty())
    {
      orig_bitvec = parse_payload (orig_pattern);
      if (orig_bitvec.empty())
        return fail (AWM_ERR_ARG);
    }
  Error err;
  auto in_stream = AudioInputStream::create (infile, err);
  if (err)
    {
      error ("audiowmark: error loading %s: %s\n", infile.c_str(), err.message());
      return fail (AWM_ERR_IO);
    }
  const int C = in_stream->n_channels();
  ResultSet result_set;
  size_t n_values = 0;
  if (int rc = get_watermark_stream (ctx, key_list, in_stream.get(), !orig_bitvec.empty(), result_set, n_values, infile))
    return rc;
  const size_t time_length = lrint (double (n_values) / (double (Params::mark_sample_rate) * C));

  /* report (reference wmget.cc:941-969) */
  if (!params().json_output.empty())
    result_set.print_json (time_length, params().json_output);
  if (params().json_output != "-")
    result_set.print();
  if (!orig_bitvec.empty())
    {
      const int match_count = result_set.print_match_count (orig_bitvec);
      result_set.print_debug_sync();
      if (params().expect_matches >= 0)
        {
          printf ("expect_matches %d\n", params().expect_matches);
          if (match_count != params().expect_matches)
            return 1;
        }
      else if (!match_count)
        return 1;
    }
  return 0;
}

/* `audiowmark test-change-speed in out speed` (reference audiowmark.cc:419-437): resample_ratio (in, 1 / speed, same rate) */
int
test_change_speed (awm_ctx *ctx, const std::string& infile, const std::string& outfile, double speed)
{
  Error err;
  auto in_stream = AudioInputStream::create (infile, err);
  if (err)
    {
      error ("audiowmark: error loading %s: %s\n", infile.c_str(), err.message());
      return fail (AWM_ERR_IO);
    }
  const int C = in_stream->n_channels();
  DevBuffer d_in, d_out;
  size_t n_values = 0;
  err = load_stream_to_device (ctx, in_stream.get(), d_in, n_values);
  if (err)
    {
      error ("audiowmark: error loading %s: %s\n", infile.c_str(), err.message());
      d_in.release();
      return fail (AWM_ERR_IO);
    }
  const size_t in_frames = n_values / C;
  const size_t out_frames = awm_resample_ratio_frames (in_frames, C, in_stream->sample_rate(), 1 / speed, -1);
  if (d_out.reserve (std::max<size_t> (16, out_frames * C * sizeof (float)))
      || awm_resample_ratio_d (ctx, d_in.as<float>(), in_frames, C, in_stream->sample_rate(), 1 / speed, -1, d_out.as<float>(), out_frames))
    {
      error ("audiowmark: failed to setup vresampler with ratio=%f\n", 1 / speed);
      d_in.release();
      d_out.release();
      return fail (AWM_ERR_ARG);
    }
  auto out_stream = AudioOutputStream::create (outfile, C, in_stream->sample_rate(), in_stream->bit_depth() < 16 ? 16 : in_stream->bit_depth(),
                                               in_stream->bit_depth() < 16 ? Encoding::SIGNED : in_stream->encoding(), out_frames, err);
  if (!err)
    err = store_device_to_stream (ctx, out_stream.get(), d_out.as<float>(), out_frames * C);
  if (!err)
    err = out_stream->close();
  d_in.release();
  d_out.release();
  if (err)
    {
      error ("audiowmark: error saving %s: %s\n", outfile.c_str(), err.message());
      return fail (AWM_ERR_ARG);
    }
  return 0;
}

} // namespace awm
