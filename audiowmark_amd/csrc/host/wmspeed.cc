// wmspeed.cc -- speed detection (reference src/wmspeed.cc) driven from the host, computed on the GPU (hip/speed.hip).
#include "wmspeed.hh"
#include "wmcommon.hh"
#include "utils.hh"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>

namespace awm {

void
SpeedWorkspace::release()
{
  for (auto& sl : table_slabs)
    sl.buf.release();
  table_slabs.clear();
  var_tables.clear();
  for (auto& t : key_tables)
    {
      t->cols.release();
      t->col_frame.release();
      t->col_first.release();
    }
  window512.release();
}

void
SpeedScratch::release()
{
  for (DevBuffer *b : { &sub, &mags, &centers, &items, &best, &gather_pos, &gather_out, &ranges, &energy, &stretched })
    b->release();
  pin.release();
}

SpeedScratch *
speed_scratch (WorkLane *lane)
{
  if (!lane->speed_scratch)
    lane->speed_scratch = new SpeedScratch();
  return lane->speed_scratch;
}

void
speed_scratch_free (WorkLane *lane)
{
  if (lane->speed_scratch)
    {
      lane->speed_scratch->release();
      delete lane->speed_scratch;
      lane->speed_scratch = nullptr;
    }
}

void
speed_workspace_free (awm_ctx *ctx)
{
  if (ctx->speed)
    {
      ctx->speed->release();
      delete ctx->speed;
      ctx->speed = nullptr;
    }
}

namespace {

SpeedWorkspace *
workspace (awm_ctx *ctx)
{
  if (!ctx->speed)
    ctx->speed = new SpeedWorkspace();
  return ctx->speed;
}

/* zita VResampler::setup (ratio, nchan, hlen = 16) geometry: taps per side, relative cutoff, phase step 256 / ratio */
struct VarGeometry
{
  bool     ok = false;
  unsigned hl = 0;
  double   frel = 0, step = 0;
};
VarGeometry
var_geometry (double ratio)
{
  VarGeometry g;
  const unsigned hlen = 16;
  if (!(16 * ratio >= 1) || ratio > 256)
    return g;
  g.frel = 1.0 - 2.6 / hlen;
  g.step = 256 / ratio;
  g.hl = hlen;
  if (ratio < 1)
    {
      g.frel *= ratio;
      g.hl = unsigned (std::ceil (hlen / ratio));
    }
  g.ok = true;
  return g;
}

/* tables for all ratios of a pass (the missing ones are computed on a few host threads: 257 x hl sinc / window values each) */
int
get_var_tables (awm_ctx *ctx, const std::vector<double>& ratios, std::vector<VarResampleTable *>& out)
{
  std::lock_guard<std::mutex> lock (ctx->speed_mutex);        // the table caches are shared by the lanes
  SpeedWorkspace *ws = workspace (ctx);
  out.assign (ratios.size(), nullptr);
  std::vector<size_t> missing;
  for (size_t i = 0; i < ratios.size(); i++)
    {
      for (auto& t : ws->var_tables)
        if (t->ratio == ratios[i])
          out[i] = t.get();
      for (size_t j = 0; j < i && !out[i]; j++)
        if (ratios[j] == ratios[i])
          out[i] = out[j];                               // filled in below through the first occurrence
      if (!out[i] && std::find_if (missing.begin(), missing.end(), [&] (size_t m) { return ratios[m] == ratios[i]; }) == missing.end())
        missing.push_back (i);
    }
  std::vector<std::vector<float>> tabs (missing.size());
  std::vector<VarGeometry> geo (missing.size());
  for (size_t k = 0; k < missing.size(); k++)
    {
      geo[k] = var_geometry (ratios[missing[k]]);
      if (!geo[k].ok)
        {
          set_error (string_printf ("failed to setup vresampler with ratio=%f", ratios[missing[k]]));
          return AWM_ERR_ARG;
        }
    }
  const size_t n_threads = std::min<size_t> (missing.size(), std::max (1u, std::min (16u, std::thread::hardware_concurrency())));
  std::vector<std::thread> threads;
  for (size_t t = 0; t < n_threads; t++)
    threads.emplace_back ([&, t] {
      for (size_t k = t; k < missing.size(); k += n_threads)
        {
          // rows padded to an odd stride (the kernel keeps the table in LDS: equal columns of different rows -> different banks)
          const std::vector<float> dense = zita_table (geo[k].frel, geo[k].hl, 256);
          const size_t hl = geo[k].hl, stride = hl | 1;
          tabs[k].assign (257 * stride, 0.f);
          for (size_t r = 0; r < 257; r++)
            std::copy (dense.begin() + r * hl, dense.begin() + (r + 1) * hl, tabs[k].begin() + r * stride);
        }
    });
  for (auto& th : threads)
    th.join();
  if (!missing.empty())
    {
      // one device allocation and one copy for all tables this call adds (every table 16 byte aligned inside the slab)
      std::vector<size_t> at (missing.size());
      size_t total = 0;
      for (size_t k = 0; k < missing.size(); k++)
        {
          at[k] = total;
          total += (tabs[k].size() + 3) & ~size_t (3);
        }
      std::vector<float> blob (total, 0.f);
      for (size_t k = 0; k < missing.size(); k++)
        std::copy (tabs[k].begin(), tabs[k].end(), blob.begin() + at[k]);
      VarTableSlab slab;
      slab.serial = ws->next_slab++;
      if (int rc = upload_sync (slab.buf, blob.data(), blob.size() * sizeof (float), ctx->stream))
        return rc;
      for (size_t k = 0; k < missing.size(); k++)
        {
          auto vt = std::make_unique<VarResampleTable>();
          vt->ratio = ratios[missing[k]];
          vt->hl = int (geo[k].hl);
          vt->ctab = slab.buf.as<float>() + at[k];
          vt->slab = slab.serial;
          ws->var_tables.push_back (std::move (vt));
        }
      ws->table_slabs.push_back (slab);
      // Bounded cache.  The grids of the first pass (57 centres each for the normal and the patient search) never change and are in
      // the first slabs ever made: those stay; beyond max_tables the OLDEST of the data dependent refinement slabs goes, with all
      // its tables.  (Nothing handed out by this call is dropped: a call either finds all its ratios, or adds the missing ones
      // last.  Up to CHUNK_LANES searches run side by side, each with ~100 ratios of its own in use: the cache is an order of
      // magnitude larger than that, so the oldest slab is never one a running search holds.  A table is ~20 KB.)
      constexpr size_t keep_slabs = 2, max_tables = 4096;
      while (ws->var_tables.size() > max_tables && ws->table_slabs.size() > keep_slabs + 1)
        {
          const size_t gone = ws->table_slabs[keep_slabs].serial;
          ws->table_slabs[keep_slabs].buf.release();
          ws->table_slabs.erase (ws->table_slabs.begin() + keep_slabs);
          ws->var_tables.erase (std::remove_if (ws->var_tables.begin(), ws->var_tables.end(),
                                                [&] (const std::unique_ptr<VarResampleTable>& t) { return t->slab == gone; }),
                                ws->var_tables.end());
        }
    }
  for (size_t i = 0; i < ratios.size(); i++)
    if (!out[i])
      for (auto& t : ws->var_tables)
        if (t->ratio == ratios[i])
          out[i] = t.get();
  return 0;
}

/* device description of one resampler run */
awmk::SpeedCenterDev
center_dev (const VarResampleTable *vt, double ratio, long long n_in, long long n_out)
{
  awmk::SpeedCenterDev cd {};
  const double step = 256 / ratio;
  int e = 0;
  const double f = std::frexp (step, &e);                  // step = f * 2^e, 0.5 <= f < 1
  cd.ctab = vt->ctab;
  cd.hl = vt->hl;
  cd.stride = vt->hl | 1;
  cd.mant = (unsigned long long) std::ldexp (f, 53);       // step = mant * 2^(e - 53)
  cd.shift = 61 - e;                                       // m * step / 256 = (m * mant) >> shift
  cd.frac_scale = std::ldexp (1.0, e - 53);
  cd.n_in = n_in;
  cd.n_out = n_out;
  cd.rows = 0;
  return cd;
}

SpeedKeyTables *
get_speed_key_tables (awm_ctx *ctx, const Key& key)
{
  std::lock_guard<std::mutex> lock (ctx->speed_mutex);
  SpeedWorkspace *ws = workspace (ctx);
  std::vector<unsigned char> kb (key.aes_key(), key.aes_key() + Key::SIZE);
  for (auto& t : ws->key_tables)
    if (t->key == kb && t->frames_per_bit == params().frames_per_bit)
      return t.get();
  KeyTables *kt = ctx->get_key_tables (key);
  if (!kt)
    return nullptr;
  const SyncTable& st = kt->sync[0].host;                  // BLOCK mode: 6 x 85 rows sorted by frame within a bit
  const int R = st.rows_per_bit;
  std::vector<unsigned int> cols (size_t (6) * R * 16, 0);
  std::vector<int> col_frame (size_t (6) * R);
  for (int col = 0; col < 6 * R; col++)
    {
      unsigned char *bytes = reinterpret_cast<unsigned char *> (&cols[size_t (col) * 16]);
      for (int i = 0; i < 30; i++)
        {
          bytes[i] = st.up[size_t (col) * 30 + i];
          bytes[30 + i] = st.down[size_t (col) * 30 + i];
        }
      col_frame[col] = st.frame[col];
    }
  const int fpb = int (mark_block_frame_count());
  std::vector<unsigned char> col_first (size_t (6) * (fpb + 2), 0);
  for (int bit = 0; bit < 6; bit++)
    for (int f = 0; f < fpb + 2; f++)
      {
        int count = 0;
        for (int r = 0; r < R; r++)
          count += st.frame[size_t (bit) * R + r] < f;
        col_first[size_t (bit) * (fpb + 2) + f] = (unsigned char) count;
      }
  auto t = std::make_unique<SpeedKeyTables>();
  t->key = kb;
  t->frames_per_bit = params().frames_per_bit;
  if (upload_sync (t->col_first, col_first.data(), col_first.size(), ctx->stream))
    return nullptr;
  if (upload_sync (t->cols, cols.data(), cols.size() * sizeof (unsigned int), ctx->stream))
    return nullptr;
  if (upload_sync (t->col_frame, col_frame.data(), col_frame.size() * sizeof (int), ctx->stream))
    return nullptr;
  ws->key_tables.push_back (std::move (t));
  return ws->key_tables.back().get();
}

int
ensure_window (awm_ctx *ctx)
{
  std::lock_guard<std::mutex> lock (ctx->speed_mutex);
  SpeedWorkspace *ws = workspace (ctx);
  if (ws->window512.ptr)
    return 0;
  const std::vector<float> win = gen_normalized_window (Params::frame_size / 2);
  return upload_sync (ws->window512, win.data(), win.size() * sizeof (float), ctx->stream);
}

/* get_speed_clip (reference wmspeed.cc:33-52): frame range of the clip */
void
speed_clip_range (double location, const DeviceWav& wav, double clip_seconds, size_t *start_point, size_t *end_point)
{
  const double end_sec = double (wav.n_frames) / wav.sample_rate;
  double start_sec = location * (end_sec - clip_seconds);
  if (start_sec < 0)
    start_sec = 0;
  *start_point = start_sec * wav.sample_rate;
  *end_point = std::min<size_t> (*start_point + clip_seconds * wav.sample_rate, wav.n_frames);
}

struct ScanCenter { double speed; VarResampleTable *table; long long n_in, n_out; int rows; };

/* resample + STFT + column sums of all centres of a pass; leaves the matrices in ws->mags */
int
prepare_mags (awm_ctx *ctx, WorkLane *lane, const Key& key, const DeviceWav& clip, double seconds, std::vector<ScanCenter>& centers,
              long long *ld_out, long long *center_stride_out)
{
  SpeedScratch *ws = speed_scratch (lane);
  SpeedKeyTables *skt = get_speed_key_tables (ctx, key);
  if (!skt)
    return AWM_ERR_HIP;
  if (int rc = ensure_window (ctx))
    return rc;
  const int C = clip.n_channels;
  std::vector<double> ratios;
  for (auto& c : centers)
    ratios.push_back (c.speed / 2);                        // "we downsample the audio by factor 2 to improve performance"
  std::vector<VarResampleTable *> tables;
  if (int rc = get_var_tables (ctx, ratios, tables))
    return rc;
  const int N = Params::frame_size / 2, hop = Params::sync_search_step / 2;
  long long max_out = 0;
  int max_rows = 0;
  std::vector<awmk::SpeedCenterDev> cds;
  for (size_t i = 0; i < centers.size(); i++)
    {
      ScanCenter& c = centers[i];
      c.table = tables[i];
      // resample_ratio_truncate (clip, center / 2, mark_sample_rate / 2, seconds / center)
      size_t in_frames = clip.n_frames;
      const double max_in_seconds = seconds / c.speed;
      if (max_in_seconds > 0)
        in_frames = std::min<size_t> (in_frames * C, C * lrint (clip.sample_rate * max_in_seconds)) / C;
      c.n_in = (long long) in_frames;
      c.n_out = lrint (in_frames * ratios[i]);
      c.rows = c.n_out > N ? int ((c.n_out - N - 1) / hop + 1) : 0;        // pos + N < n_out, pos = 0, hop, ...
      awmk::SpeedCenterDev cd = center_dev (c.table, ratios[i], c.n_in, c.n_out);
      cd.rows = c.rows;
      cds.push_back (cd);
      max_out = std::max (max_out, c.n_out);
      max_rows = std::max (max_rows, c.rows);
    }
  const long long ld = (max_rows + 15) / 16 * 16;
  const long long sub_stride = (max_out * C + 3) / 4 * 4;
  const long long center_stride = 510 * ld;
  if (int rc = ws->sub.reserve (std::max<size_t> (size_t (sub_stride) * centers.size() * sizeof (float), 16)))
    return rc;
  if (int rc = ws->mags.reserve (std::max<size_t> (size_t (center_stride) * centers.size() * sizeof (float2), 16)))
    return rc;
  if (int rc = ws->centers.reserve (cds.size() * sizeof (awmk::SpeedCenterDev)))
    return rc;
  if (int rc = ws->pin.reserve (std::max<size_t> (cds.size() * sizeof (awmk::SpeedCenterDev), 1 << 16)))
    return rc;
  hipStream_t st = lane->stream;
  std::memcpy (ws->pin.ptr, cds.data(), cds.size() * sizeof (awmk::SpeedCenterDev));
  AWM_HIP_CHECK (hipMemcpyAsync (ws->centers.ptr, ws->pin.ptr, cds.size() * sizeof (awmk::SpeedCenterDev), hipMemcpyHostToDevice, st));
  awmk::VarResampleArgs ra {};
  ra.in = clip.data;
  ra.n_channels = C;
  ra.centers = ws->centers.as<awmk::SpeedCenterDev>();
  ra.out = ws->sub.as<float>();
  ra.out_stride = sub_stride;
  for (const auto& cd : cds)
    {
      ra.max_stride = std::max (ra.max_stride, cd.stride);
      ra.max_step = std::max (ra.max_step, std::ldexp (double (cd.mant), -cd.shift));
    }
  double var_bytes = 0, mags_bytes = 0;
  for (const auto& c : centers)
    {
      var_bytes += double (c.n_in + c.n_out) * C * 4.0;                   // K12: the clip in, the resampled clip out, per centre speed
      mags_bytes += double (c.n_out) * C * 4.0 + double (c.rows) * 510 * 8.0;   // K13: the resampled clip once, { umag, dmag } per row and sync frame out
    }
  {
    ProfScope ps (ctx, PROF_RESAMPLE_VAR, var_bytes, st);
    AWM_HIP_CHECK (awmk::launch_resample_var (st, ra, max_out, int (centers.size())));
  }
  awmk::SpeedMagsArgs ma {};
  ma.sub = ws->sub.as<float>();
  ma.sub_stride = sub_stride;
  ma.n_channels = C;
  ma.centers = ra.centers;
  ma.window512 = workspace (ctx)->window512.as<float>();        // (built by ensure_window above; never replaced)
  ma.cols = skt->cols.as<unsigned int>();
  ma.mags = ws->mags.as<float2>();
  ma.mags_center_stride = center_stride;
  ma.ld = ld;
  {
    ProfScope ps (ctx, PROF_SPEED_MAGS, mags_bytes, st);
    AWM_HIP_CHECK (awmk::launch_speed_mags (st, ctx->tabs, ma, max_rows, int (centers.size())));
  }
  AWM_HIP_CHECK (hipStreamSynchronize (st));               // the pinned staging area is reused by the caller
  *ld_out = ld;
  *center_stride_out = center_stride;
  return 0;
}

} // namespace

DevBuffer&
speed_stretch_buffer (WorkLane *lane)
{
  return speed_scratch (lane)->stretched;
}

VarResampleGeometry
var_resample_geometry (double ratio)
{
  VarResampleGeometry g;
  const VarGeometry vg = var_geometry (ratio);
  if (!vg.ok)
    return g;
  int e = 0;
  const double f = std::frexp (vg.step, &e);
  g.ok = true;
  g.hl = int (vg.hl);
  g.mant = (unsigned long long) std::ldexp (f, 53);
  g.shift = 61 - e;
  return g;
}

int
resample_var_device (awm_ctx *ctx, WorkLane *lane, const float *in_d, size_t n_in, int n_channels, double ratio, float *out_d, size_t n_out)
{
  SpeedScratch *ws = speed_scratch (lane);
  std::vector<VarResampleTable *> tables;
  if (int rc = get_var_tables (ctx, { ratio }, tables))
    return rc;
  if (!n_out)
    return 0;
  const awmk::SpeedCenterDev cd = center_dev (tables[0], ratio, (long long) n_in, (long long) n_out);
  if (int rc = ws->centers.reserve (sizeof (cd)))
    return rc;
  hipStream_t st = lane->stream;
  AWM_HIP_CHECK (hipMemcpyAsync (ws->centers.ptr, &cd, sizeof (cd), hipMemcpyHostToDevice, st));
  AWM_HIP_CHECK (hipStreamSynchronize (st));
  awmk::VarResampleArgs ra {};
  ra.in = in_d;
  ra.n_channels = n_channels;
  ra.centers = ws->centers.as<awmk::SpeedCenterDev>();
  ra.out = out_d;
  ra.out_stride = 0;
  ra.max_stride = cd.stride;
  ra.max_step = std::ldexp (double (cd.mant), -cd.shift);
  {
    ProfScope ps (ctx, PROF_RESAMPLE_VAR, double (n_in + n_out) * n_channels * 4.0, st);
    AWM_HIP_CHECK (awmk::launch_resample_var (st, ra, (long long) n_out, 1));
  }
  AWM_HIP_CHECK (hipStreamSynchronize (st));
  return 0;
}

int
resample_ratio_device (awm_ctx *ctx, WorkLane *lane, const DeviceWav& wav, double ratio, double max_in_seconds, DevBuffer& out,
                       size_t *n_out_frames)
{
  const int C = wav.n_channels;
  size_t in_frames = wav.n_frames;
  if (max_in_seconds > 0)
    in_frames = std::min<size_t> (in_frames * C, C * lrint (wav.sample_rate * max_in_seconds)) / C;
  if (!var_geometry (ratio).ok)
    {
      set_error (string_printf ("failed to setup vresampler with ratio=%f", ratio));
      return AWM_ERR_ARG;
    }
  const long long n_out = lrint (in_frames * ratio);
  *n_out_frames = size_t (n_out);
  if (int rc = out.reserve (std::max<size_t> (size_t (n_out) * C * sizeof (float), 16)))
    return rc;
  return resample_var_device (ctx, lane, wav.data, in_frames, C, ratio, out.as<float>(), size_t (n_out));
}

int
speed_clip_location (awm_ctx *ctx, WorkLane *lane, const Key& key, const DeviceWav& wav, double seconds, int candidates, double *location)
{
  (void) ctx;
  SpeedScratch *ws = speed_scratch (lane);
  hipStream_t st = lane->stream;
  /* get_clip_locations (reference wmspeed.cc:533-553): hash a sparse subset of the samples */
  Random rng (key, 0, Random::Stream::speed_clip);
  std::vector<unsigned long long> pos;
  const size_t n_values = wav.n_values();
  pos.reserve (n_values / 400 + 16);
  for (size_t p = 0; p < n_values; p += rng() % 1000)
    pos.push_back (p);
  const size_t pos_bytes = pos.size() * sizeof (unsigned long long), val_bytes = pos.size() * sizeof (float);
  if (int rc = ws->gather_pos.reserve (std::max<size_t> (pos_bytes, 16)))
    return rc;
  if (int rc = ws->gather_out.reserve (std::max<size_t> (val_bytes, 16)))
    return rc;
  if (int rc = ws->pin.reserve (std::max<size_t> (std::max (pos_bytes, val_bytes), 1 << 16)))
    return rc;
  std::memcpy (ws->pin.ptr, pos.data(), pos_bytes);
  AWM_HIP_CHECK (hipMemcpyAsync (ws->gather_pos.ptr, ws->pin.ptr, pos_bytes, hipMemcpyHostToDevice, st));
  AWM_HIP_CHECK (awmk::launch_gather_values (st, wav.data, ws->gather_pos.as<unsigned long long>(), (long long) pos.size(), ws->gather_out.as<float>()));
  AWM_HIP_CHECK (hipStreamSynchronize (st));               // the staging area changes direction
  AWM_HIP_CHECK (hipMemcpyAsync (ws->pin.ptr, ws->gather_out.ptr, val_bytes, hipMemcpyDeviceToHost, st));
  AWM_HIP_CHECK (hipStreamSynchronize (st));
  unsigned char hash[20];
  sha1 (ws->pin.ptr, val_bytes, hash);                     // Random::seed_from_hash (reference random.cc:184-190)
  uint64_t seed = 0;
  for (int i = 0; i < 8; i++)
    seed = (seed << 8) | hash[i];
  rng.seed (seed, Random::Stream::speed_clip);
  std::vector<double> locations;
  for (int c = 0; c < candidates; c++)
    locations.push_back (rng.random_double());

  /* get_best_clip_location (reference wmspeed.cc:555-577): the candidate with the highest energy */
  std::vector<long long> ranges;
  for (double loc : locations)
    {
      size_t start_point, end_point;
      speed_clip_range (loc, wav, seconds, &start_point, &end_point);
      ranges.push_back ((long long) (start_point * wav.n_channels));
      ranges.push_back ((long long) (end_point * wav.n_channels));
    }
  const size_t e_bytes = size_t (candidates) * awmk::ENERGY_PARTS * sizeof (double);
  if (int rc = ws->ranges.reserve (ranges.size() * sizeof (long long)))
    return rc;
  if (int rc = ws->energy.reserve (e_bytes))
    return rc;
  std::memcpy (ws->pin.ptr, ranges.data(), ranges.size() * sizeof (long long));
  AWM_HIP_CHECK (hipMemcpyAsync (ws->ranges.ptr, ws->pin.ptr, ranges.size() * sizeof (long long), hipMemcpyHostToDevice, st));
  AWM_HIP_CHECK (awmk::launch_energy (st, wav.data, ws->ranges.as<long long>(), candidates, ws->energy.as<double>()));
  AWM_HIP_CHECK (hipStreamSynchronize (st));
  AWM_HIP_CHECK (hipMemcpyAsync (ws->pin.ptr, ws->energy.ptr, e_bytes, hipMemcpyDeviceToHost, st));
  AWM_HIP_CHECK (hipStreamSynchronize (st));
  const double *parts = ws->pin.as<double>();
  double clip_location = 0, best_energy = 0;
  for (int c = 0; c < candidates; c++)
    {
      double energy = 0;
      for (int k = 0; k < awmk::ENERGY_PARTS; k++)
        energy += parts[c * awmk::ENERGY_PARTS + k];
      if (energy > best_energy)
        {
          best_energy = energy;
          clip_location = locations[c];
        }
    }
  *location = clip_location;
  return 0;
}

int
speed_scan (awm_ctx *ctx, WorkLane *lane, const Key& key, const DeviceWav& wav, double clip_location, const SpeedScanParams& sp,
            const std::vector<double>& speeds, std::vector<SpeedScore>& scores)
{
  SpeedScratch *ws = speed_scratch (lane);
  scores.clear();
  /* SpeedSearch::get_jobs (reference wmspeed.cc:461-492): "speed is between 0.8 and 1.25, so we use a clip seconds
   * factor of 1.3 to provide enough samples" */
  size_t start_point, end_point;
  speed_clip_range (clip_location, wav, sp.seconds * 1.3, &start_point, &end_point);
  DeviceWav clip = wav;
  clip.data = wav.data + start_point * wav.n_channels;
  clip.n_frames = end_point - start_point;
  std::vector<ScanCenter> centers;
  for (double speed : speeds)
    for (int c = -sp.n_center_steps; c <= sp.n_center_steps; c++)
      centers.push_back ({ speed * std::pow (sp.step, c * (sp.n_steps * 2 + 1)), nullptr, 0, 0, 0 });
  if (centers.empty())
    return 0;
  long long ld = 0, center_stride = 0;
  if (int rc = prepare_mags (ctx, lane, key, clip, sp.seconds, centers, &ld, &center_stride))
    return rc;
  SpeedKeyTables *skt = get_speed_key_tables (ctx, key);
  /* SpeedSync::get_jobs (reference wmspeed.cc:172-192): relative speeds step^p, p = -n_steps .. n_steps, per centre */
  std::vector<awmk::SpeedItemDev> items;
  std::vector<double> item_speed;
  for (size_t ci = 0; ci < centers.size(); ci++)
    for (int p = -sp.n_steps; p <= sp.n_steps; p++)
      {
        const double center = centers[ci].speed;
        const double relative_speed = std::pow (sp.step, p) * center / center;
        awmk::SpeedItemDev it {};
        it.center = int (ci);
        it.rel_speed_inv = 1 / relative_speed;
        it.q16_scale = (1 << 16) / relative_speed;
        items.push_back (it);
        item_speed.push_back (relative_speed * center);
      }
  const size_t items_bytes = items.size() * sizeof (awmk::SpeedItemDev), best_bytes = items.size() * sizeof (unsigned long long);
  if (int rc = ws->items.reserve (items_bytes))
    return rc;
  if (int rc = ws->best.reserve (best_bytes))
    return rc;
  if (int rc = ws->pin.reserve (std::max<size_t> (std::max (items_bytes, best_bytes), 1 << 16)))
    return rc;
  hipStream_t st = lane->stream;
  std::memcpy (ws->pin.ptr, items.data(), items_bytes);
  AWM_HIP_CHECK (hipMemcpyAsync (ws->items.ptr, ws->pin.ptr, items_bytes, hipMemcpyHostToDevice, st));
  AWM_HIP_CHECK (hipMemsetAsync (ws->best.ptr, 0, best_bytes, st));
  const int steps_per_frame = Params::frame_size / Params::sync_search_step;
  const int frames_per_block = int (mark_block_frame_count());
  awmk::SpeedCompareArgs ca {};
  ca.mags = ws->mags.as<float2>();
  ca.mags_center_stride = center_stride;
  ca.ld = ld;
  ca.centers = ws->centers.as<awmk::SpeedCenterDev>();
  ca.items = ws->items.as<awmk::SpeedItemDev>();
  ca.n_centers = int (centers.size());
  ca.items_per_center = 2 * sp.n_steps + 1;
  ca.col_frame = skt->col_frame.as<int>();
  ca.col_first = skt->col_first.as<unsigned char>();
  ca.frames_per_block = frames_per_block;
  ca.steps_per_frame = steps_per_frame;
  ca.pad_start = frames_per_block * steps_per_frame + steps_per_frame;      // "a bit of overlap to handle boundaries"
  ca.rows_per_bit = Params::sync_frames_per_bit;
  ca.min_delta = std::min (params().water_delta, 0.080);
  ca.best = ws->best.as<unsigned long long>();
  {
    // K14: every centre's { umag, dmag } matrix once (its 2 n_steps + 1 relative speeds share it) + one best score per item
    double bytes = double (items.size()) * 8.0;
    for (const auto& c : centers)
      bytes += double (c.rows) * 510 * 8.0;
    ProfScope ps (ctx, PROF_SPEED_COMPARE, bytes, st);
    AWM_HIP_CHECK (awmk::launch_speed_compare (st, ca, int (items.size())));
  }
  AWM_HIP_CHECK (hipStreamSynchronize (st));
  AWM_HIP_CHECK (hipMemcpyAsync (ws->pin.ptr, ws->best.ptr, best_bytes, hipMemcpyDeviceToHost, st));
  AWM_HIP_CHECK (hipStreamSynchronize (st));
  const double *best = ws->pin.as<double>();
  for (size_t i = 0; i < items.size(); i++)
    {
      SpeedScore sc;                                       // "Score best_score": stays { 0, 0 } unless a state has quality > 0
      if (best[i] > 0)
        {
          sc.quality = best[i];
          sc.speed = item_speed[i];
        }
      scores.push_back (sc);
    }
  return 0;
}

int
speed_mags (awm_ctx *ctx, WorkLane *lane, const Key& key, const DeviceWav& wav, double clip_location, double center, double seconds,
            std::vector<float>& out, int *rows)
{
  SpeedScratch *ws = speed_scratch (lane);
  size_t start_point, end_point;
  speed_clip_range (clip_location, wav, seconds * 1.3, &start_point, &end_point);
  DeviceWav clip = wav;
  clip.data = wav.data + start_point * wav.n_channels;
  clip.n_frames = end_point - start_point;
  std::vector<ScanCenter> centers { { center, nullptr, 0, 0, 0 } };
  long long ld = 0, center_stride = 0;
  if (int rc = prepare_mags (ctx, lane, key, clip, seconds, centers, &ld, &center_stride))
    return rc;
  KeyTables *kt = ctx->get_key_tables (key);
  const SyncTable& stab = kt->sync[0].host;
  std::vector<float2> m (size_t (510) * ld);
  AWM_HIP_CHECK (hipMemcpy (m.data(), ws->mags.ptr, m.size() * sizeof (float2), hipMemcpyDeviceToHost));
  // back to the reference's column order (all 510 sync frames sorted by frame)
  std::vector<int> order (510);
  for (int i = 0; i < 510; i++)
    order[i] = i;
  std::sort (order.begin(), order.end(), [&] (int a, int b) { return stab.frame[a] < stab.frame[b]; });
  *rows = centers[0].rows;
  out.assign (size_t (*rows) * 510 * 2, 0.f);
  for (int r = 0; r < *rows; r++)
    for (int c = 0; c < 510; c++)
      {
        const float2 v = m[size_t (order[c]) * ld + r];
        out[(size_t (r) * 510 + c) * 2] = v.x;
        out[(size_t (r) * 510 + c) * 2 + 1] = v.y;
      }
  return 0;
}

void
select_n_best_scores (std::vector<SpeedScore>& scores, size_t n)
{
  std::sort (scores.begin(), scores.end(), [] (const SpeedScore& a, const SpeedScore& b) { return a.speed < b.speed; });
  const auto quality_at = [&] (int pos) { return pos >= 0 && size_t (pos) < scores.size() ? scores[pos].quality : 0.0; };
  std::vector<SpeedScore> peaks;
  for (int x = 0; size_t (x) < scores.size(); x++)
    {
      // single peak, or the first of two equal values that are larger than their outer neighbours
      if (quality_at (x - 1) <= quality_at (x) && quality_at (x) >= quality_at (x + 1))
        {
          peaks.push_back (scores[x]);
          x++;                                             // its right neighbour cannot be a local maximum
        }
    }
  std::sort (peaks.begin(), peaks.end(), [] (const SpeedScore& a, const SpeedScore& b) { return a.quality > b.quality; });
  if (peaks.size() > n)
    peaks.resize (n);
  scores = peaks;
}

double
score_smooth_find_best (const std::vector<SpeedScore>& in_scores, double step, double distance)
{
  std::vector<SpeedScore> scores = in_scores;
  std::sort (scores.begin(), scores.end(), [] (const SpeedScore& a, const SpeedScore& b) { return a.speed < b.speed; });
  const auto window_cos = [] (double x) { return std::fabs (x) > 1 ? 0.0 : 0.5 * std::cos (x * M_PI) + 0.5; };   // von Hann
  double best_speed = 0, best_quality = 0;
  for (double speed = scores.front().speed; speed < scores.back().speed; speed += 0.000001)
    {
      double quality_sum = 0, quality_div = 0;
      for (const auto& s : scores)
        {
          const double w = window_cos ((s.speed - speed) / (step * distance));
          quality_sum += s.quality * w;
          quality_div += w;
        }
      quality_sum /= quality_div;
      if (quality_sum > best_quality)
        {
          best_speed = speed;
          best_quality = quality_sum;
        }
    }
  return best_speed;
}

int
detect_speed (awm_ctx *ctx, WorkLane *lane, const std::vector<Key>& key_list, const DeviceWav& wav, std::string *report,
              std::vector<DetectSpeedResult>& results, double *best_speed_out, double *best_quality_out)
{
  results.clear();
  /* "our algorithm won't work at all for very short input files" */
  const double in_seconds = double (wav.n_frames) / wav.sample_rate;
  if (in_seconds < 0.25)
    return 0;
  const bool patient = params().detect_speed_patient;
  // first pass: grid over 0.8 .. 1.25; second: improve the n best; third: fast refinement around the best
  const SpeedScanParams scan1 = patient ? SpeedScanParams { 50, 1.00035, 11, 28 } : SpeedScanParams { 25, 1.0007, 5, 28 };
  const SpeedScanParams scan2 = patient ? SpeedScanParams { 50, 1.000175, 1, 0 } : SpeedScanParams { 50, 1.00035, 1, 0 };
  const SpeedScanParams scan3 { 50, 1.00005, 40, 0 };
  const double scan3_smooth_distance = 20;
  const double speed_sync_threshold = 0.4;
  const size_t n_best = patient ? 15 : 5;
  const int clip_candidates = 5;
  for (const Key& key : key_list)
    {
      double clip_location = 0;
      if (int rc = speed_clip_location (ctx, lane, key, wav, scan1.seconds, clip_candidates, &clip_location))
        return rc;
      std::vector<SpeedScore> scores;
      if (int rc = speed_scan (ctx, lane, key, wav, clip_location, scan1, { 1.0 }, scores))
        return rc;
      select_n_best_scores (scores, n_best);
      std::vector<double> speeds;
      for (const auto& s : scores)
        speeds.push_back (s.speed);
      if (int rc = speed_scan (ctx, lane, key, wav, clip_location, scan2, speeds, scores))
        return rc;
      select_n_best_scores (scores, 1);
      if (scores.empty())
        continue;
      if (int rc = speed_scan (ctx, lane, key, wav, clip_location, scan3, { scores[0].speed }, scores))
        return rc;
      const double best_speed = score_smooth_find_best (scores, 1 - scan3.step, scan3_smooth_distance);
      double best_quality = 0;
      for (const auto& s : scores)
        best_quality = std::max (best_quality, s.quality);
      if (report)
        {
          double delta = -1;
          if (params().test_speed > 0)
            delta = 100 * std::fabs (best_speed - params().test_speed) / params().test_speed;
          *report += string_printf ("detect_speed %f %f %.4f\n", best_speed, best_quality, delta);
        }
      if (best_speed_out)
        *best_speed_out = best_speed;
      if (best_quality_out)
        *best_quality_out = best_quality;
      if (best_quality > speed_sync_threshold)
        {
          // "speeds closer to 1.0 than this usually work without stretching before decode"
          if (best_speed < 0.9999 || best_speed > 1.0001)
            results.push_back ({ key, best_speed });
        }
    }
  return 0;
}

} // namespace awm
