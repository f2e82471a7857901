#include "syncfinder.hh"
#include "utils.hh"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace awm {

static inline int
total_frames (SyncFinder::Mode mode)
{
  return int (mark_block_frame_count()) * (mode == SyncFinder::Mode::CLIP ? 2 : 1);
}

int
SyncFinder::scan_silence (const DeviceWav& wav)
{
  if (int rc = prepare_launch (wav, Mode::CLIP))
    return rc;
  return prepare_finish();
}

int
SyncFinder::prepare_launch (const DeviceWav& wav, Mode mode, size_t scan_lo, size_t scan_hi)
{
  m_prepare_pending = false;
  if (mode != Mode::CLIP)
    {
      m_first = 0;
      m_last = wav.n_values();
      return 0;
    }
  // padding is not transformed and does not count (reference syncfinder.cc:155-169, 491-502)
  if (int rc = m_lane->ws_misc.reserve (64)) return rc;
  if (int rc = m_lane->pin_small.reserve (64)) return rc;
  auto *res = m_lane->ws_misc.as<unsigned long long>();
  scan_hi = std::min (scan_hi, wav.n_values());
  scan_lo = std::min (scan_lo, scan_hi);
  AWM_HIP_CHECK (awmk::launch_nonzero_range (m_lane->stream, wav.data + scan_lo, (long long) (scan_hi - scan_lo), res));
  AWM_HIP_CHECK (hipMemcpyAsync (m_lane->pin_small.ptr, res, 2 * sizeof (unsigned long long), hipMemcpyDeviceToHost, m_lane->stream));
  m_prepare_values = wav.n_values();
  m_prepare_lo = scan_lo;
  m_prepare_pending = true;
  return 0;
}

int
SyncFinder::prepare_finish()
{
  if (!m_prepare_pending)
    return 0;
  m_prepare_pending = false;
  AWM_HIP_CHECK (stream_wait (m_lane->stream));
  const auto *h = m_lane->pin_small.as<unsigned long long>();
  const bool none = h[0] == ~0ULL;                       // every value is zero
  m_first = none ? m_prepare_values : m_prepare_lo + h[0];
  m_last = none ? m_prepare_values : m_prepare_lo + h[1];
  return 0;
}

/* reference syncfinder.cc:171-256 */
int
SyncFinder::search_approx (KeyTables *kt, const DeviceWav& wav, Mode mode, std::vector<SearchScore>& out)
{
  out.clear();
  long long n_scores = 0;
  if (int rc = approx_device (kt, wav, mode, n_scores))
    return rc;
  return fetch_scores (n_scores, out);
}

int
SyncFinder::fetch_scores (long long n_scores, std::vector<SearchScore>& out)
{
  out.clear();
  if (n_scores <= 0)
    return 0;
  hipStream_t st = m_lane->stream;
  std::vector<double> raw (n_scores), mean (n_scores);
  AWM_HIP_CHECK (hipMemcpyAsync (raw.data(), m_lane->ws_raw.ptr, raw.size() * sizeof (double), hipMemcpyDeviceToHost, st));
  AWM_HIP_CHECK (hipMemcpyAsync (mean.data(), m_lane->ws_mean.ptr, mean.size() * sizeof (double), hipMemcpyDeviceToHost, st));
  AWM_HIP_CHECK (stream_wait (st));
  out.resize (raw.size());
  for (size_t p = 0; p < raw.size(); p++)
    {
      // sorted by index: index = start_frame * 1024 + shift * 256
      out[p].index = (p >> 2) * Params::frame_size + (p & 3) * Params::sync_search_step;
      out[p].raw_quality = raw[p];
      out[p].local_mean = mean[p];
    }
  return 0;
}

int
SyncFinder::approx_device (KeyTables *kt, const DeviceWav& wav, Mode mode, long long& n_scores, bool db_ready, bool scores_only)
{
  n_scores = 0;
  const int clip = mode == Mode::CLIP;
  const long long frame_count = wav.n_frames / Params::frame_size;
  const long long n_db = frame_count - 1;          // sync_fft_parallel drops the last frame (syncfinder.cc:632)
  const long long S = n_db - total_frames (mode);  // start frames with (start + total) * 81 < db.size()
  if (n_db <= 0 || S <= 0)
    return 0;
  hipStream_t st = m_lane->stream;
  const int n_shifts = Params::frame_size / Params::sync_search_step;
  const long long ld = (n_db + 63) & ~63LL;
  const long long plane = ld * Params::n_bands;
  const long long q_stride = (S + 63) & ~63LL;
  if (int rc = m_lane->ws_db.reserve (size_t (n_shifts) * plane * sizeof (float))) return rc;
  if (int rc = m_lane->ws_have.reserve (size_t (n_shifts) * ld)) return rc;
  if (int rc = m_lane->ws_q.reserve (size_t (n_shifts) * q_stride * sizeof (double))) return rc;
  if (int rc = m_lane->ws_raw.reserve (size_t (n_shifts) * S * sizeof (double))) return rc;
  if (int rc = m_lane->ws_mean.reserve (size_t (n_shifts) * S * sizeof (double))) return rc;

  awmk::SyncDbArgs da {};
  da.pcm = wav.data;
  da.n_frames = wav.n_frames;
  da.n_channels = wav.n_channels;
  da.per_channel = 0;
  da.base0 = 0;
  da.base_stride = Params::sync_search_step;
  da.count0 = int (n_db);
  da.n_streams = n_shifts;
  da.hop = Params::frame_size;
  da.out = m_lane->ws_db.as<float>();
  da.out_stream_stride = plane;
  da.ld = ld;
  da.have = m_lane->ws_have.as<char>();
  da.have_stream_stride = ld;
  da.first = (long long) m_first;
  da.last = (long long) m_last;
  da.tile_frames = 32;
  if (!db_ready)                                     // (else: another key of the same `get` left these matrices in the workspace)
    {
      // algorithmic bytes: the samples ONCE (the four shifts of a tile read the same samples; they share one XCD's L2) + one dB row per frame and shift
      ProfScope ps (m_ctx, PROF_SYNC_DB, double (n_db) * 4096.0 * wav.n_channels + double (n_shifts) * n_db * 324.0, st);
      AWM_HIP_CHECK (awmk::launch_sync_db (st, m_ctx->tabs, da));
    }
  if (after_db_event)
    AWM_HIP_CHECK (hipEventRecord (after_db_event, st));

  awmk::SyncScanArgs sa {};
  sa.db = m_lane->ws_db.as<float>();
  sa.have = clip ? m_lane->ws_have.as<char>() : nullptr;     // BLOCK mode never skips a frame
  sa.have_is_run = 1;                                         // (K4 skips by the non-silent range: one run, absent frames +0)
  sa.plane_stride = plane;
  sa.have_plane_stride = ld;
  sa.row_stride = 1;
  sa.band_stride = ld;
  sa.have_row_stride = 1;
  sa.n_lanes = S;
  sa.n_planes = n_shifts;
  sa.min_delta = std::min (params().water_delta, 0.080);
  sa.quality = m_lane->ws_q.as<double>();
  sa.q_stride = q_stride;
  sa.table.packed = kt->sync[clip].packed_approx.as<int>();
  sa.table.rows_per_bit = kt->sync[clip].host.rows_per_bit;
  sa.table.chains = kt->sync[clip].chains_approx.as<unsigned>();
  {
    // algorithmic HBM bytes of the scan: the dB matrix once, candidates re-read it from cache
    ProfScope ps (m_ctx, PROF_SYNC_SCAN, double (n_shifts) * n_db * 324.0 + double (n_shifts) * S * 8.0, st);
    AWM_HIP_CHECK (awmk::launch_sync_scan_window (st, sa, total_frames (mode)));
  }
  if (after_scan_event)
    AWM_HIP_CHECK (hipEventRecord (after_scan_event, st));
  n_scores = (long long) n_shifts * S;
  if (scores_only)
    return 0;
  ProfScope ps (m_ctx, PROF_LOCAL_MEAN, double (n_shifts) * S * 24.0, st);
  AWM_HIP_CHECK (awmk::launch_local_mean (st, m_lane->ws_q.as<double>(), q_stride, S, m_lane->ws_raw.as<double>(), m_lane->ws_mean.as<double>()));
  return 0;
}

int
SyncFinder::scores_loaded (long long S)
{
  if (S <= 0)
    return 0;
  const int n_shifts = Params::frame_size / Params::sync_search_step;
  const long long q_stride = (S + 63) & ~63LL;
  if (int rc = m_lane->ws_raw.reserve (size_t (n_shifts) * S * sizeof (double))) return rc;
  if (int rc = m_lane->ws_mean.reserve (size_t (n_shifts) * S * sizeof (double))) return rc;
  ProfScope ps (m_ctx, PROF_LOCAL_MEAN, double (n_shifts) * S * 24.0, m_lane->stream);
  AWM_HIP_CHECK (awmk::launch_local_mean (m_lane->stream, m_lane->ws_q.as<double>(), q_stride, S, m_lane->ws_raw.as<double>(), m_lane->ws_mean.as<double>()));
  return 0;
}

/* sync_select_local_maxima + sync_mask_avg_false_positives + sync_select_threshold_and_n_best
 * (reference syncfinder.cc:519-526).  Fast path: the three steps run on the device (K5c) and only the
 * survivors above the threshold come back; if fewer than n_best survive, the reference keeps the n_best
 * largest maxima regardless of the threshold -- that rare case (unmarked or very short material) pulls all
 * scores to the host and runs the sequential formulation. */
int
SyncFinder::select_candidates (long long n_scores, double threshold, std::vector<SearchScore>& out)
{
  if (int rc = select_launch (n_scores, threshold))
    return rc;
  return select_finish (n_scores, threshold, out);
}

namespace {
constexpr unsigned int PEAK_CAP = 16384, PEAK_HEAD = 1024;
constexpr int TOPK_MAX = 64, TOPK_SLICES = 64;

SyncFinder::SearchScore
score_of_peak (const awmk::PeakOut& pk)
{
  return { size_t (pk.p >> 2) * Params::frame_size + size_t (pk.p & 3) * Params::sync_search_step, pk.raw, pk.mean };
}

/* the peaks above the threshold (device order) -> the order the reference continues with: descending quality */
void
scores_from_threshold_list (const awmk::PeakOut *peaks, unsigned int count, std::vector<SyncFinder::SearchScore>& out)
{
  using SearchScore = SyncFinder::SearchScore;
  for (unsigned int i = 0; i < count; i++)
    out.push_back (score_of_peak (peaks[i]));
  std::sort (out.begin(), out.end(), [] (const SearchScore& a, const SearchScore& b) {
    return a.abs_quality() != b.abs_quality() ? a.abs_quality() > b.abs_quality() : a.index < b.index;
  });
}

/* Fewer than n_best peaks above the threshold: select_threshold_and_n_best keeps the n_best largest unmasked maxima.  `top` is
 * the union of the per-slice n_best + 1 largest (K5d).  false: a tie across the cut -- the complete list has to go through the
 * reference's std::sort instead. */
bool
scores_from_topk (std::vector<awmk::PeakOut> top, double threshold, std::vector<SyncFinder::SearchScore>& out)
{
  top.erase (std::remove_if (top.begin(), top.end(), [] (const awmk::PeakOut& pk) { return pk.p < 0; }), top.end());
  auto absq = [] (const awmk::PeakOut& pk) { return std::fabs (pk.raw - pk.mean); };
  std::sort (top.begin(), top.end(), [&] (const awmk::PeakOut& a, const awmk::PeakOut& b) {
    return absq (a) != absq (b) ? absq (a) > absq (b) : a.p < b.p;
  });
  const size_t nb = size_t (params().get_n_best);
  if (top.size() > nb && absq (top[nb - 1]) == absq (top[nb]))
    return false;
  if (top.size() > nb)
    top.resize (nb);
  std::sort (top.begin(), top.end(), [] (const awmk::PeakOut& a, const awmk::PeakOut& b) { return a.p < b.p; });
  for (const auto& pk : top)
    out.push_back (score_of_peak (pk));
  SyncFinder::select_threshold_and_n_best (out, threshold);
  return true;
}
}

int
SyncFinder::select_launch (long long n_scores, double threshold, bool speculate_n_best)
{
  if (n_scores <= 0)
    return 0;
  hipStream_t st = m_lane->stream;
  const unsigned int cap = PEAK_CAP;
  if (int rc = m_lane->ws_misc.reserve (256 + cap * sizeof (awmk::PeakOut))) return rc;
  auto *d_count = m_lane->ws_misc.as<unsigned int>();
  auto *d_out = reinterpret_cast<awmk::PeakOut *> (m_lane->ws_misc.as<char>() + 256);
  {
    ProfScope ps (m_ctx, PROF_LOCAL_MEAN, double (n_scores) * 16.0, st);
    AWM_HIP_CHECK (awmk::launch_peak_select (st, m_lane->ws_raw.as<double>(), m_lane->ws_mean.as<double>(), n_scores, threshold, d_count, d_out, cap));
  }
  // one round trip for the counter and the first peaks (usually all of them), through page-locked memory
  if (int rc = m_lane->pin_peaks.reserve (256 + cap * sizeof (awmk::PeakOut) + TOPK_MAX * TOPK_SLICES * sizeof (awmk::PeakOut))) return rc;
  AWM_HIP_CHECK (hipMemcpyAsync (m_lane->pin_peaks.ptr, d_count, 256 + PEAK_HEAD * sizeof (awmk::PeakOut), hipMemcpyDeviceToHost, st));
  if (speculate_n_best && params().get_n_best + 1 <= TOPK_MAX)
    {
      // Short material rarely has n_best peaks above the threshold: queue the fallback (all unmasked maxima, reduced to
      // the n_best + 1 largest per slice) right away, so that select_finish finds both answers after ONE wait.
      const unsigned int big_cap = unsigned (std::min<long long> (n_scores, 1 << 22));
      if (int rc = m_lane->ws_refine.reserve (size_t (big_cap) * sizeof (awmk::PeakOut))) return rc;
      auto *d_all = m_lane->ws_refine.as<awmk::PeakOut>();
      AWM_HIP_CHECK (awmk::launch_peak_select (st, m_lane->ws_raw.as<double>(), m_lane->ws_mean.as<double>(), n_scores, -1.0, d_count, d_all, big_cap));
      const int k = params().get_n_best + 1;
      AWM_HIP_CHECK (awmk::launch_peak_topk (st, d_all, d_count, big_cap, d_out, k, TOPK_SLICES));      // the threshold list is already on its way
      AWM_HIP_CHECK (hipMemcpyAsync (m_lane->pin_peaks.as<char>() + 256 + cap * sizeof (awmk::PeakOut), d_out,
                                     size_t (k) * TOPK_SLICES * sizeof (awmk::PeakOut), hipMemcpyDeviceToHost, st));
    }
  return 0;
}

int
SyncFinder::select_finish (long long n_scores, double threshold, std::vector<SearchScore>& out, bool speculated)
{
  out.clear();
  if (n_scores <= 0)
    return 0;
  hipStream_t st = m_lane->stream;
  const unsigned int cap = PEAK_CAP;
  auto *d_count = m_lane->ws_misc.as<unsigned int>();
  auto *d_out = reinterpret_cast<awmk::PeakOut *> (m_lane->ws_misc.as<char>() + 256);
  const unsigned int head = PEAK_HEAD;
  char *pin = m_lane->pin_peaks.as<char>();
  AWM_HIP_CHECK (stream_wait (st));
  unsigned int count = *reinterpret_cast<unsigned int *> (pin);
  if (int (count) >= params().get_n_best && count <= cap)
    {
      if (count > head)
        {
          AWM_HIP_CHECK (hipMemcpyAsync (pin + 256 + head * sizeof (awmk::PeakOut), d_out + head, (count - head) * sizeof (awmk::PeakOut),
                                         hipMemcpyDeviceToHost, st));
          AWM_HIP_CHECK (stream_wait (st));
        }
      scores_from_threshold_list (reinterpret_cast<const awmk::PeakOut *> (pin + 256), count, out);   // (atomics delivered them unordered)
      return 0;
    }
  // fewer than n_best peaks above the threshold: the reference then keeps the n_best largest unmasked maxima.
  // Fetch ALL unmasked local maxima (threshold -1) from the device and finish the selection here.
  const unsigned int big_cap = unsigned (std::min<long long> (n_scores, 1 << 22));
  speculated = speculated && params().get_n_best + 1 <= TOPK_MAX;
  if (int rc = m_lane->ws_refine.reserve (size_t (big_cap) * sizeof (awmk::PeakOut))) return rc;
  auto *d_all = m_lane->ws_refine.as<awmk::PeakOut>();
  if (!speculated)
    AWM_HIP_CHECK (awmk::launch_peak_select (st, m_lane->ws_raw.as<double>(), m_lane->ws_mean.as<double>(), n_scores, -1.0, d_count, d_all, big_cap));
  // Only the n_best largest survive select_threshold_and_n_best here (fewer than n_best are above the threshold), so
  // reduce the list on the device: n_best + 1 per slice, so that a tie across the cut is visible -- in that case
  // (degenerate input) the complete list goes through the same std::sort as in the reference instead.
  const int k = params().get_n_best + 1;
  constexpr int n_slices = TOPK_SLICES;
  const bool fewer_than_n_best = int (count) < params().get_n_best;      // (not: more than `cap` above the threshold)
  if (fewer_than_n_best && k <= TOPK_MAX)
    {
      std::vector<awmk::PeakOut> top (size_t (k) * n_slices);
      if (speculated)
        {
          // already computed and copied by select_launch
          const auto *src = reinterpret_cast<const awmk::PeakOut *> (pin + 256 + cap * sizeof (awmk::PeakOut));
          std::copy (src, src + top.size(), top.begin());
        }
      else
        {
          auto *d_top = reinterpret_cast<awmk::PeakOut *> (m_lane->ws_misc.as<char>() + 256);      // the threshold list is dead
          static_assert (sizeof (awmk::PeakOut) * TOPK_MAX * n_slices <= PEAK_CAP * sizeof (awmk::PeakOut), "ws_misc too small");
          AWM_HIP_CHECK (awmk::launch_peak_topk (st, d_all, d_count, big_cap, d_top, k, n_slices));
          AWM_HIP_CHECK (hipMemcpyAsync (top.data(), d_top, top.size() * sizeof (awmk::PeakOut), hipMemcpyDeviceToHost, st));
          AWM_HIP_CHECK (stream_wait (st));
        }
      if (scores_from_topk (std::move (top), threshold, out))
        return 0;
    }
  AWM_HIP_CHECK (hipMemcpyAsync (&count, d_count, sizeof (count), hipMemcpyDeviceToHost, st));
  AWM_HIP_CHECK (stream_wait (st));
  if (count > big_cap)
    {
      // cannot happen (at most every second score is a maximum); keep the sequential formulation as a safety net
      if (int rc = fetch_scores (n_scores, out))
        return rc;
      select_local_maxima (out);
      mask_avg_false_positives (out);
      select_threshold_and_n_best (out, threshold);
      return 0;
    }
  std::vector<awmk::PeakOut> peaks (count);
  if (count)
    {
      AWM_HIP_CHECK (hipMemcpyAsync (peaks.data(), d_all, count * sizeof (awmk::PeakOut), hipMemcpyDeviceToHost, st));
      AWM_HIP_CHECK (stream_wait (st));
    }
  std::sort (peaks.begin(), peaks.end(), [] (const awmk::PeakOut& a, const awmk::PeakOut& b) { return a.p < b.p; });   // index order, as the reference's list
  for (const auto& pk : peaks)
    out.push_back (score_of_peak (pk));
  select_threshold_and_n_best (out, threshold);
  return 0;
}

/* reference syncfinder.cc:258-281 */
void
SyncFinder::select_local_maxima (std::vector<SearchScore>& scores)
{
  std::vector<SearchScore> selected;
  for (size_t i = 0; i < scores.size(); i++)
    {
      const double q = scores[i].abs_quality();
      const double q_last = i > 0 ? scores[i - 1].abs_quality() : 0;
      const double q_next = i + 1 < scores.size() ? scores[i + 1].abs_quality() : 0;
      if (q >= q_last && q >= q_next)
        {
          selected.push_back (scores[i]);
          i++;                       // the neighbour cannot be a local maximum as well
        }
    }
  scores.swap (selected);
}

/* reference syncfinder.cc:292-332 */
void
SyncFinder::mask_avg_false_positives (std::vector<SearchScore>& scores)
{
  constexpr int    mask_distance = local_mean_distance + 3;
  constexpr double mask_factor = 3;
  auto sign = [] (const SearchScore& s) { return s.raw_quality - s.local_mean < 0 ? -1 : 1; };
  std::vector<SearchScore> kept;
  const int n = int (scores.size());
  for (int i = 0; i < n; i++)
    {
      bool mask = false;
      for (int d = -mask_distance; d <= mask_distance; d++)
        {
          const int j = i + d;
          if (j == i || j < 0 || j >= n)
            continue;
          const int distance = std::abs (int (scores[i].index) - int (scores[j].index)) / Params::sync_search_step;
          if (distance <= mask_distance
              && scores[j].abs_quality() > scores[i].abs_quality() * mask_factor
              && sign (scores[j]) != sign (scores[i]))
            mask = true;
        }
      if (!mask)
        kept.push_back (scores[i]);
    }
  scores.swap (kept);
}

/* reference syncfinder.cc:364-383 */
void
SyncFinder::select_threshold_and_n_best (std::vector<SearchScore>& scores, double threshold)
{
  std::sort (scores.begin(), scores.end(), [] (const SearchScore& a, const SearchScore& b) { return a.abs_quality() > b.abs_quality(); });
  int i = 0;
  while (i < int (scores.size()) && scores[i].abs_quality() > threshold)
    i++;
  if (i >= params().get_n_best)
    scores.resize (i);
  else if (int (scores.size()) > params().get_n_best)
    scores.resize (params().get_n_best);
}

/* reference syncfinder.cc:385-391 */
void
SyncFinder::select_truncate_n (std::vector<SearchScore>& scores, size_t n)
{
  std::sort (scores.begin(), scores.end(), [] (const SearchScore& a, const SearchScore& b) { return a.abs_quality() > b.abs_quality(); });
  if (scores.size() > n)
    scores.resize (n);
}

/* reference syncfinder.cc:393-458: every candidate is re-scored on the fine grid
 * [index - 256, index + 256] step 8 using only the sync frames */
int
SyncFinder::search_refine (KeyTables *kt, const DeviceWav& wav, Mode mode, std::vector<SearchScore>& scores)
{
  SearchJob job;
  job.candidates = scores;
  if (int rc = refine_launch (kt, wav, mode, job))
    return rc;
  return refine_finish (job, scores);
}

namespace {
constexpr int REFINE_TP = 72;            // padded fine-offset axis (<= 65 used)
constexpr int REFINE_TP_LINES = 64;      // ... of the rows K4s writes in forms 4 / 5 (fine offset 64 apart: SyncDbArgs::tail)
constexpr int REFINE_QS = 128;
}

int
SyncFinder::refine_launch (KeyTables *kt, const DeviceWav& wav, Mode mode, SearchJob& job)
{
  const int clip = mode == Mode::CLIP;
  const auto& sync = kt->sync[clip];
  const int NW = kt->slices ? kt->want_rows_of_slices() : int (sync.want_list.size());
  const size_t n_cand = job.candidates.size();
  job.refined.clear();
  job.batch_pending = false;
  if (!n_cand)
    return 0;
  const bool gathered = wav.n_channels <= 2;
  const int row_values = gathered ? 2 * int (Params::bands_per_frame) : Params::n_bands;
  // (K4s forms 4 / 5: rows of 64 fine offsets = two whole cache lines, the 65th value of every row in a compact array behind the rows)
  const bool tail_layout = gathered && awmk::sliding_rows_have_tail (wav.n_channels);
  const size_t per_cand = size_t (NW) * row_values * (tail_layout ? REFINE_TP_LINES : REFINE_TP);
  const size_t tail_per_cand = tail_layout ? size_t (NW) * row_values : 0;
  // <= 3 GiB of dB rows at a time (a group of clips: 12 GiB, every further batch is one more wait for the whole group)
  size_t batch = std::max<size_t> (1, (size_t (job.slice_frames ? 12 : 3) << 30) / (per_cand * sizeof (float)));
  batch = std::min (batch, n_cand);
  if (int rc = m_lane->ws_refine.reserve (batch * (per_cand + tail_per_cand) * sizeof (float))) return rc;
  if (int rc = m_lane->ws_refine_have.reserve (batch * NW * REFINE_TP)) return rc;
  if (int rc = m_lane->ws_q.reserve (batch * (REFINE_QS * sizeof (double) + awmk::GATHERED_SCRATCH_BYTES_PER_PLANE))) return rc;    // qualities, then K5g's chain sums
  if (int rc = m_lane->ws_idx.reserve (batch * NW * (sizeof (long long) + sizeof (int)) + 2 * batch * sizeof (int))) return rc;
  for (size_t c0 = 0; c0 < n_cand; c0 += batch)
    {
      if (job.batch_pending)
        if (int rc = refine_batch_finish (job))
          return rc;
      if (int rc = refine_batch_launch (kt, wav, mode, job, c0, std::min (batch, n_cand - c0), batch))
        return rc;
    }
  return 0;
}

int
SyncFinder::refine_batch_launch (KeyTables *kt, const DeviceWav& wav, Mode mode, SearchJob& job, size_t c0, size_t nb, size_t batch)
{
  const int clip = mode == Mode::CLIP;
  const auto& sync = kt->sync[clip];
  const int NW = kt->slices ? kt->want_rows_of_slices() : int (sync.want_list.size());
  const long long total = total_frames (mode);
  const int QS = REFINE_QS;
  hipStream_t st = m_lane->stream;
  const bool gathered = wav.n_channels <= 2;
  const int row_values = gathered ? 2 * int (Params::bands_per_frame) : Params::n_bands;
  const bool tail_layout = gathered && awmk::sliding_rows_have_tail (wav.n_channels);
  const int TP = tail_layout ? REFINE_TP_LINES : REFINE_TP;           // row length of the dB rows
  const int HP = REFINE_TP;                                           // ... of the have flags
  const size_t per_cand = size_t (NW) * row_values * TP;
  const size_t tail_per_cand = tail_layout ? size_t (NW) * row_values : 0;
  float *const d_tail = tail_layout ? m_lane->ws_refine.as<float>() + batch * per_cand : nullptr;   // (behind the rows of a full batch)

  // stream tables: [nb * NW] long long base, [nb * NW] int count, [nb] int lanes, [nb] int slice -- one page-locked block, one copy
  const size_t in_bytes = nb * NW * (sizeof (long long) + sizeof (int)) + 2 * nb * sizeof (int);
  const bool slices = job.slice_frames > 0;
  const long long span_frames = slices ? (long long) job.slice_frames : (long long) wav.n_frames;
  PinnedBuffer& pin_in = m_lane->pin_refine_in[job.slot];
  PinnedBuffer& pin_q = m_lane->pin_refine_q[job.slot];
  if (int rc = pin_in.reserve (in_bytes)) return rc;
  if (int rc = pin_q.reserve (nb * QS * sizeof (double))) return rc;
  auto *stream_base = pin_in.as<long long>();
  int *stream_count = reinterpret_cast<int *> (stream_base + nb * NW);
  int *lanes = stream_count + nb * NW;
  int *slice_of = lanes + nb;
  job.lane_count.assign (nb, 0);
  job.starts.assign (nb, 0);
  job.c0 = c0;
  job.nb = nb;
  int max_count = 0;
  long long n_items = 0;
  double db_bytes = 0;
  for (size_t c = 0; c < nb; c++)
    {
      const SearchScore& s = job.candidates[c0 + c];
      const int start = std::max (int (s.index) - Params::sync_search_step, 0);
      const int end = int (s.index) + Params::sync_search_step;
      // fine offsets for which sync_fft does not read past the end (syncfinder.cc:566-568)
      const long long limit = span_frames - total * Params::frame_size;
      const long long slice0 = slices ? (long long) job.cand_slice[c0 + c] * span_frames : 0;
      slice_of[c] = slices ? job.cand_slice[c0 + c] : 0;
      int count = 0;
      for (int fine = start; fine <= end; fine += Params::sync_search_fine)
        if (fine <= limit)
          count++;
      job.starts[c] = start;
      job.lane_count[c] = lanes[c] = count;
      max_count = std::max (max_count, count);
      n_items += (long long) count * NW;
      // per (candidate, sync frame): a (1024 + 8 (T - 1))-sample window read once, T rows of dB values written
      if (count)
        db_bytes += double (NW) * ((1024.0 + 8.0 * (count - 1)) * 4 * wav.n_channels + 4.0 * row_values * count);
      const int *want_list = kt->slices ? kt->want_of_slice (job.cand_slice[c0 + c]) : sync.want_list.data();
      for (int w = 0; w < NW; w++)
        {
          stream_base[c * NW + w] = slice0 + start + (long long) want_list[w] * Params::frame_size;
          stream_count[c * NW + w] = count;
        }
    }
  auto *d_base = m_lane->ws_idx.as<long long>();
  int *d_count = reinterpret_cast<int *> (d_base + nb * NW);
  int *d_lanes = d_count + nb * NW;
  int *d_slice_of = d_lanes + nb;
  AWM_HIP_CHECK (hipMemcpyAsync (d_base, stream_base, in_bytes, hipMemcpyHostToDevice, st));

  double *q = pin_q.as<double>();
  std::fill (q, q + nb * QS, 0.0);
  if (max_count > 0)
    {
      awmk::SyncDbArgs da {};
      da.pcm = wav.data;
      da.n_frames = wav.n_frames;
      da.n_channels = wav.n_channels;
      da.per_channel = 0;
      da.stream_base = d_base;
      da.stream_count = d_count;
      da.count0 = max_count;
      da.n_streams = (long long) nb * NW;
      da.hop = Params::sync_search_fine;
      da.out = m_lane->ws_refine.as<float>();
      da.out_stream_stride = (long long) row_values * TP;
      if (gathered)
        {
          da.row_perm = sync.refine_perm.as<int>();
          da.band_pos = sync.refine_pos.as<unsigned char>();
          da.rows_per_plane = NW;
        }
      da.ld = TP;
      da.tail = d_tail;
      da.tail_stream_stride = row_values;
      da.have = m_lane->ws_refine_have.as<char>();
      da.have_stream_stride = HP;
      da.first = (long long) m_first;
      da.last = (long long) m_last;
      if (slices)
        {
          da.stream_range = job.slice_range;
          da.range_index = d_slice_of;
          da.range_div = NW;
          da.tables_per_slice = kt->slices ? 1 : 0;          // one key per clip: refine_perm / refine_pos of the candidate's slice
        }
      da.tile_frames = TP;
      {
        ProfScope ps (m_ctx, PROF_REFINE_DB, (slices ? m_lane->prof_live_fraction : 1.0) * db_bytes, st);
        if (gathered)
          AWM_HIP_CHECK (awmk::launch_sync_db_sliding (st, m_ctx->tabs, da));       // K4s: sliding DFT over the fine offsets
        else
          AWM_HIP_CHECK (awmk::launch_sync_db (st, m_ctx->tabs, da));               // K4: one FFT per fine offset
      }
      if (gathered)
        {
          awmk::GatheredScanArgs ga {};
          ga.db = m_lane->ws_refine.as<float>();
          ga.have = clip ? m_lane->ws_refine_have.as<char>() : nullptr;
          ga.plane_stride = (long long) per_cand;
          ga.have_plane_stride = (long long) NW * HP;
          ga.ld = TP;
          ga.have_ld = HP;
          ga.tail = d_tail;
          ga.tail_plane_stride = (long long) tail_per_cand;
          ga.rows_per_bit = sync.host.rows_per_bit;
          ga.n_lanes = max_count;
          ga.lane_count = d_lanes;
          ga.n_planes = (long long) nb;
          ga.min_delta = std::min (params().water_delta, 0.080);
          ga.quality = m_lane->ws_q.as<double>();
          ga.q_stride = QS;
          ga.chain_mag = reinterpret_cast<float *> (ga.quality + batch * QS);
          ga.chain_n = reinterpret_cast<int *> (ga.chain_mag + batch * 12 * 128);
          ProfScope ps (m_ctx, PROF_REFINE_SCAN, (slices ? m_lane->prof_live_fraction : 1.0) * double (n_items) * 4.0 * row_values, st);
          AWM_HIP_CHECK (awmk::launch_sync_scan_gathered (st, ga));
        }
      else
        {
          awmk::SyncScanArgs sa {};
          sa.db = m_lane->ws_refine.as<float>();
          sa.have = clip ? m_lane->ws_refine_have.as<char>() : nullptr;
          sa.plane_stride = (long long) per_cand;
          sa.have_plane_stride = (long long) NW * HP;
          sa.row_stride = (long long) Params::n_bands * TP;
          sa.band_stride = TP;
          sa.have_row_stride = HP;
          sa.n_lanes = max_count;
          sa.lane_count = d_lanes;
          sa.n_planes = (long long) nb;
          sa.min_delta = std::min (params().water_delta, 0.080);
          sa.quality = m_lane->ws_q.as<double>();
          sa.q_stride = QS;
          sa.table.packed = sync.packed_refine.as<int>();
          sa.table.rows_per_bit = sync.host.rows_per_bit;
          ProfScope ps (m_ctx, PROF_REFINE_SCAN, double (n_items) * 324.0, st);
          AWM_HIP_CHECK (awmk::launch_sync_scan (st, sa));
        }
      AWM_HIP_CHECK (hipMemcpyAsync (q, m_lane->ws_q.ptr, nb * QS * sizeof (double), hipMemcpyDeviceToHost, st));
    }
  hipEvent_t& ev = m_lane->ev_refine[job.slot];
  if (!ev)
    AWM_HIP_CHECK (hipEventCreateWithFlags (&ev, hipEventDisableTiming));
  AWM_HIP_CHECK (hipEventRecord (ev, st));
  job.batch_pending = true;
  return 0;
}

int
SyncFinder::refine_batch_finish (SearchJob& job)
{
  if (!job.batch_pending)
    return 0;
  AWM_HIP_CHECK (event_wait (m_lane->ev_refine[job.slot]));
  job.batch_pending = false;
  const double *q = m_lane->pin_refine_q[job.slot].as<double>();
  for (size_t c = 0; c < job.nb; c++)
    {
      const SearchScore& s = job.candidates[job.c0 + c];
      double best_quality = s.raw_quality;
      size_t best_index = s.index;
      for (int t = 0; t < job.lane_count[c]; t++)
        {
          const double qt = q[c * REFINE_QS + t];
          if (std::fabs (qt - s.local_mean) > std::fabs (best_quality - s.local_mean))
            {
              best_quality = qt;
              best_index = job.starts[c] + t * Params::sync_search_fine;
            }
        }
      job.refined.push_back ({ best_index, best_quality, s.local_mean });
    }
  return 0;
}

int
SyncFinder::refine_finish (SearchJob& job, std::vector<SearchScore>& scores)
{
  if (int rc = refine_batch_finish (job))
    return rc;
  std::stable_sort (job.refined.begin(), job.refined.end(), [] (const SearchScore& a, const SearchScore& b) { return a.index < b.index; });
  scores.swap (job.refined);
  job.refined.clear();
  return 0;
}

int
SyncFinder::prepare (const DeviceWav& wav, Mode mode)
{
  if (int rc = prepare_launch (wav, mode))
    return rc;
  return prepare_finish();
}

/* reference syncfinder.cc:487-558 */
int
SyncFinder::search (const Key& key, const DeviceWav& wav, Mode mode, std::vector<Score>& out, bool db_ready)
{
  SearchJob job;
  if (int rc = search_launch (key, wav, mode, job, db_ready))
    return rc;
  return search_finish (job, out);
}

int
SyncFinder::search_launch (const Key& key, const DeviceWav& wav, Mode mode, SearchJob& job, bool db_ready)
{
  if (int rc = approx_launch (key, wav, mode, job, /* prepared */ db_ready, db_ready))
    return rc;
  return select_refine (job);
}

int
SyncFinder::approx_launch (const Key& key, const DeviceWav& wav, Mode mode, SearchJob& job, bool prepared, bool db_ready)
{
  job.out.clear();
  job.done = true;
  job.batch_pending = false;
  job.select_pending = false;
  KeyTables *kt = m_ctx->get_key_tables (key);
  if (!kt)
    return AWM_ERR_HIP;
  if (params().test_no_sync)
    {
      if (mode == Mode::BLOCK)       // fake_sync, reference syncfinder.cc:460-485
        {
          const size_t expect0 = Params::frames_pad_start * Params::frame_size;
          const size_t step = mark_block_frame_count() * Params::frame_size;
          const size_t end = (wav.n_frames / Params::frame_size) * Params::frame_size;
          int ab = 0;
          for (size_t idx = expect0; idx + step < end; idx += step)
            job.out.push_back ({ idx, 1.0, (ab++ & 1) ? ConvBlockType::b : ConvBlockType::a });
        }
      return 0;
    }
  if (!prepared)
    if (int rc = prepare (wav, mode))
      return rc;
  job.kt = kt;
  job.wav = wav;
  job.mode = mode;
  job.n_scores = 0;
  if (int rc = approx_device (kt, wav, mode, job.n_scores, db_ready))
    return rc;
  job.speculate_n_best = mode == Mode::CLIP;            // clips hold one or two sync peaks: the n_best fallback is the rule
  if (int rc = select_launch (job.n_scores, params().sync_threshold2 * 0.75, job.speculate_n_best))
    return rc;
  job.done = false;
  job.select_pending = true;
  return 0;
}

int
SyncFinder::select_refine (SearchJob& job)
{
  if (job.done || !job.select_pending)
    return 0;
  job.select_pending = false;
  if (int rc = select_finish (job.n_scores, params().sync_threshold2 * 0.75, job.candidates, job.speculate_n_best))
    return rc;
  if (job.mode == Mode::CLIP)
    select_truncate_n (job.candidates, std::max (params().get_n_best, 5));
  return refine_launch (job.kt, job.wav, job.mode, job);
}

int
SyncFinder::search_finish (SearchJob& job, std::vector<Score>& out)
{
  out.clear();
  if (job.done)
    {
      out = job.out;
      return 0;
    }
  std::vector<SearchScore> scores;
  if (int rc = refine_finish (job, scores))
    return rc;
  job.done = true;
  select_threshold_and_n_best (scores, params().sync_threshold2);
  std::sort (scores.begin(), scores.end(), [] (const SearchScore& a, const SearchScore& b) { return a.index < b.index; });
  for (const auto& s : scores)
    {
      const double q = s.raw_quality - s.local_mean;
      out.push_back ({ s.index, std::fabs (q), q > 0 ? ConvBlockType::a : ConvBlockType::b });
    }
  return 0;
}

/* refine_finish + the end of search_finish for refined scores that were collected elsewhere (candidate order) */
void
SyncFinder::finish_scores (std::vector<SearchScore> scores, std::vector<Score>& out)
{
  out.clear();
  std::stable_sort (scores.begin(), scores.end(), [] (const SearchScore& a, const SearchScore& b) { return a.index < b.index; });
  select_threshold_and_n_best (scores, params().sync_threshold2);
  std::sort (scores.begin(), scores.end(), [] (const SearchScore& a, const SearchScore& b) { return a.index < b.index; });
  for (const auto& s : scores)
    {
      const double q = s.raw_quality - s.local_mean;
      out.push_back ({ s.index, std::fabs (q), q > 0 ? ConvBlockType::a : ConvBlockType::b });
    }
}

/* ---- CLIP search for a group of padded clips (syncfinder.hh GroupJob) -------------------------------------------------------- */

namespace {
constexpr unsigned int GROUP_HEAD = 32;      // peaks above the threshold fetched per slice (a clip holds one or two sync peaks per block)
constexpr int GROUP_TOPK_SLICES = 4;
size_t align256 (size_t n) { return (n + 255) & ~size_t (255); }
struct GroupLayout
{
  size_t off_th, off_top, bytes;
  GroupLayout (int n_slices, int k)
  {
    off_th = align256 (size_t (n_slices) * 2 * sizeof (unsigned int));
    off_top = off_th + size_t (n_slices) * GROUP_HEAD * sizeof (awmk::PeakOut);
    bytes = off_top + size_t (n_slices) * GROUP_TOPK_SLICES * k * sizeof (awmk::PeakOut);
  }
};
}

int
SyncFinder::group_approx_launch (KeyTables *kt, const DeviceWav& group, int n_slices, const long long *d_range, GroupJob& gj, bool db_ready)
{
  gj.n_slices = n_slices;
  gj.kt = kt;
  gj.group = group;
  gj.slice_frames = n_slices > 0 ? group.n_frames / n_slices : 0;
  gj.n_scores = 0;
  gj.fallback.assign (std::max (n_slices, 0), 0);
  gj.refine = SearchJob();
  gj.refine.slice_frames = gj.slice_frames;
  gj.refine.slice_range = d_range;
  const Mode mode = Mode::CLIP;
  const long long frame_count = gj.slice_frames / Params::frame_size;
  const long long n_db = frame_count - 1;          // as approx_device, per slice
  const long long S = n_db - total_frames (mode);
  const int k = params().get_n_best + 1;
  if (n_slices <= 0 || n_db <= 0 || S <= 0 || params().test_no_sync)       // (--test-no-sync: CLIP mode finds nothing, see approx_launch)
    return 0;
  if (gj.slice_frames % Params::frame_size || k > TOPK_MAX)
    {
      gj.fallback.assign (n_slices, 1);            // (slices are whole frames by construction; an n_best beyond the top-k kernel)
      return 0;
    }
  hipStream_t st = m_lane->stream;
  const int n_shifts = Params::frame_size / Params::sync_search_step;
  const long long ld = (n_db + 63) & ~63LL;
  const long long plane = ld * Params::n_bands;
  const long long q_stride = (S + 63) & ~63LL;
  const long long n_planes = (long long) n_slices * n_shifts;
  const long long n_scores = (long long) n_shifts * S;
  if (int rc = m_lane->ws_db.reserve (size_t (n_planes) * plane * sizeof (float))) return rc;
  if (int rc = m_lane->ws_have.reserve (size_t (n_planes) * ld)) return rc;
  if (int rc = m_lane->ws_q.reserve (size_t (n_planes) * q_stride * sizeof (double))) return rc;
  if (int rc = m_lane->ws_raw.reserve (size_t (n_slices) * n_scores * sizeof (double))) return rc;
  if (int rc = m_lane->ws_mean.reserve (size_t (n_slices) * n_scores * sizeof (double))) return rc;

  awmk::SyncDbArgs da {};
  da.pcm = group.data;
  da.n_frames = group.n_frames;
  da.n_channels = group.n_channels;
  da.base0 = 0;
  da.base_stride = Params::sync_search_step;
  da.streams_per_slice = n_shifts;
  da.slice_stride = (long long) gj.slice_frames;
  da.stream_range = d_range;
  da.range_div = n_shifts;
  da.count0 = int (n_db);
  da.n_streams = n_planes;
  da.hop = Params::frame_size;
  da.out = m_lane->ws_db.as<float>();
  da.out_stream_stride = plane;
  da.ld = ld;
  da.have = m_lane->ws_have.as<char>();
  da.have_stream_stride = ld;
  da.tile_frames = 32;
  if (!db_ready)                                     // (else: another key of the same `get` left the group's matrices in the workspace)
    {
      // (only the frames that carry samples are transformed and written: prof_live_fraction of the padded slices)
      ProfScope ps (m_ctx, PROF_SYNC_DB, m_lane->prof_live_fraction * (double (n_slices) * n_db * 4096.0 * group.n_channels + double (n_planes) * n_db * 324.0), st);
      AWM_HIP_CHECK (awmk::launch_sync_db (st, m_ctx->tabs, da));
    }
  awmk::SyncScanArgs sa {};
  sa.db = m_lane->ws_db.as<float>();
  sa.have = m_lane->ws_have.as<char>();
  sa.have_is_run = 1;
  sa.plane_stride = plane;
  sa.have_plane_stride = ld;
  sa.row_stride = 1;
  sa.band_stride = ld;
  sa.have_row_stride = 1;
  sa.n_lanes = S;
  sa.n_planes = n_planes;
  sa.min_delta = std::min (params().water_delta, 0.080);
  sa.quality = m_lane->ws_q.as<double>();
  sa.q_stride = q_stride;
  sa.table.packed = kt->sync[1].packed_approx.as<int>();
  sa.table.rows_per_bit = kt->sync[1].host.rows_per_bit;
  sa.table.chains = kt->sync[1].chains_approx.as<unsigned>();
  if (kt->slices)
    {
      // one key per clip: the chain tables of the slices lie back to back, the planes of a slice are its shifts
      sa.table.chains_slice_stride = (long long) 12 * sa.table.rows_per_bit * 8;
      sa.table.planes_per_slice = n_shifts;
      sa.table.row_frames = kt->sync[1].row_frames.as<int>();
    }
  {
    ProfScope ps (m_ctx, PROF_SYNC_SCAN, m_lane->prof_live_fraction * double (n_planes) * n_db * 324.0 + double (n_planes) * S * 8.0, st);
    AWM_HIP_CHECK (awmk::launch_sync_scan_window (st, sa, total_frames (mode)));
  }
  {
    ProfScope ps (m_ctx, PROF_LOCAL_MEAN, double (n_slices) * n_scores * 56.0, st);
    AWM_HIP_CHECK (awmk::launch_local_mean (st, m_lane->ws_q.as<double>(), q_stride, S, m_lane->ws_raw.as<double>(), m_lane->ws_mean.as<double>(),
                                            n_slices));
    // selection (select_launch with the n_best fallback queued right away), every slice with its own counters and lists:
    // ws_misc = [slices][2] counters | [slices][GROUP_HEAD] above the threshold | [slices][4][k] largest unmasked maxima
    const GroupLayout lay (n_slices, k);
    if (int rc = m_lane->ws_misc.reserve (lay.bytes)) return rc;
    if (int rc = m_lane->pin_peaks.reserve (lay.bytes)) return rc;
    if (int rc = m_lane->ws_refine.reserve (size_t (n_slices) * n_scores * sizeof (awmk::PeakOut))) return rc;
    char *base = m_lane->ws_misc.as<char>();
    auto *d_count = reinterpret_cast<unsigned int *> (base);
    auto *d_th = reinterpret_cast<awmk::PeakOut *> (base + lay.off_th);
    auto *d_top = reinterpret_cast<awmk::PeakOut *> (base + lay.off_top);
    auto *d_all = m_lane->ws_refine.as<awmk::PeakOut>();
    const double threshold = params().sync_threshold2 * 0.75;
    AWM_HIP_CHECK (hipMemsetAsync (d_count, 0, size_t (n_slices) * 2 * sizeof (unsigned int), st));
    AWM_HIP_CHECK (awmk::launch_peak_select_slices (st, m_lane->ws_raw.as<double>(), m_lane->ws_mean.as<double>(), n_scores, threshold,
                                                    d_count, 2, d_th, GROUP_HEAD, n_slices));
    AWM_HIP_CHECK (awmk::launch_peak_select_slices (st, m_lane->ws_raw.as<double>(), m_lane->ws_mean.as<double>(), n_scores, -1.0,
                                                    d_count + 1, 2, d_all, unsigned (n_scores), n_slices));
    AWM_HIP_CHECK (awmk::launch_peak_topk_lists (st, d_all, d_count + 1, 2, unsigned (n_scores), d_top, k, GROUP_TOPK_SLICES, n_slices));
    AWM_HIP_CHECK (hipMemcpyAsync (m_lane->pin_peaks.ptr, base, lay.bytes, hipMemcpyDeviceToHost, st));
  }
  gj.n_scores = n_scores;
  return 0;
}

int
SyncFinder::group_select_refine (GroupJob& gj)
{
  if (gj.n_scores <= 0)
    return 0;
  const int k = params().get_n_best + 1;
  const GroupLayout lay (gj.n_slices, k);
  AWM_HIP_CHECK (stream_wait (m_lane->stream));
  const char *pin = m_lane->pin_peaks.as<char>();
  const auto *count = reinterpret_cast<const unsigned int *> (pin);
  const auto *th = reinterpret_cast<const awmk::PeakOut *> (pin + lay.off_th);
  const auto *top = reinterpret_cast<const awmk::PeakOut *> (pin + lay.off_top);
  const double threshold = params().sync_threshold2 * 0.75;
  SearchJob& job = gj.refine;
  for (int i = 0; i < gj.n_slices; i++)
    {
      if (gj.fallback[i])
        continue;
      std::vector<SearchScore> cands;                      // select_finish for this slice
      const unsigned int c_th = count[2 * i];
      if (int (c_th) >= params().get_n_best && c_th <= GROUP_HEAD)
        scores_from_threshold_list (th + size_t (i) * GROUP_HEAD, c_th, cands);
      else if (int (c_th) < params().get_n_best)
        {
          const awmk::PeakOut *t0 = top + size_t (i) * GROUP_TOPK_SLICES * k;
          if (!scores_from_topk (std::vector<awmk::PeakOut> (t0, t0 + size_t (GROUP_TOPK_SLICES) * k), threshold, cands))
            gj.fallback[i] = 1;
        }
      else
        gj.fallback[i] = 1;
      if (gj.fallback[i])
        continue;
      select_truncate_n (cands, std::max (params().get_n_best, 5));     // select_refine, Mode::CLIP
      for (const auto& c : cands)
        {
          job.candidates.push_back (c);
          job.cand_slice.push_back (i);
        }
    }
  return refine_launch (gj.kt, gj.group, Mode::CLIP, job);
}

int
SyncFinder::group_finish (GroupJob& gj, std::vector<std::vector<Score>>& out)
{
  out.assign (std::max (gj.n_slices, 0), {});
  if (gj.n_scores <= 0)
    return 0;
  SearchJob& job = gj.refine;
  if (int rc = refine_batch_finish (job))
    return rc;
  // job.refined is in candidate order; per slice: refine_finish + search_finish
  std::vector<std::vector<SearchScore>> per_slice (gj.n_slices);
  for (size_t c = 0; c < job.refined.size(); c++)
    per_slice[job.cand_slice[c]].push_back (job.refined[c]);
  for (int i = 0; i < gj.n_slices; i++)
    {
      auto& scores = per_slice[i];
      std::stable_sort (scores.begin(), scores.end(), [] (const SearchScore& a, const SearchScore& b) { return a.index < b.index; });
      select_threshold_and_n_best (scores, params().sync_threshold2);
      std::sort (scores.begin(), scores.end(), [] (const SearchScore& a, const SearchScore& b) { return a.index < b.index; });
      for (const auto& s : scores)
        {
          const double q = s.raw_quality - s.local_mean;
          out[i].push_back ({ s.index, std::fabs (q), q > 0 ? ConvBlockType::a : ConvBlockType::b });
        }
    }
  return 0;
}

} // namespace awm
