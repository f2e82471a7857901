/* oracle/ref_shim/ref_capi_speed.cc -- wrappers around the reference's speed detection.
 * Includes /root/reference/src/wmspeed.cc as-is (file-local SpeedSync, SpeedSearch, get_best_clip_location,
 * select_n_best_scores, score_smooth_find_best) and opens the private sections for inspection only.
 * zita-resampler is replaced by the restatement in oracle/zita_restated.h (see ref_shim/include/zita-resampler).
 * TEST INFRASTRUCTURE ONLY. */
#include <vector>
#include <string>
#include <memory>
#include <complex>
#include <algorithm>
#include <mutex>
#include <map>
#include <functional>
#include <random>
#include <array>
#include <thread>
#include <condition_variable>
#include <atomic>
#include <queue>
#include <unistd.h>

#define private public
#define protected public
#include "wmspeed.cc"        /* the reference translation unit, unmodified */
#undef private
#undef protected
#include "ref_capi.h"

Key awm_ref_make_key (const uint8_t k[16]);

extern "C" {

void
ref_set_speed_params (int detect_speed, int patient, double try_speed)
{
  Params::detect_speed = detect_speed != 0;
  Params::detect_speed_patient = patient != 0;
  Params::try_speed = try_speed;
}

size_t
ref_resample_ratio (const float *samples, size_t n_frames, int n_channels, int rate, double ratio, int new_rate,
                    double max_in_seconds, size_t max_out_frames, float *out)
{
  WavData wav (std::vector<float> (samples, samples + n_frames * n_channels), n_channels, rate, 16);
  WavData res = resample_ratio_truncate (wav, ratio, new_rate, max_in_seconds);
  const size_t n = std::min (res.n_frames(), max_out_frames);
  std::copy (res.samples().begin(), res.samples().begin() + n * n_channels, out);
  return res.n_frames();
}

double
ref_speed_clip_location (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate,
                         double seconds, int candidates)
{
  WavData wav (std::vector<float> (samples, samples + n_values), n_channels, rate, 16);
  return get_best_clip_location (awm_ref_make_key (key), wav, seconds, candidates);
}

/* SpeedSync::prepare_mags for one centre speed on the clip get_speed_clip (location, wav, seconds * 1.3); out[row][510][2] */
int
ref_speed_mags (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate,
                double clip_location, double center, double seconds, size_t max_rows, float *out)
{
  WavData wav (std::vector<float> (samples, samples + n_values), n_channels, rate, 16);
  WavData clip = get_speed_clip (clip_location, wav, seconds * 1.3);
  SpeedSync ss (awm_ref_make_key (key), clip, center);
  SpeedScanParams sp;
  sp.seconds = seconds;
  ss.prepare_mags (sp);
  const int rows = ss.sync_matrix.rows();
  const int cols = int (ss.sync_bits.size());
  for (int r = 0; r < rows && size_t (r) < max_rows; r++)
    for (int c = 0; c < cols; c++)
      {
        out[(size_t (r) * cols + c) * 2] = ss.sync_matrix (r, c).umag;
        out[(size_t (r) * cols + c) * 2 + 1] = ss.sync_matrix (r, c).dmag;
      }
  return rows;
}

/* one run_search pass (wmspeed.cc:683-719) for one key: scores sorted by speed */
int
ref_speed_scan (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate,
                double clip_location, double seconds, double step, int n_steps, int n_center_steps,
                const double *speeds, int n_speeds, size_t max_out, double *out_speed, double *out_quality)
{
  WavData wav (std::vector<float> (samples, samples + n_values), n_channels, rate, 16);
  SpeedScanParams sp;
  sp.seconds = seconds;
  sp.step = step;
  sp.n_steps = n_steps;
  sp.n_center_steps = n_center_steps;
  SpeedSearch search (wav, clip_location);
  auto jobs = search.get_jobs (awm_ref_make_key (key), sp, std::vector<double> (speeds, speeds + n_speeds));
  ThreadPool pool;
  for (auto& j : jobs)
    {
      j.prepare_job();
      for (auto& s : j.search_jobs)
        pool.add_job (s);
      pool.wait_all();
      j.free_memory();
    }
  auto scores = search.get_results();
  std::sort (scores.begin(), scores.end(), [] (auto a, auto b) { return a.speed < b.speed; });
  for (size_t i = 0; i < scores.size() && i < max_out; i++)
    {
      out_speed[i] = scores[i].speed;
      out_quality[i] = scores[i].quality;
    }
  return int (scores.size());
}

/* select_n_best_scores (wmspeed.cc:494-531), in place; returns the new count */
int
ref_speed_select_n_best (double *speed, double *quality, int count, int n)
{
  std::vector<SpeedSync::Score> scores (count);
  for (int i = 0; i < count; i++)
    {
      scores[i].speed = speed[i];
      scores[i].quality = quality[i];
    }
  select_n_best_scores (scores, n);
  for (size_t i = 0; i < scores.size(); i++)
    {
      speed[i] = scores[i].speed;
      quality[i] = scores[i].quality;
    }
  return int (scores.size());
}

double
ref_speed_smooth_best (const double *speed, const double *quality, int count, double step, double distance)
{
  std::vector<SpeedSync::Score> scores (count);
  for (int i = 0; i < count; i++)
    {
      scores[i].speed = speed[i];
      scores[i].quality = quality[i];
    }
  return score_smooth_find_best (scores, step, distance);
}

/* detect_speed (wmspeed.cc:622-781) for one key; returns the number of results (0 or 1) */
int
ref_detect_speed (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate, int patient,
                  double *speed_out)
{
  WavData wav (std::vector<float> (samples, samples + n_values), n_channels, rate, 16);
  const bool old_patient = Params::detect_speed_patient;
  Params::detect_speed_patient = patient != 0;
  auto res = detect_speed ({ awm_ref_make_key (key) }, wav, false);
  Params::detect_speed_patient = old_patient;
  if (!res.empty())
    *speed_out = res[0].speed;
  return int (res.size());
}

} /* extern "C" */
