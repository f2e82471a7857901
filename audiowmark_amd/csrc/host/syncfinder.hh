// SyncFinder -- GPU-resident counterpart of reference src/syncfinder.{hh,cc}.
// Same decisions as the reference (search_approx on 4 shifts, local mean, local maxima,
// false-positive masking, threshold / n_best selection, search_refine on a +-256 / step 8
// grid); the heavy loops run as HIP kernels (K4 sync_db, K5 sync_scan, K5b local_mean).
#pragma once
#include "context.hh"

namespace awm {

// PCM resident in HBM (the device-side WavData, reference wavdata.hh:27-74)
struct DeviceWav
{
  const float *data = nullptr;
  size_t       n_frames = 0;      // samples per channel
  int          n_channels = 0;
  int          sample_rate = 44100;
  size_t n_values() const { return n_frames * n_channels; }
};

class SyncFinder
{
public:
  enum class Mode { BLOCK, CLIP };
  struct Score
  {
    size_t        index;
    double        quality;
    ConvBlockType block_type;
  };
  struct SearchScore
  {
    size_t index;
    double raw_quality;
    double local_mean;
    double abs_quality() const { return std::fabs (raw_quality - local_mean); }
  };
  static constexpr int local_mean_distance = 20;

  explicit SyncFinder (awm_ctx *ctx, WorkLane *lane = nullptr) : m_ctx (ctx), m_lane (lane ? lane : ctx) {}
  /* recorded on the lane's stream right after the dB kernel of the approximate search (K4) has been queued: the scheduler of `get`
   * lets the next chunk start there (wmget.cc: block_decoder_run, phase offset between the lanes) */
  hipEvent_t after_db_event = nullptr;
  hipEvent_t after_scan_event = nullptr;       // the same after the scan (K5w) has been queued (measurement variant)

  // db_ready: the dB matrices (and the non-silent range) of a previous search of the SAME material by this object are still in the
  // lane's workspace -- the next key of a multi-key `get` shares them (reference syncfinder.cc:171-256)
  int search (const Key& key, const DeviceWav& wav, Mode mode, std::vector<Score>& out, bool db_ready = false);
  int prepare (const DeviceWav& wav, Mode mode);     // silence scan (CLIP) / full range (BLOCK)
  // the same without waiting for the device; [scan_lo, scan_hi) = values that can be non-zero at all (a caller that padded
  // the buffer itself knows where the data is: the silence scan then skips the padding)
  int prepare_launch (const DeviceWav& wav, Mode mode, size_t scan_lo = 0, size_t scan_hi = size_t (-1));
  int prepare_finish();                                   // ... and the wait
  int search_approx (KeyTables *kt, const DeviceWav& wav, Mode mode, std::vector<SearchScore>& out);
  // kernels of search_approx only: scores stay on the device (ws_raw / ws_mean, index order); n_scores = 4 * start frames
  // scores_only: stop after the scan (raw qualities [shift][q_stride] in ws_q, q_stride = start frames rounded up to 64): the multi-GPU
  // protocol computes a chunk's scores in parts and runs the local mean on the assembled list (scores_loaded)
  int approx_device (KeyTables *kt, const DeviceWav& wav, Mode mode, long long& n_scores, bool db_ready = false, bool scores_only = false);
  // ws_q holds the raw qualities of n_start_frames start frames x 4 shifts (assembled by the caller): local mean -> ws_raw / ws_mean
  int scores_loaded (long long n_start_frames);
  // local maxima + mask + threshold (+ n_best fallback): the candidate list search_refine works on
  int select_candidates (long long n_scores, double threshold, std::vector<SearchScore>& out);
  int select_launch (long long n_scores, double threshold, bool speculate_n_best = false);   // device part of it, not waited for
  int select_finish (long long n_scores, double threshold, std::vector<SearchScore>& out, bool speculated = false);
  int search_refine (KeyTables *kt, const DeviceWav& wav, Mode mode, std::vector<SearchScore>& scores);

  /* search() in two halves, so that a caller with several chunks can issue the next chunk's search while the
   * refinement of this one is still running on the device (the host never leaves the GPU idle between chunks):
   *   search_launch (chunk i + 1, job[(i + 1) & 1]);  search_finish (job[i & 1], scores_i);
   * At most two jobs (slots 0 and 1) may be in flight. */
  struct SearchJob
  {
    int    slot = 0;
    bool   done = true;                     // nothing on the device (result already in `out`)
    std::vector<Score> out;
    // refinement state
    std::vector<SearchScore> candidates, refined;
    std::vector<int> lane_count, starts;
    size_t c0 = 0, nb = 0;
    bool   batch_pending = false;
    // approximate search still on the device
    KeyTables *kt = nullptr;
    DeviceWav  wav;
    Mode       mode = Mode::BLOCK;
    long long  n_scores = 0;
    bool       select_pending = false;
    bool       speculate_n_best = false;    // queue the n_best fallback together with the threshold selection (one round trip)
    // batched clip search: `wav` is a row of equal slices, candidate c lives in slice cand_slice[c] (its index is relative to it)
    std::vector<int> cand_slice;
    size_t           slice_frames = 0;      // 0: no slices
    const long long *slice_range = nullptr; // device, [slices][2]: non-silent value range of every slice (kernels.hh launch_clip_pad)
  };
  int search_launch (const Key& key, const DeviceWav& wav, Mode mode, SearchJob& job, bool db_ready = false);     // = approx_launch + select_refine
  int search_finish (SearchJob& job, std::vector<Score>& out);
  // finer steps for callers that drive several lanes: approx_launch never waits for the device,
  // select_refine waits for this lane's candidate list and queues the refinement
  // db_ready: the lane workspace still holds the dB matrices of THIS wav from the previous key of the same get (they do not depend on the key)
  int approx_launch (const Key& key, const DeviceWav& wav, Mode mode, SearchJob& job, bool prepared = false, bool db_ready = false);
  int select_refine (SearchJob& job);


  /* The CLIP search for a GROUP of padded clips that lie side by side in one buffer (equal slices, kernels.hh launch_clip_pad), stage by
   * stage with ONE launch per kernel and ONE wait per stage for the whole group -- a 30 s clip alone keeps the GPU busy for a
   * fraction of the time its ~20 launches and 3 round trips take.  Results per slice are those of search (key, slice, Mode::CLIP).
   * A slice whose selection needs the rare sequential path (more peaks above the threshold than the head list holds, a tie at the
   * n_best cut) is reported in `fallback` and must go through search() on its own. */
  struct GroupJob
  {
    int        n_slices = 0;
    size_t     slice_frames = 0;
    long long  n_scores = 0;               // per slice
    KeyTables *kt = nullptr;
    DeviceWav  group;
    SearchJob  refine;
    std::vector<char> fallback;
  };
  int group_approx_launch (KeyTables *kt, const DeviceWav& group, int n_slices, const long long *d_range, GroupJob& gj, bool db_ready = false);
  int group_select_refine (GroupJob& gj);
  int group_finish (GroupJob& gj, std::vector<std::vector<Score>>& out);

  // refinement of job.candidates in two halves (job.refined: candidate order); the multi-GPU protocol refines a chunk's candidates in parts
  int refine_launch (KeyTables *kt, const DeviceWav& wav, Mode mode, SearchJob& job);
  int refine_collect (SearchJob& job) { return refine_batch_finish (job); }
  // the end of search(): refined scores in candidate order -> final sync positions (reference syncfinder.cc:527-557)
  static void finish_scores (std::vector<SearchScore> refined, std::vector<Score>& out);

  static void select_local_maxima (std::vector<SearchScore>& scores);
  static void mask_avg_false_positives (std::vector<SearchScore>& scores);
  static void select_threshold_and_n_best (std::vector<SearchScore>& scores, double threshold);
  static void select_truncate_n (std::vector<SearchScore>& scores, size_t n);
private:
  awm_ctx  *m_ctx;
  WorkLane *m_lane;                     // stream, workspaces and staging buffers of this finder
  size_t   m_first = 0, m_last = 0;     // non-silent value range [first, last)
  size_t   m_prepare_values = 0, m_prepare_lo = 0;
  bool     m_prepare_pending = false;
  int scan_silence (const DeviceWav& wav);
  int fetch_scores (long long n_scores, std::vector<SearchScore>& out);
  int refine_batch_launch (KeyTables *kt, const DeviceWav& wav, Mode mode, SearchJob& job, size_t c0, size_t nb, size_t batch);
  int refine_batch_finish (SearchJob& job);
  int refine_finish (SearchJob& job, std::vector<SearchScore>& scores);
};

} // namespace awm
