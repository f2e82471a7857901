// Speed detection on the GPU (reference src/wmspeed.{hh,cc}) and zita's variable-ratio resampler behind it
// (reference src/resample.cc:96-125).  Same entry points and result types as the reference:
//   detect_speed (key_list, wav, print_results) -> one { key, speed } per key whose best speed passes the thresholds
//   resample_ratio / resample_ratio_truncate
// The searches themselves (SpeedSync::prepare_mags / compare for every centre and relative speed of a pass) are three
// kernels per pass (hip/speed.hip); the host keeps the reference's control flow between the passes.
#pragma once
#include <memory>
#include <vector>
#include "context.hh"
#include "random.hh"
#include "syncfinder.hh"

namespace awm {

struct DetectSpeedResult
{
  Key    key;
  double speed = 0;
};

struct SpeedScanParams      // reference wmspeed.cc:54-60
{
  double seconds        = 0;
  double step           = 0;
  int    n_steps        = 0;
  int    n_center_steps = 0;
};

struct SpeedScore           // reference wmspeed.cc:99-103
{
  double speed = 0;
  double quality = 0;
};

// cached per ratio: VResampler::setup (ratio, nchan, hlen = 16)
struct VarResampleTable
{
  double       ratio = 0;
  int          hl = 0;
  const float *ctab = nullptr;      // inside a slab of SpeedWorkspace::table_slabs
  size_t       slab = 0;            // serial number of that slab
};
// the tables a call of get_var_tables had to build, side by side in ONE device allocation (a first `get --detect-speed` builds ~230
// tables of ~20 KB: as allocations of their own they were 27 ms of its 62)
struct VarTableSlab
{
  size_t    serial = 0;
  DevBuffer buf;
};

// per key: the 510 sync frames of a block in the column order of the magnitude matrix ([bit][frame ascending])
struct SpeedKeyTables
{
  std::vector<unsigned char> key;
  int       frames_per_bit = 2;     // params().frames_per_bit the block's layout was drawn for
  DevBuffer cols;           // [510][16] words: 30 up + 30 down band indices
  DevBuffer col_frame;      // [510] int
  DevBuffer col_first;      // [6][block frames + 2] uint8: columns of the bit with frame < f
};

/* tables shared by all lanes of a context (read-only once built; lookups and construction under awm_ctx::speed_mutex) */
struct SpeedWorkspace
{
  std::vector<std::unique_ptr<VarResampleTable>> var_tables;
  std::vector<VarTableSlab>                      table_slabs;
  size_t                                         next_slab = 0;
  std::vector<std::unique_ptr<SpeedKeyTables>>   key_tables;
  DevBuffer    window512;
  void release();
};

/* buffers of ONE speed search / stretch: per work lane, so that the chunks of a stream can run their searches side by side */
struct SpeedScratch
{
  DevBuffer    sub, mags, centers, items, best, gather_pos, gather_out, ranges, energy, stretched;
  PinnedBuffer pin;
  void release();
};
SpeedScratch *speed_scratch (WorkLane *lane);      // created on first use
void speed_scratch_free (WorkLane *lane);

/* VResampler::setup (ratio, nchan, 16) as the closed form the kernel uses: output m reads the input window that starts at
 * (m * mant) >> shift (in the stream "hl - 1 null frames, the input, null frames"), 2 hl taps */
struct VarResampleGeometry
{
  bool               ok = false;
  int                hl = 0;
  unsigned long long mant = 0;
  int                shift = 0;
  size_t window_start (size_t m) const { return size_t ((static_cast<unsigned __int128> (m) * mant) >> shift); }
  // outputs a streaming resampler delivers for n_in input frames followed by hl null frames: every m with window_start (m) <= n_in - 1
  size_t stream_frames (size_t n_in) const { return size_t (((static_cast<unsigned __int128> (n_in) << shift) + mant - 1) / mant); }
};
VarResampleGeometry var_resample_geometry (double ratio);
/* n_out output frames of the VResampler stream for n_in input frames (zero extended) */
int resample_var_device (awm_ctx *ctx, WorkLane *lane, const float *in_d, size_t n_in, int n_channels, double ratio, float *out_d, size_t n_out);

/* resample_ratio_truncate (reference resample.cc:96-119) on the device: `out` receives lrint (frames * ratio) frames
 * (frames = min (wav.n_frames, lrint (rate * max_in_seconds)) if max_in_seconds > 0) */
int resample_ratio_device (awm_ctx *ctx, WorkLane *lane, const DeviceWav& wav, double ratio, double max_in_seconds,
                           DevBuffer& out, size_t *n_out_frames);

/* the lane's buffer for a chunk stretched to the detected speed (decode() keeps one at a time) */
DevBuffer& speed_stretch_buffer (WorkLane *lane);

/* get_best_clip_location (reference wmspeed.cc:533-577) */
int speed_clip_location (awm_ctx *ctx, WorkLane *lane, const Key& key, const DeviceWav& wav, double seconds, int candidates, double *location);

/* one pass of run_search for one key (reference wmspeed.cc:461-492, 683-719): scores of every centre x relative speed,
 * in the order centres (speeds x -n_center_steps..n_center_steps) x steps -n_steps..n_steps */
int speed_scan (awm_ctx *ctx, WorkLane *lane, const Key& key, const DeviceWav& wav, double clip_location, const SpeedScanParams& scan_params,
                const std::vector<double>& speeds, std::vector<SpeedScore>& scores);

/* magnitude matrix of one centre speed (tests): rows x 510 x (umag, dmag), columns in the reference's order (sorted by frame) */
int speed_mags (awm_ctx *ctx, WorkLane *lane, const Key& key, const DeviceWav& wav, double clip_location, double center, double seconds,
                std::vector<float>& out, int *rows);

void   select_n_best_scores (std::vector<SpeedScore>& scores, size_t n);                          // reference wmspeed.cc:494-531
double score_smooth_find_best (const std::vector<SpeedScore>& scores, double step, double distance); // reference wmspeed.cc:397-428

/* detect_speed (reference wmspeed.cc:622-781) on a lane.  best_speed / best_quality (optional) receive the last key's values
 * whether or not they pass the thresholds.  `report` (optional) receives the "detect_speed ..." lines the reference prints
 * with print_results (the caller prints them: several chunks may be searching at the same time). */
int detect_speed (awm_ctx *ctx, WorkLane *lane, const std::vector<Key>& key_list, const DeviceWav& wav, std::string *report,
                  std::vector<DetectSpeedResult>& results, double *best_speed = nullptr, double *best_quality = nullptr);

} // namespace awm
