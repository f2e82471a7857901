// kernels.hh -- host-callable launchers of the gfx950 kernels (kernels.hip, viterbi.hip).
// Everything here works on device pointers and enqueues on the given stream.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

// The kernels are written for ONE target.  They rely on gfx950's 160 KB of LDS per workgroup (K16: 117 KB static, K5w's ring 150 KB),
// on `sc1` device-coherent loads / stores with `s_waitcnt vmcnt(0)` for the cross-XCD meetings of the one-launch Viterbi kernel, on
// `global_load_lds_dwordx4` and on gfx9 buffer descriptors (K14).  Another --offload-arch must not build silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "audiowmark_amd's HIP kernels are written for gfx950 (MI355X) only"
#endif

namespace awmk {

// constant tables resident in HBM for the lifetime of a context
struct DevTables
{
  const float2 *tw512;    // e^{-2 pi i k / 512},  k < 512
  const float2 *tw1024;   // e^{-2 pi i k / 1024}, k <= 512
  const float  *window;   // normalised von Hann analysis window, 1024 (reference wmcommon.cc:68-89)
  const float  *synth;    // synthesis window, 3072 (reference wmadd.cc:177-206)
  const double2 *slide;   // [84 bins 19..102][9]: e^{-2 pi i k j / 1024}, j = 0..7, and e^{+2 pi i 8 k / 1024} at j = 8
  const double2 *tw512d;  // e^{-2 pi i k / 512} in double (first transform of a refinement row)
  const float2 *slide32;  // [84 bins 19..102][8]: e^{-2 pi i k j / 1024} e^{+2 pi i 8 k / 1024}, j = 0..7, rounded to float (K4s with the update term in float)
};

/* K1: FFTAnalyzer::run_fft / fft_range */
hipError_t launch_stft_full (hipStream_t st, const DevTables& t, const float *pcm, int n_channels,
                             long long start_index, long long hop, long long frame_count, float2 *out);

/* K2: fused STFT -> band edit -> inverse -> overlap-add -> mix (+ per-limiter-block maxima) */
struct AddMixArgs
{
  const float *pcm_in;
  float       *out;
  long long    n_frames;          // samples per channel in this span
  int          n_channels;
  const int8_t *frame_mod;        // device, [4452][81]
  float        neg_delta_up;      // (float) (-water_delta * +1)
  float        neg_delta_down;    // (float) (-water_delta * -1)
  long long    first_frame;       // global frame index of local frame 0
  const float *halo_before;       // 1024*C samples or nullptr
  const float *halo_after;        // 1024*C samples or nullptr
  unsigned int *block_max;        // float bits, or nullptr
  long long    first_block;
  long long    n_blocks;
  int          limiter_block;     // samples per limiter block (44100)
  int          frames_per_span;   // frames each wave streams through
  int          block_frames = 2226;   // mark_block_frame_count(): sync + data frames of a block (wmcommon.cc:36-48)
  int          frames_pad_start = 250; // Params::frames_pad_start
  int          delta_only = 0;    // 1: write the watermark signal alone (out = d W..., without "+ in"): WatermarkGen::run for the resampled path
};
hipError_t launch_add_mix (hipStream_t st, const DevTables& t, const AddMixArgs& a);
/* a batch of STEREO clips in one launch: args_dev[n_clips] on the device (every clip a stream of its own: first_frame 0, no halos, its own
 * block_max), max_spans = the largest ceil (ceil (n_frames / 1024) / frames_per_span) among them; block_frames / frames_pad_start as in the arguments */
hipError_t launch_add_mix_batch (hipStream_t st, const DevTables& t, const AddMixArgs *args_dev, int n_clips, long long max_spans, int block_frames,
                                 int frames_pad_start);
int        add_mix_waves_per_simd();     // occupancy the stereo kernel is built for (sizes the spans: one round of resident waves)

/* K3: limiter ramp, in place */
hipError_t launch_limiter (hipStream_t st, float *data, long long n_frames, int n_channels, long long first_sample,
                           const float *block_max, long long first_block, long long n_blocks,
                           int limiter_block, float ceiling, float2 *scale_tab = nullptr, size_t scale_tab_entries = 0);
/* K3 for a batch of clips (1 or 2 channels, data 16 byte aligned), every clip a stream that starts at sample 0 with its own block maxima
 * and a ramp table of limiter_tab_entries (n_frames, 0, limiter_block) entries: two launches for the batch */
struct LimiterClip { float *data; long long n_frames; const float *block_max; long long n_blocks; float2 *tab; long long n_tab; };
hipError_t launch_limiter_batch (hipStream_t st, const LimiterClip *clips_dev, int n_clips, long long max_frames, int n_channels, int limiter_block,
                                 float ceiling);
/* entries launch_limiter needs in scale_tab for this span (one (scale_start, scale_step) pair per limiter block) */
size_t     limiter_tab_entries (long long n_frames, long long first_sample, int limiter_block);
hipError_t launch_fill_u32 (hipStream_t st, unsigned int *p, unsigned int v, size_t n);

/* K10: fixed-ratio polyphase resampler = zita-resampler's Resampler (restated; the reference uses it with hlen 16 for
 * every sample rate other than 44100 Hz, resample.cc:128-270).  Output frame m of the stream "hl - 1 null frames, the
 * input, null frames for ever":  b = floor (m s / np), ph = m s mod np,
 *   y[m] = (1e-20f + sum_{i < hl} (P[b + i] c1[i] + P[b + 2 hl - 1 - i] c2[i])) - 1e-20f,   c1 = ctab + hl ph, c2 = ctab + hl (np - ph)
 * in float, products and sums rounded separately in this order. */
struct ResampleArgs
{
  const float *in;
  long long    n_in;        // frames available (everything else reads as zero)
  int          n_channels;
  const float *ctab;        // device, (np + 1) * hl
  int          hl, np, step; // step = s = fs_in / gcd
  float       *out;
  long long    n_out;       // frames to produce
};
hipError_t launch_resample (hipStream_t st, const ResampleArgs& a);

/* K11: out = wm + orig (reference wmadd.cc:564-565) and the per-limiter-block maxima of the result (limiter.cc:90-97) */
hipError_t launch_mix_max (hipStream_t st, const float *orig, const float *wm, float *out, long long n_frames, int n_channels,
                           unsigned int *block_max, long long n_blocks, int limiter_block);

/* `add --snr`: acc[0] += sum (mixed - orig)^2, acc[1] += sum orig^2 in double (reference wmadd.cc:553-563) */
hipError_t launch_power_sums (hipStream_t st, const float *orig, const float *mixed, long long n_values, double *acc);

/* K4: STFT -> dB of the 81 bands, written band-major ("transposed") so that scans over the
 * frame axis are coalesced.  Stream s (0 <= s < n_streams) consists of count(s) frames starting
 * at base(s) + f * hop.  out[s * out_stream_stride + (plane * 81 + band) * ld + f]. */
struct SyncDbArgs
{
  const float *pcm;
  long long    n_frames;          // samples per channel available (bounds)
  int          n_channels;
  int          per_channel;       // 0: channels summed into one plane; 1: one plane per channel
  long long    base0, base_stride;      // base(s) = base0 + s * base_stride     (if stream_base == nullptr)
  const long long *stream_base;         // device array [n_streams] or nullptr
  int          count0;                  // count(s) = count0                      (if stream_count == nullptr)
  const int   *stream_count;            // device array [n_streams] or nullptr
  long long    n_streams;
  long long    hop;
  float       *out;
  long long    out_stream_stride;
  long long    ld;
  char        *have;                    // have[s * have_stream_stride + f] or nullptr
  long long    have_stream_stride;
  long long    first, last;             // non-silent value range [first, last) (syncfinder.cc:155-169)
  int          tile_frames;             // frames per workgroup tile (<= 72)
  int          xcd_interleave = 0;      // set by the launcher: 1-D grid, the streams of one tile on the same XCD (see kernels.hip)
  // K4s only, "gathered" output for the refinement scan (K5g): stream s = plane * rows_per_plane + w writes its
  // rows to stream slot plane * rows_per_plane + row_perm[w], and band b to row band_pos[w * 81 + b] (255: dropped)
  const int           *row_perm = nullptr;
  const unsigned char *band_pos = nullptr;
  int                  rows_per_plane = 0;
  // ... and, with forms 4 / 5 of K4s, the rows' 65th value (fine offset 64) apart from the rows: tail[slot * tail_stream_stride + row],
  // so that a row of `ld` = 64 floats is two whole cache lines (nullptr: offset 64 goes into the row like the others, ld >= 65)
  float               *tail = nullptr;
  long long            tail_stream_stride = 0;
  // a batch of clips with one KEY PER CLIP: the tables of slice i (range_index / range_div as below) follow those of slice i - 1:
  // row_perm + slice * rows_per_plane, band_pos + slice * rows_per_plane * 81
  int                  tables_per_slice = 0;
  // Slices (the batched clip search: many padded clips side by side in one buffer).  With streams_per_slice > 0 stream s is
  // stream s % streams_per_slice of slice s / streams_per_slice: base = base0 + (s % sps) * base_stride + (s / sps) * slice_stride.
  // stream_range, if set, replaces first / last by the range of the stream's slice: slice = range_index[s / range_div] if
  // range_index is set, else s / range_div; entries are absolute value indices [first, last), first < 0 = nothing but silence.
  int                  streams_per_slice = 0;
  long long            slice_stride = 0;
  const long long     *stream_range = nullptr;
  const int           *range_index = nullptr;
  int                  range_div = 1;
  // 1: a frame in the silence outside the range is not skipped but gets the dB values of a transformed frame of zeros (exactly
  // what the transform would deliver: -96 per band and channel) -- the block decoder's fft_range knows no skipping
  int                  silent_frames_are_zero = 0;
  // with silent_frames_are_zero: a tile all of whose frames, and the two frames on either side of it, lie in that silence is not
  // written at all -- for the one consumer that never reads such frames (K7 with SoftBitsArgs::stream_range)
  int                  skip_unread_silent_tiles = 0;
};
hipError_t launch_sync_db (hipStream_t st, const DevTables& t, const SyncDbArgs& a);
/* K4s: same output as K4 for streams whose frames advance by 8 samples (search_refine): instead of one FFT per fine
 * offset the 83 needed bins are carried from offset to offset by a sliding DFT in double precision. n_channels <= 2. */
hipError_t launch_sync_db_sliding (hipStream_t st, const DevTables& t, const SyncDbArgs& a);
/* does the form of K4s in force (awm_debug_set_refine_form) write rows of 64 fine offsets + SyncDbArgs::tail for streams of this many channels? */
bool sliding_rows_have_tail (int n_channels);

/* K5: sync_decode (syncfinder.cc:116-153) for many candidates.
 * value(cand, row, band) = db[plane(cand) + row * row_stride + band * band_stride + lane(cand)],
 * have(cand, row)        = have[hplane(cand) + row * have_row_stride + lane(cand)]
 * where cand = group * 64 + lane-in-wave, plane(cand) = group_plane[...] see kernels.hip. */
struct SyncTableDev
{
  // [6][rows][64] int32: [0..29] up bands, [30..59] down bands (band - 20), [60] row index
  // (frame for the approximate search, want-list position for the refinement); wave-uniform -> scalar loads
  const int *packed;
  int        rows_per_bit;
  // K5w only (approximate search): [12 chains = sync bit x (up, down)][rows][8] words = the chain's 30 bands of the row as bytes,
  // then the frame of the chain's NEXT row as u16 (0xffff: none): one 32-byte scalar load per sync frame
  const unsigned *chains = nullptr;
  // a batch of clips with one key per clip: plane p uses the chains at chains + (p / planes_per_slice) * chains_slice_stride (words)
  long long       chains_slice_stride = 0;
  int             planes_per_slice = 1;
  // ... and the frame of every row, [slice][6][rows] (K5w in CLIP mode looks rows up by frame; one key: packed[..][60] serves)
  const int      *row_frames = nullptr;
};
/* host side of SyncTableDev::chains from a [6][rows][64] packed table */
void pack_scan_chains (const int *packed, int rows_per_bit, unsigned *out /* [12 * rows_per_bit * 8] */);
struct SyncScanArgs
{
  const float *db;
  const char  *have;            // nullptr: every frame present
  int          have_is_run = 0; // 1: the present frames of every plane form one run and absent frames hold +0 in db (what K4's silence rule
                                // produces): lets launch_sync_scan_window use the streaming kernel in CLIP mode too
  long long    plane_stride;    // between blockIdx.y planes (shifts / candidates)
  long long    have_plane_stride;
  long long    row_stride, band_stride, have_row_stride;
  long long    n_lanes;         // candidates per plane (start frames, or fine offsets)
  const int   *lane_count;      // optional per-plane valid lane count (refine), else n_lanes
  long long    n_planes;
  double       min_delta;       // min (water_delta, 0.080), reference syncfinder.cc:80-92
  double      *quality;         // [plane][q_stride]
  long long    q_stride;
  SyncTableDev table;
};
hipError_t launch_sync_scan (hipStream_t st, const SyncScanArgs& a);
/* K5w (scan.hip): same result for band-major planes (row_stride == 1, band_stride % 64 == 0, table.chains set): the dB
 * matrix streams through an LDS ring, four candidates per lane.  total_frames = frames a candidate spans (2226 BLOCK / 4452 CLIP). */
hipError_t launch_sync_scan_window (hipStream_t st, const SyncScanArgs& a, int total_frames);

/* K5g: sync_decode over the gathered layout K4s writes for the refinement: plane (candidate) p holds
 * [6 bits][rows_per_bit][60 values: up 0..29, down 0..29][ld fine offsets] -- exactly the order the sums consume them,
 * so the scan is a sequential stream without any table.  have: [plane][6 * rows_per_bit][ld] or nullptr. */
struct GatheredScanArgs
{
  const float *db;
  const char  *have;
  long long    plane_stride, have_plane_stride;
  int          ld;
  int          have_ld = 0;     // row length of `have` if it is not ld
  const float *tail = nullptr;  // K4s forms 4 / 5: the values of fine offsets >= ld (there is one: 64), [plane][6 bits][rows_per_bit][60]
  long long    tail_plane_stride = 0;
  int          rows_per_bit;
  int          n_lanes;         // fine offsets (<= 128)
  const int   *lane_count;      // per plane
  long long    n_planes;
  double       min_delta;
  double      *quality;         // [plane][q_stride]
  long long    q_stride;
  float       *chain_mag;       // scratch [plane][12 chains = (bit, up / down)][128]: the chains' sums
  int         *chain_n;         // scratch [plane][6 bits][128]: sync frames that counted
  int          regular_waves, tail_first, tail_lanes;     // set by the launcher
};
constexpr size_t GATHERED_SCRATCH_BYTES_PER_PLANE = 18 * 128 * 4;
hipError_t launch_sync_scan_gathered (hipStream_t st, const GatheredScanArgs& a);

/* K5b: local mean over the index-sorted scores (syncfinder.cc:234-254); q is [4][q_stride] by shift,
 * sorted position p = 4 * start_frame + shift.  Writes raw[p], mean[p]. */
/* n_slices > 1: slice y reads q + y * 4 * q_stride and writes raw_sorted / local_mean + y * 4 * n_start_frames */
hipError_t launch_local_mean (hipStream_t st, const double *q, long long q_stride, long long n_start_frames,
                              double *raw_sorted, double *local_mean, int n_slices = 1);

/* K5c: local maxima + false-positive mask + threshold on the device (syncfinder.cc:258-332, 364-383) */
struct PeakOut { long long p; double raw, mean; };
hipError_t launch_peak_select (hipStream_t st, const double *raw_sorted, const double *local_mean, long long n, double threshold,
                               unsigned int *count, PeakOut *out, unsigned int cap);
/* the same for n_slices independent score lists of n scores each: slice y uses raw / mean + y * n, the counter count[y * count_stride]
 * (NOT cleared here) and the list out + y * cap */
hipError_t launch_peak_select_slices (hipStream_t st, const double *raw_sorted, const double *local_mean, long long n, double threshold,
                                      unsigned int *count, int count_stride, PeakOut *out, unsigned int cap, int n_slices);

/* K5d: per-slice top k (by |raw - mean|) of a peak list whose length lives in *count; out is [n_slices][k], p = -1 = empty */
hipError_t launch_peak_topk (hipStream_t st, const PeakOut *in, const unsigned int *count, unsigned int cap, PeakOut *out, int k, int n_slices);
/* the same for n_lists lists: list y = in + y * cap with its length in count[y * count_stride], result out + y * n_slices * k */
hipError_t launch_peak_topk_lists (hipStream_t st, const PeakOut *in, const unsigned int *count, int count_stride, unsigned int cap,
                                   PeakOut *out, int k, int n_slices, int n_lists);

/* K7: mix_decode (wmget.cc:67-108): db is [n_blocks][C][81][ld] (band-major), out [n_blocks][858] */
struct SoftBitsArgs
{
  const float   *db;
  long long      block_stride, ld;
  int            n_channels;
  const int16_t *mix_frame;     // [n_entries]
  const uint8_t *mix_up, *mix_down;
  int            n_data_frames; // 1716
  int            frames_per_bit;
  int            block_frames;  // 2226
  long long      n_blocks;
  float         *out;
  // one key per clip: block b takes the mix table of slice block_slice[b] (entries at + slice * n_data_frames * 30)
  const int     *block_slice = nullptr;
  // Blocks cut out of padded slices (clip batches): block b starts at sample block_base[b] of the buffer; the values of its slice that are
  // not digital silence are [stream_range[2 s], stream_range[2 s + 1]) with s = range_index[b] (first < 0: nothing but silence) -- the
  // arrays K4b got (SyncDbArgs).  A frame outside that range has -96 dB in every band and channel (K4b writes exactly that,
  // silent_frames_are_zero), so an item whose frame AND both neighbours are such frames is known without a load: (-96, -192, -96, -192).
  // Four fifths of a 30 s clip's padded block are such items.  The sums are the same operations on the same values.
  const long long *block_base = nullptr;
  const long long *stream_range = nullptr;
  const int       *range_index = nullptr;
};
hipError_t launch_soft_bits (hipStream_t st, const SoftBitsArgs& a);

/* K7b: everything between mix_decode and the Viterbi decoder, per decode job (reference wmget.cc:40-65 normalize_soft_bits,
 * wmcommon.cc:165-185 randomize_bit_order, wmget.cc:554-701 AB interleave / "all" average): the raw soft bits of the
 * blocks never leave the device.  Job j reads its sources src[src_off .. src_off + n_src) = (block slot, half) and
 * writes len = 858 (single block) or 1716 normalised soft bits to out + out_off. */
struct SoftJobDev
{
  int       mode;          // 0: one block; 1: interleave (half 0 / 1 of every pair of bits); 2: average of several blocks per half
  int       n_src, src_off;
  int       len;           // 858 or 1716
  int       norm0, norm1;  // mode 2: blocks per half
  long long out_off;       // floats
  int       order_off;     // inv_order + order_off: the bit order of this job's key (one key per clip: n_bits * slice)
  int       pad;
};
struct SoftPrepArgs
{
  const float      *raw;         // [slots][n_bits]
  int               n_bits;      // 858
  const int        *inv_order;   // [n_bits]: restored[k] = raw[inv_order[k]]
  const SoftJobDev *jobs;
  const int2       *src;         // (slot, half)
  long long         n_jobs;
  int               hard;        // Params::hard
  float            *out;
};
hipError_t launch_soft_prep (hipStream_t st, const SoftPrepArgs& a);

/* K8: soft Viterbi (convcode.cc:128-213), one workgroup per coded block */
/* index 0 / 1 / 2 = A / B / AB coded blocks (rate 6 / 6 / 12); every block has n_steps = payload + 15 trellis steps */
/* sync_ws: viterbi_sync_bytes (all blocks of the call) bytes that are ZERO on entry and zero again when the launch has finished
 * (the one-launch kernel's tickets and per-decode counters; a lane keeps one such block and zeroes it when it allocates it);
 * nullptr: the chain of 16 launches.  A decode that reports error -2 means a device-side wait gave up: the batch is invalid. */
hipError_t launch_viterbi (hipStream_t st, const float *const soft[3], const long long n_blocks[3], long long n_steps,
                           unsigned char *const decisions_ws[3], int *const bits_out[3], float *const error_out[3], unsigned int *sync_ws);
size_t viterbi_workspace_bytes (long long coded_len, int rate, long long n_blocks);
size_t viterbi_sync_bytes (long long n_blocks);
/* microseconds per launch of a chain of empty dependent launches on the (idle) stream: decides, per process, between the chain of 16
 * launches and the one-launch kernel (viterbi.hip) */
double probe_dependent_launch_us (hipStream_t st);
const char *viterbi_form_description();          // which form the batches take and why (for error messages)

/* K16 (keytab.hip): frame_mod tables built on the device, one workgroup per key (reference wmadd.cc:86-162, wmcommon.cc:143-202,
 * random.cc:97-161).  round_keys: 176 bytes per key (the AES-128 key schedule as FIPS-197's byte string, host/aes128.cc); sbox: the
 * 256 byte S-box; coded: the payload's convolutional code, 858 bytes (0 / 1) for an A block followed by 858 for a B block;
 * scratch: scratch_slots * key_table_scratch_bytes() bytes (n_keys <= scratch_slots: a launch's keys work side by side);
 * tables: n_keys * key_table_bytes() bytes, [2 (A, B)][2226 frames][81 bands] int8 KEEP 0 / UP 1 / DOWN 2. */
struct KeyTableArgs
{
  const unsigned char *round_keys;
  const unsigned char *sbox;
  const unsigned char *coded;
  unsigned char       *scratch;
  int                  scratch_slots;
  signed char         *tables;
  long long            n_keys;
};
size_t key_table_scratch_bytes();
size_t key_table_bytes();
hipError_t launch_frame_mod_tables (hipStream_t st, const KeyTableArgs& a);

/* K16g: the tables `get` needs for a clip with a key of its own (CLIP mode; reference syncfinder.cc:30-77 init_up_down, wmget.cc:52-108
 * mix_decode's entries, wmcommon.hh randomize_bit_order), from the same draws and shuffles as K16, one workgroup per key -- what the
 * host's build_clip_key_host produces (host/context.cc), table for table and byte for byte:
 *   chains      [key][12][170][8] u32   K5w: chain 2 bit + (0 up | 1 down), rows by frame: 30 band bytes + the next row's frame (u16)
 *   row_frames  [key][6][170] int       frame of every row (bit-major, by frame)
 *   want        [key][1020] int         the sync frames of the long block (two blocks), ascending
 *   perm        [key][1020] int         row w of the want list -> bit * 170 + row
 *   pos         [key][1020][81] u8      band -> 0 .. 29 (up), 30 .. 59 (down), 255 (not used) of want row w
 *   mix_frame   [key][51480] i16, mix_up / mix_down [key][51480] u8 (absolute band 20 .. 100): mix entry p
 *   inv_order   [key][858] int          inverse of the bit order permutation
 * KeyTableArgs::coded and ::tables are not used. */
struct ClipKeyTableOut
{
  unsigned int  *chains;
  int           *row_frames, *want, *perm;
  unsigned char *pos;
  short         *mix_frame;
  unsigned char *mix_up, *mix_down;
  int           *inv_order;
};
constexpr int CLIP_KEY_ROWS = 170, CLIP_KEY_WANT = 1020, CLIP_KEY_MIX = 51480, CLIP_KEY_CODED = 858;
hipError_t launch_clip_key_tables (hipStream_t st, const KeyTableArgs& a, const ClipKeyTableOut& out);

} // namespace awmk

namespace awmk {
/* I/O staging: interleaved PCM bytes <-> float32 on the device with the reference's
 * conversion rules (rawconverter.cc:155-286, rawconverter.hh:34-50).  width = bytes per value (1..4 integer, 4 / 8 float);
 * encoding 0 signed / 1 unsigned / 2 float; direct16 = the reference's native little-endian signed 16 bit rule
 * (truncate at 16 bit) instead of "clip to 32 bit, keep the top bits". */
struct PcmFormatDev { int width, encoding, big_endian, direct16; };
hipError_t launch_pcm_decode (hipStream_t st, const unsigned char *bytes, float *out, long long n_values, PcmFormatDev f);
hipError_t launch_pcm_encode (hipStream_t st, const float *in, unsigned char *bytes, long long n_values, PcmFormatDev f);

/* first / one-past-last non-zero value of an interleaved buffer (SyncFinder::scan_silence,
 * reference syncfinder.cc:155-169); result[0] = first (n_values if all zero), result[1] = last */
hipError_t launch_nonzero_range (hipStream_t st, const float *data, long long n_values, unsigned long long *result);

/* The padded copies of a group of clips (ClipDecoder, reference wmget.cc:830-867, START position) side by side in one buffer:
 * slice i = [pad_start_i zeros][clip i][pad zeros], slice_values each, and the non-silent range of every slice
 * (range[2 i] = first non-zero value, absolute index in dst, -1 if there is none; range[2 i + 1] = last + 1). */
struct ClipSrc { const float *data; long long n_values; long long pad_start; };
/* Of the padding only `margin_values` on either side of a clip are written (the rest of a slice keeps whatever the buffer held): every
 * consumer of the slices skips the frames outside the non-silent range -- K4 / K4b per frame, K4s per row of fine offsets, K7 per item --
 * and the frames it does read reach at most 1024 + 8 * 64 frames (K4s: a row of 65 fine offsets, plus its read-ahead) beyond that range.
 * margin_values >= slice_values writes whole slices. */
hipError_t launch_clip_pad (hipStream_t st, const ClipSrc *src /* device */, int n_clips, float *dst, long long slice_values, long long margin_values,
                            long long *range);
}

namespace awmk {
/* ---- speed detection (reference wmspeed.cc; speed.hip) -------------------------------------------------------------- */
/* one centre speed of a scan pass, or the one ratio of a plain resample_ratio call: zita VResampler geometry */
struct SpeedCenterDev
{
  const float       *ctab;         // (256 + 1) rows of `stride` floats, hl coefficients each
  int                hl;           // taps per side
  int                stride;       // hl | 1 (odd: LDS bank spread)
  int                shift;        // input window of output m starts at (m * mant) >> shift
  unsigned long long mant;         // 53 bit mantissa of the phase step 256 / ratio
  double             frac_scale;   // fraction of that product -> phase 0 .. 256
  long long          n_in;         // input frames (after truncation)
  long long          n_out;        // output frames
  int                rows;         // K13 / K14: STFT rows of the half-rate clip
};
/* K12: VResampler (resample.cc:96-125); blockIdx.y = centre, outputs at out + centre * out_stride */
struct VarResampleArgs
{
  const float          *in;
  int                   n_channels;
  const SpeedCenterDev *centers;
  float                *out;
  long long             out_stride;    // floats
  int                   max_stride;    // largest table row stride among the centres (sizes the LDS copy)
  double                max_step;      // largest advance of the input window per output among the centres, in frames (mant / 2^shift); 0: unknown
  int                   lds_floats;    // set by the launcher
};
hipError_t launch_resample_var (hipStream_t st, const VarResampleArgs& a, long long max_n_out, int n_centers);

/* K13: SpeedSync::prepare_mags (wmspeed.cc:204-268) */
struct SpeedMagsArgs
{
  const float          *sub;           // half-rate clips, [centre][sub_stride]
  long long             sub_stride;
  int                   n_channels;
  const SpeedCenterDev *centers;
  const float          *window512;     // FFTAnalyzer::gen_normalized_window (512)
  const unsigned int   *cols;          // [510][16] words: 30 up + 30 down band indices (0..80) per sync frame, columns [bit][frame asc]
  float2               *mags;          // [centre][510][ld] (umag, dmag)
  long long             mags_center_stride, ld;
};
hipError_t launch_speed_mags (hipStream_t st, const DevTables& t, const SpeedMagsArgs& a, int max_rows, int n_centers);

/* K14: SpeedSync::compare (wmspeed.cc:270-395) */
struct SpeedItemDev { int center; double rel_speed_inv, q16_scale; };
struct SpeedCompareArgs
{
  const float2         *mags;
  long long             mags_center_stride, ld;
  const SpeedCenterDev *centers;
  const SpeedItemDev   *items;         // [n_centers][items_per_center]
  int                   n_centers, items_per_center;
  const int            *col_frame;     // [510] frame of the column
  const unsigned char  *col_first;     // [6][frames_per_block + 2]: columns of the bit with frame < f
  int                   frames_per_block, steps_per_frame, pad_start, rows_per_bit;
  double                min_delta;
  unsigned long long   *best;          // [items] bits of the best quality (zero initialised)
  int                   fold_groups = 0, n_ranges = 0;   // set by the launcher: groups of speeds folded into blockIdx.x (see launch_speed_compare)
};
hipError_t launch_speed_compare (hipStream_t st, const SpeedCompareArgs& a, int n_items);

/* K15 */
constexpr int ENERGY_PARTS = 256;
hipError_t launch_gather_values (hipStream_t st, const float *in, const unsigned long long *pos, long long n, float *out);
hipError_t launch_energy (hipStream_t st, const float *in, const long long *range /* [n][2] value index begin, end */, int n_ranges,
                          double *out /* [n][ENERGY_PARTS] */);
}
