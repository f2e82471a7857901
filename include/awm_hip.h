/* awm_hip.h -- C ABI of the MI355X (gfx950) spectral watermark path.
 *
 * The reference (swesterfeld/audiowmark 0.6.5) has no FFI layer; the seams this library
 * replaces are C++ member functions.  Each entry point below names the reference
 * interface it stands in for (file:line relative to the reference's src/).
 *
 * Conventions
 *   - plain C, pointers + sizes only.  Pointers suffixed _d are DEVICE pointers (HBM),
 *     everything else is host memory.
 *   - return 0 on success, negative on error; awm_last_error() gives the message
 *     (thread-local).  There is NO CPU fallback: without a usable gfx950 device every
 *     compute entry point fails with AWM_ERR_NO_DEVICE.
 *   - PCM is interleaved float32, `n_frames` = samples per channel (reference WavData,
 *     wavdata.hh:27-74); all watermark arithmetic is at 44100 Hz (wmcommon.hh:68).
 *   - work is enqueued on the context's HIP stream; entry points that return host data
 *     synchronise that stream before returning, the *_async ones do not.
 */
#ifndef AWM_HIP_H
#define AWM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AWM_ERR_GENERIC    (-1)
#define AWM_ERR_NO_DEVICE  (-2)
#define AWM_ERR_ARG        (-3)
#define AWM_ERR_HIP        (-4)
#define AWM_ERR_IO         (-5)    /* a file could not be opened / read / written (file level entry points) */

#define AWM_FRAME_SIZE      1024
#define AWM_N_BANDS         81      /* bins 20..100, wmcommon.hh:39-40 */
#define AWM_BLOCK_FRAMES    2226    /* 510 sync + 1716 data frames at the default payload */
#define AWM_SOFT_BITS       858     /* conv_code_size(a,128), convcode.cc:65-75 */

typedef struct awm_ctx awm_ctx;

const char *awm_last_error (void);
const char *awm_version (void);

/* ---- context: one per GPU / per rank ------------------------------------------------ */
int   awm_ctx_create (int device, awm_ctx **ctx_out);
/* the same on a stream of the caller (NULL: the device's default stream); the context then creates no stream of its own
 * (every HIP stream costs ~190 MB of resident host memory on this runtime: the command line uses this) */
int   awm_ctx_create_on_stream (int device, void *hip_stream, awm_ctx **ctx_out);
void  awm_ctx_destroy (awm_ctx *ctx);
/* A context keeps its workspaces between calls (a call's set-up is then ~1.5 ms instead of 20 - 25): the lanes' scratch buffers, the
 * file level staging rings (4 + 4 page-locked tiles of up to 32 MiB and their device twins) and the float32 PCM of the longest stream the
 * file level `get` has decoded (1.3 GB per hour).  awm_ctx_trim gives all of that back (also for the context's helpers); key tables and
 * streams stay.  For long-lived services after an unusually long file. */
int   awm_ctx_trim (awm_ctx *ctx);
/* ... and the other way round: create NOW the HIP streams a first call would create (4 - 10 ms each in this runtime): the chunk lanes of
 * `get` (twice as many with detect_speed != 0: the plain decode runs beside the speed search), with file_level != 0 the copy stream of the
 * file level calls.  For services that create their contexts at start-up. */
int   awm_ctx_warm_up (awm_ctx *ctx, int detect_speed, int file_level);
int   awm_ctx_device (const awm_ctx *ctx);
int   awm_ctx_synchronize (awm_ctx *ctx);
/* opaque hipStream_t of the context (for callers that bracket work with HIP events) */
void *awm_ctx_stream (awm_ctx *ctx);
/* run all work of this context on an externally owned hipStream_t (e.g. torch's current stream) */
int   awm_ctx_set_stream (awm_ctx *ctx, void *hip_stream);
/* `get` decodes the 30-minute chunks of a stream concurrently on up to 4 "lanes" (HIP streams with their own workspaces);
 * n_lanes = 1 runs them one after the other on the context's stream (per-kernel timing without overlap, smallest footprint) */
int   awm_ctx_set_chunk_lanes (awm_ctx *ctx, int n_lanes);

/* ---- per-kernel timing: HIP events recorded on the context's stream around every launch.
 * ids 0..awm_prof_count()-1, awm_prof_name(id) is the kernel name as rocprofv3 shows it (plus the
 * call site in brackets); algorithmic_bytes = SURVEY.md section 8(d) bytes summed over the launches. */
int         awm_prof_enable (awm_ctx *ctx, int on);
int         awm_prof_reset (awm_ctx *ctx);
int         awm_prof_count (void);
const char *awm_prof_name (int id);
int         awm_prof_read (awm_ctx *ctx, int id, double *ms, long *launches, double *algorithmic_bytes);

/* ---- key-derived tables (pure host, callable without a GPU) -------------------------- */
/* replaces: UpDownGen::get wmcommon.hh:107-122; BitPosGen wmcommon.cc:143-165;
 *           gen_mix_entries wmcommon.cc:179-202; init_frame_mod_vec wmadd.cc:148-162;
 *           SyncFinder::get_sync_bits syncfinder.cc:30-77; randomize_bit_order wmcommon.hh:165-185 */
int awm_tab_up_down (const uint8_t key[16], int stream, int frame, int up[30], int down[30]);
int awm_tab_bit_pos (const uint8_t key[16], int pos[AWM_BLOCK_FRAMES]);
int awm_tab_mix_entries (const uint8_t key[16], int *frame_up_down /* [51480*3] */);
int awm_tab_bit_order (const uint8_t key[16], size_t n, unsigned *order);
/* out[2][2226][81]: 0 KEEP / 1 UP / 2 DOWN for the A and the B block */
int awm_tab_frame_mod (const uint8_t key[16], const char *payload_hex, int8_t *out);
/* out rows: frame, up[30], down[30] (band-20, ascending); returns rows per bit (85 / 170) */
int awm_tab_sync_bits (const uint8_t key[16], int clip_mode, int *out /* [6*rows*61] */);
int awm_tab_window (size_t n, float *out);                    /* FFTAnalyzer::gen_normalized_window wmcommon.cc:68-89 */
int awm_tab_synth_window (float *out /* [3072] */);           /* WatermarkSynth::generate_window wmadd.cc:177-206 */
int awm_conv_encode (int block_type, const int *bits, size_t n, int *out); /* conv_encode convcode.cc:100-125 */
/* `audiowmark test-gen-noise` (audiowmark.cc:399-417): n_values floats = rng.random_double() * 2 - 1 of
 * Random (key, 0, Stream::data_up_down) -- the reference's own synthetic input (interleaved stereo white noise) */
int awm_test_gen_noise (const uint8_t key[16], size_t n_values, float *out);

/* ---- kernel level (device pointers) --------------------------------------------------- */

/* FFTAnalyzer::run_fft / fft_range (wmcommon.cc:91-141): `frame_count` windowed 1024-point
 * r2c transforms per channel, frame f starting at sample start_index + f*hop.
 * out_d: [frame_count][n_channels][513] complex64 (re,im).  Fails if the range exceeds n_frames. */
int awm_stft_d (awm_ctx *ctx, const float *pcm_d, size_t n_frames, int n_channels,
                size_t start_index, size_t hop, size_t frame_count, float *out_d);

/* add: WatermarkGen::run + WatermarkSynth::run + mix + Limiter (wmadd.cc:61-84,215-250,297-317,
 * 564-568; limiter.cc:90-124) for a contiguous span of the stream.
 *   frame_mod:        host table from awm_tab_frame_mod ([2][2226][81])
 *   first_frame:      index (in 1024-sample frames) of pcm_in_d[0] inside the whole stream; spans
 *                     are how awm shards `add` across GPUs.  The span must start on a frame boundary.
 *   halo_before_d / halo_after_d: the 1024*C samples preceding / following the span (NULL = stream
 *                     start / zeros after the end), needed for the 3-frame overlap-add (wmadd.cc:228-238)
 *   use_limiter:      0 = Params::test_no_limiter
 * awm_add_mix_d writes the un-limited mix to out_d and accumulates per-limiter-block maxima into
 * block_max_d[b] (b = global limiter block index - first_block, float32, pre-initialised by
 * awm_add_init_block_max_d); awm_add_limit_d applies the limiter ramp in place.  Between the two a
 * multi-GPU caller all-reduces (max) block_max_d.  awm_add_d = both, single GPU. */
int awm_add_init_block_max_d (awm_ctx *ctx, float *block_max_d, size_t n_blocks);
int awm_add_mix_d (awm_ctx *ctx, const float *pcm_in_d, float *out_d, size_t n_frames, int n_channels,
                   const int8_t *frame_mod, double water_delta, size_t first_frame,
                   const float *halo_before_d, const float *halo_after_d,
                   float *block_max_d /* may be NULL: no limiter */, size_t first_block, size_t n_blocks);
int awm_add_limit_d (awm_ctx *ctx, float *out_d, size_t n_frames, int n_channels, size_t first_sample,
                     const float *block_max_d, size_t first_block, size_t n_blocks);
int awm_add_d (awm_ctx *ctx, const float *pcm_in_d, float *out_d, size_t n_frames, int n_channels,
               const int8_t *frame_mod, double water_delta, int use_limiter);

/* add as a tile loop with bounded memory -- the reference's streaming add_stream_watermark (wmadd.cc:520-589) keeps one
 * frame (WatermarkSynth, wmadd.cc:173,220-222) and up to two limiter blocks (limiter.cc:51-64) of state; here the unit in
 * flight is a tile of `tile_frames1024` frames (>= 128) and the same state is carried from tile to tile inside the object.
 *   create:  payload / key tables are resolved once (Params as at this call).
 *   input:   device pointer where the NEXT tile's interleaved float32 samples are to be written (tile_frames1024 * 1024 * C).
 *   push:    `n_frames` samples per channel were written there; every tile but the last must be full; last = 1 ends the
 *            stream (n_frames may be 0).  Work is enqueued on the context's stream.  Returns how many tiles became final
 *            (0..3, in stream order): out_d[i] / out_frames[i] point into the object and stay valid until the next push.
 * The concatenated output is bit-identical to awm_add_watermark_d on the whole stream (tests: spans + halo == whole).
 *   create_at: the `zero_frames` argument of add_stream_watermark (wmcommon.hh:226; wmadd.cc:501-526, 574-580, limiter.cc:69-88):
 *            the stream starts `zero_frames` samples into the frame / watermark block / limiter block grid, as if that many zeros
 *            had been pushed before the caller's first sample and cut from the output again.  Whole frames of zeros are only
 *            counted (the table row of frame m becomes (4202 + zero_frames / 1024 + m) mod 4452, the limiter blocks keep their
 *            phase); the remaining zero_frames % 1024 zeros sit in front of the caller's samples inside the object, so the
 *            LAST tile's output may be up to 1023 frames longer than its input was (what hung over the tile before it). */
typedef struct awm_add_stream awm_add_stream;
int    awm_add_stream_create (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, int n_channels, size_t tile_frames1024,
                              awm_add_stream **out);
int    awm_add_stream_create_at (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, int n_channels, size_t tile_frames1024,
                                 size_t zero_frames, awm_add_stream **out);
void   awm_add_stream_destroy (awm_add_stream *s);
float *awm_add_stream_input (awm_add_stream *s);
int    awm_add_stream_push (awm_add_stream *s, size_t n_frames, int last, const float *out_d[3], size_t out_frames[3]);

/* I/O staging: RawConverter::from_raw / to_raw (rawconverter.cc:155-286) on the device, so that files cross PCIe in
 * their own sample format.  bit_depth 8/16/24/32 (integer) or 32/64 (float); encoding 0 signed, 1 unsigned, 2 float.
 * direct16 != 0 selects the reference's native little-endian signed-16 rule (truncate at 16 bit, what its stdout / raw
 * writers do); 0 = clip to 32 bit and keep the top bits (what writing through libsndfile does). */
int awm_pcm_decode_d (awm_ctx *ctx, const void *bytes_d, size_t n_values, int bit_depth, int encoding, int big_endian, float *out_d);
int awm_pcm_encode_d (awm_ctx *ctx, const float *in_d, size_t n_values, int bit_depth, int encoding, int big_endian, int direct16,
                      void *bytes_d);

/* SyncFinder::sync_fft (syncfinder.cc:560-605): dB magnitudes of bins 20..100, channels summed.
 * db_out_d: [frame_count][81] float32, have_out_d: [frame_count] bytes.
 * want_frames (host, may be NULL) and [first,last) (value indices of the non-silent range,
 * syncfinder.cc:155-169) reproduce the reference's skip rules. */
int awm_sync_fft_d (awm_ctx *ctx, const float *pcm_d, size_t n_frames, int n_channels,
                    size_t index, size_t frame_count, const char *want_frames,
                    size_t first, size_t last, float *db_out_d, char *have_out_d);

/* SyncFinder::search (syncfinder.cc:487-558) for one key: search_approx on the 4 shifts,
 * local mean, peak selection, search_refine, final selection.  Returns the number of scores
 * (<= max_out), sorted by index.  block_type: 0 = A, 1 = B. */
int awm_sync_search_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames,
                       int n_channels, int clip_mode, size_t max_out,
                       uint64_t *index, double *quality, int *block_type);
/* search_approx only (syncfinder.cc:171-256): all candidate scores in index order */
long awm_search_approx_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames,
                          int n_channels, int clip_mode, size_t max_out,
                          uint64_t *index, double *raw_quality, double *local_mean);

/* FFTAnalyzer::fft_range(index, 2226) + mix_decode (wmget.cc:67-108) for `n_blocks` block
 * start indices: raw soft bits in mix order, out[n_blocks][858].  ok[i] = 0 where the block
 * would read past the end (fft_range returns empty, wmcommon.cc:128-130). */
int awm_block_soft_bits_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames,
                           int n_channels, const uint64_t *index, size_t n_blocks, float *out, int *ok);

/* conv_decode_soft (convcode.cc:128-213) for a batch of equally typed blocks.
 * soft: [n][coded_len] normalised soft bits; bits_out: [n][coded_len/rate - 15]; error_out[n]. */
int awm_viterbi_decode (awm_ctx *ctx, int block_type, const float *soft, size_t coded_len, size_t n,
                        int *bits_out, float *error_out);

/* ---- pipeline level ------------------------------------------------------------------- */
typedef struct
{
  double   time;           /* seconds, incl. chunk offset */
  uint64_t sync_index;     /* sample index inside its chunk */
  double   sync_quality;
  int      block_type;     /* 0 A, 1 B, 2 AB */
  int      type;           /* 0 BLOCK, 1 CLIP, 2 ALL */
  float    decode_error;
  double   speed;
  int      bits[128];
  int      n_bits;
} awm_pattern;

/* add_watermark for n_clips independent inputs with one key and payload: what n_clips `audiowmark add` runs produce
 * (wmadd.cc:620-657), the clips dealt to the context's work lanes so that their small kernels overlap.  All clips at the
 * watermark rate (44100 Hz) with n_channels channels; out_d[i] holds n_frames[i] * n_channels floats. */
int awm_add_watermark_batch_d (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, size_t n_clips, const float *const *pcm_in_d,
                               float *const *out_d, const size_t *n_frames, int n_channels);

/* The batch entry points with ONE KEY PER CLIP (keys = n_clips * 16 bytes; BASELINE configs[4]: `--test-key k` for clip k).  The key
 * tables -- frame_mod for `add`; CLIP sync tables, mix table and bit order for `get` (wmcommon.cc:143-202, wmadd.cc:86-162,
 * syncfinder.cc:30-77) -- are built group by group while the device works on the previous group and are indexed per clip inside the
 * group's launches: `add`'s frame_mod tables ON THE DEVICE (K16, hip/keytab.hip: AES-128-CTR streams and the shuffles of one key
 * per workgroup, 256 keys per launch; the host contributes the AES key schedules, 176 bytes per key), `get`'s tables on host
 * threads, 64 clips per group, one copy per group.  Results per clip equal awm_add_watermark_d / awm_get_watermark_d with that
 * clip's key. */
int awm_add_watermark_batch_keys_d (awm_ctx *ctx, const uint8_t *keys, const char *payload_hex, size_t n_clips, const float *const *pcm_in_d,
                                    float *const *out_d, const size_t *n_frames, int n_channels);
int awm_get_watermark_batch_keys_d (awm_ctx *ctx, const uint8_t *keys, size_t n_clips, const float *const *pcm_d,
                                    const size_t *n_frames, int n_channels, int n_threads, size_t max_out_per_clip,
                                    awm_pattern *out, int *n_out);

/* add_watermark core (wmadd.cc:448-618) on resident PCM: out_d gets n_frames*C samples */
int awm_add_watermark_d (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex,
                         const float *pcm_in_d, float *out_d, size_t n_frames, int n_channels,
                         int sample_rate);
/* Streams at another sample rate.  The reference resamples them to 44.1 kHz with zita-resampler (hlen 16): the fixed-ratio
 * Resampler where it takes the two rates, else the VResampler with ratio new / old (ResamplerImpl::create,
 * resample.cc:233-270).  `get` decodes the resampled stream (WavChunkLoader, wavchunkloader.cc:70-71,200-216), `add` generates the
 * watermark at 44.1 kHz and resamples the watermark signal back (WatermarkResampler, wmadd.cc:353-430) -- the latter is
 * what awm_add_watermark_d does for sample_rate != 44100.  awm_resample_d is the former: out_d receives n_out_frames
 * frames of the stream the reference's loader would hand to the decoder, awm_resample_frames tells how many there are
 * (0: neither zita class takes the ratio, i.e. it is below 1 / 16 or above 256).  zita-resampler is not part of the reference
 * tree; its algorithm is restated from the library's description, bit parity with it is unpinned. */
size_t awm_resample_frames (awm_ctx *ctx, size_t n_frames, int rate_in, int rate_out);
int awm_resample_d (awm_ctx *ctx, const float *pcm_in_d, size_t n_frames, int n_channels, int rate_in, int rate_out,
                    float *out_d, size_t n_out_frames);
/* get_watermark core (wmget.cc:886-1013) on resident 44.1 kHz PCM: chunk loop, BlockDecoder,
 * ClipDecoder, merge + sort.  Returns the pattern count (<= max_out filled). */
int awm_get_watermark_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames,
                         int n_channels, size_t max_out, awm_pattern *out);
/* add_watermark followed by get_watermark of its output, one key ("watermark, then verify that the payload decodes": wmadd.cc:448-618
 * then wmget.cc:886-1013, the two commands the reference runs one after the other in its own tests, tests/block-decoder-test.sh:8-18).
 * out_d receives the watermarked stream (pcm_in_d != out_d), `out` the pattern list of `get` on it: exactly what
 * awm_add_watermark_d + awm_get_watermark_d return.  As ONE call the library owns the order of the two halves on the context's stream:
 * `get` starts a chunk as soon as the limiter has passed the chunk's last sample instead of behind the whole `add` (two separate calls
 * cannot: the caller may have queued other work on the buffer in between).  Returns the pattern count or an error code. */
int awm_add_get_watermark_d (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, const float *pcm_in_d, float *out_d,
                             size_t n_frames, int n_channels, int sample_rate, size_t max_out, awm_pattern *out);
/* The reference's get_watermark takes a LIST of keys (wmcommon.hh:228, `--key a --key b`, tests/key-test.sh:13-37): the file is
 * read once, the approximate dB matrices of a chunk / of a padded clip are computed once and shared by the keys
 * (syncfinder.cc:171-256), patterns are sorted by time with the key list order breaking ties (wmget.cc:288-316).
 * keys = n_keys * 16 bytes; key_of_pattern[j] (may be NULL) = position in that list of the key pattern j was found with. */
int awm_get_watermark_keys_d (awm_ctx *ctx, const uint8_t *keys, int n_keys, const float *pcm_d, size_t n_frames, int n_channels,
                              size_t max_out, awm_pattern *out, int *key_of_pattern);
/* get_watermark for a batch of independent inputs (BASELINE config 5: many short clips; the reference would run one
 * `audiowmark get` process per file, wmget.cc:886-1013 each).  Clip i = n_frames[i] frames at pcm_d[i] (device pointers),
 * all with n_channels channels.  Clips shorter than one block + 2 frames (51.7 s: only the ClipDecoder's START pass has work,
 * wmget.cc:769-884) are processed in groups of 64 whose padded copies lie side by side in one buffer -- every stage of the CLIP
 * search and of the decode is one launch per group; longer clips are spread over the context's work lanes (n_threads host
 * threads, <= 0: all lanes).  Patterns of clip i go to out[i * max_out_per_clip ...], their number (possibly >
 * max_out_per_clip) to n_out[i]; every clip's result equals awm_get_watermark_d on it.  Returns 0 or an error code. */
int awm_get_watermark_batch_d (awm_ctx *ctx, const uint8_t key[16], size_t n_clips, const float *const *pcm_d,
                               const size_t *n_frames, int n_channels, int n_threads, size_t max_out_per_clip,
                               awm_pattern *out, int *n_out);
/* decode() of one chunk only (wmget.cc:886-939) -- the unit `get` is sharded by */
int awm_decode_chunk_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames,
                        int n_channels, int first_chunk, size_t max_out, awm_pattern *out);

/* decode() for several chunks that live in one resident buffer (chunk i = frames [first_frame[i], first_frame[i] +
 * chunk_frames[i]) of pcm_d); device work is batched across the chunks, every chunk is searched / combined on its own
 * exactly like awm_decode_chunk_d.  first_is_stream_start != 0 runs the ClipDecoder on chunk 0.  Pattern times are
 * relative to their chunk; chunk_of_pattern[j] tells which chunk pattern j belongs to (patterns are ordered by chunk,
 * then time).  This is the per-rank unit of a sharded `get`. */
int awm_decode_chunks_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels,
                         int n_chunks, const uint64_t *first_frame, const uint64_t *chunk_frames, int first_is_stream_start,
                         size_t max_out, awm_pattern *out, int *chunk_of_pattern);

/* File level = the reference's own entry points add_watermark (key, infile, outfile, bits) and get_watermark (key_list,
 * infile, orig_pattern) (wmcommon.hh:226-228, wmadd.cc:620-657, wmget.cc:971-1013) with a GPU context.  The file is streamed
 * through page-locked staging buffers in its own sample format (bounded host memory: tile loop for `add` at 44.1 kHz, chunked
 * staging otherwise) and converted on the device.  raw_* = NULL: WAV (RIFF / RF64) by header, like --input-format auto;
 * else headerless PCM as with --format raw --raw-rate/--raw-channels/--raw-bits/--raw-encoding/--raw-endian.
 * awm_get_watermark_file returns the number of patterns found (at most max_out are written), < 0 on error. */
typedef struct { int n_channels, sample_rate, bit_depth, encoding /* 0 signed, 1 unsigned, 2 float */, big_endian; } awm_raw_format;
int awm_add_watermark_file (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, const char *in_path, const char *out_path,
                            const awm_raw_format *raw_in, const awm_raw_format *raw_out);
/* add_stream_watermark (key, in_stream, out_stream, bits, zero_frames) (wmcommon.hh:226, wmadd.cc:448-618) on files: the input is the
 * continuation of a stream that is `zero_frames` samples in (hls.cc:279 watermarks a segment this way); zero_frames = 0 is
 * awm_add_watermark_file.  At 44.1 kHz the start offset costs nothing (a frame counter and a limiter block phase); at other rates the
 * zeros are materialised in HBM in front of the input (the resamplers see them, resample.cc:150-168). */
int awm_add_stream_watermark_file (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, const char *in_path, const char *out_path,
                                   const awm_raw_format *raw_in, const awm_raw_format *raw_out, size_t zero_frames);
int awm_get_watermark_file (awm_ctx *ctx, const uint8_t key[16], const char *in_path, const awm_raw_format *raw_in,
                            size_t max_out, awm_pattern *out);
/* add_watermark (key, infile, outfile, bits) followed by get_watermark (key, outfile) ("watermark, then verify that the payload decodes":
 * wmadd.cc:620-657 + wmget.cc:971-1013; the file twin of awm_add_get_watermark_d): the input is read once and the output is never read
 * back -- the output stage decodes the bytes it has just encoded for the file (the samples as the file holds them, after its sample format's
 * quantisation) into HBM and `get` runs there.  Same file and same pattern list as the two calls; returns the number of patterns. */
int awm_add_get_watermark_file (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, const char *in_path, const char *out_path,
                                const awm_raw_format *raw_in, const awm_raw_format *raw_out, size_t max_out, awm_pattern *out);
int awm_get_watermark_keys_file (awm_ctx *ctx, const uint8_t *keys, int n_keys, const char *in_path, const awm_raw_format *raw_in,
                                 size_t max_out, awm_pattern *out, int *key_of_pattern);

/* chunk plan of WavChunkLoader (wavchunkloader.cc:54-163) for a stream of n_frames samples per channel:
 * chunk i covers [first_frame[i], first_frame[i] + chunk_frames[i]) and reports times offset by
 * time_offset[i] seconds.  Pure host.  Returns the chunk count (<= max_out filled).  The chunks are the
 * unit `get` is sharded by across GPUs. */
int awm_plan_chunks (size_t n_frames, size_t max_out, uint64_t *first_frame, uint64_t *chunk_frames, double *time_offset);
/* ResultSet::merge + sort (wmget.cc:215-316) over per-chunk pattern lists that were decoded elsewhere
 * (other ranks): `patterns` holds the chunks' patterns back to back in chunk order with times already
 * offset, chunk_count[i] patterns for chunk i.  Result written to out (<= max_out), count returned. */
int awm_merge_patterns (const uint8_t key[16], const awm_pattern *patterns, const int *chunk_count, int n_chunks,
                        size_t max_out, awm_pattern *out);

/* ---- one stream over several GPUs -------------------------------------------------------------------------------------------
 * The stream is the concatenation of the ranks' spans (rank order; every span but the last non-empty one a whole number of
 * 1024-sample frames), one context per GPU.  What the algorithm couples across a span boundary is all that travels:
 *   add   one frame each way (3-frame overlap-add, wmadd.cc:228-238) and the limiter's per-second maxima of the seconds that
 *         straddle span edges (limiter.cc:90-124): one exchange of edge frames + one max-reduction.
 *   get   The reference's chunks (wavchunkloader.cc:75-84) stay the semantic unit -- local mean, n_best and the A / B
 *         combination are per chunk -- but the WORK of a chunk is split by position: a rank computes the sync scores
 *         (syncfinder.cc:171-256), refines (:393-458) and extracts the soft bits (wmget.cc:67-108) of the candidate starts that lie in its
 *         span, for which it needs one block + 2 frames of its successor's samples (the "overlap stitch", 18 MB for stereo);
 *         the ranks that share a chunk exchange its raw scores (32 B per frame of audio: the "score gather"), select the
 *         candidates redundantly (deterministic), exchange the few refined scores and the blocks' soft bits (3.4 KB per block),
 *         and share the Viterbi decodes.  Rank 0 merges the patterns (wmget.cc:288-316).  Work per rank is proportional to its
 *         span (balanced to within a block), results are identical to awm_get_watermark_d on the whole stream.
 * The transport is the caller's: three callbacks (RCCL / torch.distributed in audiowmark_amd/sharded.py, hipMemcpyPeer between the
 * threads of one process in awm_multi_*).  Every rank calls the entry point with the same span list; all sizes are known on both
 * sides of every transfer.  Each callback returns 0 or an error (the entry point then fails with AWM_ERR_GENERIC). */
typedef struct awm_comm
{
  void *user;
  int   rank, world;
  /* point-to-point round: all transfers posted together, complete on return.  *_d: device memory (the data is ready on the
   * context's stream when the callback is entered, and the callback's writes are visible to that stream when it returns);
   * *_h: host memory (small control data). */
  int (*exchange_d) (void *user, int n_send, const void *const *send, const size_t *send_bytes, const int *send_to,
                     int n_recv, void *const *recv, const size_t *recv_bytes, const int *recv_from);
  int (*exchange_h) (void *user, int n_send, const void *const *send, const size_t *send_bytes, const int *send_to,
                     int n_recv, void *const *recv, const size_t *recv_bytes, const int *recv_from);
  /* element-wise maximum over all ranks of n unsigned 32 bit words in device memory, in place (the limiter maxima: non-negative
   * floats order like their bit patterns) */
  int (*all_reduce_max_u32_d) (void *user, uint32_t *data, size_t n);
} awm_comm;
/* add_watermark core for this rank's span (out_d: span_frames[rank] * C floats); 44.1 kHz streams only */
int awm_sharded_add_d (awm_ctx *ctx, const uint8_t key[16], const char *payload_hex, const float *pcm_in_d, float *out_d, int n_channels,
                       const uint64_t *span_frames /* [world] */, const awm_comm *comm);
/* get_watermark core; the merged pattern list arrives on rank 0 (return value = its length, at most max_out written), the other
 * ranks return 0 */
int awm_sharded_get_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, int n_channels, const uint64_t *span_frames /* [world] */,
                       const awm_comm *comm, size_t max_out, awm_pattern *out);
/* the plan behind awm_sharded_get_d, for tests and balance checks (pure host): for every (chunk, rank) the range of candidate start
 * frames [first_sf, first_sf + n_sf) of the chunk that rank works on.  Returns the number of entries (chunks * world). */
int awm_sharded_plan (const uint64_t *span_frames, int world, size_t max_out, int *chunk, int *rank, uint64_t *first_sf, uint64_t *n_sf);

/* The same on the GPUs of ONE process (the command line's --gpus / AWM_DEVICES): one context per device, one host thread per
 * context inside the call, hipMemcpyPeer as transport.  span i lives on ctxs[i]'s device at pcm_d[i] (several contexts may share
 * a device). */
int awm_multi_add_d (awm_ctx *const *ctxs, int n_ctx, const uint8_t key[16], const char *payload_hex, const float *const *pcm_in_d,
                     float *const *out_d, int n_channels, const uint64_t *span_frames);
int awm_multi_get_d (awm_ctx *const *ctxs, int n_ctx, const uint8_t key[16], const float *const *pcm_d, int n_channels,
                     const uint64_t *span_frames, size_t max_out, awm_pattern *out);

/* Batches of independent clips over the contexts of one process (BASELINE configs[4] on several GPUs): clip i lives on the device of
 * ctxs[ctx_of_clip[i]] and is handled there -- one host thread per context runs awm_add_watermark_batch_keys_d /
 * awm_get_watermark_batch_keys_d on its share, results come back in clip order.  Replicas: nothing travels between the devices
 * (reference: n_clips independent `audiowmark add` / `get` runs, wmadd.cc:620-657, wmget.cc:764-884).  keys = n_clips * 16 bytes, or
 * NULL with one_key != NULL: that key for every clip.  The settings in force for every context are the calling thread's / ctxs[0]'s. */
int awm_multi_add_watermark_batch_d (awm_ctx *const *ctxs, int n_ctx, const int *ctx_of_clip, const uint8_t *keys, const uint8_t *one_key,
                                     const char *payload_hex, size_t n_clips, const float *const *pcm_in_d, float *const *out_d,
                                     const size_t *n_frames, int n_channels);
int awm_multi_get_watermark_batch_d (awm_ctx *const *ctxs, int n_ctx, const int *ctx_of_clip, const uint8_t *keys, const uint8_t *one_key,
                                     size_t n_clips, const float *const *pcm_d, const size_t *n_frames, int n_channels,
                                     size_t max_out_per_clip, awm_pattern *out, int *n_out);

/* Helpers of a context: contexts on other GPUs that the file level `get` (awm_get_watermark_file / _keys_file, the command line) may
 * spread a long single-key stream over (awm_multi_get_d after the stream has been read through `ctx`; its spans travel device to
 * device).  The helpers stay the caller's; n_helpers = 0 takes them away.  The command line fills them from AWM_DEVICES=0,1,... */
int awm_ctx_set_helpers (awm_ctx *ctx, awm_ctx *const *helpers, int n_helpers);

/* ---- speed detection (reference wmspeed.cc:622-781) ------------------------------------
 * `get --detect-speed`: the reference looks for the replay speed (0.8 .. 1.25) of the watermark before decoding, by
 * correlating the sync pattern with a half-rate STFT of a 25 / 50 s clip over a grid of speeds, and decodes the stream a
 * second time stretched back to speed 1 (wmget.cc:886-927).  awm_set_speed_params switches that part of decode() on for
 * awm_get_watermark_d / awm_decode_chunk(s)_d (Params::detect_speed, detect_speed_patient, try_speed, test_speed;
 * wmcommon.hh:49-52); patterns found on the stretched stream carry the speed in awm_pattern.speed.
 * The stretch is zita-resampler's VResampler (resample.cc:96-125), restated like the fixed-ratio Resampler: bit parity
 * with the library is unpinned.  On the device the resampler's phase comes from the exact product instead of zita's
 * accumulated double (differs by its accumulated rounding only), so stretched PCM agrees to ~1e-7, not bit for bit. */
void awm_set_speed_params (int detect_speed, int detect_speed_patient, double try_speed, double test_speed);
/* resample_ratio_truncate (resample.cc:96-119): frames the call produces / the call itself (out_d: n_out_frames frames) */
size_t awm_resample_ratio_frames (size_t n_frames, int n_channels, int rate, double ratio, double max_in_seconds);
int awm_resample_ratio_d (awm_ctx *ctx, const float *pcm_in_d, size_t n_frames, int n_channels, int rate, double ratio,
                          double max_in_seconds, float *out_d, size_t n_out_frames);
/* detect_speed for one key (wmspeed.cc:622-781) on resident PCM: returns 1 if decoding at *speed_out should be tried
 * (quality > 0.4 and speed outside 0.9999 .. 1.0001), 0 if not, < 0 on errors; speed / quality are filled in either way */
int awm_detect_speed_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels, int rate,
                        int patient, double *speed_out, double *quality_out);
/* the pieces, for parity tests: get_best_clip_location (wmspeed.cc:555-577); SpeedSync::prepare_mags for one centre speed
 * (:204-268; out is host memory, [row][510 sync frames sorted by frame][umag, dmag], returns the row count); one
 * run_search pass (:461-492, 683-719; scores sorted by speed, returns their number) */
int awm_speed_clip_location_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels, int rate,
                               double seconds, int candidates, double *location);
int awm_speed_mags_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels, int rate,
                      double clip_location, double center, double seconds, size_t max_rows, float *out);
int awm_speed_scan_d (awm_ctx *ctx, const uint8_t key[16], const float *pcm_d, size_t n_frames, int n_channels, int rate,
                      double clip_location, double seconds, double step, int n_steps, int n_center_steps,
                      const double *speeds, int n_speeds, size_t max_out, double *out_speed, double *out_quality);

/* host side pieces of the search (pure host, no GPU): select_n_best_scores (wmspeed.cc:494-531; in place, returns the new
 * count), score_smooth_find_best (:397-428), and the two halves of get_clip_locations (:533-553): the sample positions that
 * are hashed, and the candidate locations drawn from the generator re-seeded with the SHA-1 of those samples
 * (Random::seed_from_hash, random.cc:184-190) */
int    awm_speed_select_n_best (double *speed, double *quality, int count, int n);
double awm_speed_smooth_best (const double *speed, const double *quality, int count, double step, double distance);
size_t awm_speed_clip_positions (const uint8_t key[16], size_t n_values, size_t max_out, uint64_t *positions);
int    awm_speed_clip_candidates (const uint8_t key[16], const float *hashed_values, size_t n, int candidates, double *locations);

/* A / B hooks for measurements (tools/gpu_variants.py): the previous formulation of two kernels stays selectable so that a change can be
 * timed against it in one process, on the same data, with the same clocks.  Results are bit-identical either way (tests).
 *   viterbi_super: 1 (default) three trellis rounds per launch with the metrics exchanged through LDS / 0 one round per launch
 *   sliding3:      1 (default) refinement with three bins of one channel per lane / 0 two bins of both channels (stereo) */
void awm_debug_set_viterbi_super (int on);
void awm_debug_set_viterbi_persistent (int on); /* K8: 1 ONE launch per batch of decodes (8 resident workgroups per decode meeting at a counter in global
                                                 * memory every 12 trellis steps) | 0 the chain of 14 dependent launches | -1 (default) chosen per process from
                                                 * the measured cost of a dependent launch on this host (awm_ctx_create probes it): the chain where launches
                                                 * are cheap, the one-launch kernel where they are not.  Bits and error values are identical. */
double awm_debug_dependent_launch_us (void);    /* the probe's result (microseconds per empty dependent launch; -1 before the first context) */
int  awm_debug_viterbi_one_launch_in_use (void);
void awm_debug_set_sliding3 (int on);          /* (rounds 3 - 5: refine form 3 / 0) */
/* K4s, the refinement's sliding DFT for stereo streams (reference SyncFinder::search_refine -> sync_fft, syncfinder.cc:393-458, 560-605):
 *   0  two bins of both channels per lane | 3  three bins of one channel per lane (rounds 3 - 5) | 4 (default) the same arithmetic in a
 *   straight-line step with the wave-uniform rules on the scalar unit: outputs of 0, 3, 4 are identical to the last bit |
 *   5  the update term of the recurrence accumulated in float (state, rotation and Hann combination stay double): the level of a float
 *      FFT, which is what the reference's FFTW is; NOT bit-identical to 4 -- gated by the census in DESIGN.md section 4 */
void awm_debug_set_refine_form (int form);
int  awm_debug_refine_form (void);
void awm_debug_set_k4s_ablate (int flags);     /* measurement only (tools/gpu_k4s_alone.py): 8 = awm_debug_sync_db_sliding_d runs forms 4 / 5 without their stores */
/* K4s alone on resident PCM (stereo or mono): stream i = `count` (<= 65) windows of 1024 samples starting at base_d[i] + 8 o; writes
 * out_d[i][band 0..80][ld] (dB summed over the channels) in the form in force.  For the tests that pin forms 0 / 3 / 4 against each other.
 * ld = 64 (measurement): the refinement's gathered layout with a synthetic table instead (bands 0..59 are the rows): out_d[i][60][64], and with
 * forms 4 / 5 the 65th values at out_d + n_streams * 60 * 64 (n_streams * 60 floats more). */
int  awm_debug_sync_db_sliding_d (awm_ctx *ctx, const float *pcm_d, size_t n_frames, int n_channels, const long long *base_d, size_t n_streams,
                                  int count, int ld, float *out_d);
void awm_debug_set_soft_bits_generic (int on); /* K7: one thread per soft bit for every shape (the fallback kernel) | four bits per wave */
void awm_debug_set_chunk_stagger (int mode); /* get: phase offset between the chunk lanes -- 0 all chunks start together | 1 chunk i + 1 behind chunk i's
                                             * dB kernel | 2 behind its scan | -1 (default) 1 for streams of up to `lanes` chunks, 2 for longer ones */
void awm_debug_set_resample_phase (int on);  /* K10: 1 (default) the phase-per-thread kernel for stereo 48 <-> 44.1 kHz | 0 the generic kernel (outputs identical) */
void awm_debug_set_get_overlap (int on);    /* file level get: 1 (default) the chunks start while the rest of the stream is still crossing PCIe (a loader thread, a mark per
                                             * tile; streams of announced length at 44.1 kHz with two chunks or more) | 0 the whole stream first (rounds 1 - 5) */
void awm_debug_set_speed_compare_wide (int on); /* K14: 1 all relative speeds of a centre (<= 12) in one thread / 0 (default) groups of six: measured slower, see hip/speed.hip */
void awm_debug_set_speed_compare_fold (int on); /* K14: 1 (default) the groups of six of a (state range, centre) are neighbouring workgroups on one XCD / 0 a grid dimension of their own */
void awm_debug_set_resample_var_mode (int mode); /* K12: bit 0 the stereo input window of a tile through LDS | bit 1 a workgroup keeps its coefficient
                                             * table for several tiles (default 2; outputs identical) */
void awm_debug_set_speed_overlap (int on);   /* get with a speed search: 1 (default) the plain decode of the chunks runs beside the speed part, on lanes
                                             * of its own | 0 after it (the reference's order of work; the pattern lists are the same) */
int  awm_debug_frame_mod_tables_d (awm_ctx *ctx, const uint8_t *keys, size_t n_keys, const char *payload_hex, int8_t *tables_out);
                                             /* the frame_mod tables as K16 builds them (wmadd.cc:86-162), n_keys x 2 x 2226 x 81 bytes to host memory:
                                              * for the test that they equal awm_tab_frame_mod key by key */
int  awm_debug_clip_key_tables_check_d (awm_ctx *ctx, const uint8_t *keys, size_t n_keys, long long mismatch_out[9]);
                                             /* the tables `get` needs per key of a clip batch (K16g: sync chains, row frames, want list, gathered layout,
                                              * mix entries, bit order) built on the device, group by group, against the host's build of the same tables:
                                              * mismatch_out[i] = differing elements per table (all 0 = identical) */
void awm_debug_set_key_tables_on_device (int on);   /* batches with one key per clip: 0 the tables from host threads | `add`'s frame_mod tables (K16) and `get`'s sync / mix / bit order tables (K16g) built on the device: 1 (default) `get` builds a group's tables one group ahead of its lane, 2 the tables of all (up to 4096) keys first (measured: the same time) */
void awm_debug_set_merge_decodes (int on);  /* get of a stream of 2 - 4 chunks: the chunks' Viterbi jobs as ONE batch at the end | per chunk (default: the step is faster) */
void awm_debug_set_add_batched (int on);   /* add of a batch of stereo clips: 2 (default) ONE launch per stage for many clips (block maxima, K2, limiter table, limiter: blockIdx.y =
                                            * the clip, spans sized for the batch), with a key per clip after the tables of all (up to 4096) keys | 1 the same with a
                                            * group's tables built while the previous group of 256 clips is watermarked | 0 four launches per clip on eight lanes; the
                                            * outputs are the same */
void awm_debug_set_add_slab_mb (int mb);   /* add: 0 (default) one fused add over the stream, then the limiter | > 0: in slabs of that many MB (cache experiment) */
void awm_debug_set_fft_pair (int on);      /* stereo add: both channels' transforms pipelined in one wave (default) | one after the other */
void awm_debug_set_clip_poison (int on);    /* clip batches: the padded slices are filled with NaNs before the copies are written (the copy writes a clip and 2048
                                            * frames of zeros on either side, not the rest of the padding: a consumer that read further would change its result) */
void awm_debug_set_clip_pad_margin (int frames); /* clip batches: frames of zeros written on either side of a clip (default and minimum 2048; one slice = 6693 frames
                                            * or more: whole slices, the A side of the measurement in tools/gpu_clip_margin_ab.py) */
void awm_debug_set_staged_threads (int n);  /* clip batches: host threads (= lanes) working on groups of clips; 0 = default (4; 2 with a key per clip only where the
                                            * tables cannot come from the device and host threads build them) */
void awm_debug_clip_key_timing (double us_out[3]); /* get with a key per clip, summed over the batch's host threads since the last call: [0] microseconds waiting for
                                            * a group's key tables (built on host threads one group ahead), [1] packing + uploading them, [2] groups */
double awm_debug_time_group_key_tables (int n_keys, int threads); /* host only: wall milliseconds the key tables of one group of n_keys clips take to build
                                            * (threads <= 0: 64 as shipped) */
void awm_debug_set_group_fallback (int on); /* clip batches: every clip takes the sequential peak selection on its own slice (normally the rare clips whose peak
                                            * lists overflow the grouped selection) */
void awm_debug_alloc_stats (long *dev_allocs, double *dev_ms, long *pinned_allocs, double *pinned_ms);
                                           /* process-wide census of hipMalloc / hipHostMalloc calls made by the library's grow-only buffers and the
                                            * time the runtime took for them (what a first call pays); any pointer may be NULL */
void awm_debug_file_timing (double ms_out[8]);
                                           /* where the calling thread's time went in its last file level `add` at the watermark rate (milliseconds):
                                            * [0] set-up (add stream, rings), [1] waiting for input tiles, [2] waiting for a free output slot, [3] queueing GPU
                                            * work, [4] the final wait for the GPU, [5] the final wait for the writers, [6] tear-down, [7] handing output tiles on
                                            * (includes [2]) */
void awm_debug_set_io_flags (int flags);   /* file level add / get, host side (host/wmfile.cc): bit 0 regular INPUT files are read by the I/O workers through the
                                            * stream's raw_region (else one reader thread in stream order, as for pipes) | bit 3 regular OUTPUT files of known
                                            * length are written by the workers (else one writer thread in stream order) -- then bit 1: through ONE shared mapping
                                            * of the file (else pwrite per part), bit 2: MADV_POPULATE_WRITE before the copy.  Default 1: creating the pages of
                                            * one file is serial in the kernel, the writer thread is at that floor (profiles/r05/io_probe.txt); the bytes written
                                            * are the same either way. */

/* Host threads that copy between the page cache and the page-locked staging rings of the file level calls (awm_add_watermark_file,
 * awm_get_watermark_file, the command line): 0 (default) = min (16, cores); one pool per process.  The reference reads and writes on
 * its one thread (wavchunkloader.cc:196-222, rawconverter.cc, stdoutwavoutputstream.cc). */
void awm_set_io_threads (int n);

/* --quiet (reference audiowmark.cc:1020-1023): the "Input: / Output: / Message: ..." information lines of add_watermark off */
void awm_set_quiet (int quiet);

/* ---- parameters (reference Params, wmcommon.hh:33-89) ------------------------------------------------------------------
 * The reference keeps its settings in static members of Params: one set per process.  This library has one process-wide set
 * with the same names and defaults -- awm_set_params / awm_set_speed_params / awm_set_global_params change it -- and every
 * context may carry its OWN set (awm_ctx_set_params), which is in force for all work entered through that context, also on the
 * helper threads the library starts for it.  Two contexts with different strength / thresholds / --hard can therefore work side
 * by side in one process; a context without its own set follows the process-wide one as of each call. */
void awm_set_params (double water_delta, int mix, int frames_per_bit, int test_no_limiter,
                     double sync_threshold2, int n_best, double chunk_size_min);
typedef struct
{
  size_t struct_size;            /* sizeof (awm_params), filled by awm_params_init / awm_ctx_get_params */
  double water_delta;            /* --strength / 1000            (Params::water_delta, default 0.01) */
  int    mix;                    /* 0 = --linear                 (Params::mix, 1) */
  int    hard;                   /* --hard: hard decode bits     (Params::hard, 0; wmget.cc:40-65) */
  int    strict;                 /* --strict: message length must equal payload_size, `add` refuses clipped input (Params::strict, 0) */
  int    snr;                    /* --snr: `add` reports the SNR (Params::snr, 0; file level only) */
  int    payload_size;           /* bits (Params::payload_size, 128; other sizes are refused by the compute entry points) */
  int    frames_per_bit;         /* (Params::frames_per_bit, 2: --frames-per-bit, reference audiowmark.cc:675; the compute entry points take 1 .. 8) */
  double sync_threshold2;        /* --sync-threshold (Params::sync_threshold2, 0.35) */
  int    get_n_best;             /* (Params::get_n_best, 8) */
  double get_chunk_size;         /* --chunk-size, minutes (Params::get_chunk_size, 30) */
  int    detect_speed, detect_speed_patient;   /* --detect-speed, --detect-speed-patient */
  double try_speed, test_speed;  /* --try-speed, --test-speed (-1: unset) */
  int    test_cut, test_no_sync, test_no_limiter, test_truncate;     /* the reference's --test-* knobs */
} awm_params;
void awm_params_init (awm_params *p);                               /* the reference's defaults */
int  awm_set_global_params (const awm_params *p);                   /* the process-wide set */
int  awm_ctx_set_params (awm_ctx *ctx, const awm_params *p);        /* p == NULL: back to the process-wide set */
int  awm_ctx_get_params (const awm_ctx *ctx, awm_params *out);      /* the set in force for ctx (ctx == NULL: the process-wide one) */

/* `add --snr` (reference wmadd.cc:553-563, 591-592): between begin and end every `add` of the context accumulates the power of its
 * input and of the watermark signal (mix - input, BEFORE the limiter, sums in double); SNR = 10 log10 (signal / delta). */
int  awm_ctx_snr_begin (awm_ctx *ctx);
int  awm_ctx_snr_end (awm_ctx *ctx, double *signal_power, double *delta_power);

#ifdef __cplusplus
}
#endif
#endif /* AWM_HIP_H */
