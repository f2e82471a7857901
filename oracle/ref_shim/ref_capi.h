/* oracle/ref_shim/ref_capi.h -- C view of the UNMODIFIED reference (oracle/_ref/libawm_ref.so).
 *
 * Every entry point is a thin wrapper (ref_capi_*.cc) that calls the reference's own
 * functions/classes compiled from /root/reference/src where they lie.
 * TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* global Params knobs of the reference (wmcommon.hh:33-89) */
void   ref_set_params (double water_delta, int mix, int frames_per_bit, int test_no_limiter,
                       double sync_threshold2, int n_best, double chunk_size_min);
void   ref_set_quiet (int quiet);

/* random.cc: Random(key, seed, stream)() x n */
void   ref_random_u64 (const uint8_t key[16], uint64_t seed, int stream, size_t n, uint64_t *out);
void   ref_random_double (const uint8_t key[16], uint64_t seed, int stream, size_t n, double *out);
/* audiowmark.cc:399-417 test-gen-noise, float samples before quantisation */
void   ref_gen_noise (const uint8_t key[16], size_t n_values, float *out);

/* wmcommon.{hh,cc} tables */
void   ref_up_down (const uint8_t key[16], int stream, int f, int up[30], int down[30]);
void   ref_bit_pos (const uint8_t key[16], int *pos /* [2226]: sync_frame(0..509) then data_frame(0..1715) */);
size_t ref_mix_entries (const uint8_t key[16], int *frame_up_down /* [51480*3] */);
void   ref_window (size_t n, float *out);
size_t ref_bit_order (const uint8_t key[16], size_t n, unsigned *order);
/* convcode.cc */
size_t ref_conv_encode (int block_type, const int *bits, size_t n, int *out);
size_t ref_conv_decode_soft (int block_type, const float *coded, size_t n, int *out, float *error_out);
/* wmadd.cc:148-162 init_frame_mod_vec; out[2226*101] in {0 KEEP,1 UP,2 DOWN} */
int    ref_frame_mod (const uint8_t key[16], const char *payload_hex, int ab, uint8_t *out);
/* syncfinder.cc:30-77; out rows: frame, up[30], down[30]  -> returns rows per bit (85 or 170) */
int    ref_sync_bits (const uint8_t key[16], int clip_mode, int *out /* [6*rows*61] */);

/* wmcommon.cc:91-141 */
int    ref_fft_range (const float *samples, size_t n_values, int n_channels, size_t start_index, size_t frame_count,
                      float *out /* [frame_count*C*513*2] */);
/* fft.cc c2r (unnormalised) of one 513-bin spectrum */
void   ref_ifft (size_t n, const float *spect /* [(n/2+1)*2] */, float *out /* [n] */);

/* wmadd.cc:448-618 through in-memory streams; returns rc of add_stream_watermark; out has n_frames*C values */
int    ref_add (const uint8_t key[16], const float *samples, size_t n_frames, int n_channels, int sample_rate,
                const char *payload_hex, float *out, size_t *out_frames, double *snr_db);
/* the same with add_stream_watermark's zero_frames argument (wmadd.cc:448, 501-526, 574-580) */
int    ref_add_at (const uint8_t key[16], const float *samples, size_t n_frames, int n_channels, int sample_rate,
                   const char *payload_hex, size_t zero_frames, float *out, size_t *out_frames);

/* RawConverter (rawconverter.cc:73-286): encoding 0 signed / 1 unsigned / 2 float; to_raw: floats -> bytes, else bytes -> floats */
int    ref_raw_convert (int bit_depth, int encoding, int big_endian, int to_raw, const void *in, void *out, size_t n_values);

/* syncfinder.cc private pieces */
int    ref_sync_fft (const float *samples, size_t n_values, int n_channels, size_t index, size_t frame_count,
                     const char *want_frames /* may be NULL */, size_t first, size_t last,
                     float *db_out /* [frame_count*81] */, char *have_out);
double ref_sync_decode (const uint8_t key[16], int clip_mode, size_t start_frame,
                        const float *db, size_t n_db, const char *have, size_t n_have);
/* SyncFinder::search; returns count; arrays sized max_out */
int    ref_sync_search (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int clip_mode,
                        size_t max_out, uint64_t *index, double *quality, int *block_type);
/* search_approx only (scores sorted by index, local mean filled): returns count */
size_t ref_search_approx (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int clip_mode,
                          size_t max_out, uint64_t *index, double *raw_quality, double *local_mean);

/* wmget.cc:67-108 mix_decode over fft_range(index, 2226); out[858] raw soft bits (mix order, before randomize_bit_order) */
int    ref_mix_decode (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, size_t index, float *out);

typedef struct
{
  double   time;
  uint64_t sync_index;
  double   sync_quality;
  int      block_type;    /* 0 a, 1 b, 2 ab */
  int      type;          /* 0 BLOCK, 1 CLIP, 2 ALL */
  float    decode_error;
  double   speed;
  int      bits[128];
  int      n_bits;
} ref_pattern;

/* wmget.cc:886-939 decode() on one chunk (BlockDecoder + ClipDecoder if first_chunk); returns count */
int    ref_decode_chunk (const uint8_t key[16], const float *samples, size_t n_values, int n_channels,
                         int first_chunk, size_t max_out, ref_pattern *out);
/* whole get_watermark chunk loop (wavchunkloader.cc:101-163 arithmetic, wmget.cc:971-1013) on in-memory
 * 44.1 kHz data; patterns merged + sorted like the reference; returns count */
int    ref_get (const uint8_t key[16], const float *samples, size_t n_values, int n_channels,
                size_t max_out, ref_pattern *out);
/* the same for a file of another sample rate: the reference's WavChunkLoader resamples to 44.1 kHz (restated zita) */
int    ref_get_rate (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int sample_rate,
                     size_t max_out, ref_pattern *out);

/* ---- speed detection (wmspeed.cc) and VResampler paths (resample.cc:96-125); zita-resampler restated ---------- */
void   ref_set_speed_params (int detect_speed, int patient, double try_speed);
/* resample_ratio_truncate (resample.cc:96-119); returns the output length in frames */
size_t ref_resample_ratio (const float *samples, size_t n_frames, int n_channels, int rate, double ratio, int new_rate,
                           double max_in_seconds, size_t max_out_frames, float *out);
/* get_best_clip_location (wmspeed.cc:555-577) */
double ref_speed_clip_location (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate,
                                double seconds, int candidates);
/* SpeedSync::prepare_mags (wmspeed.cc:204-268) for one centre speed; out[row][510][2] = umag, dmag; returns rows */
int    ref_speed_mags (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate,
                       double clip_location, double center, double seconds, size_t max_rows, float *out);
/* one run_search pass (wmspeed.cc:683-719): scores of all centre / relative speeds sorted by speed; returns count */
int    ref_speed_scan (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate,
                       double clip_location, double seconds, double step, int n_steps, int n_center_steps,
                       const double *speeds, int n_speeds, size_t max_out, double *out_speed, double *out_quality);
int    ref_speed_select_n_best (double *speed, double *quality, int count, int n);               /* wmspeed.cc:494-531 */
double ref_speed_smooth_best (const double *speed, const double *quality, int count, double step, double distance); /* :397-428 */
/* detect_speed (wmspeed.cc:622-781) for one key; returns the number of results (0 or 1) */
int    ref_detect_speed (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate, int patient,
                         double *speed_out);

#ifdef __cplusplus
}
#endif
