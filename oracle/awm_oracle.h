/* oracle/awm_oracle.h -- C view of the CPU restatement of audiowmark's spectral path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (audiowmark_amd/) never does.
 *
 * The entry points mirror oracle/ref_shim/ref_capi.h one to one (orc_X <-> ref_X) so that the same
 * test can be checked against the restatement (always buildable from this repository) and against
 * the compiled reference (oracle/_ref, only where /root/reference exists).  Parity of the
 * restatement itself is pinned by tests/test_oracle_*.py: known-answer vectors captured from the
 * reference's own programs, fixtures generated from oracle/_ref
 * (tests/golden/, script tests/golden/make_golden.py) and, where oracle/_ref is present, direct
 * comparison.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

void   orc_set_params (double water_delta, int mix, int frames_per_bit, int test_no_limiter,
                       double sync_threshold2, int n_best, double chunk_size_min);
void   orc_set_threads (int n_threads);     /* worker threads for the sync search (results do not depend on it) */

void   orc_random_u64 (const uint8_t key[16], uint64_t seed, int stream, size_t n, uint64_t *out);
void   orc_random_double (const uint8_t key[16], uint64_t seed, int stream, size_t n, double *out);
void   orc_gen_noise (const uint8_t key[16], size_t n_values, float *out);

void   orc_up_down (const uint8_t key[16], int stream, int f, int up[30], int down[30]);
void   orc_bit_pos (const uint8_t key[16], int *pos);
size_t orc_mix_entries (const uint8_t key[16], int *frame_up_down);
void   orc_window (size_t n, float *out);
void   orc_synth_window (float *out /* [3072] */);
size_t orc_bit_order (const uint8_t key[16], size_t n, unsigned *order);
size_t orc_conv_encode (int block_type, const int *bits, size_t n, int *out);
size_t orc_conv_decode_soft (int block_type, const float *coded, size_t n, int *out, float *error_out);
int    orc_frame_mod (const uint8_t key[16], const char *payload_hex, int ab, uint8_t *out /* [2226*101] */);
int    orc_sync_bits (const uint8_t key[16], int clip_mode, int *out);

int    orc_fft_range (const float *samples, size_t n_values, int n_channels, size_t start_index, size_t frame_count, float *out);
void   orc_ifft (size_t n, const float *spect, float *out);

int    orc_add (const uint8_t key[16], const float *samples, size_t n_frames, int n_channels, int sample_rate,
                const char *payload_hex, float *out, size_t *out_frames, double *snr_db);

/* sample rates other than 44100 Hz (zita-resampler restated, PARITY UNPINNED): orc_add accepts them; this is the
 * 44.1 kHz stream `get` decodes for such a file */
size_t orc_resample (const float *samples, size_t n_frames, int n_channels, int rate_in, int rate_out, size_t max_out_frames, float *out);

int    orc_sync_fft (const float *samples, size_t n_values, int n_channels, size_t index, size_t frame_count,
                     const char *want_frames, size_t first, size_t last, float *db_out, char *have_out);
double orc_sync_decode (const uint8_t key[16], int clip_mode, size_t start_frame,
                        const float *db, size_t n_db, const char *have, size_t n_have);
int    orc_sync_search (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int clip_mode,
                        size_t max_out, uint64_t *index, double *quality, int *block_type);
size_t orc_search_approx (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int clip_mode,
                          size_t max_out, uint64_t *index, double *raw_quality, double *local_mean);
int    orc_mix_decode (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, size_t index, float *out);

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
} orc_pattern;

int    orc_decode_chunk (const uint8_t key[16], const float *samples, size_t n_values, int n_channels,
                         int first_chunk, size_t max_out, orc_pattern *out);
int    orc_get (const uint8_t key[16], const float *samples, size_t n_values, int n_channels,
                size_t max_out, orc_pattern *out);

/* ---- speed detection (wmspeed.cc) and the VResampler paths (resample.cc:96-125); zita-resampler restated ---------- */
void   orc_set_speed_params (int detect_speed, int patient, double try_speed);      /* --detect-speed, --detect-speed-patient, --try-speed */
size_t orc_resample_ratio (const float *samples, size_t n_frames, int n_channels, int rate, double ratio, double max_in_seconds,
                           size_t max_out_frames, float *out);                        /* resample_ratio_truncate */
double orc_speed_clip_location (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate,
                                double seconds, int candidates);                      /* wmspeed.cc:533-577 */
int    orc_speed_mags (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate,
                       double clip_location, double center, double seconds, size_t max_rows, float *out /* [row][510][2] */);
int    orc_speed_scan (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate,
                       double clip_location, double seconds, double step, int n_steps, int n_center_steps,
                       const double *speeds, int n_speeds, size_t max_out, double *out_speed, double *out_quality);
int    orc_speed_select_n_best (double *speed, double *quality, int count, int n);    /* wmspeed.cc:494-531 */
double orc_speed_smooth_best (const double *speed, const double *quality, int count, double step, double distance);
/* detect_speed for one key (wmspeed.cc:622-781): returns 1 if the speed passes the thresholds (quality > 0.4, more than
 * 1e-4 away from 1); speed / quality are filled in either way.  -1: the reference would exit with "failed to setup
 * vresampler" (every score 0, e.g. digital silence: the second pass is asked to search around speed 0) */
int    orc_detect_speed (const uint8_t key[16], const float *samples, size_t n_values, int n_channels, int rate, int patient,
                         double *speed_out, double *quality_out);

#ifdef __cplusplus
}
#endif
