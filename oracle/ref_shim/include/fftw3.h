/* oracle/ref_shim: the FFTW3 single-precision entry points the reference calls
 * (fft.cc:57-67,85,91).  FFTW itself is not available in this image; the
 * implementation in ../fftw_shim.cc evaluates the same transforms (r2c forward,
 * c2r backward UNNORMALISED, FFTW sign convention) in double precision and
 * rounds the result to float once.  TEST INFRASTRUCTURE ONLY. */
#pragma once
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef float fftwf_complex[2];
typedef struct awm_shim_plan *fftwf_plan;
#define FFTW_ESTIMATE (1U << 6)
#define FFTW_PRESERVE_INPUT (1U << 4)
void *fftwf_malloc (size_t n);
void fftwf_free (void *p);
fftwf_plan fftwf_plan_dft_r2c_1d (int n, float *in, fftwf_complex *out, unsigned flags);
fftwf_plan fftwf_plan_dft_c2r_1d (int n, fftwf_complex *in, float *out, unsigned flags);
void fftwf_execute_dft_r2c (const fftwf_plan p, float *in, fftwf_complex *out);
void fftwf_execute_dft_c2r (const fftwf_plan p, fftwf_complex *in, float *out);
void fftwf_destroy_plan (fftwf_plan p);
#ifdef __cplusplus
}
#endif
