/* oracle/ref_shim/fftw_shim.cc -- stand-in for the FFTW3f calls of the reference
 * (fft.cc:57-67 plan creation, :85 r2c execute, :91 c2r execute).
 *
 * FFTW is not installed in this image.  The transforms are
 * evaluated in DOUBLE precision (radix-2 on the half-length complex sequence + real
 * split) and rounded to float once, i.e. the result is the correctly-rounded-ish
 * "ideal" single precision FFT; FFTW's own float result differs from it only by
 * float rounding noise (~1e-7 relative), see SURVEY.md Appendix C.
 * Conventions follow FFTW: forward exponent -1, c2r unnormalised, imaginary parts of
 * the DC and Nyquist bins ignored by c2r.   TEST INFRASTRUCTURE ONLY.
 */
#include "fftw3.h"
#include <math.h>
#include <stdlib.h>
#include <vector>
#include <complex>
#include <map>
#include <mutex>

namespace {
typedef std::complex<double> cd;

struct HalfPlan
{
  int n = 0;                 /* real length */
  int h = 0;                 /* n / 2 */
  std::vector<int> rev;      /* bit reversal for length h */
  std::vector<cd>  tw;       /* e^{-2 pi i k / h}, k < h/2 */
  std::vector<cd>  split;    /* e^{-2 pi i k / n}, k <= h */
};

std::mutex plan_mutex;
std::map<int, HalfPlan *> plans;

HalfPlan *
get_plan (int n)
{
  std::lock_guard<std::mutex> lg (plan_mutex);
  HalfPlan *&p = plans[n];
  if (!p)
    {
      p = new HalfPlan();
      p->n = n;
      p->h = n / 2;
      int bits = 0;
      while ((1 << bits) < p->h) bits++;
      p->rev.resize (p->h);
      for (int i = 0; i < p->h; i++)
        {
          int r = 0;
          for (int b = 0; b < bits; b++)
            if (i & (1 << b)) r |= 1 << (bits - 1 - b);
          p->rev[i] = r;
        }
      p->tw.resize (p->h / 2);
      for (int k = 0; k < p->h / 2; k++)
        p->tw[k] = cd (cos (-2 * M_PI * k / p->h), sin (-2 * M_PI * k / p->h));
      p->split.resize (p->h + 1);
      for (int k = 0; k <= p->h; k++)
        p->split[k] = cd (cos (-2 * M_PI * k / n), sin (-2 * M_PI * k / n));
    }
  return p;
}

/* in-place forward complex FFT of length h (input already bit-reversed) */
void
cfft (const HalfPlan *p, cd *a)
{
  const int h = p->h;
  for (int len = 2; len <= h; len <<= 1)
    {
      const int half = len / 2, step = h / len;
      for (int i = 0; i < h; i += len)
        for (int j = 0; j < half; j++)
          {
            cd u = a[i + j], v = a[i + j + half] * p->tw[j * step];
            a[i + j] = u + v;
            a[i + j + half] = u - v;
          }
    }
}
} // namespace

struct awm_shim_plan { int n; int dir; };

extern "C" {

void *fftwf_malloc (size_t n) { void *p = nullptr; if (posix_memalign (&p, 64, n ? n : 1)) return nullptr; return p; }
void fftwf_free (void *p) { free (p); }

fftwf_plan
fftwf_plan_dft_r2c_1d (int n, float *, fftwf_complex *, unsigned)
{
  if (n < 4 || (n & (n - 1))) return nullptr;
  get_plan (n);
  return new awm_shim_plan { n, -1 };
}
fftwf_plan
fftwf_plan_dft_c2r_1d (int n, fftwf_complex *, float *, unsigned)
{
  if (n < 4 || (n & (n - 1))) return nullptr;
  get_plan (n);
  return new awm_shim_plan { n, +1 };
}
void fftwf_destroy_plan (fftwf_plan p) { delete p; }

void
fftwf_execute_dft_r2c (const fftwf_plan pl, float *in, fftwf_complex *out)
{
  const HalfPlan *p = get_plan (pl->n);
  const int h = p->h;
  std::vector<cd> z (h);
  for (int i = 0; i < h; i++)
    z[p->rev[i]] = cd (in[2 * i], in[2 * i + 1]);
  cfft (p, z.data());
  /* real split: X[k] = (Z[k] + conj Z[h-k]) / 2 - i/2 * W^k * (Z[k] - conj Z[h-k]) */
  for (int k = 0; k <= h; k++)
    {
      cd zk = z[k % h], zc = std::conj (z[(h - k) % h]);
      cd even = 0.5 * (zk + zc);
      cd odd  = cd (0, -0.5) * (zk - zc);
      cd x = even + p->split[k] * odd;
      out[k][0] = (float) x.real();
      out[k][1] = (float) x.imag();
    }
}

void
fftwf_execute_dft_c2r (const fftwf_plan pl, fftwf_complex *in, float *out)
{
  const HalfPlan *p = get_plan (pl->n);
  const int h = p->h;
  std::vector<cd> z (h);
  for (int k = 0; k < h; k++)
    {
      cd xk (in[k][0], in[k][1]), xm (in[h - k][0], in[h - k][1]);
      if (k == 0)
        {
          xk = cd (in[0][0], 0);   /* FFTW ignores Im of DC and Nyquist */
          xm = cd (in[h][0], 0);
        }
      cd e = xk + std::conj (xm);
      cd o = (xk - std::conj (xm)) * std::conj (p->split[k]);
      /* z_time[n] = sum_k (E + iO) e^{+2 pi i k n / h}; use conj trick with forward FFT */
      cd v = e + cd (0, 1) * o;
      z[p->rev[k]] = std::conj (v);
    }
  cfft (p, z.data());
  for (int i = 0; i < h; i++)
    {
      out[2 * i]     = (float) z[i].real();
      out[2 * i + 1] = (float) -z[i].imag();
    }
}

} /* extern "C" */
