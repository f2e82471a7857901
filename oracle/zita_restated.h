/* oracle/zita_restated.h -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference resamples with zita-resampler (Resampler and VResampler, hlen = 16; resample.cc:29-270), a third-party
 * library that is neither part of /root/reference nor installed in this image: PARITY UNPINNED for everything that goes
 * through these classes.  They restate zita-resampler 1.x's published algorithm (Resampler_table: a polyphase
 * windowed-sinc FIR with 2 hl taps and np phases, window 0.384 + 0.5 cos x + 0.116 cos 2x; Resampler: rational ratio,
 * integer phase; VResampler: arbitrary ratio, 256 phases, coefficients interpolated linearly between two phases, phase
 * accumulated in double).  The member names are zita's public interface (inp_count / inp_data / out_count / out_data /
 * setup / process / inpsize / nchan), so that the same classes serve
 *   - the oracle's restatement of the reference's call sequences (awm_oracle.cc), and
 *   - oracle/ref_shim/include/zita-resampler/{resampler,vresampler}.h, the stand-in the unmodified reference
 *     sources are compiled against in oracle/_ref.
 */
#pragma once
#include <math.h>
#include <string.h>
#include <memory>
#include <vector>

struct ZitaTable
{
  unsigned hl = 0, np = 0;
  std::vector<float> ctab;                 /* (np + 1) * hl */
  ZitaTable (double fr, unsigned hl_, unsigned np_) : hl (hl_), np (np_), ctab (size_t (hl_) * (np_ + 1))
  {
    auto sinc = [] (double x) { x = fabs (x); if (x < 1e-6) return 1.0; x *= M_PI; return sin (x) / x; };
    auto wind = [] (double x) { x = fabs (x); if (x >= 1.0) return 0.0; x *= M_PI; return 0.384 + 0.500 * cos (x) + 0.116 * cos (2 * x); };
    float *p = ctab.data();
    for (unsigned j = 0; j <= np; j++)
      {
        double t = double (j) / double (np);
        for (unsigned i = 0; i < hl; i++)
          {
            p[hl - i - 1] = float (fr * sinc (t * fr) * wind (t / hl));
            t += 1;
          }
        p += hl;
      }
  }
};

static unsigned zita_gcd (unsigned a, unsigned b) { while (b) { const unsigned t = a % b; a = b; b = t; } return a; }

class ZitaResampler
{
  std::unique_ptr<ZitaTable> table;
  unsigned nchan_ = 0, inmax = 0, index = 0, nread = 0, nzero = 0, phase = 0, pstep = 0;
  std::vector<float> buff;
public:
  unsigned     inp_count = 0, out_count = 0;
  const float *inp_data = nullptr;
  float       *out_data = nullptr;
  unsigned nchan() const { return nchan_; }
  unsigned inpsize() const { return table ? 2 * table->hl : 0; }
  int
  setup (unsigned fs_inp, unsigned fs_out, unsigned nchan, unsigned hlen)
  {
    double frel = 1.0 - 2.6 / hlen;
    if (!fs_inp || !fs_out || !nchan)
      return 1;
    const double r = double (fs_out) / double (fs_inp);
    const unsigned g = zita_gcd (fs_out, fs_inp), n = fs_out / g, s = fs_inp / g;
    if (!(16 * r >= 1 && n <= 1000))
      return 1;
    unsigned h = hlen, k = 250;
    if (r < 1)
      {
        frel *= r;
        h = unsigned (ceil (h / r));
        k = unsigned (ceil (k / r));
      }
    table = std::make_unique<ZitaTable> (frel, h, n);
    buff.assign (size_t (nchan) * (2 * h - 1 + k), 0.f);
    nchan_ = nchan;
    inmax = k;
    pstep = s;
    index = 0; nzero = 0; phase = 0;
    nread = 2 * h;
    return 0;
  }
  void
  process()
  {
    if (!table)
      return;
    const unsigned hl = table->hl, np = table->np, dp = pstep;
    unsigned in = index, nr = nread, ph = phase, nz = nzero;
    unsigned n = (2 * hl - nr) * nchan_;
    float *p1 = buff.data() + in * nchan_;
    float *p2 = p1 + n;
    while (out_count)
      {
        if (nr)
          {
            if (inp_count == 0)
              break;
            if (inp_data)
              {
                for (unsigned c = 0; c < nchan_; c++)
                  p2[c] = inp_data[c];
                inp_data += nchan_;
                nz = 0;
              }
            else
              {
                for (unsigned c = 0; c < nchan_; c++)
                  p2[c] = 0;
                if (nz < 2 * hl)
                  nz++;
              }
            nr--;
            p2 += nchan_;
            inp_count--;
          }
        else
          {
            if (out_data)
              {
                if (nz < 2 * hl)
                  {
                    const float *c1 = table->ctab.data() + hl * ph;
                    const float *c2 = table->ctab.data() + hl * (np - ph);
                    for (unsigned c = 0; c < nchan_; c++)
                      {
                        const float *q1 = p1 + c;
                        const float *q2 = p2 + c;
                        float sum = 1e-20f;
                        for (unsigned i = 0; i < hl; i++)
                          {
                            q2 -= nchan_;
                            sum += *q1 * c1[i] + *q2 * c2[i];
                            q1 += nchan_;
                          }
                        *out_data++ = sum - 1e-20f;
                      }
                  }
                else
                  for (unsigned c = 0; c < nchan_; c++)
                    *out_data++ = 0;
              }
            out_count--;
            ph += dp;
            if (ph >= np)
              {
                nr = ph / np;
                ph -= nr * np;
                in += nr;
                p1 += nr * nchan_;
                if (in >= inmax)
                  {
                    n = (2 * hl - nr) * nchan_;
                    memmove (buff.data(), p1, n * sizeof (float));
                    in = 0;
                    p1 = buff.data();
                    p2 = p1 + n;
                  }
              }
          }
      }
    index = in; nread = nr; phase = ph; nzero = nz;
  }
};

/* VResampler (variable ratio): setup (ratio, nchan, hlen) -> 256 phases, step np / ratio kept in double */
class ZitaVResampler
{
  std::unique_ptr<ZitaTable> table;
  unsigned nchan_ = 0, inmax = 0, index = 0, nread = 0, nzero = 0;
  double   phase = 0, pstep = 0, qstep = 0, wstep = 1;
  std::vector<float> buff, c1, c2;
public:
  enum { NPHASE = 256 };
  unsigned     inp_count = 0, out_count = 0;
  const float *inp_data = nullptr;
  float       *out_data = nullptr;
  unsigned nchan() const { return nchan_; }
  unsigned inpsize() const { return table ? 2 * table->hl : 0; }
  double   step() const { return pstep; }
  unsigned half_length() const { return table ? table->hl : 0; }
  const ZitaTable *coefficients() const { return table.get(); }
  int
  setup (double ratio, unsigned nchan, unsigned hlen)
  {
    double frel = 1.0 - 2.6 / hlen;
    if (!nchan || hlen < 8 || hlen > 96 || 16 * ratio < 1 || ratio > 256)
      return 1;
    const unsigned n = NPHASE;
    const double s = n / ratio;
    unsigned h = hlen, k = 250;
    if (ratio < 1)
      {
        frel *= ratio;
        h = unsigned (ceil (h / ratio));
        k = unsigned (ceil (k / ratio));
      }
    table = std::make_unique<ZitaTable> (frel, h, n);
    buff.assign (size_t (nchan) * (2 * h - 1 + k), 0.f);
    c1.assign (2 * h, 0.f);
    c2.assign (2 * h, 0.f);
    nchan_ = nchan;
    inmax = k;
    pstep = qstep = s;
    wstep = 1;
    index = 0; nzero = 0; phase = 0;
    nread = 2 * h;
    return 0;
  }
  int
  process()
  {
    if (!table)
      return 1;
    const unsigned hl = table->hl, np = table->np;
    unsigned in = index, nr = nread, nz = nzero;
    double ph = phase, dp = pstep;
    unsigned n = (2 * hl - nr) * nchan_;
    float *p1 = buff.data() + in * nchan_;
    float *p2 = p1 + n;
    while (out_count)
      {
        if (nr)
          {
            if (inp_count == 0)
              break;
            if (inp_data)
              {
                for (unsigned c = 0; c < nchan_; c++)
                  p2[c] = inp_data[c];
                inp_data += nchan_;
                nz = 0;
              }
            else
              {
                for (unsigned c = 0; c < nchan_; c++)
                  p2[c] = 0;
                if (nz < 2 * hl)
                  nz++;
              }
            nr--;
            p2 += nchan_;
            inp_count--;
          }
        else
          {
            if (out_data)
              {
                if (nz < 2 * hl)
                  {
                    const unsigned k = unsigned (ph);
                    float b = float (ph - k);
                    float a = 1.0f - b;
                    const float *q1 = table->ctab.data() + hl * k;
                    const float *q2 = table->ctab.data() + hl * (np - k);
                    for (unsigned i = 0; i < hl; i++)
                      {
                        c1[i] = a * q1[i] + b * q1[i + hl];
                        c2[i] = a * q2[i] + b * q2[int (i) - int (hl)];
                      }
                    for (unsigned c = 0; c < nchan_; c++)
                      {
                        const float *r1 = p1 + c;
                        const float *r2 = p2 + c;
                        float sum = 1e-25f;
                        for (unsigned i = 0; i < hl; i++)
                          {
                            r2 -= nchan_;
                            sum += *r1 * c1[i] + *r2 * c2[i];
                            r1 += nchan_;
                          }
                        *out_data++ = sum - 1e-25f;
                      }
                  }
                else
                  for (unsigned c = 0; c < nchan_; c++)
                    *out_data++ = 0;
              }
            out_count--;
            const double dd = qstep - dp;
            if (fabs (dd) < 1e-20)
              dp = qstep;
            else
              dp += wstep * dd;
            ph += dp;
            if (ph >= np)
              {
                nr = unsigned (floor (ph / np));
                ph -= nr * np;
                in += nr;
                p1 += nr * nchan_;
                if (in >= inmax)
                  {
                    n = (2 * hl - nr) * nchan_;
                    memmove (buff.data(), p1, n * sizeof (float));
                    in = 0;
                    p1 = buff.data();
                    p2 = p1 + n;
                  }
              }
          }
      }
    index = in; nread = nr; nzero = nz; phase = ph; pstep = dp;
    return 0;
  }
};
