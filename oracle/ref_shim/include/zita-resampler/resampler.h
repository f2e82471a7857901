/* oracle/ref_shim: zita-resampler is not available in this image; setup() reports
 * "unsupported" so every reference path that needs a sample-rate change fails
 * cleanly (resample.cc:81-94,250-262).  The oracle is therefore 44.1 kHz only
 * (SURVEY.md section 8c).  TEST INFRASTRUCTURE ONLY. */
#pragma once
class Resampler
{
public:
  unsigned int inp_count = 0, out_count = 0;
  float *inp_data = nullptr, *out_data = nullptr;
  int setup (unsigned int, unsigned int, unsigned int, unsigned int) { return 1; }
  int process () { return 1; }
  int nchan () const { return 1; }
  int inpsize () const { return 2; }
  double inpdist () const { return 0; }
};
