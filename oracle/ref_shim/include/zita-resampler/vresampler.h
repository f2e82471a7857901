/* oracle/ref_shim: see resampler.h */
#pragma once
class VResampler
{
public:
  unsigned int inp_count = 0, out_count = 0;
  float *inp_data = nullptr, *out_data = nullptr;
  int setup (double, unsigned int, unsigned int) { return 1; }
  int process () { return 1; }
  int nchan () const { return 1; }
  int inpsize () const { return 2; }
  double inpdist () const { return 0; }
};
