/* oracle/zita_probe.cc -- TEST INFRASTRUCTURE ONLY (tests/test_zita_library.py; nothing of the product links or loads it).
 *
 * zita-resampler is the one third-party algorithm on the path whose parity is UNPINNED: the library is neither in the reference tree nor
 * in this image, so K10 / K12 are pinned to the restatement of its published algorithm (oracle/zita_restated.h) and to nothing else.
 * This file is the driver that closes the gap the day a libzita-resampler is there: the reference's use of the two classes
 * (resample.cc:30-126, wavchunkloader.cc:200-216: hl - 1 null frames, the input in pieces of 1024 output frames, hl null frames)
 * written ONCE against the library's public interface and compiled TWICE -- against the installed headers with -lzita-resampler and
 * against the restated classes (-I oracle/ref_shim/include, where <zita-resampler/...> are the stand-ins).  The test compares the two
 * outputs bit for bit. */
#include <zita-resampler/resampler.h>
#include <zita-resampler/vresampler.h>
#include <algorithm>
#include <cstddef>
#include <vector>

namespace {
template<class R> size_t
run (R& rs, const float *in, size_t n_frames, int C, float *out, size_t max_out_frames)
{
  constexpr unsigned PIECE = 1024;
  std::vector<float> chunk (size_t (PIECE) * C);
  size_t produced = 0;
  auto feed = [&] (const float *data, size_t frames) {
    size_t done = 0;
    while (done < frames)
      {
        rs.out_count = PIECE;
        rs.out_data = chunk.data();
        const unsigned given = unsigned (std::min<size_t> (frames - done, 1u << 30));
        rs.inp_count = given;
        rs.inp_data = data ? const_cast<float *> (data + done * C) : nullptr;
        rs.process();
        const size_t count = PIECE - rs.out_count;
        for (size_t f = 0; f < count && produced + f < max_out_frames; f++)
          std::copy (chunk.begin() + f * C, chunk.begin() + (f + 1) * C, out + (produced + f) * C);
        produced += count;
        done += given - rs.inp_count;
      }
  };
  rs.inp_count = rs.inpsize() / 2 - 1;          // priming: no output yet
  rs.inp_data = nullptr;
  rs.out_count = 1000000;
  rs.out_data = nullptr;
  rs.process();
  feed (in, n_frames);
  feed (nullptr, rs.inpsize() / 2);
  return produced;
}
}

extern "C" size_t
zita_probe_fixed (const float *in, size_t n_frames, int n_channels, unsigned fs_in, unsigned fs_out, unsigned hlen, float *out, size_t max_out_frames)
{
  Resampler rs;
  if (rs.setup (fs_in, fs_out, unsigned (n_channels), hlen) != 0)
    return 0;
  return run (rs, in, n_frames, n_channels, out, max_out_frames);
}

extern "C" size_t
zita_probe_var (const float *in, size_t n_frames, int n_channels, double ratio, unsigned hlen, float *out, size_t max_out_frames)
{
  VResampler rs;
  if (rs.setup (ratio, unsigned (n_channels), hlen) != 0)
    return 0;
  return run (rs, in, n_frames, n_channels, out, max_out_frames);
}
