// Audio stream "plugin surface" of the reference, re-implemented for raw and WAV data:
//   AudioStream / AudioInputStream / AudioOutputStream + create()   reference src/audiostream.hh:31-62, audiostream.cc:34-121
//   RawFormat, sample conversion rules                              reference src/rawinputstream.hh:27-55, rawconverter.{hh,cc}
//   RawInputStream / RawOutputStream                                reference src/rawinputstream.cc, rawoutputstream.cc
//   WavInputStream  (RIFF / RF64 parser, files and pipes)           reference src/wavpipeinputstream.cc:69-173
//   WavOutputStream (stdout and named files)                        reference src/stdoutwavoutputstream.cc:75-146
//   WavData                                                         reference src/wavdata.hh:27-74
// libsndfile / mpg123 backed streams are outside the scope of this path:
// Format::AUTO opens WAV files with the built-in parser.
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>
#include "utils.hh"

namespace awm {

enum class Encoding { SIGNED, UNSIGNED, FLOAT };

class RawFormat
{
public:
  enum Endian { LITTLE, BIG };
  int      n_channels  = 2;
  int      sample_rate = 0;
  int      bit_depth   = 16;
  Endian   endian      = LITTLE;
  Encoding encoding    = Encoding::SIGNED;
};

// float <-> PCM bytes with the reference's normalisation / clipping / truncation rules (rawconverter.cc:155-286)
class PcmCodec
{
  RawFormat m_format;
  bool      m_libsndfile_int_rule = false;   // int32 clip, then keep the top bits (what writing through libsndfile does)
public:
  static std::unique_ptr<PcmCodec> create (const RawFormat& format, Error& err, bool libsndfile_int_rule = false);
  int  sample_width() const { return m_format.bit_depth / 8; }
  const RawFormat& format() const { return m_format; }
  bool direct16() const { return !m_libsndfile_int_rule && m_format.endian == RawFormat::LITTLE && m_format.encoding == Encoding::SIGNED && m_format.bit_depth == 16; }
  void decode (const unsigned char *bytes, float *samples, size_t n_samples) const;
  void encode (const float *samples, unsigned char *bytes, size_t n_samples) const;
};

class AudioStream
{
public:
  virtual int bit_depth() const = 0;
  virtual int sample_rate() const = 0;
  virtual int n_channels() const = 0;
  virtual ~AudioStream() {}
};

class AudioInputStream : public AudioStream
{
public:
  static std::unique_ptr<AudioInputStream> create (const std::string& filename, Error& err);
  static constexpr size_t N_FRAMES_UNKNOWN = ~size_t (0);
  virtual size_t   n_frames() const = 0;
  virtual Encoding encoding() const = 0;
  // up to `count` interleaved frames normalised to [-1,1); a short / empty vector means EOF
  virtual Error    read_frames (std::vector<float>& samples, size_t count) = 0;
  // optional fast path for the GPU pipeline: hand out the undecoded sample bytes (decoded on the device by
  // awm_pcm_decode_d with the same rules as read_frames).  Default: not available.
  virtual bool     raw_access (RawFormat& format) const { (void) format; return false; }
  virtual Error    read_raw (unsigned char *dst, size_t max_frames, size_t& got_frames) { (void) dst; (void) max_frames; got_frames = 0; return Error ("raw access not supported"); }
  // optional, for streams whose sample bytes lie in a REGULAR file: the descriptor, the file offset of the next unread frame and
  // how many frames lie behind it (bounded by the file's size).  The caller then reads the bytes itself -- pread, any thread, any
  // order (host/wmfile.cc: several workers fill a ring of page-locked tiles ahead of the GPU) -- and tells the stream how many
  // frames it took with raw_region_consume(), after which read_raw / read_frames continue behind them.
  virtual bool     raw_region (int& fd, uint64_t& byte_offset, size_t& frames_available) { (void) fd; (void) byte_offset; (void) frames_available; return false; }
  virtual void     raw_region_consume (size_t n_frames) { (void) n_frames; }
};

class AudioOutputStream : public AudioStream
{
public:
  static std::unique_ptr<AudioOutputStream> create (const std::string& filename, int n_channels, int sample_rate, int bit_depth,
                                                    Encoding encoding, size_t n_frames, Error& err);
  virtual Error write_frames (const std::vector<float>& frames) = 0;
  virtual Error close() = 0;
  // optional fast path: accept sample bytes that were encoded on the device (awm_pcm_encode_d); direct16 tells which
  // of the reference's two 16 bit rules this stream's own write_frames applies
  virtual bool  raw_access (RawFormat& format, bool& direct16) const { (void) format; (void) direct16; return false; }
  virtual Error write_raw (const unsigned char *bytes, size_t n_frames) { (void) bytes; (void) n_frames; return Error ("raw access not supported"); }
  // optional, for streams that write to a REGULAR file of their own: everything written so far is flushed, `byte_offset` is where
  // the next sample byte belongs.  The caller places the bytes of the following frames there itself (any thread, any order:
  // host/wmfile.cc copies into a shared mapping from several workers -- buffered write() calls on one file serialise on the
  // inode) and reports them with raw_region_written(), after which write_raw / write_frames / close() continue behind them.
  virtual bool  raw_region (int& fd, uint64_t& byte_offset) { (void) fd; (void) byte_offset; return false; }
  virtual Error raw_region_written (size_t n_frames) { (void) n_frames; return Error ("raw access not supported"); }
};

// global stream format selection, as in the reference's Params (wmcommon.hh:79-83)
struct StreamParams
{
  static RawFormat raw_input_format;
  static RawFormat raw_output_format;
};

class WavData
{
  std::vector<float> m_samples;
  int m_sample_rate = 0, m_n_channels = 0, m_bit_depth = 0;
public:
  WavData() {}
  WavData (const std::vector<float>& samples, int n_channels, int sample_rate, int bit_depth) :
    m_samples (samples), m_sample_rate (sample_rate), m_n_channels (n_channels), m_bit_depth (bit_depth) {}
  Error load (AudioInputStream *in_stream);
  Error load (const std::string& filename);
  Error save (const std::string& filename) const;
  int sample_rate() const { return m_sample_rate; }
  int bit_depth() const { return m_bit_depth; }
  int n_channels() const { return m_n_channels; }
  size_t n_values() const { return m_samples.size(); }
  size_t n_frames() const { return m_samples.size() / m_n_channels; }
  const std::vector<float>& samples() const { return m_samples; }
  std::vector<float>& mutable_samples() { return m_samples; }
  void set_samples (const std::vector<float>& samples) { m_samples = samples; }
};

} // namespace awm
