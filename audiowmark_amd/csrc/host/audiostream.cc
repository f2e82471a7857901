#include "audiostream.hh"
#include <sys/stat.h>
#include <unistd.h>
#include <thread>
#include "wmcommon.hh"
#include <algorithm>
#include <cerrno>
#include <cstring>

namespace awm {

RawFormat StreamParams::raw_input_format;
RawFormat StreamParams::raw_output_format;

/* ---- sample conversion ----------------------------------------------------------------- */

std::unique_ptr<PcmCodec>
PcmCodec::create (const RawFormat& format, Error& err, bool libsndfile_int_rule)
{
  err = Error::Code::NONE;
  if (format.encoding == Encoding::FLOAT)
    {
      if (format.bit_depth != 32 && format.bit_depth != 64)
        {
          err = Error (string_printf ("unsupported bit depth %d for float encoding", format.bit_depth));
          return nullptr;
        }
    }
  else if (format.bit_depth != 8 && format.bit_depth != 16 && format.bit_depth != 24 && format.bit_depth != 32)
    {
      err = Error (string_printf ("unsupported bit depth %d", format.bit_depth));
      return nullptr;
    }
  std::unique_ptr<PcmCodec> codec (new PcmCodec());
  codec->m_format = format;
  codec->m_libsndfile_int_rule = libsndfile_int_rule;
  return codec;
}

template<int BITS> static inline int
float_to_int_clip (float f)
{
  // scale in float, saturate, truncate toward zero (reference rawconverter.hh:34-50)
  const int64_t inorm = 1LL << (BITS - 1);
  const float snorm = f * float (inorm);
  if (snorm >= float (inorm - 1))
    return int (inorm - 1);
  if (snorm <= float (-inorm))
    return int (-inorm);
  return int (snorm);
}

static inline float
float_clip (float f)
{
  return f >= 1 ? 1.f : (f <= -1 ? -1.f : f);
}

void
PcmCodec::decode (const unsigned char *bytes, float *samples, size_t n) const
{
  const int width = sample_width();
  const bool big = m_format.endian == RawFormat::BIG;
  if (m_format.encoding == Encoding::FLOAT)
    {
      for (size_t i = 0; i < n; i++)
        {
          unsigned char tmp[8];
          for (int b = 0; b < width; b++)
            tmp[b] = bytes[i * width + (big ? width - 1 - b : b)];
          if (width == 4)
            std::memcpy (&samples[i], tmp, 4);
          else
            {
              double d;
              std::memcpy (&d, tmp, 8);
              samples[i] = d;
            }
        }
      return;
    }
  const float norm = 1.0 / 0x80000000LL;
  for (size_t i = 0; i < n; i++)
    {
      // left-align the sample in 32 bits
      uint32_t u = 0;
      for (int b = 0; b < width; b++)
        {
          const unsigned char byte = bytes[i * width + b];
          const int significance = big ? width - 1 - b : b;          // 0 = least significant byte of the sample
          u |= uint32_t (byte) << (8 * (4 - width + significance));
        }
      if (m_format.encoding == Encoding::UNSIGNED)
        u ^= 0x80000000u;
      samples[i] = int32_t (u) * norm;
    }
}

void
PcmCodec::encode (const float *samples, unsigned char *bytes, size_t n) const
{
  const int width = sample_width();
  const bool big = m_format.endian == RawFormat::BIG;
  if (m_format.encoding == Encoding::FLOAT)
    {
      for (size_t i = 0; i < n; i++)
        {
          unsigned char tmp[8];
          const float f = float_clip (samples[i]);
          if (width == 4)
            std::memcpy (tmp, &f, 4);
          else
            {
              const double d = f;
              std::memcpy (tmp, &d, 8);
            }
          for (int b = 0; b < width; b++)
            bytes[i * width + (big ? width - 1 - b : b)] = tmp[b];
        }
      return;
    }
  // the reference converts little-endian signed 16 / 32 bit directly (truncation toward zero at that width);
  // every other layout goes through a 32-bit value whose top bits are kept (rawconverter.cc:187-215)
  const bool direct = !m_libsndfile_int_rule && !big && m_format.encoding == Encoding::SIGNED && (width == 2 || width == 4);
  for (size_t i = 0; i < n; i++)
    {
      uint32_t u;
      if (direct && width == 2)
        u = uint32_t (float_to_int_clip<16> (samples[i])) << 16;
      else
        u = uint32_t (float_to_int_clip<32> (samples[i]));
      if (m_format.encoding == Encoding::UNSIGNED)
        u ^= 0x80000000u;
      for (int b = 0; b < width; b++)
        {
          const int significance = big ? width - 1 - b : b;
          bytes[i * width + b] = (unsigned char) (u >> (8 * (4 - width + significance)));
        }
    }
}

/* ---- raw streams -------------------------------------------------------------------------- */

namespace {

/* Bulk file input for the GPU staging path: a chunk of tens of megabytes read from a REGULAR file is split over a few threads
 * using pread -- a single thread copies out of the page cache at ~8 GB/s, which bounded the file level `get` of an hour of
 * audio (87 -> 64 ms).  Pipes and small requests take the plain stdio path; writes stay on the writer thread's fwrite
 * (splitting them too made `add` slower: 139 -> 183 ms, the page allocations of a growing file do not parallelise). */
constexpr size_t BULK_IO_MIN = size_t (4) << 20;
constexpr int    BULK_IO_THREADS = 4;

bool
is_regular (FILE *f)
{
  struct stat st;
  return f && fstat (fileno (f), &st) == 0 && S_ISREG (st.st_mode);
}

/* returns bytes read (short only at EOF / on error, `failed` tells which) */
size_t
bulk_read (FILE *f, unsigned char *buf, size_t bytes, bool& failed, int *err_out = nullptr)
{
  failed = false;
  if (err_out)
    *err_out = 0;
  const off_t pos = ftello (f);
  if (pos < 0)
    {
      failed = true;
      if (err_out)
        *err_out = errno;
      return 0;
    }
  const int fd = fileno (f);
  const size_t part = (bytes / BULK_IO_THREADS + 4095) & ~size_t (4095);
  size_t done[BULK_IO_THREADS] = {};
  bool bad[BULK_IO_THREADS] = {};
  int  err[BULK_IO_THREADS] = {};                          // errno is per thread: every worker keeps its own
  auto work = [&] (int t) {
    const size_t lo = std::min (bytes, part * t), hi = std::min (bytes, part * (t + 1));
    size_t n = 0;
    while (lo + n < hi)
      {
        const ssize_t r = pread (fd, buf + lo + n, hi - lo - n, pos + off_t (lo + n));
        if (r < 0 && errno == EINTR)
          continue;
        if (r <= 0)
          {
            bad[t] = r < 0;
            err[t] = r < 0 ? errno : 0;
            break;
          }
        n += size_t (r);
      }
    done[t] = n;
  };
  std::thread threads[BULK_IO_THREADS - 1];
  for (int t = 1; t < BULK_IO_THREADS; t++)
    threads[t - 1] = std::thread (work, t);
  work (0);
  for (auto& th : threads)
    th.join();
  size_t total = 0;
  for (int t = 0; t < BULK_IO_THREADS; t++)
    {
      if (bad[t] && !failed && err_out)
        *err_out = err[t];
      failed = failed || bad[t];
      const size_t want = std::min (bytes, part * (t + 1)) - std::min (bytes, part * t);
      total += done[t];
      if (done[t] < want)
        break;                                             // EOF inside this part: later parts lie beyond it
    }
  if (!failed)                                             // (after an error the caller discards the data: the position stays)
    fseeko (f, pos + off_t (total), SEEK_SET);
  return total;
}

/* raw_region of the input streams: a regular file, positioned at the next sample byte */
bool
input_region (FILE *f, size_t frame_bytes, size_t frames_left /* N_FRAMES_UNKNOWN: until the end of the file */, int& fd, uint64_t& byte_offset,
              size_t& frames_available)
{
  struct stat st;
  if (!f || !frame_bytes || fstat (fileno (f), &st) != 0 || !S_ISREG (st.st_mode))
    return false;
  const off_t pos = ftello (f);
  if (pos < 0 || st.st_size < pos)
    return false;
  fd = fileno (f);
  byte_offset = uint64_t (pos);
  frames_available = size_t (st.st_size - pos) / frame_bytes;
  if (frames_left != AudioInputStream::N_FRAMES_UNKNOWN)
    frames_available = std::min (frames_available, frames_left);
  return true;
}

/* raw_region of the output streams: a regular file of our own; stdio's buffer is flushed so that the descriptor can be used beside it */
bool
output_region (FILE *f, bool own_file, int& fd, uint64_t& byte_offset)
{
  struct stat st;
  if (!f || !own_file || fflush (f) != 0 || fstat (fileno (f), &st) != 0 || !S_ISREG (st.st_mode))
    return false;
  const off_t pos = ftello (f);
  if (pos < 0)
    return false;
  fd = fileno (f);
  byte_offset = uint64_t (pos);
  return true;
}

class RawInputStream : public AudioInputStream
{
  RawFormat m_format;
  FILE     *m_file = nullptr;
  bool      m_close = false;
  std::unique_ptr<PcmCodec> m_codec;
public:
  ~RawInputStream() { if (m_close && m_file) fclose (m_file); }
  Error
  open (const std::string& filename, const RawFormat& format)
  {
    if (!format.n_channels)  return Error ("RawInputStream: input format: missing number of channels");
    if (!format.bit_depth)   return Error ("RawInputStream: input format: missing bit depth");
    if (!format.sample_rate) return Error ("RawInputStream: input format: missing sample rate");
    Error err;
    m_codec = PcmCodec::create (format, err);
    if (err)
      return err;
    if (filename == "-")
      m_file = stdin;
    else
      {
        m_file = fopen (filename.c_str(), "r");
        if (!m_file)
          return Error (strerror (errno));
        m_close = true;
      }
    m_format = format;
    return Error::Code::NONE;
  }
  int bit_depth() const override { return m_format.bit_depth; }
  int sample_rate() const override { return m_format.sample_rate; }
  int n_channels() const override { return m_format.n_channels; }
  size_t n_frames() const override { return N_FRAMES_UNKNOWN; }
  Encoding encoding() const override { return m_format.encoding; }
  Error
  read_frames (std::vector<float>& samples, size_t count) override
  {
    const size_t frame_bytes = size_t (m_format.n_channels) * m_codec->sample_width();
    std::vector<unsigned char> bytes (count * frame_bytes);
    const size_t got = fread (bytes.data(), frame_bytes, count, m_file);
    if (ferror (m_file))
      return Error ("error reading sample data");
    samples.resize (got * m_format.n_channels);
    m_codec->decode (bytes.data(), samples.data(), samples.size());
    return Error::Code::NONE;
  }
  bool raw_access (RawFormat& format) const override { format = m_format; return true; }
  Error
  read_raw (unsigned char *dst, size_t max_frames, size_t& got_frames) override
  {
    const size_t frame_bytes = size_t (m_format.n_channels) * m_codec->sample_width();
    if (max_frames * frame_bytes >= BULK_IO_MIN && is_regular (m_file))
      {
        bool failed;
        got_frames = bulk_read (m_file, dst, max_frames * frame_bytes, failed) / frame_bytes;
        return failed ? Error ("error reading sample data") : Error (Error::Code::NONE);
      }
    got_frames = fread (dst, frame_bytes, max_frames, m_file);
    if (ferror (m_file))
      return Error ("error reading sample data");
    return Error::Code::NONE;
  }
  bool
  raw_region (int& fd, uint64_t& byte_offset, size_t& frames_available) override
  {
    const size_t frame_bytes = size_t (m_format.n_channels) * m_codec->sample_width();
    return input_region (m_file, frame_bytes, N_FRAMES_UNKNOWN, fd, byte_offset, frames_available);
  }
  void
  raw_region_consume (size_t n_frames) override
  {
    fseeko (m_file, off_t (n_frames * size_t (m_format.n_channels) * m_codec->sample_width()), SEEK_CUR);
  }
};

class RawOutputStream : public AudioOutputStream
{
  RawFormat m_format;
  FILE     *m_file = nullptr;
  bool      m_close = false;
  std::unique_ptr<PcmCodec> m_codec;
public:
  ~RawOutputStream() { close(); }
  Error
  open (const std::string& filename, const RawFormat& format)
  {
    if (!format.n_channels)  return Error ("RawOutputStream: output format: missing number of channels");
    if (!format.bit_depth)   return Error ("RawOutputStream: output format: missing bit depth");
    if (!format.sample_rate) return Error ("RawOutputStream: output format: missing sample rate");
    Error err;
    m_codec = PcmCodec::create (format, err);
    if (err)
      return err;
    if (filename == "-")
      m_file = stdout;
    else
      {
        m_file = fopen (filename.c_str(), "w");
        if (!m_file)
          return Error (strerror (errno));
        m_close = true;
      }
    m_format = format;
    return Error::Code::NONE;
  }
  int bit_depth() const override { return m_format.bit_depth; }
  int sample_rate() const override { return m_format.sample_rate; }
  int n_channels() const override { return m_format.n_channels; }
  Error
  write_frames (const std::vector<float>& samples) override
  {
    if (samples.empty())
      return Error::Code::NONE;
    std::vector<unsigned char> bytes (samples.size() * m_codec->sample_width());
    m_codec->encode (samples.data(), bytes.data(), samples.size());
    fwrite (bytes.data(), 1, bytes.size(), m_file);
    if (ferror (m_file))
      return Error ("write sample data failed");
    return Error::Code::NONE;
  }
  bool raw_access (RawFormat& format, bool& direct16) const override { format = m_format; direct16 = m_codec->direct16(); return true; }
  Error
  write_raw (const unsigned char *bytes, size_t n_frames) override
  {
    const size_t n = n_frames * m_format.n_channels * m_codec->sample_width();
    fwrite (bytes, 1, n, m_file);
    if (ferror (m_file))
      return Error ("write sample data failed");
    return Error::Code::NONE;
  }
  bool raw_region (int& fd, uint64_t& byte_offset) override { return output_region (m_file, m_close, fd, byte_offset); }
  Error
  raw_region_written (size_t n_frames) override
  {
    if (fseeko (m_file, off_t (n_frames * m_format.n_channels * m_codec->sample_width()), SEEK_CUR) != 0)
      return Error ("write sample data failed");
    return Error::Code::NONE;
  }
  Error
  close() override
  {
    if (m_file)
      {
        fflush (m_file);
        const bool failed = ferror (m_file);
        if (m_close)
          fclose (m_file);
        m_file = nullptr;
        if (failed)
          return Error ("error during flush");
      }
    return Error::Code::NONE;
  }
};

/* ---- WAV ------------------------------------------------------------------------------------- */

uint32_t u32le (const unsigned char *b) { return b[0] | (b[1] << 8) | (b[2] << 16) | (uint32_t (b[3]) << 24); }
uint16_t u16le (const unsigned char *b) { return uint16_t (b[0] | (b[1] << 8)); }
uint64_t u64le (const unsigned char *b) { return uint64_t (u32le (b)) | (uint64_t (u32le (b + 4)) << 32); }

class WavInputStream : public AudioInputStream
{
  RawFormat m_format;
  FILE     *m_file = nullptr;
  bool      m_close = false;
  size_t    m_n_frames = N_FRAMES_UNKNOWN;
  size_t    m_frames_left = N_FRAMES_UNKNOWN;
  std::unique_ptr<PcmCodec> m_codec;
  Error
  read_error (const std::string& message)
  {
    if (ferror (m_file))
      return Error (string_printf ("wav input read error: %s", strerror (errno)));
    return Error (message);
  }
public:
  ~WavInputStream() { if (m_close && m_file) fclose (m_file); }
  // pipe_mode: length fields of the header are not trusted (wav-pipe format); data runs until EOF
  Error
  open (const std::string& filename, bool pipe_mode)
  {
    if (filename == "-")
      m_file = stdin;
    else
      {
        m_file = fopen (filename.c_str(), "r");
        if (!m_file)
          return Error (strerror (errno));
        m_close = true;
      }
    unsigned char riff[12];
    const bool riff_ok = fread (riff, sizeof (riff), 1, m_file);
    const std::string tag (reinterpret_cast<char *> (riff), 4);
    if (!riff_ok || (tag != "RIFF" && tag != "RF64") || std::string (reinterpret_cast<char *> (riff + 8), 4) != "WAVE")
      return read_error ("input file is not a valid wav file");
    RawFormat format;
    bool have_fmt = false, in_data = false;
    uint64_t data_size = 0, ds64_data_size = 0;
    while (!in_data)
      {
        unsigned char chunk[8];
        if (!fread (chunk, sizeof (chunk), 1, m_file))
          return read_error ("wav input is incomplete (no data chunk found)");
        const std::string id (reinterpret_cast<char *> (chunk), 4);
        uint32_t chunk_size = u32le (chunk + 4);
        if (id == "fmt " && chunk_size >= 16 && chunk_size <= 64 * 1024 && !have_fmt)
          {
            std::vector<unsigned char> buffer (chunk_size);
            if (!fread (buffer.data(), buffer.size(), 1, m_file))
              return read_error ("wav input is incomplete (error reading fmt chunk)");
            const int format_type = u16le (&buffer[0]);
            if (format_type == 3)
              format.encoding = Encoding::FLOAT;
            else if (format_type != 1)
              {
                static const unsigned char pcm_guid[16] = { 0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71 };
                if (format_type == 0xFFFE && chunk_size >= 40)
                  {
                    if (memcmp (pcm_guid, &buffer[24], 16) != 0)
                      return Error ("wav input has unsupported extended format type, expected PCM");
                  }
                else
                  return Error (string_printf ("wav input has unsupported format type (%d), expected PCM", format_type));
              }
            format.n_channels = u16le (&buffer[2]);
            format.sample_rate = int (u32le (&buffer[4]));
            format.bit_depth = u16le (&buffer[14]);
            if (format.bit_depth == 8)
              format.encoding = Encoding::UNSIGNED;        // 8 bit wav is always unsigned
            // a header that would divide by zero later (libsndfile rejects such files in the reference)
            if (format.n_channels < 1 || format.sample_rate < 1)
              return Error (string_printf ("wav input has an invalid fmt chunk (%d channels, %d Hz)", format.n_channels, format.sample_rate));
            if (format.bit_depth % 8 == 0 && u16le (&buffer[12]) != format.n_channels * format.bit_depth / 8)
              return Error (string_printf ("wav input has an inconsistent fmt chunk (block align %d for %d channels of %d bits)",
                                           u16le (&buffer[12]), format.n_channels, format.bit_depth));
            if ((chunk_size & 1) && fgetc (m_file) == EOF)                 // chunks are word aligned
              return read_error ("wav input is incomplete (error reading fmt chunk)");
            have_fmt = true;
          }
        else if (id == "ds64" && chunk_size >= 24 && chunk_size <= 4096)
          {
            std::vector<unsigned char> buffer (chunk_size);
            if (!fread (buffer.data(), buffer.size(), 1, m_file))
              return read_error ("wav input is incomplete (error reading ds64 chunk)");
            ds64_data_size = u64le (&buffer[8]);
            if ((chunk_size & 1) && fgetc (m_file) == EOF)
              return read_error ("wav input is incomplete (error reading ds64 chunk)");
          }
        else if (id == "data")
          {
            data_size = chunk_size == 0xFFFFFFFFu ? ds64_data_size : chunk_size;
            in_data = true;
          }
        else
          {
            char junk[1024];
            uint64_t todo = uint64_t (chunk_size) + (chunk_size & 1);      // chunks are word aligned
            while (todo)
              {
                const size_t n = std::min<uint64_t> (todo, sizeof (junk));
                if (!fread (junk, n, 1, m_file))
                  return read_error ("wav input is incomplete (error skipping unknown chunk)");
                todo -= n;
              }
          }
      }
    if (!have_fmt)
      return Error ("wav input is incomplete (missing fmt chunk)");
    Error err;
    m_codec = PcmCodec::create (format, err);
    if (err)
      return err;
    m_format = format;
    if (!pipe_mode && data_size && is_regular (m_file))
      {
        // a header may claim more than the file holds (truncated files, crafted ds64 sizes): what can be read is what counts
        struct stat sb;
        const off_t pos = ftello (m_file);
        if (pos >= 0 && fstat (fileno (m_file), &sb) == 0 && sb.st_size >= pos)
          data_size = std::min<uint64_t> (data_size, uint64_t (sb.st_size - pos));
      }
    if (!pipe_mode && data_size && format.n_channels)
      m_n_frames = m_frames_left = data_size / (size_t (format.n_channels) * m_codec->sample_width());
    return Error::Code::NONE;
  }
  int bit_depth() const override { return m_format.bit_depth; }
  int sample_rate() const override { return m_format.sample_rate; }
  int n_channels() const override { return m_format.n_channels; }
  size_t n_frames() const override { return m_n_frames; }
  Encoding encoding() const override { return m_format.encoding; }
  Error
  read_frames (std::vector<float>& samples, size_t count) override
  {
    const size_t block = 8192;
    const size_t frame_bytes = size_t (m_format.n_channels) * m_codec->sample_width();
    std::vector<unsigned char> bytes (block * frame_bytes);
    size_t pos = 0;
    samples.clear();
    if (m_frames_left != N_FRAMES_UNKNOWN)
      count = std::min (count, m_frames_left);
    while (count)
      {
        const size_t todo = std::min (count, block);
        const size_t got = fread (bytes.data(), frame_bytes, todo, m_file);
        if (ferror (m_file))
          return Error (string_printf ("error reading wav input sample data: %s", strerror (errno)));
        if (!got)
          break;
        samples.resize ((pos + got) * m_format.n_channels);
        m_codec->decode (bytes.data(), samples.data() + pos * m_format.n_channels, got * m_format.n_channels);
        pos += got;
        count -= got;
        if (m_frames_left != N_FRAMES_UNKNOWN)
          m_frames_left -= got;
      }
    return Error::Code::NONE;
  }
  bool raw_access (RawFormat& format) const override { format = m_format; return true; }
  Error
  read_raw (unsigned char *dst, size_t max_frames, size_t& got_frames) override
  {
    if (m_frames_left != N_FRAMES_UNKNOWN)
      max_frames = std::min (max_frames, m_frames_left);
    const size_t frame_bytes = size_t (m_format.n_channels) * m_codec->sample_width();
    if (max_frames * frame_bytes >= BULK_IO_MIN && is_regular (m_file))
      {
        bool failed;
        int err = 0;
        got_frames = bulk_read (m_file, dst, max_frames * frame_bytes, failed, &err) / frame_bytes;
        if (failed)
          return Error (string_printf ("error reading wav input sample data: %s", strerror (err)));
      }
    else
      got_frames = max_frames ? fread (dst, frame_bytes, max_frames, m_file) : 0;
    if (ferror (m_file))
      return Error (string_printf ("error reading wav input sample data: %s", strerror (errno)));
    if (m_frames_left != N_FRAMES_UNKNOWN)
      m_frames_left -= got_frames;
    return Error::Code::NONE;
  }
  bool
  raw_region (int& fd, uint64_t& byte_offset, size_t& frames_available) override
  {
    const size_t frame_bytes = size_t (m_format.n_channels) * m_codec->sample_width();
    return input_region (m_file, frame_bytes, m_frames_left, fd, byte_offset, frames_available);
  }
  void
  raw_region_consume (size_t n_frames) override
  {
    fseeko (m_file, off_t (n_frames * size_t (m_format.n_channels) * m_codec->sample_width()), SEEK_CUR);
    if (m_frames_left != N_FRAMES_UNKNOWN)
      m_frames_left -= std::min (m_frames_left, n_frames);
  }
};

class WavOutputStream : public AudioOutputStream
{
  int    m_bit_depth = 0, m_sample_rate = 0, m_n_channels = 0;
  FILE  *m_file = nullptr;
  bool   m_close = false, m_fix_header = false, m_rf64 = false;
  size_t m_close_padding = 0, m_bytes_written = 0;
  static constexpr uint64_t RF64_HEADER_BYTES = 12 + 36 + 24 + 8;      // "RF64" size "WAVE", ds64, fmt, data chunk header
  std::unique_ptr<PcmCodec> m_codec;
public:
  ~WavOutputStream() { close(); }
  Error
  open (const std::string& filename, int n_channels, int sample_rate, int bit_depth, Encoding encoding, size_t n_frames, bool wav_pipe,
        bool rf64 = false)
  {
    if (encoding == Encoding::FLOAT)
      {
        if (bit_depth != 32 && bit_depth != 64)
          return Error (string_printf ("WavOutputStream::open: unsupported floating point bit depth %d", bit_depth));
      }
    else if (bit_depth != 16 && bit_depth != 24 && bit_depth != 32)
      return Error (string_printf ("WavOutputStream::open: unsupported bit depth %d", bit_depth));
    const bool to_stdout = filename == "-";
    if (n_frames == AudioInputStream::N_FRAMES_UNKNOWN && !wav_pipe && to_stdout)
      return Error ("unable to write wav format to standard out without input length information");
    RawFormat format;
    format.bit_depth = bit_depth;
    format.encoding = encoding;
    Error err;
    // named files take the conversion rule of the reference's libsndfile path (sfoutputstream.cc:148-155)
    m_codec = PcmCodec::create (format, err, /* libsndfile_int_rule */ !to_stdout);
    if (err)
      return err;
    if (to_stdout)
      m_file = stdout;
    else
      {
        m_file = fopen (filename.c_str(), "w");
        if (!m_file)
          return Error (strerror (errno));
        m_close = true;
        m_fix_header = n_frames == AudioInputStream::N_FRAMES_UNKNOWN;    // sizes are patched in close()
      }
    const bool unknown = wav_pipe || n_frames == AudioInputStream::N_FRAMES_UNKNOWN;
    const uint64_t data_size = unknown ? 0 : uint64_t (n_frames) * n_channels * ((bit_depth + 7) / 8);
    m_close_padding = data_size & 1;
    m_rf64 = rf64 && !to_stdout && !wav_pipe;
    if (!m_rf64 && !unknown && 36 + data_size + m_close_padding > 0xFFFFFFFFull)
      return Error (string_printf ("wav output of %llu bytes does not fit a RIFF header (use --output-format rf64)", (unsigned long long) data_size));
    std::vector<unsigned char> h;
    auto str = [&] (const char *s) { h.insert (h.end(), s, s + 4); };
    auto u32 = [&] (uint32_t u) { for (int i = 0; i < 4; i++) h.push_back ((unsigned char) (u >> (8 * i))); };
    auto u64 = [&] (uint64_t u) { for (int i = 0; i < 8; i++) h.push_back ((unsigned char) (u >> (8 * i))); };
    auto u16 = [&] (uint16_t u) { h.push_back ((unsigned char) u); h.push_back ((unsigned char) (u >> 8)); };
    if (m_rf64)
      {
        // EBU Tech 3306: 32 bit sizes read 0xFFFFFFFF, the 64 bit ones live in the ds64 chunk that follows "WAVE"
        // (what libsndfile writes for SF_FORMAT_RF64, reference sfoutputstream.cc:73)
        m_fix_header = unknown;
        str ("RF64");
        u32 (0xFFFFFFFFu);
        str ("WAVE");
        str ("ds64");
        u32 (28);
        u64 (unknown ? 0 : RF64_HEADER_BYTES - 8 + data_size + m_close_padding);    // riff size
        u64 (data_size);
        u64 (unknown ? 0 : n_frames);                                               // sample count
        u32 (0);                                                                    // no chunk size table
      }
    else
      {
        str ("RIFF");
        u32 (unknown ? 0xFFFFFFFFu : uint32_t (36 + data_size + m_close_padding));
        str ("WAVE");
      }
    str ("fmt ");
    u32 (16);
    u16 (encoding == Encoding::FLOAT ? 3 : 1);
    u16 (n_channels);
    u32 (sample_rate);
    u32 (sample_rate * n_channels * bit_depth / 8);
    u16 (n_channels * bit_depth / 8);
    u16 (bit_depth);
    str ("data");
    u32 (unknown || m_rf64 ? 0xFFFFFFFFu : uint32_t (data_size));
    fwrite (h.data(), 1, h.size(), m_file);
    if (ferror (m_file))
      return Error ("write wav header failed");
    m_bit_depth = bit_depth;
    m_sample_rate = sample_rate;
    m_n_channels = n_channels;
    return Error::Code::NONE;
  }
  int bit_depth() const override { return m_bit_depth; }
  int sample_rate() const override { return m_sample_rate; }
  int n_channels() const override { return m_n_channels; }
  Error
  write_frames (const std::vector<float>& samples) override
  {
    if (samples.empty())
      return Error::Code::NONE;
    const size_t block = 8192 * size_t (m_n_channels);
    const int width = m_bit_depth / 8;
    std::vector<unsigned char> bytes (block * width);
    for (size_t pos = 0; pos < samples.size(); pos += block)
      {
        const size_t todo = std::min (block, samples.size() - pos);
        m_codec->encode (samples.data() + pos, bytes.data(), todo);
        fwrite (bytes.data(), 1, todo * width, m_file);
        if (ferror (m_file))
          return Error (string_printf ("write sample data failed (%s)", strerror (errno)));
        m_bytes_written += todo * width;
      }
    return Error::Code::NONE;
  }
  bool
  raw_access (RawFormat& format, bool& direct16) const override
  {
    format = m_codec->format();
    format.n_channels = m_n_channels;
    format.sample_rate = m_sample_rate;
    direct16 = m_codec->direct16();
    return true;
  }
  Error
  write_raw (const unsigned char *bytes, size_t n_frames) override
  {
    const size_t n = n_frames * m_n_channels * (m_bit_depth / 8);
    fwrite (bytes, 1, n, m_file);
    if (ferror (m_file))
      return Error (string_printf ("write sample data failed (%s)", strerror (errno)));
    m_bytes_written += n;
    return Error::Code::NONE;
  }
  bool raw_region (int& fd, uint64_t& byte_offset) override { return output_region (m_file, m_close, fd, byte_offset); }
  Error
  raw_region_written (size_t n_frames) override
  {
    const size_t n = n_frames * m_n_channels * (m_bit_depth / 8);
    if (fseeko (m_file, off_t (n), SEEK_CUR) != 0)
      return Error (string_printf ("write sample data failed (%s)", strerror (errno)));
    m_bytes_written += n;
    return Error::Code::NONE;
  }
  Error
  close() override
  {
    if (!m_file)
      return Error::Code::NONE;
    if (m_fix_header)
      {
        m_close_padding = m_bytes_written & 1;
      }
    for (size_t i = 0; i < m_close_padding; i++)
      fputc (0, m_file);
    Error size_error = Error::Code::NONE;
    if (m_fix_header)
      {
        unsigned char b[8];
        auto put = [&] (long pos, uint64_t v, int n) {
          for (int i = 0; i < n; i++) b[i] = (unsigned char) (v >> (8 * i));
          fseek (m_file, pos, SEEK_SET);
          fwrite (b, 1, n, m_file);
        };
        if (m_rf64)
          {
            put (20, RF64_HEADER_BYTES - 8 + m_bytes_written + m_close_padding, 8);
            put (28, m_bytes_written, 8);
            put (36, m_bytes_written / (size_t (m_n_channels) * (m_bit_depth / 8)), 8);
          }
        else if (36 + m_bytes_written + m_close_padding <= 0xFFFFFFFFull)
          {
            put (4, 36 + m_bytes_written + m_close_padding, 4);
            put (40, m_bytes_written, 4);
          }
        else
          size_error = Error (string_printf ("wav output of %llu bytes does not fit a RIFF header (use --output-format rf64)",
                                             (unsigned long long) m_bytes_written));
      }
    fflush (m_file);
    const bool failed = ferror (m_file);
    if (m_close)
      fclose (m_file);
    m_file = nullptr;
    if (failed)
      return Error ("error during flush");
    return size_error;
  }
};

} // namespace

std::unique_ptr<AudioInputStream>
AudioInputStream::create (const std::string& filename, Error& err)
{
  if (params().input_format == Format::RAW)
    {
      auto s = std::make_unique<RawInputStream>();
      err = s->open (filename, StreamParams::raw_input_format);
      if (err)
        return nullptr;
      return s;
    }
  if (params().input_format == Format::AUTO || params().input_format == Format::WAV_PIPE)
    {
      auto s = std::make_unique<WavInputStream>();
      err = s->open (filename, params().input_format == Format::WAV_PIPE);
      if (err)
        return nullptr;
      return s;
    }
  err = Error ("selected format is not supported as input format");
  return nullptr;
}

std::unique_ptr<AudioOutputStream>
AudioOutputStream::create (const std::string& filename, int n_channels, int sample_rate, int bit_depth, Encoding encoding,
                           size_t n_frames, Error& err)
{
  if (params().output_format == Format::RAW)
    {
      auto s = std::make_unique<RawOutputStream>();
      err = s->open (filename, StreamParams::raw_output_format);
      if (err)
        return nullptr;
      return s;
    }
  auto s = std::make_unique<WavOutputStream>();
  err = s->open (filename, n_channels, sample_rate, bit_depth, encoding, n_frames, params().output_format == Format::WAV_PIPE,
                 params().output_format == Format::RF64);
  if (err)
    return nullptr;
  return s;
}

Error
WavData::load (AudioInputStream *in_stream)
{
  m_samples.clear();
  if (in_stream->n_frames() != AudioInputStream::N_FRAMES_UNKNOWN)
    m_samples.reserve (in_stream->n_frames() * in_stream->n_channels());
  std::vector<float> buffer;
  while (true)
    {
      Error err = in_stream->read_frames (buffer, 65536);
      if (err)
        return err;
      if (buffer.empty())
        break;
      m_samples.insert (m_samples.end(), buffer.begin(), buffer.end());
    }
  m_sample_rate = in_stream->sample_rate();
  m_n_channels = in_stream->n_channels();
  m_bit_depth = in_stream->bit_depth();
  return Error::Code::NONE;
}

Error
WavData::load (const std::string& filename)
{
  Error err;
  auto in_stream = AudioInputStream::create (filename, err);
  if (err)
    return err;
  return load (in_stream.get());
}

Error
WavData::save (const std::string& filename) const
{
  Error err;
  auto out_stream = AudioOutputStream::create (filename, m_n_channels, m_sample_rate, m_bit_depth, Encoding::SIGNED,
                                               m_samples.size() / m_n_channels, err);
  if (err)
    return err;
  err = out_stream->write_frames (m_samples);
  if (err)
    return err;
  return out_stream->close();
}

} // namespace awm
