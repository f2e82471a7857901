#include "wmfile.hh"
#include "context.hh"
#include "wmget.hh"
#include "utils.hh"
#include <algorithm>
#include <cmath>

namespace awm {

namespace {

// pinned host staging buffer (PCIe transfers at full rate), grows geometrically
struct PinnedBytes
{
  unsigned char *ptr = nullptr;
  size_t capacity = 0;
  ~PinnedBytes() { if (ptr) (void) hipHostFree (ptr); }
  bool
  reserve (size_t n, size_t keep)
  {
    if (n <= capacity)
      return true;
    unsigned char *np = nullptr;
    const size_t cap = std::max<size_t> (n, capacity * 2);
    if (hipHostMalloc (reinterpret_cast<void **> (&np), cap, hipHostMallocDefault) != hipSuccess)
      return false;
    if (ptr)
      {
        std::copy (ptr, ptr + keep, np);
        (void) hipHostFree (ptr);
      }
    ptr = np;
    capacity = cap;
    return true;
  }
};

bool
device_codec_supported (const RawFormat& f)
{
  if (f.encoding == Encoding::FLOAT)
    return f.bit_depth == 32 || f.bit_depth == 64;
  return f.bit_depth == 8 || f.bit_depth == 16 || f.bit_depth == 24 || f.bit_depth == 32;
}

int encoding_id (Encoding e) { return e == Encoding::SIGNED ? 0 : (e == Encoding::UNSIGNED ? 1 : 2); }

/* Whole stream -> float32 PCM in HBM.  Streams that can hand out their sample bytes are read straight into pinned
 * memory, cross PCIe in their own format and are converted on the device (awm_pcm_decode_d, same rules as the host
 * codec); everything else goes through read_frames. */
Error
load_stream_to_device (awm_ctx *ctx, AudioInputStream *in_stream, DevBuffer& d_pcm, size_t& n_values)
{
  const int C = in_stream->n_channels();
  n_values = 0;
  PinnedBytes host;
  RawFormat fmt;
  if (in_stream->raw_access (fmt) && device_codec_supported (fmt))
    {
      const size_t frame_bytes = size_t (C) * (fmt.bit_depth / 8);
      size_t frames = 0;
      if (in_stream->n_frames() != AudioInputStream::N_FRAMES_UNKNOWN && !host.reserve ((in_stream->n_frames() + 1) * frame_bytes, 0))
        return Error ("out of (pinned) host memory");
      while (true)
        {
          const size_t block = size_t (1) << 22;                       // frames per read
          if (!host.reserve ((frames + block) * frame_bytes, frames * frame_bytes))
            return Error ("out of (pinned) host memory");
          size_t got = 0;
          Error err = in_stream->read_raw (host.ptr + frames * frame_bytes, block, got);
          if (err)
            return err;
          if (!got)
            break;
          frames += got;
        }
      n_values = frames * C;
      if (!n_values)
        return Error::Code::NONE;
      DevBuffer d_bytes;
      if (d_bytes.reserve (frames * frame_bytes) || d_pcm.reserve (n_values * sizeof (float)))
        return Error (awm_last_error());
      bool ok = hipMemcpyAsync (d_bytes.ptr, host.ptr, frames * frame_bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess
             && awm_pcm_decode_d (ctx, d_bytes.ptr, n_values, fmt.bit_depth, encoding_id (fmt.encoding), fmt.endian == RawFormat::BIG, d_pcm.as<float>()) == 0
             && hipStreamSynchronize (ctx->stream) == hipSuccess;
      d_bytes.release();
      return ok ? Error (Error::Code::NONE) : Error (std::string ("GPU staging failed: ") + awm_last_error());
    }
  std::vector<float> tile;
  while (true)
    {
      Error err = in_stream->read_frames (tile, 1 << 20);       // the stream surface accepts any count
      if (err)
        return err;
      if (tile.empty())
        break;
      if (!host.reserve ((n_values + tile.size()) * sizeof (float), n_values * sizeof (float)))
        return Error ("out of (pinned) host memory");
      std::copy (tile.begin(), tile.end(), reinterpret_cast<float *> (host.ptr) + n_values);
      n_values += tile.size();
    }
  if (!n_values)
    return Error::Code::NONE;
  if (d_pcm.reserve (n_values * sizeof (float)))
    return Error (awm_last_error());
  if (hipMemcpyAsync (d_pcm.ptr, host.ptr, n_values * sizeof (float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess
      || hipStreamSynchronize (ctx->stream) != hipSuccess)
    return Error ("GPU transfer failed");
  return Error::Code::NONE;
}

/* float32 PCM in HBM -> output stream, mirror image of load_stream_to_device */
Error
store_device_to_stream (awm_ctx *ctx, AudioOutputStream *out_stream, const float *d_pcm, size_t n_values)
{
  if (!n_values)
    return Error::Code::NONE;
  const int C = out_stream->n_channels();
  PinnedBytes host;
  RawFormat fmt;
  bool direct16 = false;
  if (out_stream->raw_access (fmt, direct16) && device_codec_supported (fmt))
    {
      const size_t bytes = n_values * (fmt.bit_depth / 8);
      DevBuffer d_bytes;
      if (d_bytes.reserve (bytes) || !host.reserve (bytes, 0))
        return Error ("out of memory for output staging");
      bool ok = awm_pcm_encode_d (ctx, d_pcm, n_values, fmt.bit_depth, encoding_id (fmt.encoding), fmt.endian == RawFormat::BIG, direct16, d_bytes.ptr) == 0
             && hipMemcpyAsync (host.ptr, d_bytes.ptr, bytes, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess
             && hipStreamSynchronize (ctx->stream) == hipSuccess;
      d_bytes.release();
      if (!ok)
        return Error (std::string ("GPU staging failed: ") + awm_last_error());
      return out_stream->write_raw (host.ptr, n_values / C);
    }
  if (!host.reserve (n_values * sizeof (float), 0))
    return Error ("out of (pinned) host memory");
  if (hipMemcpyAsync (host.ptr, d_pcm, n_values * sizeof (float), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess
      || hipStreamSynchronize (ctx->stream) != hipSuccess)
    return Error ("GPU transfer failed");
  const float *result = reinterpret_cast<const float *> (host.ptr);
  const size_t tile = size_t (1 << 20) * C;                     // write in tiles through the unchanged stream surface
  for (size_t pos = 0; pos < n_values; pos += tile)
    {
      std::vector<float> part (result + pos, result + std::min (n_values, pos + tile));
      Error err = out_stream->write_frames (part);
      if (err)
        return err;
    }
  return Error::Code::NONE;
}

/* "Data Blocks" counter of WatermarkGen (reference wmadd.cc:311-313, 346-351): depends only on how many frames the
 * streaming loop of the reference pushes through the generator, i.e. on the latency of synth + limiter */
int
count_data_blocks (size_t n_frames, int sample_rate, bool limiter)
{
  const size_t N = Params::frame_size, block = mark_block_frame_count();
  const size_t lim_block = size_t (sample_rate) * size_t (Params::limiter_block_size_ms) / 1000;
  size_t total_in = 0, total_out = 0, frame_number = 2 * block - Params::frames_pad_start, data_blocks = 0, lim_buffer = 0;
  bool first_frame = true;
  while (true)
    {
      const size_t got = std::min (N, n_frames - total_in);
      total_in += got;
      if (got < N && total_in == total_out)
        break;
      frame_number++;
      if (frame_number % block == 0)
        data_blocks++;
      size_t out = first_frame ? 0 : N;                    // WatermarkSynth emits nothing for the first frame
      first_frame = false;
      if (limiter)
        {
          lim_buffer += out;
          const size_t buffered_blocks = lim_buffer / lim_block;
          out = buffered_blocks < 2 ? 0 : (buffered_blocks - 1) * lim_block;
          lim_buffer -= out;
        }
      total_out += std::min (out, total_in - total_out);
    }
  return std::max (int (data_blocks) - 1, 0);
}

} // namespace

int
add_stream_watermark (awm_ctx *ctx, const Key& key, AudioInputStream *in_stream, AudioOutputStream *out_stream,
                      const std::string& bits, size_t zero_frames)
{
  auto bitvec = parse_payload (bits);
  if (bitvec.empty())
    return 1;
  if (in_stream->sample_rate() != out_stream->sample_rate())
    {
      error ("audiowmark: input sample rate (%d) and output sample rate (%d) don't match\n", in_stream->sample_rate(), out_stream->sample_rate());
      return 1;
    }
  if (in_stream->n_channels() != out_stream->n_channels())
    {
      error ("audiowmark: input channels (%d) and output channels (%d) don't match\n", in_stream->n_channels(), out_stream->n_channels());
      return 1;
    }
  if (zero_frames)
    {
      error ("audiowmark: zero_frames (HLS segment watermarking) is not supported by the GPU path\n");
      return 1;
    }
  if (in_stream->sample_rate() != Params::mark_sample_rate
      && (!awm_resample_frames (ctx, 1024, in_stream->sample_rate(), Params::mark_sample_rate)
          || !awm_resample_frames (ctx, 1024, Params::mark_sample_rate, in_stream->sample_rate())))
    {
      // the reference falls back to zita's VResampler for such ratios (resample.cc:233-270)
      error ("audiowmark: resampling from old_rate=%d to new_rate=%d not implemented\n", in_stream->sample_rate(), Params::mark_sample_rate);
      return 1;
    }
  info ("Message:      %s\n", bit_vec_to_str (bitvec).c_str());
  info ("Strength:     %.6g\n\n", Params::water_delta * 1000);
  if (in_stream->n_frames() == AudioInputStream::N_FRAMES_UNKNOWN)
    info ("Time:         unknown\n");
  else
    {
      const size_t orig_seconds = in_stream->n_frames() / in_stream->sample_rate();
      info ("Time:         %zd:%02zd\n", orig_seconds / 60, orig_seconds % 60);
    }
  info ("Sample Rate:  %d\n", in_stream->sample_rate());
  info ("Channels:     %d\n", in_stream->n_channels());

  const int C = in_stream->n_channels();
  DevBuffer d_in, d_out;
  auto cleanup = [&] { d_in.release(); d_out.release(); };
  size_t n_values = 0;
  Error err = load_stream_to_device (ctx, in_stream, d_in, n_values);
  if (err)
    {
      error ("audiowmark: input stream read failed: %s\n", err.message());
      cleanup();
      return 1;
    }
  const size_t n_frames = n_values / C;
  if (n_values)
    {
      if (d_out.reserve (n_values * sizeof (float))
          || awm_add_watermark_d (ctx, key.aes_key(), bit_vec_to_str (bitvec).c_str(), d_in.as<float>(), d_out.as<float>(), n_frames, C,
                                  in_stream->sample_rate()) != 0)
        {
          error ("audiowmark: GPU watermarking failed: %s\n", awm_last_error());
          cleanup();
          return 1;
        }
      if (Params::snr)
        {
          // the reference measures the watermark before the limiter (wmadd.cc:553-563); with the limiter
          // active this is the power of (output - original), which includes the limiter's gain change
          std::vector<float> orig (n_values), result (n_values);
          if (hipMemcpy (orig.data(), d_in.ptr, n_values * sizeof (float), hipMemcpyDeviceToHost) != hipSuccess
              || hipMemcpy (result.data(), d_out.ptr, n_values * sizeof (float), hipMemcpyDeviceToHost) != hipSuccess)
            {
              error ("audiowmark: GPU transfer failed\n");
              cleanup();
              return 1;
            }
          double delta_power = 0, signal_power = 0;
          for (size_t i = 0; i < n_values; i++)
            {
              const double o = orig[i], d = double (result[i]) - o;
              delta_power += d * d;
              signal_power += o * o;
            }
          info ("SNR:          %f dB\n", 10 * log10 (signal_power / delta_power));
        }
      err = store_device_to_stream (ctx, out_stream, d_out.as<float>(), n_values);
      if (err)
        {
          error ("audiowmark output write failed: %s\n", err.message());
          cleanup();
          return 1;
        }
    }
  cleanup();
  info ("Data Blocks:  %d\n", count_data_blocks (n_frames, in_stream->sample_rate(), !Params::test_no_limiter));
  if (in_stream->n_frames() != AudioInputStream::N_FRAMES_UNKNOWN && n_frames != in_stream->n_frames())
    {
      auto msg = string_printf ("unexpected EOF; input frames (%zd) != output frames (%zd)", in_stream->n_frames(), n_frames);
      if (Params::strict)
        {
          error ("audiowmark: error: %s\n", msg.c_str());
          return 1;
        }
      warning ("audiowmark: warning: %s\n", msg.c_str());
    }
  err = out_stream->close();
  if (err)
    {
      error ("audiowmark: closing output stream failed: %s\n", err.message());
      return 1;
    }
  return 0;
}

static void
info_format (const std::string& label, const RawFormat& format)
{
  const char *e = format.encoding == Encoding::SIGNED ? "signed" : format.encoding == Encoding::UNSIGNED ? "unsigned" : "float";
  info ("%-13s %d Hz, %d Channels, %d Bit (%s %s-endian)\n", (label + ":").c_str(), format.sample_rate, format.n_channels,
        format.bit_depth, e, format.endian == RawFormat::LITTLE ? "little" : "big");
}

int
add_watermark (awm_ctx *ctx, const Key& key, const std::string& infile, const std::string& outfile, const std::string& bits)
{
  Error err;
  auto in_stream = AudioInputStream::create (infile, err);
  if (err)
    {
      error ("audiowmark: error opening %s: %s\n", infile.c_str(), err.message());
      return 1;
    }
  int out_bit_depth = in_stream->bit_depth();
  Encoding out_encoding = in_stream->encoding();
  if (in_stream->bit_depth() < 16)
    {
      out_bit_depth = 16;
      out_encoding = Encoding::SIGNED;
    }
  auto out_stream = AudioOutputStream::create (outfile, in_stream->n_channels(), in_stream->sample_rate(), out_bit_depth, out_encoding,
                                               in_stream->n_frames(), err);
  if (err)
    {
      error ("audiowmark: error writing to %s: %s\n", outfile.c_str(), err.message());
      return 1;
    }
  info ("Input:        %s\n", infile.c_str());
  if (Params::input_format == Format::RAW)
    info_format ("Raw Input", StreamParams::raw_input_format);
  info ("Output:       %s\n", outfile.c_str());
  if (Params::output_format == Format::RAW)
    info_format ("Raw Output", StreamParams::raw_output_format);
  return add_stream_watermark (ctx, key, in_stream.get(), out_stream.get(), bits, 0);
}

int
get_watermark (awm_ctx *ctx, const std::vector<Key>& key_list, const std::string& infile, const std::string& orig_pattern)
{
  std::vector<int> orig_bitvec;
  if (!orig_pattern.empty())
    {
      orig_bitvec = parse_payload (orig_pattern);
      if (orig_bitvec.empty())
        return 1;
    }
  Error err;
  auto in_stream = AudioInputStream::create (infile, err);
  if (err)
    {
      error ("audiowmark: error loading %s: %s\n", infile.c_str(), err.message());
      return 1;
    }
  const int C = in_stream->n_channels();
  DevBuffer d_in;
  size_t n_values = 0;
  err = load_stream_to_device (ctx, in_stream.get(), d_in, n_values);
  if (err)
    {
      error ("audiowmark: error loading %s: %s\n", infile.c_str(), err.message());
      d_in.release();
      return 1;
    }
  if (in_stream->sample_rate() != Params::mark_sample_rate)
    {
      // WavChunkLoader resamples the whole stream to the watermark rate before anything else (wavchunkloader.cc:70-71, 200-216)
      const size_t in_frames = n_values / C;
      const size_t out_frames = awm_resample_frames (ctx, in_frames, in_stream->sample_rate(), Params::mark_sample_rate);
      DevBuffer d_res;
      if ((in_frames && !out_frames) || d_res.reserve (std::max<size_t> (1, out_frames * C * sizeof (float)))
          || awm_resample_d (ctx, d_in.as<float>(), in_frames, C, in_stream->sample_rate(), Params::mark_sample_rate, d_res.as<float>(), out_frames))
        {
          error ("audiowmark: resampling from old_rate=%d to new_rate=%d not implemented\n", in_stream->sample_rate(), Params::mark_sample_rate);
          d_in.release();
          d_res.release();
          return 1;
        }
      (void) hipStreamSynchronize (ctx->stream);
      d_in.release();
      d_in = d_res;
      n_values = out_frames * C;
    }
  if (Params::test_truncate)
    n_values = std::min (n_values, size_t (Params::mark_sample_rate) * C * Params::test_truncate);
  const size_t n_frames = n_values / C;
  ResultSet result_set;
  if (n_frames)
    {
      DeviceWav wav;
      wav.data = d_in.as<float>();
      wav.n_frames = n_frames;
      wav.n_channels = C;
      wav.sample_rate = Params::mark_sample_rate;
      speed_print_results = !orig_bitvec.empty();      // decode (..., orig_bits, ...): the detect_speed report line of `cmp`
      const int rc = get_watermark_device (ctx, key_list, wav, result_set);
      if (rc)
        {
          error ("audiowmark: GPU detection failed: %s\n", awm_last_error());
          d_in.release();
          return 1;
        }
    }
  else
    result_set.sort (key_list);
  d_in.release();
  const size_t time_length = lrint (double (n_values) / (double (Params::mark_sample_rate) * C));

  /* report (reference wmget.cc:941-969) */
  if (!Params::json_output.empty())
    result_set.print_json (time_length, Params::json_output);
  if (Params::json_output != "-")
    result_set.print();
  if (!orig_bitvec.empty())
    {
      const int match_count = result_set.print_match_count (orig_bitvec);
      result_set.print_debug_sync();
      if (Params::expect_matches >= 0)
        {
          printf ("expect_matches %d\n", Params::expect_matches);
          if (match_count != Params::expect_matches)
            return 1;
        }
      else if (!match_count)
        return 1;
    }
  return 0;
}

/* `audiowmark test-change-speed in out speed` (reference audiowmark.cc:419-437): resample_ratio (in, 1 / speed, same rate) */
int
test_change_speed (awm_ctx *ctx, const std::string& infile, const std::string& outfile, double speed)
{
  Error err;
  auto in_stream = AudioInputStream::create (infile, err);
  if (err)
    {
      error ("audiowmark: error loading %s: %s\n", infile.c_str(), err.message());
      return 1;
    }
  const int C = in_stream->n_channels();
  DevBuffer d_in, d_out;
  size_t n_values = 0;
  err = load_stream_to_device (ctx, in_stream.get(), d_in, n_values);
  if (err)
    {
      error ("audiowmark: error loading %s: %s\n", infile.c_str(), err.message());
      d_in.release();
      return 1;
    }
  const size_t in_frames = n_values / C;
  const size_t out_frames = awm_resample_ratio_frames (in_frames, C, in_stream->sample_rate(), 1 / speed, -1);
  if (d_out.reserve (std::max<size_t> (16, out_frames * C * sizeof (float)))
      || awm_resample_ratio_d (ctx, d_in.as<float>(), in_frames, C, in_stream->sample_rate(), 1 / speed, -1, d_out.as<float>(), out_frames))
    {
      error ("audiowmark: failed to setup vresampler with ratio=%f\n", 1 / speed);
      d_in.release();
      d_out.release();
      return 1;
    }
  auto out_stream = AudioOutputStream::create (outfile, C, in_stream->sample_rate(), in_stream->bit_depth() < 16 ? 16 : in_stream->bit_depth(),
                                               in_stream->bit_depth() < 16 ? Encoding::SIGNED : in_stream->encoding(), out_frames, err);
  if (!err)
    err = store_device_to_stream (ctx, out_stream.get(), d_out.as<float>(), out_frames * C);
  if (!err)
    err = out_stream->close();
  d_in.release();
  d_out.release();
  if (err)
    {
      error ("audiowmark: error saving %s: %s\n", outfile.c_str(), err.message());
      return 1;
    }
  return 0;
}

} // namespace awm
