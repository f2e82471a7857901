// audiowmark -- command line front end of the MI355X watermark path.
//
// Same commands, option names, messages and exit codes as the reference's command line (reference src/audiowmark.cc:
// usage :46-90, options :661-881, commands :911-1079) for the subset in scope: add / get / cmp / gen-key /
// test-gen-noise / test-snr / test-change-speed on raw and WAV data.  The structure is this file's own: every command
// is described by a table of option specifications (name, kind, action); one generic pass extracts the options a
// command knows from the argument list, whatever is left must be its positional arguments.
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <cstring>
#include <functional>
#include "audiostream.hh"
#include "context.hh"
#include "utils.hh"
#include "wmfile.hh"

using namespace awm;
using std::string;
using std::vector;

namespace {

[[noreturn]] void
die (const string& message)
{
  error ("%s", message.c_str());
  exit (1);
}

/* ---- values ------------------------------------------------------------------------------------------------------ */
int
to_int (const string& s)
{
  char *end = nullptr;
  const long v = strtol (s.c_str(), &end, 0);
  if (end && *end)
    die ("audiowmark: error during string->int conversion: " + s + "\n");
  return int (v);
}

float
to_float (const string& s)
{
  char *end = nullptr;
  const double v = strtod (s.c_str(), &end);
  if (end && *end)
    die ("audiowmark: error during string->float conversion: " + s + "\n");
  return float (v);
}

template<class T> T
from_names (const string& what, const string& s, std::initializer_list<std::pair<const char *, T>> names)
{
  for (const auto& n : names)
    if (s == n.first)
      return n.second;
  die ("audiowmark: unsupported " + what + " '" + s + "'\n");
}

Format to_format (const string& s) { return from_names<Format> ("format", s, { { "raw", Format::RAW }, { "auto", Format::AUTO }, { "rf64", Format::RF64 }, { "wav-pipe", Format::WAV_PIPE } }); }
RawFormat::Endian to_endian (const string& s) { return from_names<RawFormat::Endian> ("endianness", s, { { "little", RawFormat::LITTLE }, { "big", RawFormat::BIG } }); }

void
set_encoding (RawFormat& fmt, const string& s)
{
  struct Choice { Encoding encoding; int float_bits; };
  const Choice c = from_names<Choice> ("encoding", s, { { "signed", { Encoding::SIGNED, 0 } }, { "unsigned", { Encoding::UNSIGNED, 0 } },
                                                         { "float", { Encoding::FLOAT, 32 } }, { "double", { Encoding::FLOAT, 64 } } });
  fmt.encoding = c.encoding;
  if (c.float_bits)
    fmt.bit_depth = c.float_bits;
}

void
set_bits (RawFormat& fmt, int bits)
{
  if (fmt.encoding == Encoding::FLOAT)
    die ("audiowmark: bit depth can not be changed for float / double encoding\n");
  fmt.bit_depth = bits;
}

/* ---- option tables ------------------------------------------------------------------------------------------------- */
struct Option
{
  enum Kind { FLAG, VALUE, MULTI } kind;
  const char *name;
  std::function<void (const string&)> action;       // FLAG: called with "" when present; VALUE: last occurrence; MULTI: every one
};
typedef vector<Option> Options;

bool looks_like_option (const string& s) { return s.size() > 1 && s[0] == '-'; }

/* removes every occurrence of the option from `args` ("--opt value" and "--opt=value"), returns its values in order */
vector<string>
extract (vector<string>& args, const string& name, bool takes_value)
{
  vector<string> values;
  for (size_t i = 0; i < args.size();)
    {
      if (args[i] == name && !takes_value)
        {
          values.push_back ("");
          args.erase (args.begin() + i);
        }
      else if (args[i] == name && i + 1 < args.size())
        {
          values.push_back (args[i + 1]);
          args.erase (args.begin() + i, args.begin() + i + 2);
        }
      else if (takes_value && args[i].compare (0, name.size() + 1, name + "=") == 0)
        {
          values.push_back (args[i].substr (name.size() + 1));
          args.erase (args.begin() + i);
        }
      else
        i++;
    }
  return values;
}

void
apply_options (vector<string>& args, const Options& options)
{
  for (const Option& o : options)
    {
      const auto values = extract (args, o.name, o.kind != Option::FLAG);
      if (values.empty())
        continue;
      if (o.kind == Option::MULTI)
        for (const auto& v : values)
          o.action (v);
      else
        o.action (values.back());
    }
}

/* what is left must be exactly the positional arguments of the command */
vector<string>
positional (const string& command, const vector<string>& args, std::initializer_list<const char *> names)
{
  for (const auto& a : args)
    if (looks_like_option (a))
      die ("audiowmark: unsupported option '" + a + "' for command '" + command + "' (use audiowmark -h)\n");
  if (args.size() == names.size())
    return args;
  error ("audiowmark: error parsing arguments for command '%s' (use audiowmark -h)\n\n", command.c_str());
  string usage = "usage: audiowmark " + command + " [options...]";
  for (const char *n : names)
    usage += string (" <") + n + ">";
  die (usage + "\n");
}

Options
shared_options()
{
  return {
    { Option::VALUE, "--short", [] (const string&) { die ("audiowmark: --short payloads are not supported by the GPU path\n"); } },
    { Option::VALUE, "--frames-per-bit", [] (const string& v) {
        // (reference audiowmark.cc:675: any integer; the compute entry points take 1 .. 8 and say so otherwise)
        params().frames_per_bit = to_int (v);
      } },
    { Option::FLAG, "--linear", [] (const string&) { params().mix = false; } },
  };
}

Options
stream_options (bool with_output)
{
  RawFormat& in = StreamParams::raw_input_format;
  RawFormat& out = StreamParams::raw_output_format;
  Options o {
    { Option::VALUE, "--input-format", [] (const string& v) { params().input_format = to_format (v); } },
  };
  if (with_output)
    o.push_back ({ Option::VALUE, "--output-format", [] (const string& v) { params().output_format = to_format (v); } });
  o.push_back ({ Option::VALUE, "--format", [with_output] (const string& v) {
      params().input_format = to_format (v);
      if (with_output)
        params().output_format = params().input_format;
    } });
  const Options raw {
    { Option::VALUE, "--raw-input-endian", [&in] (const string& v) { in.endian = to_endian (v); } },
    { Option::VALUE, "--raw-output-endian", [&out] (const string& v) { out.endian = to_endian (v); } },
    { Option::VALUE, "--raw-endian", [&in, &out] (const string& v) { in.endian = out.endian = to_endian (v); } },
    { Option::VALUE, "--raw-input-encoding", [&in] (const string& v) { set_encoding (in, v); } },
    { Option::VALUE, "--raw-output-encoding", [&out] (const string& v) { set_encoding (out, v); } },
    { Option::VALUE, "--raw-encoding", [&in, &out] (const string& v) { set_encoding (in, v); set_encoding (out, v); } },
    { Option::VALUE, "--raw-input-bits", [&in] (const string& v) { set_bits (in, to_int (v)); } },
    { Option::VALUE, "--raw-output-bits", [&out] (const string& v) { set_bits (out, to_int (v)); } },
    { Option::VALUE, "--raw-bits", [&in, &out] (const string& v) { set_bits (in, to_int (v)); set_bits (out, to_int (v)); } },
    { Option::VALUE, "--raw-channels", [&in, &out] (const string& v) { in.n_channels = out.n_channels = to_int (v); } },
    { Option::VALUE, "--raw-rate", [&in, &out] (const string& v) { in.sample_rate = out.sample_rate = to_int (v); } },
  };
  o.insert (o.end(), raw.begin(), raw.end());
  return o;
}

void
check_stream_options()
{
  if (params().input_format == Format::RF64)
    die ("audiowmark: using rf64 as input format has no effect\n");
}

Options
key_options (vector<Key>& keys)
{
  return {
    { Option::MULTI, "--key", [&keys] (const string& file) { Key k; k.load_key (file); keys.push_back (k); } },
    { Option::MULTI, "--test-key", [&keys] (const string& n) { Key k; k.set_test_key (to_int (n)); keys.push_back (k); } },
  };
}

Options
add_options()
{
  Options o { { Option::FLAG, "--snr", [] (const string&) { params().snr = true; } } };
  const Options s = stream_options (true);
  o.insert (o.end(), s.begin(), s.end());
  o.push_back ({ Option::FLAG, "--test-no-limiter", [] (const string&) { params().test_no_limiter = true; } });
  o.push_back ({ Option::VALUE, "--strength", [] (const string& v) { params().water_delta = to_float (v) / 1000; } });
  return o;
}

Options
get_options (int& speed_options)
{
  Options o {
    { Option::VALUE, "--test-cut", [] (const string& v) { params().test_cut = to_int (v); } },
    { Option::VALUE, "--test-truncate", [] (const string& v) { params().test_truncate = to_int (v); } },
    { Option::FLAG, "--hard", [] (const string&) { params().hard = true; } },
    { Option::FLAG, "--test-no-sync", [] (const string&) { params().test_no_sync = true; } },
    { Option::FLAG, "--detect-speed", [&speed_options] (const string&) { params().detect_speed = true; speed_options++; } },
    { Option::FLAG, "--detect-speed-patient", [&speed_options] (const string&) { params().detect_speed_patient = true; speed_options++; } },
    { Option::VALUE, "--try-speed", [&speed_options] (const string& v) { params().try_speed = to_float (v); speed_options++; } },
    { Option::VALUE, "--test-speed", [] (const string& v) { params().test_speed = to_float (v); } },
    { Option::VALUE, "--json", [] (const string& v) { params().json_output = v; } },
    { Option::VALUE, "--chunk-size", [] (const string& v) {
        const float minutes = to_float (v);
        if (minutes < 10)
          die ("audiowmark: --chunk-size needs to be at least 10 minutes\n");
        params().get_chunk_size = minutes;
      } },
    { Option::VALUE, "--sync-threshold", [] (const string& v) { params().sync_threshold2 = to_float (v); } },
    { Option::VALUE, "--n-best", [] (const string& v) {
        const int n = to_int (v);
        if (n < 0)
          die ("audiowmark: --n-best should not be a negative number\n");
        params().get_n_best = n;
      } },
    { Option::VALUE, "--strength", [] (const string& v) { params().water_delta = to_float (v) / 1000; } },
  };
  // the reference's get / cmp always open the input through libsndfile; this build has the input options of add instead
  const Options s = stream_options (false);
  o.insert (o.end(), s.begin(), s.end());
  return o;
}

/* ---- commands ------------------------------------------------------------------------------------------------------ */
void
print_usage()
{
  static const char *const text =
    "usage: audiowmark <command> [ <args>... ]\n\n"
    "Commands:\n"
    "  * create a watermarked wav file with a message\n"
    "    audiowmark add <input_wav> <watermarked_wav> <message_hex>\n\n"
    "  * retrieve message\n"
    "    audiowmark get <watermarked_wav>\n\n"
    "  * compare watermark message with expected message\n"
    "    audiowmark cmp <watermarked_wav> <message_hex>\n\n"
    "  * generate 128-bit watermarking key, to be used with --key option\n"
    "    audiowmark gen-key <key_file> [ --name <key_name> ]\n\n"
    "Global options:\n"
    "  -q, --quiet             disable information messages\n"
    "  --strict                treat (minor) problems as errors\n\n"
    "Options for get / cmp:\n"
    "  --json <file>           write JSON results into file\n"
    "  --input-format <f>      raw | wav-pipe | auto (GPU build: get / cmp accept the input options of add)\n\n"
    "Options for add / get / cmp:\n"
    "  --key <file>            load watermarking key from file\n";
  fputs (text, stdout);
  printf ("  --strength <s>          set watermark strength              [%.6g]\n\n", params().water_delta * 1000);
  fputs ("  --input-format raw      use raw stream as input\n"
         "  --output-format raw     use raw stream as output\n"
         "  --format raw            use raw stream as input and output\n\n"
         "The options to set the raw stream parameters (such as --raw-rate\n"
         "or --raw-channels) are the reference's.\n", stdout);
}

// AWM_TIMING=1: milliseconds since the start of main() on stderr -- at the first HIP call's return (runtime initialisation), when the
// context exists, and at the end of the command (bench.py's e2e.cli leg: what of the wall time is the HIP runtime's, what is ours)
static const std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();
static void
timing_mark (const char *what)
{
  static const bool on = getenv ("AWM_TIMING") != nullptr;
  if (on)
    fprintf (stderr, "awm_timing %s %.3f\n", what, std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - g_t0).count());
}

struct Gpu
{
  awm_ctx *ctx = nullptr;
  Gpu()
  {
    timing_mark ("main");
    {
      int n = 0;
      (void) hipGetDeviceCount (&n);              // the first HIP call initialises the runtime (and loads the code object)
      timing_mark ("hip_runtime_up");
    }
    // AWM_DEVICE=n: which GPU of the node (default 0); AWM_DEVICES=a,b,...: `get` spreads long files over these GPUs (the file is
    // read through the first one, awm_ctx_set_helpers)
    const char *dev = getenv ("AWM_DEVICE"), *devs = getenv ("AWM_DEVICES");
    std::vector<int> ids;
    for (const char *p = devs; p && *p; )
      {
        char *end = nullptr;
        const long v = strtol (p, &end, 10);
        if (end == p)
          die ("audiowmark: AWM_DEVICES must be a comma separated list of device numbers\n");
        ids.push_back (int (v));
        p = *end == ',' ? end + 1 : end;
      }
    if (ids.empty())
      ids.push_back (dev ? atoi (dev) : 0);
    // on the default stream, chunks of `get` on two lanes: a file level run is bound by file I/O, and every additional HIP stream
    // costs ~190 MB of resident host memory
    if (awm_ctx_create_on_stream (ids[0], nullptr, &ctx) != 0)
      die (string ("audiowmark: ") + awm_last_error() + "\n");
    awm_ctx_set_chunk_lanes (ctx, 2);
    std::vector<awm_ctx *> helpers;
    for (size_t i = 1; i < ids.size(); i++)
      {
        awm_ctx *h = nullptr;
        if (awm_ctx_create (ids[i], &h) != 0)
          die (string ("audiowmark: ") + awm_last_error() + "\n");
        awm_ctx_set_chunk_lanes (h, 2);
        helpers.push_back (h);
      }
    if (!helpers.empty())
      awm_ctx_set_helpers (ctx, helpers.data(), int (helpers.size()));
    timing_mark ("context_ready");
  }
  // no destructor: the process ends right after the command (finish() below), which returns everything at once; tearing down the
  // context and the HIP runtime piece by piece costs 70-90 ms -- a third of an `add` of one hour of audio
};

/* end of a command: every file is closed by now, flush the standard streams and leave without the teardown of context and runtime */
[[noreturn]] void
finish (int rc)
{
  timing_mark ("command_done");
  if (getenv ("AWM_TIMING"))
    {
      double ms[8];
      awm_debug_file_timing (ms);       // (`add` at the watermark rate only; zeros otherwise)
      fprintf (stderr, "awm_timing add_calling_thread setup %.2f wait_input %.2f wait_output_slot %.2f queue_gpu_work %.2f final_gpu_wait %.2f "
                       "final_writer_wait %.2f teardown %.2f hand_on_output %.2f\n", ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[6], ms[7]);
    }
  fflush (nullptr);
  _exit (rc);
}

vector<Key>
keys_or_default (vector<Key> keys)
{
  if (keys.empty())
    keys.push_back (Key());
  return keys;
}

Key
single_key (const string& command, const vector<Key>& keys)
{
  if (keys.size() > 1)
    die ("audiowmark " + command + ": watermark key can at most be set once (--key / --test-key option)\n");
  return keys_or_default (keys)[0];
}

int
cmd_add (vector<string>& args)
{
  vector<Key> keys;
  apply_options (args, shared_options());
  apply_options (args, add_options());
  check_stream_options();
  apply_options (args, key_options (keys));
  const Key key = single_key ("add", keys);
  const auto pos = positional ("add", args, { "input_wav", "watermarked_wav", "message_hex" });
  Gpu gpu;
  return add_watermark (gpu.ctx, key, pos[0], pos[1], pos[2]);
}

int
cmd_get (vector<string>& args, bool cmp)
{
  vector<Key> keys;
  int speed_options = 0;
  apply_options (args, shared_options());
  apply_options (args, get_options (speed_options));
  if (speed_options > 1)
    die ("audiowmark: can only use one option: --detect-speed or --detect-speed-patient or --try-speed\n");
  check_stream_options();
  if (cmp)
    apply_options (args, { { Option::VALUE, "--expect-matches", [] (const string& v) { params().expect_matches = to_int (v); } } });
  apply_options (args, key_options (keys));
  const auto pos = cmp ? positional ("cmp", args, { "watermarked_wav", "message_hex" }) : positional ("get", args, { "watermarked_wav" });
  Gpu gpu;
  return get_watermark (gpu.ctx, keys_or_default (keys), pos[0], cmp ? pos[1] : "");
}

int
cmd_gen_key (vector<string>& args)
{
  string name;
  apply_options (args, { { Option::VALUE, "--name", [&name] (const string& v) { name = v; } } });
  const auto pos = positional ("gen-key", args, { "key_file" });
  FILE *f = fopen (pos[0].c_str(), "w");
  if (!f)
    {
      error ("audiowmark: error writing to file %s\n", pos[0].c_str());
      return 1;
    }
  fprintf (f, "# watermarking key for audiowmark\n\nkey %s\n", Random::gen_key().c_str());
  if (!name.empty())
    fprintf (f, "name %s\n", name.c_str());
  fclose (f);
  return 0;
}

int
cmd_test_change_speed (vector<string>& args)
{
  apply_options (args, shared_options());
  apply_options (args, stream_options (true));
  check_stream_options();
  const auto pos = positional ("test-change-speed", args, { "input_wav", "output_wav", "speed" });
  Gpu gpu;
  return test_change_speed (gpu.ctx, pos[0], pos[1], to_float (pos[2]));
}

int
cmd_test_gen_noise (vector<string>& args)
{
  // reference audiowmark.cc:399-417: stereo, uniform [-1, 1) from the key's data_up_down stream
  vector<Key> keys;
  int bits = 16;
  apply_options (args, shared_options());
  apply_options (args, { { Option::VALUE, "--bits", [&bits] (const string& v) { bits = to_int (v); } } });
  apply_options (args, key_options (keys));
  const Key key = single_key ("test-gen-noise", keys);
  const auto pos = positional ("test-gen-noise", args, { "output_wav", "seconds", "sample_rate" });
  const int rate = to_int (pos[2]);
  vector<float> noise (size_t (rate * to_float (pos[1])) * 2);
  awm_test_gen_noise (key.aes_key(), noise.size(), noise.data());
  const Error err = WavData (noise, 2, rate, bits).save (pos[0]);
  if (err)
    {
      error ("audiowmark: error saving %s: %s\n", pos[0].c_str(), err.message());
      return 1;
    }
  return 0;
}

int
cmd_test_snr (vector<string>& args)
{
  const auto pos = positional ("test-snr", args, { "orig_wav", "watermarked_wav" });
  WavData orig, marked;
  Error err = orig.load (pos[0]);
  if (!err)
    err = marked.load (pos[1]);
  if (err)
    {
      error ("audiowmark: error loading: %s\n", err.message());
      return 1;
    }
  if (orig.n_values() != marked.n_values())
    {
      error ("audiowmark: files have different length\n");
      return 1;
    }
  double noise_power = 0, signal_power = 0;
  for (size_t i = 0; i < orig.n_values(); i++)
    {
      const double s = orig.samples()[i], n = s - marked.samples()[i];
      noise_power += n * n;
      signal_power += s * s;
    }
  printf ("snr_db %f\n", 10 * log10 (signal_power / noise_power));
  return 0;
}

} // namespace

int
main (int argc, char **argv)
{
  // This process moves one file through the GPU and is bound by file I/O: tell the HIP runtime (before its first call) to keep all
  // streams on one hardware queue and to copy with shader blits instead of the SDMA engines.  Measured on the 60 min file: resident host
  // memory 1.07 -> 0.52 GB (add) and 1.19 -> 0.47 GB (get), run time unchanged (profiles/r02/rss.txt).  A value set by the caller wins.
  setenv ("GPU_MAX_HW_QUEUES", "1", 0);
  setenv ("HSA_ENABLE_SDMA", "0", 0);
  vector<string> args (argv + 1, argv + argc);
  bool help = false, version = false;
  apply_options (args, {
    { Option::FLAG, "--help", [&help] (const string&) { help = true; } },
    { Option::FLAG, "-h", [&help] (const string&) { help = true; } },
    { Option::FLAG, "--version", [&version] (const string&) { version = true; } },
    { Option::FLAG, "-v", [&version] (const string&) { version = true; } },
  });
  if (help)
    {
      print_usage();
      return 0;
    }
  if (version)
    {
      printf ("audiowmark 0.6.5 (%s)\n", awm_version());
      return 0;
    }
  apply_options (args, {
    { Option::FLAG, "--quiet", [] (const string&) { set_log_level (Log::WARNING); } },
    { Option::FLAG, "-q", [] (const string&) { set_log_level (Log::WARNING); } },
    { Option::FLAG, "--strict", [] (const string&) { params().strict = true; } },
  });
  if (args.empty())
    {
      error ("audiowmark: error parsing commandline args (use audiowmark -h)\n");
      return 1;
    }
  const string command = args.front();
  const std::pair<const char *, std::function<int (vector<string>&)>> commands[] = {
    { "add", cmd_add },
    { "get", [] (vector<string>& a) { return cmd_get (a, false); } },
    { "cmp", [] (vector<string>& a) { return cmd_get (a, true); } },
    { "gen-key", cmd_gen_key },
    { "test-change-speed", cmd_test_change_speed },
    { "test-gen-noise", cmd_test_gen_noise },
    { "test-snr", cmd_test_snr },
  };
  for (const auto& c : commands)
    if (command == c.first)
      {
        args.erase (args.begin());
        finish (c.second (args));
      }
  if (looks_like_option (command))
    error ("audiowmark: unsupported global option '%s' (use audiowmark -h)\n", command.c_str());
  else
    error ("audiowmark: unsupported command '%s' (use audiowmark -h)\n", command.c_str());
  return 1;
}
