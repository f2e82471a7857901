// audiowmark -- command line front end of the MI355X watermark path.  Commands, option names, messages
// and exit codes follow reference src/audiowmark.cc (print_usage :46-90, ArgParser :540-659, option parsing
// :661-881, main :911-1079) for the subset in scope: add / get / cmp / gen-key / test-gen-noise / test-snr
// on raw and WAV data at 44.1 kHz.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "audiostream.hh"
#include "context.hh"
#include "utils.hh"
#include "wmfile.hh"

using namespace awm;
using std::string;
using std::vector;

namespace {

void
print_usage()
{
  printf ("usage: audiowmark <command> [ <args>... ]\n\n");
  printf ("Commands:\n");
  printf ("  * create a watermarked wav file with a message\n");
  printf ("    audiowmark add <input_wav> <watermarked_wav> <message_hex>\n\n");
  printf ("  * retrieve message\n");
  printf ("    audiowmark get <watermarked_wav>\n\n");
  printf ("  * compare watermark message with expected message\n");
  printf ("    audiowmark cmp <watermarked_wav> <message_hex>\n\n");
  printf ("  * generate 128-bit watermarking key, to be used with --key option\n");
  printf ("    audiowmark gen-key <key_file> [ --name <key_name> ]\n\n");
  printf ("Global options:\n");
  printf ("  -q, --quiet             disable information messages\n");
  printf ("  --strict                treat (minor) problems as errors\n\n");
  printf ("Options for get / cmp:\n");
  printf ("  --json <file>           write JSON results into file\n");
  printf ("  --input-format <f>      raw | wav-pipe | auto (GPU build: get / cmp accept the input options of add)\n\n");
  printf ("Options for add / get / cmp:\n");
  printf ("  --key <file>            load watermarking key from file\n");
  printf ("  --strength <s>          set watermark strength              [%.6g]\n\n", Params::water_delta * 1000);
  printf ("  --input-format raw      use raw stream as input\n");
  printf ("  --output-format raw     use raw stream as output\n");
  printf ("  --format raw            use raw stream as input and output\n\n");
  printf ("The options to set the raw stream parameters (such as --raw-rate\n");
  printf ("or --raw-channels) are the reference's.\n");
}

bool is_option (const string& s) { return s.size() > 1 && s[0] == '-'; }

int
atoi_or_die (const string& s)
{
  char *e = nullptr;
  const int i = strtol (s.c_str(), &e, 0);
  if (e && e[0])
    {
      error ("audiowmark: error during string->int conversion: %s\n", s.c_str());
      exit (1);
    }
  return i;
}

float
atof_or_die (const string& s)
{
  char *e = nullptr;
  const float f = strtod (s.c_str(), &e);
  if (e && e[0])
    {
      error ("audiowmark: error during string->float conversion: %s\n", s.c_str());
      exit (1);
    }
  return f;
}

class ArgParser
{
  vector<string> m_args;
  string         m_command;
public:
  ArgParser (int argc, char **argv) : m_args (argv + 1, argv + argc) {}
  bool
  parse_cmd (const string& cmd)
  {
    if (m_args.empty() || m_args[0] != cmd)
      return false;
    m_args.erase (m_args.begin());
    m_command = cmd;
    return true;
  }
  vector<string>
  parse_multi_opt (const string& option)
  {
    vector<string> values;
    for (size_t i = 0; i < m_args.size();)
      {
        if (m_args[i] == option && i + 1 < m_args.size())
          {
            values.push_back (m_args[i + 1]);
            m_args.erase (m_args.begin() + i, m_args.begin() + i + 2);
          }
        else if (m_args[i].compare (0, option.size() + 1, option + "=") == 0)
          {
            values.push_back (m_args[i].substr (option.size() + 1));
            m_args.erase (m_args.begin() + i);
          }
        else
          i++;
      }
    return values;
  }
  bool
  parse_opt (const string& option, string& out)
  {
    const auto values = parse_multi_opt (option);
    if (values.empty())
      return false;
    out = values.back();
    return true;
  }
  bool parse_opt (const string& option, int& out) { string s; if (!parse_opt (option, s)) return false; out = atoi_or_die (s); return true; }
  bool parse_opt (const string& option, float& out) { string s; if (!parse_opt (option, s)) return false; out = atof_or_die (s); return true; }
  bool
  parse_opt (const string& option)
  {
    for (size_t i = 0; i < m_args.size(); i++)
      if (m_args[i] == option)
        {
          m_args.erase (m_args.begin() + i);
          return true;
        }
    return false;
  }
  bool
  parse_args (size_t expected, vector<string>& out)
  {
    if (m_args.size() != expected)
      return false;
    for (const auto& a : m_args)
      if (is_option (a))
        return false;
    out = m_args;
    return true;
  }
  const vector<string>& remaining_args() const { return m_args; }
  const string& command() const { return m_command; }
};

Format
parse_format (const string& str)
{
  if (str == "raw") return Format::RAW;
  if (str == "auto") return Format::AUTO;
  if (str == "rf64") return Format::RF64;
  if (str == "wav-pipe") return Format::WAV_PIPE;
  error ("audiowmark: unsupported format '%s'\n", str.c_str());
  exit (1);
}

RawFormat::Endian
parse_endian (const string& str)
{
  if (str == "little") return RawFormat::LITTLE;
  if (str == "big") return RawFormat::BIG;
  error ("audiowmark: unsupported endianness '%s'\n", str.c_str());
  exit (1);
}

void
parse_encoding (const string& str, RawFormat& fmt)
{
  if (str == "signed") fmt.encoding = Encoding::SIGNED;
  else if (str == "unsigned") fmt.encoding = Encoding::UNSIGNED;
  else if (str == "float") { fmt.encoding = Encoding::FLOAT; fmt.bit_depth = 32; }
  else if (str == "double") { fmt.encoding = Encoding::FLOAT; fmt.bit_depth = 64; }
  else
    {
      error ("audiowmark: unsupported encoding '%s'\n", str.c_str());
      exit (1);
    }
}

void
update_raw_bits (RawFormat& fmt, int bits)
{
  if (fmt.encoding == Encoding::FLOAT)
    {
      error ("audiowmark: bit depth can not be changed for float / double encoding\n");
      exit (1);
    }
  fmt.bit_depth = bits;
}

void
parse_shared_options (ArgParser& ap)
{
  int i;
  if (ap.parse_opt ("--short", i))
    {
      error ("audiowmark: --short payloads are not supported by the GPU path\n");
      exit (1);
    }
  if (ap.parse_opt ("--frames-per-bit", i) && i != Params::frames_per_bit)
    {
      error ("audiowmark: --frames-per-bit other than %d is not supported by the GPU path\n", Params::frames_per_bit);
      exit (1);
    }
  if (ap.parse_opt ("--linear"))
    Params::mix = false;
}

vector<Key>
parse_key_list (ArgParser& ap)
{
  vector<Key> key_list;
  for (const auto& f : ap.parse_multi_opt ("--key"))
    {
      Key key;
      key.load_key (f);
      key_list.push_back (key);
    }
  for (const auto& t : ap.parse_multi_opt ("--test-key"))
    {
      Key key;
      key.set_test_key (atoi_or_die (t));
      key_list.push_back (key);
    }
  if (key_list.empty())
    key_list.push_back (Key());
  return key_list;
}

Key
parse_key (ArgParser& ap)
{
  auto key_list = parse_key_list (ap);
  if (key_list.size() > 1)
    {
      error ("audiowmark %s: watermark key can at most be set once (--key / --test-key option)\n", ap.command().c_str());
      exit (1);
    }
  return key_list[0];
}

void
parse_stream_options (ArgParser& ap, bool with_output)
{
  string s;
  int i;
  if (ap.parse_opt ("--input-format", s)) Params::input_format = parse_format (s);
  if (with_output && ap.parse_opt ("--output-format", s)) Params::output_format = parse_format (s);
  if (ap.parse_opt ("--format", s))
    {
      Params::input_format = parse_format (s);
      if (with_output)
        Params::output_format = Params::input_format;
    }
  auto& rin = StreamParams::raw_input_format;
  auto& rout = StreamParams::raw_output_format;
  if (ap.parse_opt ("--raw-input-endian", s)) rin.endian = parse_endian (s);
  if (ap.parse_opt ("--raw-output-endian", s)) rout.endian = parse_endian (s);
  if (ap.parse_opt ("--raw-endian", s)) rin.endian = rout.endian = parse_endian (s);
  if (ap.parse_opt ("--raw-input-encoding", s)) parse_encoding (s, rin);
  if (ap.parse_opt ("--raw-output-encoding", s)) parse_encoding (s, rout);
  if (ap.parse_opt ("--raw-encoding", s)) { parse_encoding (s, rin); parse_encoding (s, rout); }
  if (ap.parse_opt ("--raw-input-bits", i)) update_raw_bits (rin, i);
  if (ap.parse_opt ("--raw-output-bits", i)) update_raw_bits (rout, i);
  if (ap.parse_opt ("--raw-bits", i)) { update_raw_bits (rin, i); update_raw_bits (rout, i); }
  if (ap.parse_opt ("--raw-channels", i)) rin.n_channels = rout.n_channels = i;
  if (ap.parse_opt ("--raw-rate", i)) rin.sample_rate = rout.sample_rate = i;
  if (Params::input_format == Format::RF64)
    {
      error ("audiowmark: using rf64 as input format has no effect\n");
      exit (1);
    }
}

void
parse_add_options (ArgParser& ap)
{
  float f;
  if (ap.parse_opt ("--snr")) Params::snr = true;
  parse_stream_options (ap, true);
  if (ap.parse_opt ("--test-no-limiter")) Params::test_no_limiter = true;
  if (ap.parse_opt ("--strength", f)) Params::water_delta = f / 1000;
}

void
parse_get_options (ArgParser& ap)
{
  string s;
  float f;
  int i;
  ap.parse_opt ("--test-cut", Params::test_cut);
  ap.parse_opt ("--test-truncate", Params::test_truncate);
  if (ap.parse_opt ("--hard")) Params::hard = true;
  if (ap.parse_opt ("--test-no-sync")) Params::test_no_sync = true;
  int speed_options = 0;
  if (ap.parse_opt ("--detect-speed"))
    {
      Params::detect_speed = true;
      speed_options++;
    }
  if (ap.parse_opt ("--detect-speed-patient"))
    {
      Params::detect_speed_patient = true;
      speed_options++;
    }
  if (ap.parse_opt ("--try-speed", f))
    {
      Params::try_speed = f;
      speed_options++;
    }
  if (speed_options > 1)
    {
      error ("audiowmark: can only use one option: --detect-speed or --detect-speed-patient or --try-speed\n");
      exit (1);
    }
  if (ap.parse_opt ("--test-speed", f))
    Params::test_speed = f;
  if (ap.parse_opt ("--json", s)) Params::json_output = s;
  if (ap.parse_opt ("--chunk-size", f))
    {
      if (f < 10)
        {
          error ("audiowmark: --chunk-size needs to be at least 10 minutes\n");
          exit (1);
        }
      Params::get_chunk_size = f;
    }
  if (ap.parse_opt ("--sync-threshold", f)) Params::sync_threshold2 = f;
  if (ap.parse_opt ("--n-best", i))
    {
      if (i < 0)
        {
          error ("audiowmark: --n-best should not be a negative number\n");
          exit (1);
        }
      Params::get_n_best = i;
    }
  if (ap.parse_opt ("--strength", f)) Params::water_delta = f / 1000;
  // the reference's get / cmp always open the input through libsndfile; this build has the input options of add instead
  parse_stream_options (ap, false);
}

template<class... Args> vector<string>
parse_positional (ArgParser& ap, Args... arg_names)
{
  const vector<string> names { arg_names... };
  vector<string> args;
  if (ap.parse_args (names.size(), args))
    return args;
  for (const auto& arg : ap.remaining_args())
    if (is_option (arg))
      {
        error ("audiowmark: unsupported option '%s' for command '%s' (use audiowmark -h)\n", arg.c_str(), ap.command().c_str());
        exit (1);
      }
  error ("audiowmark: error parsing arguments for command '%s' (use audiowmark -h)\n\n", ap.command().c_str());
  string msg = "usage: audiowmark " + ap.command() + " [options...]";
  for (const auto& s : names)
    msg += " <" + s + ">";
  error ("%s\n", msg.c_str());
  exit (1);
}

awm_ctx *
open_gpu()
{
  awm_ctx *ctx = nullptr;
  const char *dev = getenv ("AWM_DEVICE");
  if (awm_ctx_create (dev ? atoi (dev) : 0, &ctx) != 0)
    {
      error ("audiowmark: %s\n", awm_last_error());
      exit (1);
    }
  return ctx;
}

int
gen_key (const string& outfile, const string& key_name)
{
  FILE *f = fopen (outfile.c_str(), "w");
  if (!f)
    {
      error ("audiowmark: error writing to file %s\n", outfile.c_str());
      return 1;
    }
  fprintf (f, "# watermarking key for audiowmark\n\nkey %s\n", Random::gen_key().c_str());
  if (!key_name.empty())
    fprintf (f, "name %s\n", key_name.c_str());
  fclose (f);
  return 0;
}

int
test_gen_noise (const Key& key, const string& out_file, double seconds, int rate, int bits)
{
  // reference audiowmark.cc:399-417
  const int channels = 2;
  vector<float> noise;
  Random rng (key, 0, Random::Stream::data_up_down);
  for (size_t i = 0; i < size_t (rate * seconds) * channels; i++)
    noise.push_back (rng.random_double() * 2 - 1);
  WavData out (noise, channels, rate, bits);
  Error err = out.save (out_file);
  if (err)
    {
      error ("audiowmark: error saving %s: %s\n", out_file.c_str(), err.message());
      return 1;
    }
  return 0;
}

int
test_snr (const string& orig_file, const string& wm_file)
{
  WavData orig, wm;
  Error err = orig.load (orig_file);
  if (!err)
    err = wm.load (wm_file);
  if (err)
    {
      error ("audiowmark: error loading: %s\n", err.message());
      return 1;
    }
  if (orig.n_values() != wm.n_values())
    {
      error ("audiowmark: files have different length\n");
      return 1;
    }
  double delta_power = 0, signal_power = 0;
  for (size_t i = 0; i < orig.n_values(); i++)
    {
      const double o = orig.samples()[i], d = o - wm.samples()[i];
      delta_power += d * d;
      signal_power += o * o;
    }
  printf ("snr_db %f\n", 10 * log10 (signal_power / delta_power));
  return 0;
}

} // namespace

int
main (int argc, char **argv)
{
  ArgParser ap (argc, argv);
  vector<string> args;
  if (ap.parse_opt ("--help") || ap.parse_opt ("-h"))
    {
      print_usage();
      return 0;
    }
  if (ap.parse_opt ("--version") || ap.parse_opt ("-v"))
    {
      printf ("audiowmark 0.6.5 (%s)\n", awm_version());
      return 0;
    }
  if (ap.parse_opt ("--quiet") || ap.parse_opt ("-q"))
    set_log_level (Log::WARNING);
  if (ap.parse_opt ("--strict"))
    Params::strict = true;

  if (ap.parse_cmd ("add"))
    {
      parse_shared_options (ap);
      parse_add_options (ap);
      Key key = parse_key (ap);
      args = parse_positional (ap, "input_wav", "watermarked_wav", "message_hex");
      awm_ctx *ctx = open_gpu();
      const int rc = add_watermark (ctx, key, args[0], args[1], args[2]);
      awm_ctx_destroy (ctx);
      return rc;
    }
  else if (ap.parse_cmd ("get") || ap.parse_cmd ("cmp"))
    {
      const bool cmp = ap.command() == "cmp";
      parse_shared_options (ap);
      parse_get_options (ap);
      if (cmp)
        ap.parse_opt ("--expect-matches", Params::expect_matches);
      vector<Key> key_list = parse_key_list (ap);
      if (cmp)
        args = parse_positional (ap, "watermarked_wav", "message_hex");
      else
        args = parse_positional (ap, "watermarked_wav");
      awm_ctx *ctx = open_gpu();
      const int rc = get_watermark (ctx, key_list, args[0], cmp ? args[1] : "");
      awm_ctx_destroy (ctx);
      return rc;
    }
  else if (ap.parse_cmd ("gen-key"))
    {
      string key_name;
      ap.parse_opt ("--name", key_name);
      args = parse_positional (ap, "key_file");
      return gen_key (args[0], key_name);
    }
  else if (ap.parse_cmd ("test-change-speed"))
    {
      parse_shared_options (ap);
      parse_stream_options (ap, true);
      args = parse_positional (ap, "input_wav", "output_wav", "speed");
      awm_ctx *ctx = open_gpu();
      const int rc = test_change_speed (ctx, args[0], args[1], atof_or_die (args[2]));
      awm_ctx_destroy (ctx);
      return rc;
    }
  else if (ap.parse_cmd ("test-gen-noise"))
    {
      parse_shared_options (ap);
      int bits = 16;
      ap.parse_opt ("--bits", bits);
      Key key = parse_key (ap);
      args = parse_positional (ap, "output_wav", "seconds", "sample_rate");
      return test_gen_noise (key, args[0], atof_or_die (args[1]), atoi_or_die (args[2]), bits);
    }
  else if (ap.parse_cmd ("test-snr"))
    {
      args = parse_positional (ap, "orig_wav", "watermarked_wav");
      return test_snr (args[0], args[1]);
    }
  else if (!ap.remaining_args().empty())
    {
      const string s = ap.remaining_args().front();
      if (is_option (s))
        error ("audiowmark: unsupported global option '%s' (use audiowmark -h)\n", s.c_str());
      else
        error ("audiowmark: unsupported command '%s' (use audiowmark -h)\n", s.c_str());
      return 1;
    }
  error ("audiowmark: error parsing commandline args (use audiowmark -h)\n");
  return 1;
}
