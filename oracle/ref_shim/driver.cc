/* oracle/ref_shim/driver.cc -- main() of oracle/_ref/audiowmark_ref.
 * The reference's `get`/`cmp` have no --input-format option (audiowmark.cc:812-881) and
 * always go through libsndfile, which the oracle build does not have.  This driver
 * strips two private flags and presets Params::input_format before handing over to the
 * reference's unmodified main (compiled with -Dmain=ref_main):
 *   --x-in-wav-pipe          read input with the reference's WavPipeInputStream
 *   --x-in-raw               read input with RawInputStream (use with --raw-* options of add)
 * TEST INFRASTRUCTURE ONLY. */
#include <string.h>
#include <vector>
#include "wmcommon.hh"

int ref_main (int argc, char **argv);

int
main (int argc, char **argv)
{
  std::vector<char *> args;
  for (int i = 0; i < argc; i++)
    {
      if (!strcmp (argv[i], "--x-in-wav-pipe"))
        Params::input_format = Format::WAV_PIPE;
      else if (!strcmp (argv[i], "--x-in-raw"))
        Params::input_format = Format::RAW;
      else
        args.push_back (argv[i]);
    }
  args.push_back (nullptr);
  return ref_main ((int) args.size() - 1, args.data());
}
