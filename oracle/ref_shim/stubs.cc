/* oracle/ref_shim/stubs.cc -- failing stand-ins for libraries that are out of scope for
 * the hot path (libsndfile, libmpg123; SURVEY.md section 2 rows 16,17).  TEST INFRASTRUCTURE ONLY. */
#include "sndfile.h"
#include "mp3inputstream.hh"

extern "C" {
SNDFILE *sf_open (const char *, int, SF_INFO *) { return nullptr; }
SNDFILE *sf_open_fd (int, int, SF_INFO *, int) { return nullptr; }
SNDFILE *sf_open_virtual (SF_VIRTUAL_IO *, int, SF_INFO *, void *) { return nullptr; }
int sf_error (SNDFILE *) { return 1; }
const char *sf_strerror (SNDFILE *) { return "libsndfile is not available in the oracle build (use --format wav-pipe / raw)"; }
int sf_close (SNDFILE *) { return 0; }
sf_count_t sf_readf_float (SNDFILE *, float *, sf_count_t) { return 0; }
sf_count_t sf_readf_int (SNDFILE *, int *, sf_count_t) { return 0; }
sf_count_t sf_writef_float (SNDFILE *, const float *, sf_count_t) { return 0; }
sf_count_t sf_writef_int (SNDFILE *, const int *, sf_count_t) { return 0; }
int sf_command (SNDFILE *, int, void *, int) { return 0; }
}

MP3InputStream::~MP3InputStream() {}
Error MP3InputStream::open (const std::string&) { return Error ("mp3 is not available in the oracle build"); }
Error MP3InputStream::read_frames (std::vector<float>&, size_t) { return Error ("mp3 is not available in the oracle build"); }
void MP3InputStream::close() {}
int MP3InputStream::bit_depth() const { return 0; }
int MP3InputStream::sample_rate() const { return 0; }
int MP3InputStream::n_channels() const { return 0; }
size_t MP3InputStream::n_frames() const { return 0; }
Encoding MP3InputStream::encoding() const { return Encoding::SIGNED; }
bool MP3InputStream::detect (const std::string&) { return false; }
