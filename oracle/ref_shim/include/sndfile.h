/* oracle/ref_shim: type/constant surface of libsndfile used by sfinputstream.cc /
 * sfoutputstream.cc.  All functions are stubs that fail (../stubs.cc): the oracle
 * only uses the reference's raw / wav-pipe / in-memory streams.  TEST INFRASTRUCTURE ONLY. */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct SNDFILE_tag SNDFILE;
typedef int64_t sf_count_t;
#define SF_COUNT_MAX 0x7FFFFFFFFFFFFFFFLL
typedef struct { sf_count_t frames; int samplerate; int channels; int format; int sections; int seekable; } SF_INFO;
typedef sf_count_t (*sf_vio_get_filelen) (void *);
typedef sf_count_t (*sf_vio_seek) (sf_count_t, int, void *);
typedef sf_count_t (*sf_vio_read) (void *, sf_count_t, void *);
typedef sf_count_t (*sf_vio_write) (const void *, sf_count_t, void *);
typedef sf_count_t (*sf_vio_tell) (void *);
typedef struct { sf_vio_get_filelen get_filelen; sf_vio_seek seek; sf_vio_read read; sf_vio_write write; sf_vio_tell tell; } SF_VIRTUAL_IO;
enum { SF_FALSE = 0, SF_TRUE = 1, SFM_READ = 0x10, SFM_WRITE = 0x20, SFM_RDWR = 0x30 };
enum {
  SF_FORMAT_WAV = 0x010000, SF_FORMAT_W64 = 0x0B0000, SF_FORMAT_FLAC = 0x170000, SF_FORMAT_RF64 = 0x220000,
  SF_FORMAT_PCM_S8 = 0x0001, SF_FORMAT_PCM_16 = 0x0002, SF_FORMAT_PCM_24 = 0x0003, SF_FORMAT_PCM_32 = 0x0004,
  SF_FORMAT_PCM_U8 = 0x0005, SF_FORMAT_FLOAT = 0x0006, SF_FORMAT_DOUBLE = 0x0007,
  SF_FORMAT_VORBIS = 0x0060, SF_FORMAT_OPUS = 0x0064,
  SF_FORMAT_ALAC_16 = 0x0070, SF_FORMAT_ALAC_20 = 0x0071, SF_FORMAT_ALAC_24 = 0x0072, SF_FORMAT_ALAC_32 = 0x0073,
  SF_FORMAT_MPEG_LAYER_I = 0x0080, SF_FORMAT_MPEG_LAYER_II = 0x0081, SF_FORMAT_MPEG_LAYER_III = 0x0082,
  SF_FORMAT_SUBMASK = 0x0000FFFF, SF_FORMAT_TYPEMASK = 0x0FFF0000
};
SNDFILE *sf_open (const char *path, int mode, SF_INFO *info);
SNDFILE *sf_open_fd (int fd, int mode, SF_INFO *info, int close_desc);
SNDFILE *sf_open_virtual (SF_VIRTUAL_IO *io, int mode, SF_INFO *info, void *user);
int sf_error (SNDFILE *);
const char *sf_strerror (SNDFILE *);
int sf_close (SNDFILE *);
sf_count_t sf_readf_float (SNDFILE *, float *, sf_count_t);
sf_count_t sf_readf_int (SNDFILE *, int *, sf_count_t);
sf_count_t sf_writef_float (SNDFILE *, const float *, sf_count_t);
sf_count_t sf_writef_int (SNDFILE *, const int *, sf_count_t);
int sf_command (SNDFILE *, int, void *, int);
#ifdef __cplusplus
}
#endif
