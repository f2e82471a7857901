/* oracle/ref_shim: minimal libgcrypt API surface used by the reference
 * (random.cc:38-48,101-111,130-135,151,180,188 and testrawconverter.cc:71).
 * Implemented in ../gcrypt_shim.cc with a from-scratch FIPS-197 AES-128 and
 * FIPS-180 SHA-1, so the oracle does not depend on any crypto library.
 * TEST INFRASTRUCTURE ONLY. */
#pragma once
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef unsigned int gcry_error_t;
typedef struct awm_shim_cipher *gcry_cipher_hd_t;
#define GCRYPT_VERSION "1.9.4"
enum { GCRY_CIPHER_AES128 = 7 };
enum { GCRY_CIPHER_MODE_ECB = 1, GCRY_CIPHER_MODE_CTR = 6 };
enum { GCRY_MD_SHA1 = 2 };
enum { GCRYCTL_DISABLE_SECMEM = 37, GCRYCTL_INITIALIZATION_FINISHED = 38 };
enum { GCRY_VERY_STRONG_RANDOM = 2 };
const char *gcry_check_version (const char *req);
gcry_error_t gcry_control (int cmd, ...);
gcry_error_t gcry_cipher_open (gcry_cipher_hd_t *h, int algo, int mode, unsigned flags);
void gcry_cipher_close (gcry_cipher_hd_t h);
gcry_error_t gcry_cipher_setkey (gcry_cipher_hd_t h, const void *key, size_t len);
gcry_error_t gcry_cipher_setctr (gcry_cipher_hd_t h, const void *ctr, size_t len);
gcry_error_t gcry_cipher_encrypt (gcry_cipher_hd_t h, void *out, size_t outsize, const void *in, size_t inlen);
const char *gcry_strsource (gcry_error_t e);
const char *gcry_strerror (gcry_error_t e);
void gcry_randomize (void *buf, size_t len, int level);
void gcry_md_hash_buffer (int algo, void *digest, const void *buf, size_t len);
#ifdef __cplusplus
}
#endif
