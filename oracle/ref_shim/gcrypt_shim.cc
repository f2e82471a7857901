/* oracle/ref_shim/gcrypt_shim.cc -- stand-in for the libgcrypt calls of the reference
 * (random.cc).  AES-128 ECB / CTR (big-endian 128-bit counter increment, as libgcrypt
 * and NIST SP 800-38A do), SHA-1, and a /dev/urandom randomize.  TEST INFRASTRUCTURE ONLY. */
#include "gcrypt.h"
#include "../aes128.h"
#include <stdio.h>
#include <stdlib.h>

struct awm_shim_cipher
{
  AwmAes128 aes;
  int       mode;
  uint8_t   ctr[16];
  uint8_t   ks[16];
  int       ks_used; /* bytes of ks consumed; 16 = none available */
};

extern "C" {

const char *gcry_check_version (const char *) { return GCRYPT_VERSION; }
gcry_error_t gcry_control (int, ...) { return 0; }

gcry_error_t
gcry_cipher_open (gcry_cipher_hd_t *h, int algo, int mode, unsigned)
{
  if (algo != GCRY_CIPHER_AES128 || (mode != GCRY_CIPHER_MODE_ECB && mode != GCRY_CIPHER_MODE_CTR))
    return 1;
  awm_shim_cipher *c = new awm_shim_cipher();
  c->mode = mode;
  memset (c->ctr, 0, 16);
  c->ks_used = 16;
  *h = c;
  return 0;
}
void gcry_cipher_close (gcry_cipher_hd_t h) { delete h; }
gcry_error_t
gcry_cipher_setkey (gcry_cipher_hd_t h, const void *key, size_t len)
{
  if (len != 16) return 1;
  h->aes.set_key ((const uint8_t *) key);
  return 0;
}
gcry_error_t
gcry_cipher_setctr (gcry_cipher_hd_t h, const void *ctr, size_t len)
{
  if (len != 16) return 1;
  memcpy (h->ctr, ctr, 16);
  h->ks_used = 16;
  return 0;
}
gcry_error_t
gcry_cipher_encrypt (gcry_cipher_hd_t h, void *out, size_t outsize, const void *in, size_t inlen)
{
  if (outsize < inlen) return 1;
  const uint8_t *ip = (const uint8_t *) in;
  uint8_t *op = (uint8_t *) out;
  if (h->mode == GCRY_CIPHER_MODE_ECB)
    {
      if (inlen % 16) return 1;
      for (size_t i = 0; i < inlen; i += 16)
        h->aes.encrypt_block (ip + i, op + i);
      return 0;
    }
  for (size_t i = 0; i < inlen; i++)
    {
      if (h->ks_used == 16)
        {
          h->aes.encrypt_block (h->ctr, h->ks);
          for (int k = 15; k >= 0; k--)   /* big-endian increment */
            if (++h->ctr[k]) break;
          h->ks_used = 0;
        }
      op[i] = ip[i] ^ h->ks[h->ks_used++];
    }
  return 0;
}
const char *gcry_strsource (gcry_error_t) { return "awm-shim"; }
const char *gcry_strerror (gcry_error_t e) { return e ? "error" : "ok"; }
void
gcry_randomize (void *buf, size_t len, int)
{
  FILE *f = fopen ("/dev/urandom", "rb");
  if (!f || fread (buf, 1, len, f) != len) { fprintf (stderr, "awm-shim: gcry_randomize failed\n"); exit (1); }
  fclose (f);
}

static inline uint32_t rol32 (uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
void
gcry_md_hash_buffer (int algo, void *digest, const void *buf, size_t len)
{
  if (algo != GCRY_MD_SHA1) { fprintf (stderr, "awm-shim: unsupported digest\n"); exit (1); }
  uint32_t h0 = 0x67452301, h1 = 0xEFCDAB89, h2 = 0x98BADCFE, h3 = 0x10325476, h4 = 0xC3D2E1F0;
  const uint8_t *p = (const uint8_t *) buf;
  size_t total = len + 1 + 8;
  size_t padded = (total + 63) / 64 * 64;
  for (size_t off = 0; off < padded; off += 64)
    {
      uint8_t blk[64];
      for (size_t i = 0; i < 64; i++)
        {
          size_t pos = off + i;
          if (pos < len) blk[i] = p[pos];
          else if (pos == len) blk[i] = 0x80;
          else if (pos >= padded - 8) blk[i] = (uint8_t) (((uint64_t) len * 8) >> (8 * (padded - 1 - pos)));
          else blk[i] = 0;
        }
      uint32_t w[80];
      for (int i = 0; i < 16; i++)
        w[i] = (uint32_t (blk[4 * i]) << 24) | (uint32_t (blk[4 * i + 1]) << 16) | (uint32_t (blk[4 * i + 2]) << 8) | blk[4 * i + 3];
      for (int i = 16; i < 80; i++)
        w[i] = rol32 (w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
      uint32_t a = h0, b = h1, c = h2, d = h3, e = h4;
      for (int i = 0; i < 80; i++)
        {
          uint32_t f, k;
          if (i < 20)      { f = (b & c) | (~b & d);          k = 0x5A827999; }
          else if (i < 40) { f = b ^ c ^ d;                   k = 0x6ED9EBA1; }
          else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDC; }
          else             { f = b ^ c ^ d;                   k = 0xCA62C1D6; }
          uint32_t t = rol32 (a, 5) + f + e + k + w[i];
          e = d; d = c; c = rol32 (b, 30); b = a; a = t;
        }
      h0 += a; h1 += b; h2 += c; h3 += d; h4 += e;
    }
  uint32_t hs[5] = { h0, h1, h2, h3, h4 };
  uint8_t *o = (uint8_t *) digest;
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 4; j++)
      o[4 * i + j] = (uint8_t) (hs[i] >> (24 - 8 * j));
}

} /* extern "C" */
