/* oracle/sha1.h -- FIPS 180-4 SHA-1 (the reference hashes a sample subset with libgcrypt's GCRY_MD_SHA1 to seed the
 * speed clip selection, random.cc:184-190).  TEST INFRASTRUCTURE ONLY. */
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static inline void
awm_sha1 (const void *data, size_t len, uint8_t digest[20])
{
  uint32_t h[5] = { 0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u, 0xc3d2e1f0u };
  auto rol = [] (uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
  auto block = [&] (const uint8_t *p) {
    uint32_t w[80];
    for (int i = 0; i < 16; i++)
      w[i] = uint32_t (p[4 * i]) << 24 | uint32_t (p[4 * i + 1]) << 16 | uint32_t (p[4 * i + 2]) << 8 | p[4 * i + 3];
    for (int i = 16; i < 80; i++)
      w[i] = rol (w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
    for (int i = 0; i < 80; i++)
      {
        uint32_t f, k;
        if (i < 20)      { f = (b & c) | (~b & d);          k = 0x5a827999u; }
        else if (i < 40) { f = b ^ c ^ d;                   k = 0x6ed9eba1u; }
        else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8f1bbcdcu; }
        else             { f = b ^ c ^ d;                   k = 0xca62c1d6u; }
        const uint32_t t = rol (a, 5) + f + e + k + w[i];
        e = d; d = c; c = rol (b, 30); b = a; a = t;
      }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
  };
  const uint8_t *p = static_cast<const uint8_t *> (data);
  size_t left = len;
  for (; left >= 64; left -= 64, p += 64)
    block (p);
  uint8_t tail[128] = { 0 };
  memcpy (tail, p, left);
  tail[left] = 0x80;
  const size_t total = left + 9 <= 64 ? 64 : 128;
  const uint64_t bits = uint64_t (len) * 8;
  for (int i = 0; i < 8; i++)
    tail[total - 1 - i] = uint8_t (bits >> (8 * i));
  block (tail);
  if (total == 128)
    block (tail + 64);
  for (int i = 0; i < 5; i++)
    for (int k = 0; k < 4; k++)
      digest[4 * i + k] = uint8_t (h[i] >> (24 - 8 * k));
}
