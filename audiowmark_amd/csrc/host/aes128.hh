// AES-128 block encryption (FIPS-197), encrypt direction only.
//
// The reference draws every key-dependent table from an AES-128-CTR keystream
// (reference src/random.cc:97-161, via libgcrypt).  AES is a public standard, so a
// conforming implementation reproduces that stream bit for bit.
#pragma once
#include <cstdint>
#include <cstring>

namespace awm {

class Aes128
{
  uint8_t m_rk[176];
  static uint8_t xtime (uint8_t x) { return uint8_t ((x << 1) ^ ((x & 0x80) ? 0x1B : 0)); }
public:
  static const uint8_t *sbox();                                   // the 256 byte S-box
  const uint8_t *round_keys() const { return m_rk; }              // the key schedule: 11 x 16 bytes, FIPS-197 byte order
  void set_key (const uint8_t key[16]);
  void encrypt_block (const uint8_t in[16], uint8_t out[16]) const;
  void encrypt_blocks (const uint8_t *in, uint8_t *out, size_t n_blocks) const;   // independent blocks (CTR keystream)
};

} // namespace awm
