/* oracle/aes128.h -- from-scratch AES-128 (FIPS-197) block encryption, header-only.
 *
 * TEST INFRASTRUCTURE ONLY (oracle): used by the libgcrypt stand-in of oracle/_ref and
 * by the CPU restatement (awm_oracle.cc).  The reference obtains AES from libgcrypt
 * (random.cc:97-161); AES is a public standard, so any conforming implementation
 * reproduces the reference's PRNG stream bit for bit (KAT: SURVEY.md Appendix A,
 * checked in tests/test_oracle_kat.py).
 */
#pragma once
#include <stdint.h>
#include <string.h>

struct AwmAes128
{
  uint8_t rk[176];

  static const uint8_t *
  sbox()
  {
    static uint8_t s[256];
    static bool init = false;
    if (!init)
      {
        /* generate the S-box from the GF(2^8) inverse + affine map */
        uint8_t p = 1, q = 1;
        do
          {
            p = p ^ (uint8_t) (p << 1) ^ ((p & 0x80) ? 0x1B : 0);  /* p *= 3 */
            q ^= q << 1; q ^= q << 2; q ^= q << 4;                 /* q /= 3 */
            if (q & 0x80) q ^= 0x09;
            uint8_t x = q ^ (uint8_t) ((q << 1) | (q >> 7)) ^ (uint8_t) ((q << 2) | (q >> 6))
                          ^ (uint8_t) ((q << 3) | (q >> 5)) ^ (uint8_t) ((q << 4) | (q >> 4));
            s[p] = x ^ 0x63;
          }
        while (p != 1);
        s[0] = 0x63;
        init = true;
      }
    return s;
  }
  static uint8_t xtime (uint8_t x) { return (uint8_t) ((x << 1) ^ ((x & 0x80) ? 0x1B : 0)); }

  void
  set_key (const uint8_t key[16])
  {
    const uint8_t *S = sbox();
    memcpy (rk, key, 16);
    uint8_t rcon = 1;
    for (int i = 16; i < 176; i += 4)
      {
        uint8_t t[4] = { rk[i - 4], rk[i - 3], rk[i - 2], rk[i - 1] };
        if (i % 16 == 0)
          {
            uint8_t t0 = t[0];
            t[0] = S[t[1]] ^ rcon; t[1] = S[t[2]]; t[2] = S[t[3]]; t[3] = S[t0];
            rcon = xtime (rcon);
          }
        for (int j = 0; j < 4; j++)
          rk[i + j] = rk[i - 16 + j] ^ t[j];
      }
  }
  void
  encrypt_block (const uint8_t in[16], uint8_t out[16]) const
  {
    const uint8_t *S = sbox();
    uint8_t st[16];
    for (int i = 0; i < 16; i++)
      st[i] = in[i] ^ rk[i];
    for (int round = 1; round <= 10; round++)
      {
        uint8_t t[16];
        /* SubBytes + ShiftRows (state is column-major: st[4*c + r]) */
        for (int c = 0; c < 4; c++)
          for (int r = 0; r < 4; r++)
            t[4 * c + r] = S[st[4 * ((c + r) & 3) + r]];
        if (round < 10)
          {
            for (int c = 0; c < 4; c++)
              {
                uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                uint8_t all = a0 ^ a1 ^ a2 ^ a3;
                st[4 * c + 0] = a0 ^ all ^ xtime (a0 ^ a1);
                st[4 * c + 1] = a1 ^ all ^ xtime (a1 ^ a2);
                st[4 * c + 2] = a2 ^ all ^ xtime (a2 ^ a3);
                st[4 * c + 3] = a3 ^ all ^ xtime (a3 ^ a0);
              }
          }
        else
          memcpy (st, t, 16);
        for (int i = 0; i < 16; i++)
          st[i] ^= rk[16 * round + i];
      }
    memcpy (out, st, 16);
  }
};
