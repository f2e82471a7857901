#include "aes128.hh"
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define AWM_X86_CRYPTO 1
#endif

namespace awm {

const uint8_t *
Aes128::sbox()
{
  // S-box built from the multiplicative inverse in GF(2^8) followed by the affine map.
  static uint8_t table[256];
  static const bool ready = [] {
    uint8_t p = 1, q = 1;
    do
      {
        p = uint8_t (p ^ (p << 1) ^ ((p & 0x80) ? 0x1B : 0));      // p *= 3
        q ^= uint8_t (q << 1); q ^= uint8_t (q << 2); q ^= uint8_t (q << 4);
        if (q & 0x80) q ^= 0x09;                                    // q /= 3
        auto rotl = [] (uint8_t v, int n) { return uint8_t ((v << n) | (v >> (8 - n))); };
        table[p] = uint8_t (q ^ rotl (q, 1) ^ rotl (q, 2) ^ rotl (q, 3) ^ rotl (q, 4) ^ 0x63);
      }
    while (p != 1);
    table[0] = 0x63;
    return true;
  }();
  (void) ready;
  return table;
}

void
Aes128::set_key (const uint8_t key[16])
{
  const uint8_t *S = sbox();
  std::memcpy (m_rk, key, 16);
  uint8_t rcon = 1;
  for (int i = 16; i < 176; i += 4)
    {
      uint8_t t[4] = { m_rk[i - 4], m_rk[i - 3], m_rk[i - 2], m_rk[i - 1] };
      if (i % 16 == 0)
        {
          const uint8_t t0 = t[0];
          t[0] = S[t[1]] ^ rcon; t[1] = S[t[2]]; t[2] = S[t[3]]; t[3] = S[t0];
          rcon = xtime (rcon);
        }
      for (int j = 0; j < 4; j++)
        m_rk[i + j] = m_rk[i - 16 + j] ^ t[j];
    }
}

#ifdef AWM_X86_CRYPTO
// the same cipher on the CPU's AES unit (the round keys are FIPS-197's byte string, which is what AESENC consumes);
// the speed search draws ~300 000 positions per 30 minute chunk from the key stream (reference wmspeed.cc:533-553)
__attribute__ ((target ("aes,sse2"))) static void
encrypt_block_aesni (const uint8_t *rk, const uint8_t in[16], uint8_t out[16])
{
  __m128i b = _mm_xor_si128 (_mm_loadu_si128 (reinterpret_cast<const __m128i *> (in)), _mm_loadu_si128 (reinterpret_cast<const __m128i *> (rk)));
  for (int round = 1; round < 10; round++)
    b = _mm_aesenc_si128 (b, _mm_loadu_si128 (reinterpret_cast<const __m128i *> (rk + 16 * round)));
  b = _mm_aesenclast_si128 (b, _mm_loadu_si128 (reinterpret_cast<const __m128i *> (rk + 160)));
  _mm_storeu_si128 (reinterpret_cast<__m128i *> (out), b);
}
// eight independent blocks at a time: AESENC has a latency of several cycles but issues every cycle
__attribute__ ((target ("aes,sse2"))) static void
encrypt_blocks_aesni (const uint8_t *rk, const uint8_t *in, uint8_t *out, size_t n_blocks)
{
  __m128i k[11];
  for (int r = 0; r < 11; r++)
    k[r] = _mm_loadu_si128 (reinterpret_cast<const __m128i *> (rk + 16 * r));
  size_t i = 0;
  for (; i + 8 <= n_blocks; i += 8)
    {
      __m128i b[8];
      for (int j = 0; j < 8; j++)
        b[j] = _mm_xor_si128 (_mm_loadu_si128 (reinterpret_cast<const __m128i *> (in + 16 * (i + j))), k[0]);
      for (int r = 1; r < 10; r++)
        for (int j = 0; j < 8; j++)
          b[j] = _mm_aesenc_si128 (b[j], k[r]);
      for (int j = 0; j < 8; j++)
        _mm_storeu_si128 (reinterpret_cast<__m128i *> (out + 16 * (i + j)), _mm_aesenclast_si128 (b[j], k[10]));
    }
  for (; i < n_blocks; i++)
    encrypt_block_aesni (rk, in + 16 * i, out + 16 * i);
}
static const bool have_aesni = [] { __builtin_cpu_init(); return bool (__builtin_cpu_supports ("aes")); }();
#endif

void
Aes128::encrypt_blocks (const uint8_t *in, uint8_t *out, size_t n_blocks) const
{
#ifdef AWM_X86_CRYPTO
  if (have_aesni)
    {
      encrypt_blocks_aesni (m_rk, in, out, n_blocks);
      return;
    }
#endif
  for (size_t i = 0; i < n_blocks; i++)
    encrypt_block (in + 16 * i, out + 16 * i);
}

void
Aes128::encrypt_block (const uint8_t in[16], uint8_t out[16]) const
{
#ifdef AWM_X86_CRYPTO
  if (have_aesni)
    {
      encrypt_block_aesni (m_rk, in, out);
      return;
    }
#endif
  const uint8_t *S = sbox();
  uint8_t st[16], t[16];
  for (int i = 0; i < 16; i++)
    st[i] = in[i] ^ m_rk[i];
  for (int round = 1; round <= 10; round++)
    {
      for (int c = 0; c < 4; c++)            // SubBytes + ShiftRows, column-major state
        for (int r = 0; r < 4; r++)
          t[4 * c + r] = S[st[4 * ((c + r) & 3) + r]];
      if (round < 10)
        for (int c = 0; c < 4; c++)          // MixColumns
          {
            const uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
            const uint8_t all = a0 ^ a1 ^ a2 ^ a3;
            st[4 * c + 0] = a0 ^ all ^ xtime (a0 ^ a1);
            st[4 * c + 1] = a1 ^ all ^ xtime (a1 ^ a2);
            st[4 * c + 2] = a2 ^ all ^ xtime (a2 ^ a3);
            st[4 * c + 3] = a3 ^ all ^ xtime (a3 ^ a0);
          }
      else
        std::memcpy (st, t, 16);
      for (int i = 0; i < 16; i++)
        st[i] ^= m_rk[16 * round + i];
    }
  std::memcpy (out, st, 16);
}

} // namespace awm
