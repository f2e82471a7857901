#include "utils.hh"
#include <cstdio>
#include <cstdlib>
#include <sys/time.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define AWM_X86_CRYPTO 1
#endif

namespace awm {

static Log g_log_level = Log::INFO;

void set_log_level (Log level) { g_log_level = level; }

static void
vlog (Log level, const char *format, va_list ap)
{
  if (int (level) >= int (g_log_level))
    {
      vfprintf (stderr, format, ap);
      fflush (stderr);
    }
}
#define AWM_LOG_FN(name, level) \
  void name (const char *format, ...) { va_list ap; va_start (ap, format); vlog (level, format, ap); va_end (ap); }
AWM_LOG_FN (error, Log::ERROR)
AWM_LOG_FN (warning, Log::WARNING)
AWM_LOG_FN (info, Log::INFO)
AWM_LOG_FN (debug, Log::DEBUG)

std::string
string_printf (const char *format, ...)
{
  va_list ap;
  va_start (ap, format);
  char *str = nullptr;
  std::string s;
  if (vasprintf (&str, format, ap) >= 0 && str)
    {
      s = str;
      free (str);
    }
  else
    s = format;
  va_end (ap);
  return s;
}

static int
hex_value (char c)
{
  if (c >= '0' && c <= '9') return c - '0';
  if (c >= 'a' && c <= 'f') return c - 'a' + 10;
  if (c >= 'A' && c <= 'F') return c - 'A' + 10;
  return -1;
}

std::vector<int>
bit_str_to_vec (const std::string& bits)
{
  std::vector<int> v;
  v.reserve (bits.size() * 4);
  for (char c : bits)
    {
      const int h = hex_value (c);
      if (h < 0)
        return {};
      for (int shift = 3; shift >= 0; shift--)
        v.push_back ((h >> shift) & 1);
    }
  return v;
}

std::string
bit_vec_to_str (const std::vector<int>& bits)
{
  static const char digits[] = "0123456789abcdef";
  std::string s;
  for (size_t pos = 0; pos + 3 < bits.size(); pos += 4)
    {
      const int nibble = (bits[pos] ? 8 : 0) | (bits[pos + 1] ? 4 : 0) | (bits[pos + 2] ? 2 : 0) | (bits[pos + 3] ? 1 : 0);
      s += digits[nibble];
    }
  return s;
}

std::vector<unsigned char>
hex_str_to_vec (const std::string& str)
{
  if (str.size() % 2)
    return {};
  std::vector<unsigned char> v;
  for (size_t i = 0; i < str.size(); i += 2)
    {
      const int h = hex_value (str[i]), l = hex_value (str[i + 1]);
      if (h < 0 || l < 0)
        return {};
      v.push_back ((unsigned char) (h * 16 + l));
    }
  return v;
}

std::string
vec_to_hex_str (const std::vector<unsigned char>& vec)
{
  std::string s;
  for (auto b : vec)
    s += string_printf ("%02x", b);
  return s;
}

double
get_time()
{
  timeval tv;
  gettimeofday (&tv, nullptr);
  return tv.tv_sec + tv.tv_usec / 1e6;
}

#ifdef AWM_X86_CRYPTO
// SHA-1 compression of `blocks` 64 byte blocks on the CPU's SHA unit (four rounds per SHA1RNDS4; the message schedule runs
// through SHA1MSG1 / SHA1MSG2, the round constant group is the instruction's immediate)
__attribute__ ((target ("sha,sse4.1,ssse3"))) static void
sha1_blocks_shani (uint32_t state[5], const unsigned char *data, size_t blocks)
{
  const __m128i be = _mm_set_epi64x (0x0001020304050607ll, 0x08090a0b0c0d0e0fll);   // big endian words, word 0 in the top lane
  __m128i abcd = _mm_shuffle_epi32 (_mm_loadu_si128 (reinterpret_cast<const __m128i *> (state)), 0x1b);
  __m128i e0 = _mm_set_epi32 (int (state[4]), 0, 0, 0);
  for (; blocks; blocks--, data += 64)
    {
      const __m128i abcd_in = abcd, e_in = e0;
      __m128i m[4], e1;
      for (int i = 0; i < 4; i++)
        m[i] = _mm_shuffle_epi8 (_mm_loadu_si128 (reinterpret_cast<const __m128i *> (data + 16 * i)), be);
      // 20 groups of four rounds; group g uses W[4g .. 4g + 3] = m[g & 3] (extended in place from group 4 on)
#define AWM_SHA1_GROUP(g, fn)                                                                   \
      {                                                                                         \
        if (g >= 4)                                                                             \
          {                                                                                     \
            __m128i w = _mm_sha1msg1_epu32 (m[g & 3], m[(g + 1) & 3]);                           \
            w = _mm_xor_si128 (w, m[(g + 2) & 3]);                                              \
            m[g & 3] = _mm_sha1msg2_epu32 (w, m[(g + 3) & 3]);                                   \
          }                                                                                     \
        if (g == 0)                                                                             \
          e1 = _mm_add_epi32 (e0, m[0]);                                                        \
        else                                                                                    \
          e1 = _mm_sha1nexte_epu32 (e0, m[g & 3]);                                              \
        e0 = abcd;                                                                              \
        abcd = _mm_sha1rnds4_epu32 (abcd, e1, fn);                                              \
      }
      AWM_SHA1_GROUP (0, 0) AWM_SHA1_GROUP (1, 0) AWM_SHA1_GROUP (2, 0) AWM_SHA1_GROUP (3, 0) AWM_SHA1_GROUP (4, 0)
      AWM_SHA1_GROUP (5, 1) AWM_SHA1_GROUP (6, 1) AWM_SHA1_GROUP (7, 1) AWM_SHA1_GROUP (8, 1) AWM_SHA1_GROUP (9, 1)
      AWM_SHA1_GROUP (10, 2) AWM_SHA1_GROUP (11, 2) AWM_SHA1_GROUP (12, 2) AWM_SHA1_GROUP (13, 2) AWM_SHA1_GROUP (14, 2)
      AWM_SHA1_GROUP (15, 3) AWM_SHA1_GROUP (16, 3) AWM_SHA1_GROUP (17, 3) AWM_SHA1_GROUP (18, 3) AWM_SHA1_GROUP (19, 3)
#undef AWM_SHA1_GROUP
      e0 = _mm_sha1nexte_epu32 (e0, e_in);
      abcd = _mm_add_epi32 (abcd, abcd_in);
    }
  _mm_storeu_si128 (reinterpret_cast<__m128i *> (state), _mm_shuffle_epi32 (abcd, 0x1b));
  state[4] = uint32_t (_mm_extract_epi32 (e0, 3));
}
static const bool have_shani = [] { __builtin_cpu_init(); return bool (__builtin_cpu_supports ("sha")) && bool (__builtin_cpu_supports ("sse4.1")); }();
#endif

void
sha1 (const void *data, size_t len, unsigned char digest[20])
{
  uint32_t state[5] = { 0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u, 0xc3d2e1f0u };
  const auto rotl = [] (uint32_t x, int s) { return (x << s) | (x >> (32 - s)); };
  const auto compress = [&] (const unsigned char *chunk)
    {
      uint32_t w[80];
      for (int t = 0; t < 16; t++)
        w[t] = uint32_t (chunk[4 * t]) << 24 | uint32_t (chunk[4 * t + 1]) << 16 | uint32_t (chunk[4 * t + 2]) << 8 | uint32_t (chunk[4 * t + 3]);
      for (int t = 16; t < 80; t++)
        w[t] = rotl (w[t - 3] ^ w[t - 8] ^ w[t - 14] ^ w[t - 16], 1);
      uint32_t a = state[0], b = state[1], c = state[2], d = state[3], e = state[4];
      for (int t = 0; t < 80; t++)
        {
          uint32_t f, k;
          switch (t / 20)
            {
            case 0:  f = (b & c) | (~b & d);          k = 0x5a827999u; break;
            case 1:  f = b ^ c ^ d;                   k = 0x6ed9eba1u; break;
            case 2:  f = (b & c) | (b & d) | (c & d); k = 0x8f1bbcdcu; break;
            default: f = b ^ c ^ d;                   k = 0xca62c1d6u; break;
            }
          const uint32_t tmp = rotl (a, 5) + f + e + k + w[t];
          e = d;
          d = c;
          c = rotl (b, 30);
          b = a;
          a = tmp;
        }
      state[0] += a; state[1] += b; state[2] += c; state[3] += d; state[4] += e;
    };
  const unsigned char *bytes = static_cast<const unsigned char *> (data);
  size_t rest = len;
#ifdef AWM_X86_CRYPTO
  if (have_shani && rest >= 64)
    {
      sha1_blocks_shani (state, bytes, rest / 64);
      bytes += rest / 64 * 64;
      rest %= 64;
    }
#endif
  while (rest >= 64)
    {
      compress (bytes);
      bytes += 64;
      rest -= 64;
    }
  unsigned char last[128] = { 0 };
  for (size_t i = 0; i < rest; i++)
    last[i] = bytes[i];
  last[rest] = 0x80;
  const size_t padded = rest + 9 <= 64 ? 64 : 128;
  const uint64_t bit_len = uint64_t (len) * 8;
  for (int i = 0; i < 8; i++)
    last[padded - 1 - i] = (unsigned char) (bit_len >> (8 * i));
  compress (last);
  if (padded == 128)
    compress (last + 64);
  for (int i = 0; i < 20; i++)
    digest[i] = (unsigned char) (state[i / 4] >> (24 - 8 * (i % 4)));
}

} // namespace awm
