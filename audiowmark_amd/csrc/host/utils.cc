#include "utils.hh"
#include <cstdio>
#include <cstdlib>
#include <sys/time.h>

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
