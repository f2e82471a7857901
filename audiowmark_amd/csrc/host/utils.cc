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

} // namespace awm
