// Small host utilities: Error value type, logging, bit/hex string helpers.
// Mirrors the surface of reference src/utils.hh:87-130 (Error), utils.cc:95-164
// (bit_str_to_vec / bit_vec_to_str / hex helpers) and utils.cc:195-254 (log levels).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <string>
#include <vector>

namespace awm {

class Error
{
public:
  enum class Code { NONE, STR };
  Error (Code code = Code::NONE) : m_code (code), m_message (code == Code::NONE ? "OK" : "Unknown error") {}
  explicit Error (const std::string& message) : m_code (Code::STR), m_message (message) {}
  Code code() const { return m_code; }
  const char *message() const { return m_message.c_str(); }
  operator bool() const { return m_code != Code::NONE; }   // truthy == failure, as in the reference
private:
  Code        m_code;
  std::string m_message;
};

enum class Log { ERROR = 3, WARNING = 2, INFO = 1, DEBUG = 0 };
void set_log_level (Log level);
void error (const char *format, ...) __attribute__ ((format (printf, 1, 2)));
void warning (const char *format, ...) __attribute__ ((format (printf, 1, 2)));
void info (const char *format, ...) __attribute__ ((format (printf, 1, 2)));
void debug (const char *format, ...) __attribute__ ((format (printf, 1, 2)));

std::string string_printf (const char *format, ...) __attribute__ ((format (printf, 1, 2)));

std::vector<int>            bit_str_to_vec (const std::string& bits);      // hex nibbles -> bits, MSB first; empty on error
std::string                 bit_vec_to_str (const std::vector<int>& bits); // groups of 4 bits -> hex; trailing partial nibble dropped
std::vector<unsigned char>  hex_str_to_vec (const std::string& str);
std::string                 vec_to_hex_str (const std::vector<unsigned char>& vec);

double get_time();

// FIPS 180-4 SHA-1 (the reference seeds the speed clip selection from libgcrypt's GCRY_MD_SHA1, random.cc:184-190)
void sha1 (const void *data, size_t len, unsigned char digest[20]);

} // namespace awm
