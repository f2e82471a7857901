// Key + AES-CTR pseudo random generator used for every key-dependent table.
// Interface and stream numbering follow reference src/random.hh:31-115; the
// generator itself (reference src/random.cc:97-161) is:
//   counter block = AES-ECB_key( seed as big-endian u64 || stream id || 7 zero bytes )
//   output        = AES-CTR keystream from that counter block, consumed in 256-byte
//                   refills as 32 big-endian u64 words.
#pragma once
#include <cstdint>
#include <random>
#include <string>
#include <vector>
#include "aes128.hh"

namespace awm {

class Key
{
  std::vector<unsigned char> m_aes_key;
  std::string                m_name;
public:
  static constexpr size_t SIZE = 16;
  Key() : m_aes_key (SIZE) {}
  ~Key();
  bool operator== (const Key& o) const { return m_aes_key == o.m_aes_key && m_name == o.m_name; }
  void set_test_key (uint64_t key);                 // reference random.cc:204-209
  void set_raw (const uint8_t key[SIZE], const std::string& name = "");
  void load_key (const std::string& filename);      // reference random.cc:295-360 (exits on parse errors)
  const unsigned char *aes_key() const { return m_aes_key.data(); }
  const std::string& name() const { return m_name; }
};

class Random
{
public:
  enum class Stream { data_up_down = 1, sync_up_down = 2, speed_clip = 3, mix = 4, bit_order = 5, frame_position = 6 };
  typedef uint64_t result_type;

  Random (const Key& key, uint64_t seed, Stream stream);
  void seed (uint64_t seed, Stream stream);

  result_type operator()()
  {
    if (m_pos == WORDS)
      refill();
    return m_words[m_pos++];
  }
  static constexpr result_type min() { return 0; }
  static constexpr result_type max() { return UINT64_MAX; }
  double random_double() { return m_double_dist (*this); }   // [0,1), libstdc++ generate_canonical as in the reference

  // Fisher-Yates exactly as reference random.hh:102-113 (j = i + rng() % (n - i)).  The 64 bit remainder is the cost of a key's tables
  // (235 000 draws per key, a hardware division each): for the divisors that occur (<= 51 480) it is taken by multiplication with a
  // precomputed 128 bit reciprocal instead (Lemire, Kaser, Kurz: "Faster remainder by direct computation", exact for all 64 bit x)
  template<class V> void
  shuffle (V& v)
  {
    const size_t n = v.size();
    if (n > MAX_FAST_DIVISOR)
      {
        for (size_t i = 0; i < n; i++)
          std::swap (v[i], v[i + size_t ((*this)() % (n - i))]);
        return;
      }
    const unsigned __int128 *const rec = reciprocals();
    for (size_t i = 0; i < n; i++)
      {
        const uint64_t d = n - i;
        const unsigned __int128 lowbits = rec[d] * (*this)();                      // mod 2^128
        const unsigned __int128 bottom = ((lowbits & 0xFFFFFFFFFFFFFFFFULL) * d) >> 64;
        const unsigned __int128 top = (lowbits >> 64) * d;
        std::swap (v[i], v[i + size_t ((bottom + top) >> 64)]);
      }
  }
  static constexpr size_t MAX_FAST_DIVISOR = 51480;
  static const unsigned __int128 *reciprocals();      // [d] = floor ((2^128 - 1) / d) + 1
  static std::string gen_key();
private:
  static constexpr size_t WORDS = 32;   // 256-byte refill
  Aes128   m_aes;
  uint8_t  m_counter[16];
  uint64_t m_words[WORDS];
  size_t   m_pos = WORDS;
  std::uniform_real_distribution<double> m_double_dist;
  void refill();
};

} // namespace awm
