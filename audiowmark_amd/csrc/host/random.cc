#include "random.hh"
#include <cstring>
#include <vector>
#include "utils.hh"
#include <cinttypes>
#include <cstdio>
#include <cstdlib>

namespace awm {

Key::~Key()
{
  std::fill (m_aes_key.begin(), m_aes_key.end(), 0);
}

void
Key::set_test_key (uint64_t key)
{
  for (int i = 0; i < 8; i++)
    m_aes_key[i] = (unsigned char) (key >> (56 - 8 * i));
  std::fill (m_aes_key.begin() + 8, m_aes_key.end(), 0);
  m_name = string_printf ("test-key-%" PRId64, (int64_t) key);
}

void
Key::set_raw (const uint8_t key[SIZE], const std::string& name)
{
  m_aes_key.assign (key, key + SIZE);
  m_name = name;
}

namespace {
// key file tokenizer: words made of [A-Za-z0-9.:=/_-], "quoted strings" with \ escapes, # comments
bool
word_char (char c)
{
  return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9')
      || c == '.' || c == ':' || c == '=' || c == '/' || c == '-' || c == '_';
}
bool
space_char (char c)
{
  return c == ' ' || c == '\n' || c == '\t' || c == '\r';
}
bool
tokenize_line (const std::string& line, std::vector<std::string>& tokens)
{
  enum { BLANK, WORD, QUOTED, ESCAPED, COMMENT } state = BLANK;
  std::string cur;
  tokens.clear();
  for (char c : line + '\n')
    {
      switch (state)
        {
        case BLANK:
          if (word_char (c))        { state = WORD; cur += c; }
          else if (c == '"')        state = QUOTED;
          else if (space_char (c))  ;
          else if (c == '#')        state = COMMENT;
          else                      return false;
          break;
        case WORD:
          if (word_char (c))        cur += c;
          else if (space_char (c))  { tokens.push_back (cur); cur.clear(); state = BLANK; }
          else if (c == '#')        state = COMMENT;   // NB: reference drops the unfinished word here too
          else                      return false;
          break;
        case QUOTED:
          if (c == '"')             { tokens.push_back (cur); cur.clear(); state = BLANK; }
          else if (c == '\\')       state = ESCAPED;
          else                      cur += c;
          break;
        case ESCAPED:
          cur += c;
          state = QUOTED;
          break;
        case COMMENT:
          break;
        }
    }
  return state == BLANK || state == COMMENT;
}
} // namespace

void
Key::load_key (const std::string& filename)
{
  FILE *f = fopen (filename.c_str(), "r");
  if (!f)
    {
      error ("audiowmark: error opening key file: '%s'\n", filename.c_str());
      exit (1);
    }
  m_name = filename;
  const size_t sep = m_name.find_last_of ("\\/");
  if (sep != std::string::npos)
    m_name = m_name.substr (sep + 1);

  char buffer[1024];
  int line = 1, keys = 0;
  while (fgets (buffer, sizeof (buffer), f))
    {
      std::vector<std::string> tokens;
      bool ok = false;
      if (tokenize_line (buffer, tokens))
        {
          if (tokens.size() == 2 && tokens[0] == "key")
            {
              auto k = hex_str_to_vec (tokens[1]);
              if (k.size() != SIZE)
                {
                  error ("audiowmark: wrong key length in key file '%s', line %d\n => required key length is %zd bits\n", filename.c_str(), line, SIZE * 8);
                  exit (1);
                }
              m_aes_key = k;
              keys++;
              ok = true;
            }
          if (tokens.size() == 2 && tokens[0] == "name")
            {
              m_name = tokens[1];
              ok = true;
            }
          if (tokens.empty())
            ok = true;
        }
      if (!ok)
        {
          error ("audiowmark: parse error in key file '%s', line %d\n", filename.c_str(), line);
          exit (1);
        }
      line++;
    }
  fclose (f);
  if (keys > 1)
    {
      error ("audiowmark: key file '%s' contains more than one key\n", filename.c_str());
      exit (1);
    }
  if (keys == 0)
    {
      error ("audiowmark: key file '%s' contains no key\n", filename.c_str());
      exit (1);
    }
}

const unsigned __int128 *
Random::reciprocals()
{
  static const std::vector<unsigned __int128> table = [] {
    std::vector<unsigned __int128> t (MAX_FAST_DIVISOR + 1, 0);
    for (size_t d = 1; d <= MAX_FAST_DIVISOR; d++)
      t[d] = ~(unsigned __int128) 0 / d + 1;
    return t;
  }();
  return table.data();
}

Random::Random (const Key& key, uint64_t start_seed, Stream stream)
{
  m_aes.set_key (key.aes_key());
  seed (start_seed, stream);
}

void
Random::seed (uint64_t seed, Stream stream)
{
  uint8_t plain[16] = { 0 };
  for (int i = 0; i < 8; i++)
    plain[i] = uint8_t (seed >> (56 - 8 * i));
  plain[8] = uint8_t (stream);
  m_aes.encrypt_block (plain, m_counter);
  m_pos = WORDS;                       // discard buffered words
}

void
Random::refill()
{
  // CTR mode over an all-zero plaintext == raw keystream; 128-bit big-endian counter increment.  The counter blocks of
  // a refill are independent: encrypted as one batch.
  constexpr size_t BLOCKS = WORDS / 2;
  uint8_t counters[16 * BLOCKS], ks[16 * BLOCKS];
  for (size_t blk = 0; blk < BLOCKS; blk++)
    {
      std::memcpy (counters + 16 * blk, m_counter, 16);
      for (int k = 15; k >= 0; k--)
        if (++m_counter[k])
          break;
    }
  m_aes.encrypt_blocks (counters, ks, BLOCKS);
  for (size_t w = 0; w < WORDS; w++)
    {
      uint64_t v;
      std::memcpy (&v, ks + 8 * w, 8);
#if __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
      v = __builtin_bswap64 (v);                      // the stream is defined in big endian words
#endif
      m_words[w] = v;
    }
  m_pos = 0;
}

std::string
Random::gen_key()
{
  std::vector<unsigned char> key (Key::SIZE);
  FILE *f = fopen ("/dev/urandom", "rb");
  if (!f || fread (key.data(), 1, key.size(), f) != key.size())
    {
      error ("audiowmark: cannot read /dev/urandom\n");
      exit (1);
    }
  fclose (f);
  return vec_to_hex_str (key);
}

} // namespace awm
