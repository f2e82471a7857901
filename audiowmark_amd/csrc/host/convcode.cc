#include "convcode.hh"
#include <cassert>

namespace awm {

// generator polynomials, octal, interleaved A,B,A,B,... (reference convcode.cc:42-46)
static const unsigned ab_generators[conv_ab_rate] = {
  066561, 075211, 071545, 054435, 063635, 052475,
  063543, 075307, 052547, 045627, 067657, 051757
};

size_t
conv_code_size (ConvBlockType block_type, size_t msg_size)
{
  const size_t n = (msg_size + conv_order) * conv_ab_rate;
  return block_type == ConvBlockType::ab ? n : n / 2;
}

std::vector<unsigned>
conv_generators (ConvBlockType block_type)
{
  std::vector<unsigned> g;
  for (unsigned i = 0; i < conv_ab_rate; i++)
    {
      const bool is_b = i & 1;
      if (block_type == ConvBlockType::ab || (block_type == ConvBlockType::a && !is_b) || (block_type == ConvBlockType::b && is_b))
        g.push_back (ab_generators[i]);
    }
  return g;
}

std::vector<int>
conv_encode (ConvBlockType block_type, const std::vector<int>& in_bits)
{
  const auto generators = conv_generators (block_type);
  std::vector<int> out;
  out.reserve ((in_bits.size() + conv_order) * generators.size());
  unsigned reg = 0;
  const size_t total = in_bits.size() + conv_order;      // zero tail terminates the trellis
  for (size_t i = 0; i < total; i++)
    {
      const unsigned bit = i < in_bits.size() ? (in_bits[i] & 1) : 0;
      reg = (reg << 1) | bit;
      for (unsigned poly : generators)
        out.push_back (__builtin_parity (reg & poly));
    }
  return out;
}

size_t code_size (ConvBlockType block_type, size_t msg_size) { return conv_code_size (block_type, msg_size); }
std::vector<int> code_encode (ConvBlockType block_type, const std::vector<int>& in_bits) { return conv_encode (block_type, in_bits); }

} // namespace awm
