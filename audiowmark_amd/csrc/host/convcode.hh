// Rate-1/6 (A or B block) / 1/12 (AB) K=15 convolutional code.
// Interface follows reference src/convcode.hh:23-32; generators and termination follow
// reference src/convcode.cc:42-49,100-125.  The soft Viterbi decoder of the reference
// (convcode.cc:128-213) runs on the GPU here (csrc/hip/viterbi.hip, awm_viterbi_decode).
#pragma once
#include <cstddef>
#include <vector>

namespace awm {

enum class ConvBlockType { a, b, ab };

constexpr unsigned int conv_order = 15;
constexpr unsigned int conv_ab_rate = 12;

size_t                conv_code_size (ConvBlockType block_type, size_t msg_size);
std::vector<unsigned> conv_generators (ConvBlockType block_type);
std::vector<int>      conv_encode (ConvBlockType block_type, const std::vector<int>& in_bits);

// payload code dispatch (reference src/shortcode.cc:112-133); the deprecated --short
// block codes are out of scope, so these forward to the conv code
size_t                code_size (ConvBlockType block_type, size_t msg_size);
std::vector<int>      code_encode (ConvBlockType block_type, const std::vector<int>& in_bits);

} // namespace awm
