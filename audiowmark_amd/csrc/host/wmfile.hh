// File / stream level entry points with the reference's signatures (reference src/wmcommon.hh:226-228):
//   add_stream_watermark, add_watermark   reference src/wmadd.cc:448-657
//   get_watermark                         reference src/wmget.cc:971-1013 (+ report :941-969)
#pragma once
#include "audiostream.hh"
#include "wmcommon.hh"

struct awm_ctx;

namespace awm {

int add_stream_watermark (awm_ctx *ctx, const Key& key, AudioInputStream *in_stream, AudioOutputStream *out_stream,
                          const std::string& bits, size_t zero_frames);
int add_watermark (awm_ctx *ctx, const Key& key, const std::string& infile, const std::string& outfile, const std::string& bits);
int add_watermark_at (awm_ctx *ctx, const Key& key, const std::string& infile, const std::string& outfile, const std::string& bits,
                      size_t zero_frames);
int get_watermark (awm_ctx *ctx, const std::vector<Key>& key_list, const std::string& infile, const std::string& orig_pattern);
class ResultSet;
int get_watermark_stream (awm_ctx *ctx, const std::vector<Key>& key_list, AudioInputStream *in_stream, bool print_speed, ResultSet& result_set,
                          size_t& n_values_out, const std::string& what);
int get_watermark_loaded (awm_ctx *ctx, const std::vector<Key>& key_list, size_t n_values, int n_channels, int sample_rate, bool print_speed,
                          ResultSet& result_set, size_t& n_values_out);
// add_watermark, and get_watermark of what it wrote without reading the file back (the output stage keeps the quantised samples in HBM)
int add_get_watermark (awm_ctx *ctx, const Key& key, const std::string& infile, const std::string& outfile, const std::string& bits,
                       ResultSet& result_set);

// kind of the last failure of the functions above on this thread (AWM_ERR_ARG / AWM_ERR_IO / AWM_ERR_HIP) for the C ABI
int  file_fail_kind();
void file_fail_reset();

int test_change_speed (awm_ctx *ctx, const std::string& infile, const std::string& outfile, double speed);   // reference audiowmark.cc:419-437

} // namespace awm
