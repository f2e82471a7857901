// resultset.cc -- what `get` reports: the list of decoded patterns, how chunk results are merged, ranked and printed.
//
// Observable behaviour == reference ResultSet (src/wmget.cc:163-474): two findings are the same if key, payload, block type,
// pattern type agree and time / speed are within a frame / 0.01; payloads are ranked by the float sum of their sync
// qualities ("all" patterns count twice); the text and JSON reports are byte-compatible (tests/test_cli_gpu.py compares them
// with the compiled reference binary line by line).
#include "wmget.hh"
#include "utils.hh"
#include <algorithm>
#include <cmath>
#include <tuple>
#include <unordered_map>

namespace awm {

namespace {

constexpr double SAME_TIME_SECONDS = Params::frame_size / double (Params::mark_sample_rate);
constexpr double SAME_SPEED = 0.01;

int
block_order (ConvBlockType t)          // A before B before AB
{
  return t == ConvBlockType::a ? 0 : t == ConvBlockType::b ? 1 : t == ConvBlockType::ab ? 2 : 99;
}

/* "A" / "B" / "AB", with the CLIP- prefix and -SPEED suffix of the reports; `all_label` replaces the block for "all" patterns */
std::string
type_label (const ResultSet::Pattern& p, const char *all_label)
{
  static const char *const block[] = { "A", "B", "AB" };
  std::string label = p.type == ResultSet::Type::ALL && all_label ? all_label : block[std::min (block_order (p.sync_score.block_type), 2)];
  if (p.type == ResultSet::Type::CLIP)
    label = "CLIP-" + label;
  if (p.speed != 1)
    label += "-SPEED";
  return label;
}

std::string
minutes_seconds (double time)
{
  const int seconds = int (time);
  return string_printf ("%d:%02d", seconds / 60, seconds % 60);
}

std::string
json_string (const std::string& s)
{
  std::string out;
  for (unsigned char c : s)
    if (c == '"' || c == '\\')
      out += std::string ("\\") + char (c);
    else if (c < 0x20)
      out += string_printf ("\\u%04x", c);
    else
      out += char (c);
  return out;
}

} // namespace

bool
ResultSet::Pattern::approx_match (const Pattern& other) const
{
  if (!(key == other.key) || type != other.type || sync_score.block_type != other.sync_score.block_type || bit_vec != other.bit_vec)
    return false;
  const bool same_place = type == Type::ALL || std::fabs (time - other.time) < SAME_TIME_SECONDS;
  return same_place && std::fabs (speed - other.speed) < SAME_SPEED;
}

void
ResultSet::add_pattern (const Key& key, double time, SyncFinder::Score sync_score, const std::vector<int>& bit_vec,
                        float decode_error, Type type, double speed)
{
  patterns.emplace_back();
  Pattern& p = patterns.back();
  p.key = key;
  p.time = time;
  p.sync_score = sync_score;
  p.bit_vec = bit_vec;
  p.decode_error = decode_error;
  p.type = type;
  p.speed = speed;
}

void
ResultSet::apply_time_offset (double time_offset)
{
  for (Pattern& p : patterns)
    p.time += time_offset;
}

/* a chunk's findings join the list unless an equivalent one is already there (chunks overlap by two blocks); the chunk's
 * patterns are taken in time order */
void
ResultSet::merge (ResultSet& other)
{
  std::vector<const Pattern *> incoming;
  for (const Pattern& p : other.patterns)
    incoming.push_back (&p);
  std::stable_sort (incoming.begin(), incoming.end(), [] (const Pattern *a, const Pattern *b) { return a->time < b->time; });
  for (const Pattern *p : incoming)
    if (std::none_of (patterns.begin(), patterns.end(), [p] (const Pattern& known) { return known.approx_match (*p); }))
      patterns.push_back (*p);
  if (m_debug_sync.empty())
    m_debug_sync = other.m_debug_sync;
}

/* rating of a payload = sum of the sync qualities of the patterns that carry it under this key, "all" patterns twice;
 * accumulated in float in list order (the value is printed in the JSON report) */
void
ResultSet::rate_patterns (const Key& key)
{
  std::unordered_map<std::string, float> score;
  std::vector<std::pair<Pattern *, float *>> mine;
  for (Pattern& p : patterns)
    if (p.key == key)
      {
        float& s = score[bit_vec_to_str (p.bit_vec)];
        s += p.sync_score.quality * (p.type == Type::ALL ? 2.f : 1.f);
        mine.emplace_back (&p, &s);
      }
  for (auto& m : mine)
    m.first->rating = *m.second;
}

void
ResultSet::sort (const std::vector<Key>& key_list)
{
  for (const Key& key : key_list)
    rate_patterns (key);
  // by key name; best rated payload first; within a payload the "all" pattern last, else by time, A < B < AB, payload text
  auto rank = [] (const Pattern& p) {
    return std::make_tuple (p.key.name(), -p.rating, p.type == Type::ALL, p.time, block_order (p.sync_score.block_type), bit_vec_to_str (p.bit_vec));
  };
  std::sort (patterns.begin(), patterns.end(), [&rank] (const Pattern& a, const Pattern& b) { return rank (a) < rank (b); });
}

void
ResultSet::print() const
{
  std::string listed_key;              // name of the key whose patterns are being listed (the default key has none: no "key" line)
  bool speed_line_due = true;
  for (const Pattern& p : patterns)
    {
      if (p.key.name() != listed_key)
        {
          listed_key = p.key.name();
          printf ("key %s\n", listed_key.c_str());
          speed_line_due = true;
        }
      if (speed_line_due)
        {
          // one "speed" line per key: the speed of its first pattern that was found on a stretched stream
          auto stretched = std::find_if (patterns.begin(), patterns.end(), [&p] (const Pattern& q) { return q.key == p.key && q.speed != 1; });
          if (stretched != patterns.end())
            printf ("speed %.6f\n", stretched->speed);
          speed_line_due = false;
        }
      const std::string bits = bit_vec_to_str (p.bit_vec);
      if (p.type == Type::ALL)
        printf ("pattern   all %s %.3f %.3f%s\n", bits.c_str(), p.sync_score.quality, p.decode_error, p.speed != 1 ? " SPEED" : "");
      else
        {
          const int seconds = int (p.time);
          printf ("pattern %2d:%02d %s %.3f %.3f %s\n", seconds / 60, seconds % 60, bits.c_str(), p.sync_score.quality, p.decode_error,
                  type_label (p, nullptr).c_str());
        }
    }
}

void
ResultSet::print_json (size_t time_length, const std::string& json_file) const
{
  FILE *out = fopen (json_file == "-" ? "/dev/stdout" : json_file.c_str(), "w");
  if (!out)
    {
      perror (("audiowmark: failed to open \"" + json_file + "\":").c_str());
      exit (127);
    }
  std::string matches;
  for (const Pattern& p : patterns)
    {
      if (!matches.empty())
        matches += ",\n";
      matches += string_printf ("    { \"key\": \"%s\", \"pos\": \"%s\", \"bits\": \"%s\", \"quality\": %.5f, \"error\": %.6f, \"rating\": %.5f, \"type\": \"%s\", \"speed\": %.6f }",
                                json_string (p.key.name()).c_str(), minutes_seconds (p.time).c_str(), bit_vec_to_str (p.bit_vec).c_str(),
                                p.sync_score.quality, p.decode_error, p.rating, type_label (p, "ALL").c_str(), p.speed);
    }
  fprintf (out, "{ \"length\": \"%ld:%02ld\",\n  \"matches\": [\n%s ]\n}\n", long (time_length / 60), long (time_length % 60), matches.c_str());
  fclose (out);
}

int
ResultSet::print_match_count (const std::vector<int>& orig_bits) const
{
  const long matches = std::count_if (patterns.begin(), patterns.end(), [&] (const Pattern& p) { return p.bit_vec == orig_bits; });
  printf ("match_count %ld %zd\n", matches, patterns.size());
  return int (matches);
}

} // namespace awm
