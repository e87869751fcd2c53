// Synthetic blocks of the BASELINE.json configurations, the same
// bytes as bwtc_amd/synth.py produces with numpy: splitmix64, integer only.  Here so that the
// bench and the C++ host tools can make a 256 MiB block in a second instead of half a minute.
#include <cstdint>
#include <cstring>

#include "bwtc_hip.h"

namespace {

inline uint64_t splitmix(uint64_t seed, uint64_t k) {          // output number k (1-based) of the stream
  uint64_t z = seed + k * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

const char* const kTokens[64] = {
    "the ", "of ", "and ", "in ", "to ", "a ", "is ", "that ", "for ", "it ", "as ", "was ",
    "with ", "be ", "by ", "on ", "not ", "he ", "this ", "are ", "or ", "his ", "from ", "at ",
    "which ", "but ", "have ", "an ", "had ", "they ", "you ", "were ", "their ", "one ", "all ",
    "we ", "can ", "her ", "has ", "there ", "been ", "if ", "more ", "when ", "will ", "would ",
    "who ", "so ", "no ", "[[", "]]", "&quot;", "<page>\n", "</page>\n", "==", "'''",
    "[[Category:", "<title>", "</title>\n", "<text>", "</text>\n", "http://", ".\n", ", ",
};

}  // namespace

extern "C" int bwtc_hip_synth(char kind, uint64_t seed, uint64_t size, uint8_t* out) {
  if (!out && size) return -1;
  if (kind == 'r') {                                             // C1: little-endian bytes of successive outputs
    for (uint64_t i = 0, k = 1; i < size; i += 8, ++k) {
      const uint64_t z = splitmix(seed, k);
      const uint64_t n = size - i < 8 ? size - i : 8;
      for (uint64_t b = 0; b < n; ++b) out[i + b] = static_cast<uint8_t>(z >> (8 * b));
    }
    return 0;
  }
  if (kind == 'd') {                                             // C2: symbol j of a word = "ACGT"[(z >> 2j) & 3]
    for (uint64_t i = 0, k = 1; i < size; i += 32, ++k) {
      const uint64_t z = splitmix(seed, k);
      const uint64_t n = size - i < 32 ? size - i : 32;
      for (uint64_t j = 0; j < n; ++j) out[i + j] = static_cast<uint8_t>("ACGT"[(z >> (2 * j)) & 3]);
    }
    return 0;
  }
  if (kind == 't') {                                             // C3: four draws per token
    size_t len[64];
    for (int t = 0; t < 64; ++t) len[t] = std::strlen(kTokens[t]);
    uint64_t pos = 0, k = 1;
    while (pos < size) {
      const uint64_t z1 = splitmix(seed, k), z2 = splitmix(seed, k + 1), z3 = splitmix(seed, k + 2), z4 = splitmix(seed, k + 3);
      k += 4;
      const unsigned tok = static_cast<unsigned>(((z1 % 64) * (z2 % 64)) / 64);
      const char* s = kTokens[tok];
      for (size_t b = 0; b < len[tok] && pos < size; ++b) out[pos++] = static_cast<uint8_t>(s[b]);
      if ((z3 & 31) == 0 && pos < size) out[pos++] = static_cast<uint8_t>('a' + z4 % 26);
    }
    return 0;
  }
  return -1;
}
