// Host-side small-table work of the 'H' coder.  See entropy_host.hpp.
#include "entropy_host.hpp"
#include <algorithm>
#include <utility>

namespace bwtc {
namespace utils {

unsigned packInteger(uint64_t value, std::vector<uint8_t>& out) {
  unsigned n = 0;
  do {
    uint8_t b = static_cast<uint8_t>(value & 0x7F);
    value >>= 7;
    if (value) b |= 0x80;
    out.push_back(b);
    ++n;
  } while (value);
  return n;
}

// Moffat & Katajainen, "In-place calculation of minimum-redundancy codes": three sweeps
// over one array holding, in turn, weights -> parent links -> internal depths -> leaf
// depths.  Leaves win ties against internal nodes, as in the reference (Utils.cpp:440-455).
void calculateHuffmanLengths(const uint64_t freqs[256], uint8_t clen[256]) {
  std::pair<uint64_t, uint32_t> order[256];
  size_t n = 0;
  for (uint32_t c = 0; c < 256; ++c) {
    clen[c] = 0;
    if (freqs[c]) order[n++] = std::make_pair(freqs[c], c);
  }
  if (n == 0) return;
  if (n == 1) { clen[order[0].second] = 1; return; }
  std::sort(order, order + n);
  uint64_t w[256];
  for (size_t i = 0; i < n; ++i) w[i] = order[i].first;

  size_t leaf = 0, node = 0;          // next unused leaf / internal node
  for (size_t next = 0; next + 1 < n; ++next) {
    uint64_t sum = 0;
    for (int pick = 0; pick < 2; ++pick) {
      const bool take_node = leaf >= n || (node < next && w[node] < w[leaf]);
      if (take_node) { sum += w[node]; w[node++] = next; }   // link child -> parent
      else sum += w[leaf++];
    }
    w[next] = sum;
  }
  w[n - 2] = 0;                                              // root depth
  for (size_t i = n - 2; i-- > 0;) w[i] = w[w[i]] + 1;       // internal node depths

  long avail = 1, used = 0, leaf_slot = static_cast<long>(n) - 1, inode = static_cast<long>(n) - 2;
  for (uint64_t depth = 0; avail > 0; ++depth) {
    while (inode >= 0 && w[inode] == depth) { ++used; --inode; }
    while (avail > used) { w[leaf_slot--] = depth; --avail; }
    avail = 2 * used;
    used = 0;
  }
  for (size_t i = 0; i < n; ++i) clen[order[i].second] = static_cast<uint8_t>(w[i]);
}

void computeHuffmanCodes(const uint8_t clen[256], uint32_t code[256]) {
  uint32_t count[256] = {0};
  int max_len = 0;
  for (int c = 0; c < 256; ++c) {
    ++count[clen[c]];
    if (clen[c] > max_len) max_len = clen[c];
  }
  // the first code of each length, walking from the longest length up to the root
  uint32_t first[256] = {0};
  uint32_t next_first = 0;
  for (int len = max_len; len >= 0; --len) {
    first[len] = next_first;
    next_first = (next_first + count[len]) >> 1;
  }
  for (int c = 0; c < 256; ++c) code[c] = clen[c] ? first[clen[c]]++ : 0;
}

void BitSink::push(uint64_t value, unsigned nbits) {
  for (unsigned i = nbits; i-- > 0;) {
    if ((nbits_ & 7) == 0) bytes_.push_back(0);
    if ((value >> i) & 1) bytes_.back() |= static_cast<uint8_t>(0x80u >> (nbits_ & 7));
    ++nbits_;
  }
}

void BitSink::flushTo(std::vector<uint8_t>& out) {
  out.insert(out.end(), bytes_.begin(), bytes_.end());
  bytes_.clear();
  nbits_ = 0;
}

static unsigned ceilLog2(size_t n) {
  unsigned l = 0;
  while ((static_cast<size_t>(1) << l) < n) ++l;
  return l;
}

// Minimal binary code for n in [lo, hi]: the 2^k - range shortest codewords sit in the
// middle of the range (Utils.hpp:239-252).
void binaryCode(size_t n, size_t lo, size_t hi, BitSink& bits) {
  const size_t range = hi - lo + 1;
  if (range == 1) return;
  const unsigned k = ceilLog2(range);
  const size_t n_short = (static_cast<size_t>(1) << k) - range;
  const size_t n_long_left = (range - n_short) / 2;
  const size_t v = n - lo;
  if (v < n_long_left) bits.push(v, k);
  else if (v < n_long_left + n_short) bits.push(v, k - 1);
  else bits.push(v - n_short, k);
}

static void interpolative(const std::vector<uint32_t>& list, size_t begin, size_t end, size_t lo,
                          size_t hi, BitSink& bits) {
  // [begin, end] inclusive; nothing to say when the values are forced
  if (begin > end || end - begin == hi - lo) return;
  if (begin == end) { binaryCode(list[begin], lo, hi, bits); return; }
  const size_t h = (end - begin) / 2, mid = begin + h;
  binaryCode(list[mid], lo + h, hi - (end - mid), bits);
  if (mid > begin) interpolative(list, begin, mid - 1, lo, list[mid] - 1, bits);
  interpolative(list, mid + 1, end, list[mid] + 1, hi, bits);
}

void binaryInterpolativeCode(const std::vector<uint32_t>& list, size_t maxValue, BitSink& bits) {
  if (list.empty()) return;
  interpolative(list, 0, list.size() - 1, 0, maxValue, bits);
}

void unaryCode(BitSink& bits, size_t n) {
  for (size_t i = 1; i < n; ++i) bits.pushBit(false);
  bits.pushBit(true);
}

}  // namespace utils

void serializeShape(const uint8_t clen[256], std::vector<uint8_t>& out) {
  std::vector<uint32_t> symbols;
  unsigned max_len = 0;
  for (uint32_t c = 0; c < 256; ++c) {
    if (clen[c]) {
      symbols.push_back(c);
      if (clen[c] > max_len) max_len = clen[c];
    }
  }
  utils::BitSink bits;
  bits.push(symbols.back(), 8);              // largest symbol
  bits.push(symbols.size(), 8);              // number of symbols (256 -> 0)
  std::vector<uint8_t> packed;
  utils::packInteger(max_len, packed);       // low byte first == pushBits(packedInt, 8*bytes)
  for (size_t i = packed.size(); i-- > 0;) bits.push(packed[i], 8);
  utils::binaryInterpolativeCode(symbols, symbols.back(), bits);
  for (size_t i = 0; i < symbols.size(); ++i) utils::unaryCode(bits, max_len - clen[symbols[i]] + 1);
  bits.flushTo(out);
}

std::vector<uint32_t> deduceSections(const uint32_t freqs[256]) {
  std::vector<uint32_t> sections;
  uint64_t sum = 0;
  for (int c = 0; c < 256; ++c) {
    sum += freqs[c];
    if (sum >= 10000) { sections.push_back(static_cast<uint32_t>(sum)); sum = 0; }
  }
  if (sum) {
    if (sections.empty()) sections.push_back(static_cast<uint32_t>(sum));
    else sections.back() += static_cast<uint32_t>(sum);
  }
  return sections;
}

void writeBWTBlockHeader(const uint32_t* lf, uint32_t n_lf, std::vector<uint8_t>& out) {
  out.push_back(static_cast<uint8_t>(n_lf - 1));
  utils::BitSink bits;
  for (uint32_t i = 0; i < n_lf; ++i) bits.push(lf[i] & 0x7FFFFFFFu, 31);
  bits.flushTo(out);
}

}  // namespace bwtc
