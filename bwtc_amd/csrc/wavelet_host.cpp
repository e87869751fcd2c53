// Host half of the wavelet coder 'B'.  See wavelet_host.hpp.  Reference quirks that decide
// output bytes are kept on purpose and marked "sic".
#include "wavelet_host.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

#include "entropy_host.hpp"
#include "wavelet_rc.hpp"

namespace bwtc {
namespace wavelet {
namespace {

typedef std::pair<uint64_t, uint32_t> LenSym;     // <length or frequency, symbol>
typedef std::vector<uint8_t> Bits;                // one bit per element

// ---- small pieces of WaveletTree.hpp ---------------------------------------------------
unsigned logFloor(uint64_t n) { unsigned l = 0; while (n > 1) { n >>= 1; ++l; } return l; }

uint32_t lengthOfFixedCode(uint32_t x, uint32_t w) {               // :528-533
  const uint64_t B = logFloor(static_cast<uint64_t>(x) - 1 + (uint64_t(1) << w)) - w;
  return static_cast<uint32_t>(2 * B + w + 1);
}

void fixedIntegerCode(Bits& bits, uint32_t x, uint32_t w) {        // :514-526
  const uint64_t B = logFloor(static_cast<uint64_t>(x) - 1 + (uint64_t(1) << w)) - w;
  bits.insert(bits.end(), B, 1);
  bits.push_back(0);
  const uint64_t y = static_cast<uint64_t>(x) - (1 + (((uint64_t(1) << B) - 1) << w));
  for (int i = static_cast<int>(w + B) - 1; i >= 0; --i) bits.push_back((y >> i) & 1);
}

// utils::calculateCodeLengths (Utils.cpp:422-473) on <weight, symbol> pairs; the weights are
// replaced by code lengths.  `sorted` = the caller claims ascending order (it is not always
// true, WaveletTree.hpp:1622-1625 -- sic).
void calculateCodeLengths(std::vector<LenSym>& v, bool sorted) {
  const size_t n = v.size();
  if (n == 1) { v[0].first = 1; return; }
  if (!sorted) std::sort(v.begin(), v.end());
  std::vector<uint64_t> a(n);
  for (size_t i = 0; i < n; ++i) a[i] = v[i].first;
  // The selection only looks at leaves not yet consumed (index >= leaf > next) and at
  // finished internal nodes (index < next), so it is well defined for unsorted input too.
  size_t leaf = 0, inner = 0;
  for (size_t next = 0; next + 1 < n; ++next) {
    uint64_t sum = 0;
    for (int pick = 0; pick < 2; ++pick) {
      const bool take_inner = leaf >= n || (inner < next && a[inner] < a[leaf]);
      if (take_inner) { sum += a[inner]; a[inner++] = next; }
      else sum += a[leaf++];
    }
    a[next] = sum;
  }
  a[n - 2] = 0;
  for (size_t i = n - 2; i-- > 0;) a[i] = a[a[i]] + 1;
  long avail = 1, used = 0, slot = static_cast<long>(n) - 1, in = static_cast<long>(n) - 2;
  for (uint64_t depth = 0; avail > 0; ++depth) {
    while (in >= 0 && a[in] == depth) { ++used; --in; }
    while (avail > used) { a[slot--] = depth; --avail; }
    avail = 2 * used;
    used = 0;
  }
  for (size_t i = 0; i < n; ++i) v[i].first = a[i];
}

// bitsForIntegers, :1552-1641.  Returns false when the reference would step a reverse
// iterator past rend() (undefined behaviour there; the candidate is dropped here).
bool bitsForIntegers(uint32_t w, uint32_t depth, const std::vector<LenSym>& freqs,
                     std::vector<LenSym>& codeLengths, uint64_t totalFreq, uint64_t* bits) {
  *bits = 0;
  if (freqs.size() < depth) return true;
  if (depth == 0) {
    for (size_t i = 0; i < freqs.size(); ++i) *bits += freqs[i].first * lengthOfFixedCode(freqs[i].second, w);
    return true;
  }
  std::vector<bool> removed(freqs.size(), false);
  std::map<uint64_t, uint32_t> notRemoved;          // keyed by frequency: equal ones collapse (sic)
  uint64_t minFreq = 0;
  for (size_t i = 0; i < freqs.size(); ++i) {
    const LenSym& p = freqs[i];
    const double optimalLength = std::log(totalFreq / static_cast<double>(p.first));   // natural log (sic)
    const double price = static_cast<double>(p.first * (depth + lengthOfFixedCode(p.second, w)));
    if (p.first * optimalLength + 64 < price && p.first >= 10) {
      removed[i] = true;
      codeLengths.push_back(p);
      if (minFreq == 0 || minFreq > p.first) minFreq = p.first;
    } else {
      notRemoved[p.first] = static_cast<uint32_t>(i);
    }
  }
  if (depth > codeLengths.size()) {
    const size_t need = depth - codeLengths.size();
    if (need > notRemoved.size()) return false;
    std::map<uint64_t, uint32_t>::const_reverse_iterator it = notRemoved.rbegin();
    for (size_t k = 0; k < need; ++k, ++it) {
      removed[it->second] = true;
      codeLengths.push_back(freqs[it->second]);
      if (minFreq == 0 || minFreq > it->first) minFreq = it->first;
    }
    codeLengths.push_back(LenSym(minFreq, 0));      // the node of the fixed codes
    std::sort(codeLengths.begin(), codeLengths.end());
  } else {
    codeLengths.push_back(LenSym(minFreq, 0));      // not re-sorted (sic)
  }
  calculateCodeLengths(codeLengths, true);
  std::sort(codeLengths.begin(), codeLengths.end());
  size_t j = codeLengths.size() - 1;
  while (codeLengths[j].second != 0) --j;
  while (j > 0 && codeLengths[j - 1].first >= depth) { std::swap(codeLengths[j].second, codeLengths[j - 1].second); --j; }
  const uint32_t realDepth = codeLengths[j].second;                 // the symbol, i.e. 0 (sic)
  std::map<uint32_t, uint64_t> huffmanLengths;
  for (size_t i = 0; i < codeLengths.size(); ++i) huffmanLengths[codeLengths[i].second] = codeLengths[i].first;
  for (size_t i = 0; i < freqs.size(); ++i) {
    const LenSym& p = freqs[i];
    *bits += removed[i] ? p.first * huffmanLengths[p.second]
                        : p.first * (realDepth + lengthOfFixedCode(p.second, w));
  }
  return true;
}

// findParametersForSemiFixedCodes, :1644-1680
uint32_t findParametersForSemiFixedCodes(std::vector<LenSym>& freqs, uint64_t totalFreq) {
  std::sort(freqs.begin(), freqs.end());
  std::vector<LenSym> codeLengths;
  uint32_t bestW = 0;
  uint64_t leastBits = 0, bits = 0;
  bitsForIntegers(0, 0, freqs, codeLengths, totalFreq, &leastBits);
  for (uint32_t w = 1; w < 16; ++w) {
    bitsForIntegers(w, 0, freqs, codeLengths, totalFreq, &bits);
    if (bits < leastBits) bestW = w;                                 // leastBits stays (sic)
  }
  for (uint32_t w = 0; w < 16; ++w) {
    for (uint32_t depth = 1; depth <= 3; ++depth) {
      std::vector<LenSym> tmp;
      if (!bitsForIntegers(w, depth, freqs, tmp, totalFreq, &bits)) continue;
      if (bits == 0) break;
      if (bits < leastBits) { codeLengths = tmp; leastBits = bits; bestW = w; }
    }
  }
  freqs = codeLengths;
  return bestW;
}

// ---- the tree ----------------------------------------------------------------------------
// bit sequence of a tree node (and of a node's "gap" vector), 64 bits per word
class PackedBits {
 public:
  PackedBits() : n_(0) {}
  void push(bool b) {
    if ((n_ & 63) == 0) w_.push_back(0);
    w_.back() |= static_cast<uint64_t>(b) << (n_ & 63);
    ++n_;
  }
  bool operator[](size_t i) const { return (w_[i >> 6] >> (i & 63)) & 1; }
  size_t size() const { return n_; }
  void release() { std::vector<uint64_t>().swap(w_); n_ = 0; }
 private:
  std::vector<uint64_t> w_;
  size_t n_;
};

struct Node {
  int left, right;
  bool hasSymbol;
  uint32_t symbol;
  PackedBits bits;
  PackedBits gaps;     // filled by the parent during the breadth-first walk
  Node() : left(-1), right(-1), hasSymbol(false), symbol(0) {}
};

class Tree {
 public:
  std::vector<Node> nodes;
  int root;
  Bits codes[256];
  std::map<uint32_t, Bits> integerCodes;
  uint32_t W;

  int add(bool hasSymbol, uint32_t symbol) {
    nodes.push_back(Node());
    nodes.back().hasSymbol = hasSymbol;
    nodes.back().symbol = symbol;
    return static_cast<int>(nodes.size()) - 1;
  }

  // static assignPrefixCodes(lengths, node, elem, bits), :1446-1482, on any node pool
  static size_t assignPrefixCodes(std::vector<Node>& pool, const std::vector<LenSym>& lengths, int nd,
                                  size_t elem, size_t depth) {
    if (elem >= lengths.size()) return elem;
    if (depth == lengths[elem].first - 1) {
      Node leaf; leaf.hasSymbol = true; leaf.symbol = lengths[elem].second;
      pool.push_back(leaf);
      const int c = static_cast<int>(pool.size()) - 1;
      if (pool[nd].left < 0) { pool[nd].left = c; return assignPrefixCodes(pool, lengths, nd, elem + 1, depth); }
      pool[nd].right = c;
      return elem + 1;
    }
    if (pool[nd].left < 0) {
      pool.push_back(Node());
      const int c = static_cast<int>(pool.size()) - 1;
      pool[nd].left = c;
      elem = assignPrefixCodes(pool, lengths, c, elem, depth + 1);
    }
    if (elem < lengths.size()) {
      pool.push_back(Node());
      const int c = static_cast<int>(pool.size()) - 1;
      pool[nd].right = c;
      elem = assignPrefixCodes(pool, lengths, c, elem, depth + 1);
    }
    return elem;
  }

  template <typename Sink>
  static void collectCodes(const std::vector<Node>& pool, int nd, Bits& path, Sink& sink) {   // :1533-1550
    if (pool[nd].left < 0 && pool[nd].right < 0) sink(pool[nd].symbol, path);
    if (pool[nd].left >= 0) { path.push_back(0); collectCodes(pool, pool[nd].left, path, sink); path.pop_back(); }
    if (pool[nd].right >= 0) { path.push_back(1); collectCodes(pool, pool[nd].right, path, sink); path.pop_back(); }
  }

  struct SymbolSink { Bits* codes; void operator()(uint32_t s, const Bits& p) { codes[s] = p; } };
  struct IntegerSink { std::map<uint32_t, Bits>* m; void operator()(uint32_t s, const Bits& p) { (*m)[s] = p; } };

  // WaveletTree(const byte*, size_t), :294-360, from the GPU's run statistics: the two code
  // sets and the symbol tree
  void buildCodes(const SectionRuns& in) {
    std::vector<LenSym> lengths;
    for (uint32_t c = 0; c < 256; ++c) if (in.run_freqs[c]) lengths.push_back(LenSym(in.run_freqs[c], c));
    calculateCodeLengths(lengths, false);
    std::sort(lengths.begin(), lengths.end());                       // assignPrefixCodes(lengths), :1440-1444
    root = add(false, 0);
    assignPrefixCodes(nodes, lengths, root, 0, 0);

    std::vector<LenSym> integerLengths;
    for (size_t i = 0; i < in.n_dist; ++i) integerLengths.push_back(LenSym(in.dist[i].second, in.dist[i].first));
    W = findParametersForSemiFixedCodes(integerLengths, in.n_runs);
    if (!integerLengths.empty()) {
      std::vector<Node> ipool(1);
      assignPrefixCodes(ipool, integerLengths, 0, 0, 0);
      Bits path;
      IntegerSink sink = {&integerCodes};
      collectCodes(ipool, 0, path, sink);
    } else {
      integerCodes[0] = Bits();
    }
    Bits path;
    SymbolSink sink = {codes};
    collectCodes(nodes, root, path, sink);
  }

  // the code of a run length: its own prefix code, or the escape code + the fixed code
  Bits lengthCodeOf(uint32_t len) {
    std::map<uint32_t, Bits>::const_iterator it = integerCodes.find(len);
    if (it != integerCodes.end()) return it->second;
    Bits code = integerCodes[0];
    fixedIntegerCode(code, len, W);
    return code;
  }

  void build(const SectionRuns& in) {
    buildCodes(in);
    // pushMessage / pushRun, :1484-1494, :1251-1275.  Every run appends its symbol's code bits
    // to the nodes on the root->leaf path and its length code to the leaf's integer subtree.
    // Both paths are resolved once per symbol / per (symbol, small length) and then replayed.
    std::vector<std::pair<int, uint8_t> > symPath[256];
    int leafOf[256];
    for (uint32_t c = 0; c < 256; ++c) {
      leafOf[c] = -1;
      if (codes[c].empty()) continue;
      int nd = root;
      for (size_t i = 0; i < codes[c].size(); ++i) {
        symPath[c].push_back(std::make_pair(nd, codes[c][i]));
        nd = codes[c][i] ? nodes[nd].right : nodes[nd].left;
      }
      leafOf[c] = nd;
    }
    const uint32_t kCachedLengths = 512;
    std::vector<std::pair<int, uint8_t> > steps;                  // replay pool
    std::vector<std::pair<uint32_t, uint32_t> > cache(256 * kCachedLengths, std::make_pair(0u, 0u));
    std::vector<std::pair<int, uint8_t> > scratch;
    Bits lengthCode;
    for (uint64_t r = 0; r < in.n_runs; ++r) {
      const uint32_t c = in.symbols[r];
      const std::vector<std::pair<int, uint8_t> >& sp = symPath[c];
      for (size_t i = 0; i < sp.size(); ++i) nodes[sp[i].first].bits.push(sp[i].second);
      const uint32_t len = in.starts[r + 1] - in.starts[r];
      std::pair<uint32_t, uint32_t>* slot = len < kCachedLengths ? &cache[c * kCachedLengths + len] : 0;
      if (slot && slot->second) {
        const std::pair<int, uint8_t>* q = &steps[slot->first];
        for (uint32_t i = 0; i < slot->second; ++i) nodes[q[i].first].bits.push(q[i].second);
        continue;
      }
      lengthCode = lengthCodeOf(len);
      scratch.clear();
      pushBelow(leafOf[c], lengthCode, len, scratch);
      if (slot) {
        slot->first = static_cast<uint32_t>(steps.size());
        slot->second = static_cast<uint32_t>(scratch.size());
        steps.insert(steps.end(), scratch.begin(), scratch.end());
      }
    }
  }

  // pushBits(node, bits, symbol), :1228-1249; `path` receives the (node, bit) steps taken
  void pushBelow(int nd, const Bits& bits, uint32_t symbol, std::vector<std::pair<int, uint8_t> >& path) {
    for (size_t i = 0; i + 1 < bits.size(); ++i) {
      nodes[nd].bits.push(bits[i]);
      path.push_back(std::make_pair(nd, bits[i]));
      int next = bits[i] ? nodes[nd].right : nodes[nd].left;
      if (next < 0) {
        next = add(false, 0);
        if (bits[i]) nodes[nd].right = next; else nodes[nd].left = next;
      }
      nd = next;
    }
    nodes[nd].bits.push(bits.back());
    path.push_back(std::make_pair(nd, bits.back()));
    if (bits.back()) { if (nodes[nd].right < 0) { const int c = add(true, symbol); nodes[nd].right = c; } }
    else if (nodes[nd].left < 0) { const int c = add(true, symbol); nodes[nd].left = c; }
  }

  // treeShape, :553-635, byte padded as WaveletEncoder::encodeData writes it (:124-135)
  void shape(std::vector<uint8_t>& out) const {
    utils::BitSink sink;
    std::vector<uint32_t> symbols;
    size_t maxLen = 0;
    for (uint32_t c = 0; c < 256; ++c) if (!codes[c].empty()) { symbols.push_back(c); maxLen = std::max(maxLen, codes[c].size()); }
    sink.push(symbols.back(), 8);
    sink.push(symbols.size(), 8);
    pushPackedRev(sink, maxLen);
    utils::binaryInterpolativeCode(symbols, symbols.back(), sink);
    for (size_t i = 0; i < symbols.size(); ++i) utils::unaryCode(sink, maxLen - codes[symbols[i]].size() + 1);
    std::vector<uint32_t> integers;
    maxLen = 0;
    for (std::map<uint32_t, Bits>::const_iterator it = integerCodes.begin(); it != integerCodes.end(); ++it) {
      integers.push_back(it->first);
      maxLen = std::max(maxLen, it->second.size());
    }
    pushPackedRev(sink, integers.back());
    if (integers.back() > 0) {
      pushPackedRev(sink, integers.size());
      pushPackedRev(sink, maxLen);
      utils::binaryInterpolativeCode(integers, integers.back(), sink);
      for (std::map<uint32_t, Bits>::const_iterator it = integerCodes.begin(); it != integerCodes.end(); ++it)
        utils::unaryCode(sink, maxLen - it->second.size() + 1);
    }
    sink.push(W & 15u, 4);
    sink.flushTo(out);
  }

  static void pushPackedRev(utils::BitSink& sink, uint64_t value) {  // packInteger + pushBitsRev
    std::vector<uint8_t> packed;
    utils::packInteger(value, packed);
    for (size_t b = 0; b < packed.size(); ++b)
      for (int i = 0; i < 8; ++i) sink.pushBit((packed[b] >> i) & 1);
  }

  // encodeTreeBF, :637-809, as a traversal that reports every coded bit to a visitor:
  //   v.pm(bit)            bit coded with the main model
  //   v.gap(bit, state)    bit coded with the gap model; state = also advance the main model's state
  //   v.integer(bit)       bit coded with the integer model
  //   v.resetInternal()    before every internal node;  v.resetIntegerLevel() before every level
  template <typename Visitor>
  void walk(Visitor& v) {
    std::deque<int> queue, integerNodes;
    {
      Node& r = nodes[root];
      PackedBits* leftGaps = r.left >= 0 ? &nodes[r.left].gaps : 0;
      PackedBits* rightGaps = r.right >= 0 ? &nodes[r.right].gaps : 0;
      bool prev = !r.bits[0];
      for (size_t i = 0, n = r.bits.size(); i < n; ++i) {
        const bool bit = r.bits[i];
        v.pm(bit);
        (bit ? rightGaps : leftGaps)->push(prev != bit);
        prev = bit;
      }
      if (r.left >= 0) { if (nodes[r.left].hasSymbol) integerNodes.push_back(r.left); else queue.push_back(r.left); }
      if (r.right >= 0) { if (nodes[r.right].hasSymbol) integerNodes.push_back(r.right); else queue.push_back(r.right); }
    }
    while (!queue.empty()) {
      v.resetInternal();
      Node& nd = nodes[queue.front()];
      queue.pop_front();
      const PackedBits& gaps = nd.gaps;
      const bool leftSym = nodes[nd.left].hasSymbol, rightSym = nodes[nd.right].hasSymbol;
      const size_t n = nd.bits.size();
      bool prev = !nd.bits[0];
      if (leftSym && rightSym) {
        for (size_t i = 0; i < n; ++i) if (gaps[i]) v.gap(nd.bits[i], false);
        integerNodes.push_back(nd.left);
        integerNodes.push_back(nd.right);
      } else if (leftSym) {
        PackedBits& rightGaps = nodes[nd.right].gaps;
        for (size_t i = 0; i < n; ++i) {
          const bool bit = nd.bits[i], g = gaps[i];
          if (bit) rightGaps.push(prev != bit || g);
          if (prev || g) { if (g) v.gap(bit, true); else v.pm(bit); }
          prev = bit;
        }
        queue.push_back(nd.right);
        integerNodes.push_back(nd.left);
      } else if (!rightSym) {
        PackedBits& leftGaps = nodes[nd.left].gaps;
        PackedBits& rightGaps = nodes[nd.right].gaps;
        for (size_t i = 0; i < n; ++i) {
          const bool bit = nd.bits[i], g = gaps[i];
          if (g) v.gap(bit, true); else v.pm(bit);
          (bit ? rightGaps : leftGaps).push(prev != bit || g);
          prev = bit;
        }
        queue.push_back(nd.left);
        queue.push_back(nd.right);
      }
      nd.gaps.release();
    }
    for (size_t i = 0; i < nodes.size(); ++i) nodes[i].gaps.release();   // symbol leaves' unused vectors
    std::deque<int> lefts, rights;
    while (!integerNodes.empty() || !lefts.empty() || !rights.empty()) {
      integerNodes.insert(integerNodes.end(), lefts.begin(), lefts.end());
      integerNodes.insert(integerNodes.end(), rights.begin(), rights.end());
      lefts.clear(); rights.clear();
      v.resetIntegerLevel();
      while (!integerNodes.empty()) {
        const Node& nd = nodes[integerNodes.front()];
        integerNodes.pop_front();
        for (size_t i = 0, n = nd.bits.size(); i < n; ++i) v.integer(nd.bits[i]);
        if (nd.left >= 0 && (!nodes[nd.left].hasSymbol || nodes[nd.left].symbol == 0)) lefts.push_back(nd.left);
        if (nd.right >= 0 && (!nodes[nd.right].hasSymbol || nodes[nd.right].symbol == 0)) rights.push_back(nd.right);
      }
    }
  }
};

// ---- models (probmodels/) and the range coder (BitCoders.cpp) --------------------------------
inline uint32_t nextState(uint32_t states, uint32_t cur, bool bit) {  // FSM.hpp:42-67
  if (states == 3) {
    if (cur == 1) return bit ? 2 : 0;
    if (cur == 2 && bit) return 2;
    if (cur == 0 && !bit) return 0;
    return 1;
  }
  if (bit) return cur >= states / 2 ? std::min(cur + 1, states - 1) : states / 2;
  return cur < states / 2 ? (cur ? cur - 1 : 0) : (states - 1) / 2;
}

// The three models of the 'B' coder (probmodels/ProbabilityModel.cpp:38-75) and the range
// coder (BitEncoder, BitCoders.cpp:59-113) in one flat state: 15 predictors
// (UnbiasedPredictor, BitPredictors.hpp:37-65: p moves 1/2^delay of the way to its bound)
// selected by three small state machines (FSM.hpp:42-67).  Written without data-dependent
// branches, the coded bits are close to random.
//   main     FSM8: 8 predictors, delay 4 for the two outer states, 5 otherwise; the upper four
//            predict the inverted bit (InversePredictor); reset keeps the state (FSM.hpp:196-205)
//   gaps     FSM<4>, delay 5;  integers  FSM<3>, delay 5, probabilities kept within [100, 3996]
// FSM.hpp:42-67 as tables: [state][bit] -> state
const uint8_t kNext8[8][2] = {{0, 4}, {0, 4}, {1, 4}, {2, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 7}};
const uint8_t kNext4[4][2] = {{0, 2}, {0, 2}, {1, 3}, {1, 3}};
const uint8_t kNext3[3][2] = {{0, 1}, {0, 2}, {1, 2}};

// FSM<6, ...> of coder 'b' (generic nextState<6>, FSM.hpp:42-54)
const uint8_t kNext6[6][2] = {{0, 3}, {0, 3}, {1, 3}, {2, 4}, {2, 5}, {2, 5}};

// EvenIntervalPredictor<4> (BitPredictors.hpp:95-125): steps of 1024, never 0 or 4096
inline uint32_t evenMoved(uint32_t q, uint32_t bit) {
  const uint32_t up = q + 1024u < 4096u ? q + 1024u : q, down = q > 1024u ? q - 1024u : q;
  return bit ? up : down;
}

struct Coder {
  enum { kMain = 0, kGaps = 8, kInts = 12 };
  char model;                                                          // 'B', 'b' or 'u': the main model
  uint16_t p[16];
  uint8_t next8[8][2], next4[4][2], next3[3][2];
  uint32_t mcur, gcur, icur;
  uint32_t low, high;
  std::vector<uint8_t>* out;
  size_t used;                                                         // bytes of *out that are final

  Coder(char model_, uint32_t startState, std::vector<uint8_t>* o)
      : model(model_), mcur(startState), gcur(2), icur(1), low(0), high(0xFFFFFFFFu), out(o), used(o->size()), w(0) {
    for (uint32_t c = 0; c < 8; ++c) for (uint32_t b = 0; b < 2; ++b) next8[c][b] = static_cast<uint8_t>(nextState(8, c, b));
    for (uint32_t c = 0; c < 4; ++c) for (uint32_t b = 0; b < 2; ++b) next4[c][b] = static_cast<uint8_t>(nextState(4, c, b));
    for (uint32_t c = 0; c < 3; ++c) for (uint32_t b = 0; b < 2; ++b) next3[c][b] = static_cast<uint8_t>(nextState(3, c, b));
    resetMain(); resetGaps(); resetInts();
  }
  void resetMain() {                                                   // resetModel of the main model
    static const uint16_t init[8] = {2400, 2300, 2200, 2100, 2100, 2200, 2300, 2400};
    if (model == 'B') { for (int i = 0; i < 8; ++i) p[kMain + i] = init[i]; return; }   // FSM8 keeps its state
    for (int i = 0; i < 8; ++i) p[kMain + i] = 2048;
    mcur = model == 'b' ? 3 : 0;
  }
  void resetGaps() { for (int i = 0; i < 4; ++i) p[kGaps + i] = 2048; gcur = 2; }
  void resetInts() { for (int i = 0; i < 3; ++i) p[kInts + i] = 2048; icur = 1; }

  // room for `elements` more coded bits (at most four bytes leave the coder per bit)
  void ensure(size_t elements) { if (out->size() - used < 4 * elements + 8) out->resize(used + 4 * elements + 8 + out->size() / 2); }

  uint8_t* w;                                                          // write position (hot loops only)
  inline void encode(uint32_t bit, uint32_t p1) {
    const uint32_t size = high - low - 1;
    const uint32_t split = low + (size >> 12) * p1 + (((size & 4095u) * p1 + 2048u) >> 12);
    high = bit ? split : high;
    low = bit ? low : split + 1;
    while (((low ^ high) & 0xFF000000u) == 0) { *w++ = static_cast<uint8_t>(low >> 24); low <<= 8; high = (high << 8) + 255; }
  }
  static inline uint16_t moved(uint32_t q, uint32_t toward_one, uint32_t lo, uint32_t delay) {
    const uint32_t up = q + (((4096u - lo) - q) >> delay), down = q - ((q - lo) >> delay);
    return static_cast<uint16_t>(toward_one ? up : down);
  }
  inline void mainState(uint32_t bit) {
    if (model == 'B') mcur = next8[mcur][bit];
    else if (model == 'b') mcur = kNext6[mcur][bit];
  }
  inline void pmFast(uint32_t bit) {
    const uint32_t q = p[kMain + mcur];
    if (model == 'B') {
      const uint32_t inv = mcur >> 2;
      encode(bit, inv ? 4096u - q : q);
      p[kMain + mcur] = moved(q, bit ^ inv, 2, (mcur == 0 || mcur == 7) ? 4 : 5);
    } else {
      encode(bit, q);
      p[kMain + mcur] = static_cast<uint16_t>(evenMoved(q, bit));
    }
    mainState(bit);
  }
  inline void gapFast(uint32_t bit, bool state) {
    const uint32_t q = p[kGaps + gcur];
    encode(bit, q);
    p[kGaps + gcur] = moved(q, bit, 2, 5);
    gcur = next4[gcur][bit];
    if (state) mainState(bit);
  }
  inline void integerFast(uint32_t bit) {
    const uint32_t q = p[kInts + icur];
    encode(bit, q);
    p[kInts + icur] = moved(q, bit, 100, 5);
    icur = next3[icur][bit];
  }
  // visitor interface of Tree::walk
  void pm(bool bit) { ensure(1); w = out->data() + used; pmFast(bit); used = static_cast<size_t>(w - out->data()); }
  void gap(bool bit, bool state) { ensure(1); w = out->data() + used; gapFast(bit, state); used = static_cast<size_t>(w - out->data()); }
  void integer(bool bit) { ensure(1); w = out->data() + used; integerFast(bit); used = static_cast<size_t>(w - out->data()); }
  void resetInternal() { resetMain(); resetGaps(); }
  void resetIntegerLevel() { resetInts(); }
  void finish() {                                                      // BitEncoder::finish, :100-113
    ensure(1);
    uint8_t* o = out->data();
    o[used++] = static_cast<uint8_t>(low >> 24);
    o[used++] = 255; o[used++] = 255; o[used++] = 255;
    out->resize(used);
    low = 0; high = 0xFFFFFFFFu;
  }
};

// visitor 1: only the main model's state, as a function of the (unknown) starting state.  The
// eight images collapse to one value after the first change of bit value, from then on a
// single state is tracked.
struct StateTracker {
  uint32_t to[8];
  bool uniform;
  StateTracker() : uniform(false) { for (uint32_t i = 0; i < 8; ++i) to[i] = i; }
  void step(bool bit) {
    if (uniform) { to[0] = nextState(8, to[0], bit); return; }
    bool same = true;
    for (uint32_t i = 0; i < 8; ++i) { to[i] = nextState(8, to[i], bit); same = same && to[i] == to[0]; }
    uniform = same;
  }
  uint32_t image(uint32_t start) const { return uniform ? to[0] : to[start]; }
  void pm(bool bit) { step(bit); }
  void gap(bool bit, bool state) { if (state) step(bit); }
  void integer(bool) {}
  void resetInternal() {}
  void resetIntegerLevel() {}
};

// runs f(i) for every i on a small thread pool, heaviest items first
template <typename F>
void parallelFor(size_t count, unsigned threads, const uint64_t* weight, F f) {
  if (threads <= 1 || count <= 1) { for (size_t i = 0; i < count; ++i) f(i); return; }
  std::vector<std::thread> pool;
  std::vector<size_t> order(count);
  for (size_t i = 0; i < count; ++i) order[i] = i;
  if (weight) std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weight[a] > weight[b]; });
  size_t next = 0;
  std::mutex* m = new std::mutex();
  auto worker = [&]() {
    for (;;) {
      size_t i;
      { std::lock_guard<std::mutex> g(*m); if (next >= count) return; i = order[next++]; }
      f(i);
    }
  };
  for (unsigned t = 0; t < std::min<size_t>(threads, count); ++t) pool.push_back(std::thread(worker));
  for (size_t t = 0; t < pool.size(); ++t) pool[t].join();
  delete m;
}

}  // namespace

void encodeSections(const std::vector<SectionRuns>& sections, unsigned threads, uint32_t* fsm8_state,
                    std::vector<SectionOutput>* out, char model) {
  const bool debug = std::getenv("BWTC_HIP_DEBUG") != 0;
  const auto t0 = std::chrono::steady_clock::now();
  const size_t n = sections.size();
  std::vector<Tree> trees(n);
  std::vector<StateTracker> trans(n);
  out->assign(n, SectionOutput());
  // pass 1 (parallel): trees, headers, state transitions
  std::vector<uint64_t> weight(n);
  for (size_t s = 0; s < n; ++s) weight[s] = sections[s].n_runs;
  parallelFor(n, threads, weight.data(), [&](size_t s) {
    if (sections[s].n_runs == 0) return;
    trees[s].build(sections[s]);
    std::vector<uint8_t>& bytes = (*out)[s].bytes;
    utils::packInteger(trees[s].nodes[trees[s].root].bits.size(), bytes);   // bitsInRoot
    trees[s].shape(bytes);
    trees[s].walk(trans[s]);
  });
  const auto t1 = std::chrono::steady_clock::now();
  // the one carried value: the main model's state at the start of every section
  std::vector<uint32_t> start(n);
  uint32_t state = *fsm8_state;
  for (size_t s = 0; s < n; ++s) { start[s] = state; if (sections[s].n_runs) state = trans[s].image(state); }
  *fsm8_state = state;
  // pass 2 (parallel): range coding
  parallelFor(n, threads, weight.data(), [&](size_t s) {
    if (sections[s].n_runs == 0) return;
    Coder coder(model, model == 'B' ? start[s] : 0u, &(*out)[s].bytes);
    trees[s].walk(coder);
    coder.finish();                                                   // endContextBlock, WaveletCoders.cpp:62-68
    Tree().nodes.swap(trees[s].nodes);
  });
  if (debug) {
    const auto t2 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "wavelet host: build+state pass %.3f s, coding pass %.3f s, %u threads, %zu sections\n",
                 std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(),
                 threads, n);
  }
}


// ---- device-built streams -------------------------------------------------------------------
namespace {

inline uint32_t codeAt(const uint8_t* codes, uint64_t i) { return (codes[i >> 2] >> ((i & 3) * 2)) & 3u; }

// union of the length codes of one section as a binary trie
struct LengthTrie {
  struct N { int child[2]; bool terminal; uint32_t group; N() : terminal(false), group(0) { child[0] = child[1] = -1; } };
  std::vector<N> n;
  LengthTrie() : n(1) {}
  // false: the code is not prefix free against the ones already inserted
  bool insert(const Bits& code) {
    int nd = 0;
    for (size_t i = 0; i < code.size(); ++i) {
      if (n[nd].terminal) return false;
      int next = n[nd].child[code[i]];
      if (next < 0) { n.push_back(N()); next = static_cast<int>(n.size()) - 1; n[nd].child[code[i]] = next; }
      nd = next;
    }
    if (n[nd].child[0] >= 0 || n[nd].child[1] >= 0) return false;
    n[nd].terminal = true;
    return true;
  }
};


// Stage 1 of the stream coder: the probability every coded bit of one group is coded with.
// A group starts with fresh predictors (resetModel before every inner node and every integer
// level, a new encoder for the root; WaveletTree.hpp:651-653, :707, :792-797), so groups are
// independent once the main model's state `mc` at the group's start is known.
// The upper four predictors of the main model predict the inverted bit (InversePredictor):
// the probability of a one is 4096 - p and the predictor moves the other way.  Kept as
// p' = 4096 - p they follow the same rule as the others (4094 - p = p' - 2), so all fifteen
// are handled alike: probability of a one = q[slot], moved towards the coded bit.
// MODEL = the main model's letter: 'B' FSM8 of UnbiasedPredictors (slots 0..7), 'b' FSM<6> of
// EvenIntervalPredictor<4> (slots 0..5), 'u' one EvenIntervalPredictor<4> (slot 0).
template <int TYPE, char MODEL>
struct GroupModel {
  uint32_t q[16];
  uint32_t mc, gc, ic;
  explicit GroupModel(uint32_t mainState) : mc(mainState), gc(2), ic(1) {
    static const uint32_t init[8] = {2400, 2300, 2200, 2100, 4096 - 2100, 4096 - 2200, 4096 - 2300, 4096 - 2400};
    for (int k = 0; k < 8; ++k) q[k] = MODEL == 'B' ? init[k] : 2048u;
    for (int k = 8; k < 16; ++k) q[k] = 2048;
  }
  inline uint16_t step(uint32_t v) {
    static const uint8_t kDelay[16] = {4, 5, 5, 5, 5, 5, 5, 4, 5, 5, 5, 5, 5, 5, 5, 5};
    const uint32_t bit = v & 1u;
    uint32_t slot, floor = 2;
    if (TYPE == kRoot) slot = mc;
    else if (TYPE == kBothLeaves) slot = 8 + gc;
    else if (TYPE == kInteger) { slot = 12 + ic; floor = 100; }
    else slot = (v >> 1) ? 8 + gc : mc;
    const uint32_t pr = q[slot], delay = kDelay[slot];
    const uint32_t up = pr + (((4096u - floor) - pr) >> delay), down = pr - ((pr - floor) >> delay);
    uint32_t moved = bit ? up : down;
    if (MODEL != 'B' && (TYPE == kRoot || TYPE == kInner)) moved = slot < 8 ? evenMoved(pr, bit) : moved;
    q[slot] = moved;
    if (TYPE == kRoot) mc = nextMain(bit);
    else if (TYPE == kBothLeaves) gc = kNext4[gc][bit];
    else if (TYPE == kInteger) ic = kNext3[ic][bit];
    else { gc = (v >> 1) ? kNext4[gc][bit] : gc; mc = nextMain(bit); }
    return static_cast<uint16_t>(pr);
  }
  inline uint32_t nextMain(uint32_t bit) const {
    return MODEL == 'B' ? kNext8[mc][bit] : MODEL == 'b' ? kNext6[mc][bit] : 0u;
  }
};

template <int TYPE, char MODEL>
void modelGroupWith(const uint8_t* codes, uint64_t b, uint64_t e, uint32_t mc, uint16_t* prob) {
  GroupModel<TYPE, MODEL> m(mc);
  uint64_t i = b;
  for (; i < e && (i & 3); ++i) prob[i] = m.step(codeAt(codes, i));
  for (; i + 4 <= e; i += 4) {                                         // four elements per byte of codes
    const uint32_t four = codes[i >> 2];
    prob[i] = m.step(four & 3u);
    prob[i + 1] = m.step((four >> 2) & 3u);
    prob[i + 2] = m.step((four >> 4) & 3u);
    prob[i + 3] = m.step(four >> 6);
  }
  for (; i < e; ++i) prob[i] = m.step(codeAt(codes, i));
}

// Two groups of coder 'B' stepped alternately by one thread: each is a serial chain that leaves
// most of the core idle (used for a block's few huge groups, which do not go through the lanes).
template <int TA, int TB>
void modelTwo(const uint8_t* codes, uint64_t ia, uint64_t ea, uint32_t mca, uint64_t ib, uint64_t eb, uint32_t mcb, uint16_t* prob) {
  GroupModel<TA, 'B'> A(mca);
  GroupModel<TB, 'B'> B(mcb);
  for (uint64_t n = std::min(ea - ia, eb - ib); n > 0; --n) {
    prob[ia] = A.step(codeAt(codes, ia)); ++ia;
    prob[ib] = B.step(codeAt(codes, ib)); ++ib;
  }
  for (; ia < ea; ++ia) prob[ia] = A.step(codeAt(codes, ia));
  for (; ib < eb; ++ib) prob[ib] = B.step(codeAt(codes, ib));
}

template <int TA>
void modelTwoB(int tb, const uint8_t* codes, uint64_t ia, uint64_t ea, uint32_t mca, uint64_t ib, uint64_t eb, uint32_t mcb, uint16_t* prob) {
  switch (tb) {
    case kRoot: modelTwo<TA, kRoot>(codes, ia, ea, mca, ib, eb, mcb, prob); break;
    case kBothLeaves: modelTwo<TA, kBothLeaves>(codes, ia, ea, mca, ib, eb, mcb, prob); break;
    case kInteger: modelTwo<TA, kInteger>(codes, ia, ea, mca, ib, eb, mcb, prob); break;
    default: modelTwo<TA, kInner>(codes, ia, ea, mca, ib, eb, mcb, prob); break;
  }
}

template <int TYPE>
void modelGroup(char model, const uint8_t* codes, uint64_t b, uint64_t e, uint32_t mc, uint16_t* prob) {
  if (model == 'b') modelGroupWith<TYPE, 'b'>(codes, b, e, mc, prob);
  else if (model == 'u') modelGroupWith<TYPE, 'u'>(codes, b, e, mc, prob);
  else modelGroupWith<TYPE, 'B'>(codes, b, e, mc, prob);
}

}  // namespace

namespace {

// One section's share of the plan with group numbers and pool offsets local to the section;
// sections are planned in parallel and then laid end to end.
struct SectionPlan {
  bool ok;
  std::vector<uint8_t> prefix;
  uint32_t n_nodes, n_groups, n_leaves;
  std::vector<uint32_t> level_first;
  std::vector<uint8_t> types;
  std::vector<uint32_t> pool;
  uint32_t symtab[512];
  uint64_t symcode[256];
  struct Length { uint32_t len, off, steps; };
  std::vector<Length> lengths;
  uint64_t max_elements;
  SectionPlan() : ok(true), n_nodes(0), n_groups(0), n_leaves(0), max_elements(0) {
    std::memset(symtab, 0, sizeof symtab);
    std::memset(symcode, 0, sizeof symcode);
  }
};

void planSection(const SectionRuns& in, SectionPlan* sec) {
  if (in.n_runs == 0) { sec->level_first.assign(1, 0); return; }
  Tree tree;
  tree.buildCodes(in);
  utils::packInteger(in.n_runs, sec->prefix);                        // bitsInRoot: one bit per run
  tree.shape(sec->prefix);

  // the nodes encodeTreeBF visits, in its order, and the order the symbol leaves are met in
  std::vector<int> groupOf(tree.nodes.size(), -1);
  std::vector<int> leafRank(tree.nodes.size(), -1);
  std::vector<uint8_t>& types = sec->types;
  int leaves = 0;
  {
    std::deque<int> queue;
    const Node& r = tree.nodes[tree.root];
    groupOf[tree.root] = 0; types.push_back(kRoot);
    if (r.left >= 0) { if (tree.nodes[r.left].hasSymbol) leafRank[r.left] = leaves++; else queue.push_back(r.left); }
    if (r.right >= 0) { if (tree.nodes[r.right].hasSymbol) leafRank[r.right] = leaves++; else queue.push_back(r.right); }
    while (!queue.empty()) {
      const int id = queue.front();
      queue.pop_front();
      const Node& nd = tree.nodes[id];
      if (nd.left < 0 || nd.right < 0) { sec->ok = false; return; }   // the reference would read out of bounds
      const bool leftSym = tree.nodes[nd.left].hasSymbol, rightSym = tree.nodes[nd.right].hasSymbol;
      if (leftSym && rightSym) {
        groupOf[id] = static_cast<int>(types.size()); types.push_back(kBothLeaves);
        leafRank[nd.left] = leaves++; leafRank[nd.right] = leaves++;
      } else if (leftSym) {
        groupOf[id] = static_cast<int>(types.size()); types.push_back(kLeftLeaf);
        queue.push_back(nd.right);
        leafRank[nd.left] = leaves++;
      } else if (!rightSym) {
        groupOf[id] = static_cast<int>(types.size()); types.push_back(kInner);
        queue.push_back(nd.left); queue.push_back(nd.right);
      }                                                              // right leaf only: dropped with its subtree (sic)
    }
  }
  if (leaves > 256) { sec->ok = false; return; }
  sec->n_leaves = static_cast<uint32_t>(leaves);
  sec->n_nodes = static_cast<uint32_t>(types.size());

  // the integer levels: union trie of the section's length codes
  LengthTrie trie;
  std::vector<Bits> lengthCodes(in.n_dist);
  for (size_t i = 0; i < in.n_dist; ++i) {
    lengthCodes[i] = tree.lengthCodeOf(in.dist[i].first);
    if (lengthCodes[i].empty() || !trie.insert(lengthCodes[i])) { sec->ok = false; return; }
  }
  uint32_t nextGroup = sec->n_nodes;
  {
    std::vector<int> level(1, 0), lefts, rights;                     // next level = all lefts, then all rights
    while (!level.empty()) {
      sec->level_first.push_back(nextGroup);
      lefts.clear(); rights.clear();
      for (size_t i = 0; i < level.size(); ++i) {
        LengthTrie::N& t = trie.n[level[i]];
        t.group = nextGroup++;
        if (t.child[0] >= 0 && !trie.n[t.child[0]].terminal) lefts.push_back(t.child[0]);
        if (t.child[1] >= 0 && !trie.n[t.child[1]].terminal) rights.push_back(t.child[1]);
      }
      level = lefts;
      level.insert(level.end(), rights.begin(), rights.end());
    }
    sec->level_first.push_back(nextGroup);
  }
  sec->n_groups = nextGroup;
  types.resize(nextGroup, static_cast<uint8_t>(kInteger));

  // steps of every symbol: the visited nodes on its path
  for (uint32_t c = 0; c < 256; ++c) {
    if (tree.codes[c].empty()) continue;
    if (tree.codes[c].size() > 64) { sec->ok = false; return; }
    uint64_t packed = 0;
    for (size_t i = 0; i < tree.codes[c].size(); ++i) packed |= static_cast<uint64_t>(tree.codes[c][i]) << (63 - i);
    sec->symcode[c] = packed;
    const uint32_t off = static_cast<uint32_t>(sec->pool.size());
    int nd = tree.root;
    for (size_t i = 0; i < tree.codes[c].size(); ++i) {
      const uint32_t bit = tree.codes[c][i];
      if (groupOf[nd] >= 0) sec->pool.push_back((static_cast<uint32_t>(groupOf[nd]) << kStepGroupShift) | bit);
      nd = bit ? tree.nodes[nd].right : tree.nodes[nd].left;
    }
    const uint32_t steps = static_cast<uint32_t>(sec->pool.size()) - off;
    const bool live = leafRank[nd] >= 0;
    sec->symtab[c * 2] = off;
    sec->symtab[c * 2 + 1] = steps | (live ? static_cast<uint32_t>(leafRank[nd]) << 8 : 0u) | (live ? 1u << 16 : 0u);
    sec->max_elements += static_cast<uint64_t>(steps) * in.run_freqs[c];
  }
  // steps of every run length
  for (size_t i = 0; i < in.n_dist; ++i) {
    const uint32_t off = static_cast<uint32_t>(sec->pool.size());
    int nd = 0;
    for (size_t k = 0; k < lengthCodes[i].size(); ++k) {
      const uint32_t bit = lengthCodes[i][k];
      sec->pool.push_back((trie.n[nd].group << kStepGroupShift) | bit);
      nd = trie.n[nd].child[bit];
    }
    const SectionPlan::Length entry = {in.dist[i].first, off, static_cast<uint32_t>(lengthCodes[i].size())};
    sec->lengths.push_back(entry);                                   // dist is ascending by length
    sec->max_elements += static_cast<uint64_t>(entry.steps) * in.dist[i].second;
  }
}

}  // namespace

bool planStreams(const std::vector<SectionRuns>& sections, StreamPlan* plan) {
  const size_t nsec = sections.size();
  // the sections' codes and shapes are independent (and the semi-fixed-code search is the
  // expensive part): planned on a few threads, then laid end to end
  std::vector<SectionPlan> local(nsec);
  std::vector<uint64_t> weight(nsec);
  for (size_t s = 0; s < nsec; ++s) weight[s] = sections[s].n_dist;
  const unsigned hc = std::thread::hardware_concurrency();
  parallelFor(nsec, std::max(1u, std::min(8u, hc ? hc : 1u)), weight.data(), [&](size_t s) { planSection(sections[s], &local[s]); });

  // dense ids when they fit: one per symbol-tree node, one per (integer node, symbol leaf)
  uint64_t n_ids = 0;
  for (size_t s = 0; s < nsec; ++s)
    if (local[s].ok) n_ids += local[s].n_nodes + static_cast<uint64_t>(local[s].n_groups - local[s].n_nodes) * std::max<uint32_t>(1, local[s].n_leaves);
  // Opt-in (BWTC_HIP_DENSE_STEP_KEYS=1).  Measured on the 256 MiB text block: 66 000 ids are 17
  // bits, still three 8-bit passes like the 19 bits of the two-field layout, and the id -> group
  // look-ups cost the select kernel half a millisecond; it pays only below 65 536 ids.
  const bool dense = n_ids > 0 && n_ids <= kMaxDenseIds && std::getenv("BWTC_HIP_DENSE_STEP_KEYS") != nullptr;
  plan->id_group.clear();
  uint32_t id_next = 0;
  plan->sections.assign(nsec, StreamPlan::Section());
  plan->group_type.clear();
  plan->symtab.assign(nsec * 512, 0);
  plan->symcode.assign(nsec * 256, 0);
  plan->lendense.assign(nsec * 2 * static_cast<size_t>(kLenDense), 0);
  plan->over_first.assign(nsec + 1, 0);
  plan->over.clear();
  plan->pool.clear();
  plan->max_elements = 0;
  for (size_t s = 0; s < nsec; ++s) {
    SectionPlan& in = local[s];
    if (!in.ok) return false;
    StreamPlan::Section& sec = plan->sections[s];
    sec.group_base = static_cast<uint32_t>(plan->group_type.size());
    sec.n_nodes = in.n_nodes;
    sec.n_leaves = in.n_leaves;
    sec.prefix.swap(in.prefix);
    sec.level_first.swap(in.level_first);
    plan->over_first[s] = static_cast<uint32_t>(plan->over.size() / 4);
    if (static_cast<uint64_t>(sec.group_base) + in.n_groups >= kMaxGroups) return false;
    plan->group_type.insert(plan->group_type.end(), in.types.begin(), in.types.end());
    const uint32_t poolBase = static_cast<uint32_t>(plan->pool.size());
    const uint32_t groupBits = sec.group_base << kStepGroupShift;
    if (dense) {
      const uint32_t span = std::max<uint32_t>(1, in.n_leaves), id0 = id_next;
      for (uint32_t g = 0; g < in.n_groups; ++g) {
        const uint32_t width = g < in.n_nodes ? 1u : span;
        plan->id_group.insert(plan->id_group.end(), width, sec.group_base + g);
        id_next += width;
      }
      for (size_t i = 0; i < in.pool.size(); ++i) {
        const uint32_t g = in.pool[i] >> kStepGroupShift;
        const uint32_t id = id0 + (g < in.n_nodes ? g : in.n_nodes + (g - in.n_nodes) * span);
        plan->pool.push_back((id << kStepLeafShift) | (in.pool[i] & 1u));
      }
    } else {
      for (size_t i = 0; i < in.pool.size(); ++i) plan->pool.push_back(in.pool[i] + groupBits);
    }
    for (uint32_t c = 0; c < 256; ++c) {
      if (!(in.symtab[c * 2 + 1] & 255u)) continue;
      plan->symtab[(s * 256 + c) * 2] = in.symtab[c * 2] + poolBase;
      plan->symtab[(s * 256 + c) * 2 + 1] = in.symtab[c * 2 + 1];
      plan->symcode[s * 256 + c] = in.symcode[c];
    }
    for (size_t i = 0; i < in.lengths.size(); ++i) {
      const SectionPlan::Length& l = in.lengths[i];
      if (l.len < kLenDense) {
        plan->lendense[(s * kLenDense + l.len) * 2] = l.off + poolBase;
        plan->lendense[(s * kLenDense + l.len) * 2 + 1] = l.steps;
      } else {
        plan->over.push_back(l.len); plan->over.push_back(l.off + poolBase); plan->over.push_back(l.steps); plan->over.push_back(0);
      }
    }
    plan->max_elements += in.max_elements;
  }
  plan->over_first[nsec] = static_cast<uint32_t>(plan->over.size() / 4);
  return true;
}

StreamCoder::StreamCoder(const StreamPlan& plan, const uint32_t* coded_pos, const uint8_t* codes, uint32_t fsm8_state, char model)
    : plan_(plan), pos_(coded_pos), codes_(codes), end_state_(fsm8_state), model_(model) {
  const size_t nsec = plan.sections.size();
  // The main model's state is the one value carried from section to section.  It is a
  // function of the last few bits that advanced it: two different bits in a row, or four
  // equal ones, fix the state whatever it was before.
  start_.assign(nsec, 0);
  std::vector<uint64_t> weight(nsec, 0);
  for (size_t s = 0; s < nsec; ++s) {
    const StreamPlan::Section& sec = plan.sections[s];
    start_[s] = end_state_;
    if (sec.level_first.size() < 2) continue;
    weight[s] = coded_pos[sec.group_base + sec.level_first.back()] - coded_pos[sec.group_base];
    if (weight[s]) end_state_ = stateBefore(s, sec.n_nodes);
  }
  for (size_t s = 0; s < nsec; ++s) {
    const StreamPlan::Section& sec = plan.sections[s];
    if (sec.level_first.size() < 2) continue;
    sections_.push_back(static_cast<uint32_t>(s));
    const uint32_t* pos = coded_pos + sec.group_base;
    for (uint32_t g = 0; g < sec.n_nodes; ++g) {
      const Task t = {static_cast<uint32_t>(s), g, pos[g], pos[g + 1], plan.group_type[sec.group_base + g]};
      if (t.end > t.begin) tasks_.push_back(t);
    }
    for (size_t l = 0; l + 1 < sec.level_first.size(); ++l) {
      const Task t = {static_cast<uint32_t>(s), sec.level_first[l], pos[sec.level_first[l]], pos[sec.level_first[l + 1]], kInteger};
      if (t.end > t.begin) tasks_.push_back(t);
    }
  }
  std::stable_sort(tasks_.begin(), tasks_.end(), [](const Task& x, const Task& y) { return x.end - x.begin > y.end - y.begin; });
  std::stable_sort(sections_.begin(), sections_.end(), [&](uint32_t x, uint32_t y) { return weight[x] > weight[y]; });
}

StreamCoder::StreamCoder(const StreamPlan& plan, const uint32_t* coded_pos, uint32_t end_state, Modelled)
    : plan_(plan), pos_(coded_pos), codes_(nullptr), end_state_(end_state), model_('B') {
  const size_t nsec = plan.sections.size();
  std::vector<uint64_t> weight(nsec, 0);
  for (size_t s = 0; s < nsec; ++s) {
    const StreamPlan::Section& sec = plan.sections[s];
    if (sec.level_first.size() < 2) continue;
    weight[s] = coded_pos[sec.group_base + sec.level_first.back()] - coded_pos[sec.group_base];
    sections_.push_back(static_cast<uint32_t>(s));
  }
  std::stable_sort(sections_.begin(), sections_.end(), [&](uint32_t x, uint32_t y) { return weight[x] > weight[y]; });
}

// state of the main model before group `group` of section s (group = n_nodes: after the section)
uint32_t StreamCoder::stateBefore(size_t s, uint32_t group) const {
  // 'b' and 'u' reset their state machine with the model: every group starts from the same state
  if (model_ == 'b') return 3;
  if (model_ == 'u') return 0;
  const StreamPlan::Section& sec = plan_.sections[s];
  uint8_t tail[8];
  int have = 0;
  for (uint32_t g = group; g-- > 0 && have < 8;) {
    if (plan_.group_type[sec.group_base + g] == kBothLeaves) continue;   // gaps that leave the main model alone
    const uint64_t b = pos_[sec.group_base + g], e = pos_[sec.group_base + g + 1];
    for (uint64_t i = e; i-- > b && have < 8;) tail[have++] = codeAt(codes_, i) & 1u;
  }
  StateTracker t;
  while (have > 0) t.step(tail[--have]);
  return t.image(start_[s]);
}

uint64_t StreamCoder::elements() const { return pos_[plan_.group_type.size()]; }

uint64_t StreamCoder::largestSectionElements() const {
  if (sections_.empty()) return 0;
  const StreamPlan::Section& sec = plan_.sections[sections_[0]];          // sorted largest first
  return pos_[sec.group_base + sec.level_first.back()] - pos_[sec.group_base];
}

uint64_t StreamCoder::sectionElements(size_t k) const {
  const StreamPlan::Section& sec = plan_.sections[sections_[k]];
  return pos_[sec.group_base + sec.level_first.back()] - pos_[sec.group_base];
}

void StreamCoder::model(size_t k, uint16_t* prob) const {
  const Task& t = tasks_[k];
  if (t.type == kInteger) { modelGroup<kInteger>(model_, codes_, t.begin, t.end, 0, prob); return; }
  if (t.type == kBothLeaves) { modelGroup<kBothLeaves>(model_, codes_, t.begin, t.end, 0, prob); return; }
  const uint32_t mc = stateBefore(t.section, t.group);
  if (t.type == kRoot) modelGroup<kRoot>(model_, codes_, t.begin, t.end, mc, prob);
  else modelGroup<kInner>(model_, codes_, t.begin, t.end, mc, prob);
}

void StreamCoder::modelPair(size_t k1, size_t k2, uint16_t* prob) const {
  if (model_ != 'B') { model(k1, prob); model(k2, prob); return; }
  const Task& a = tasks_[k1];
  const Task& b = tasks_[k2];
  const uint32_t mca = (a.type == kInteger || a.type == kBothLeaves) ? 0u : stateBefore(a.section, a.group);
  const uint32_t mcb = (b.type == kInteger || b.type == kBothLeaves) ? 0u : stateBefore(b.section, b.group);
  switch (a.type) {
    case kRoot: modelTwoB<kRoot>(b.type, codes_, a.begin, a.end, mca, b.begin, b.end, mcb, prob); break;
    case kBothLeaves: modelTwoB<kBothLeaves>(b.type, codes_, a.begin, a.end, mca, b.begin, b.end, mcb, prob); break;
    case kInteger: modelTwoB<kInteger>(b.type, codes_, a.begin, a.end, mca, b.begin, b.end, mcb, prob); break;
    default: modelTwoB<kInner>(b.type, codes_, a.begin, a.end, mca, b.begin, b.end, mcb, prob); break;
  }
}

void StreamCoder::describe(size_t k, uint16_t* prob, ModelGroupDesc* d) const {
  const Task& t = tasks_[k];
  d->codes = codes_;
  d->prob = prob;
  d->begin = t.begin;
  d->end = t.end;
  d->type = t.type;
  d->mainState = (t.type == kInteger || t.type == kBothLeaves) ? 0u : stateBefore(t.section, t.group);
  d->cookie = 0;
}

void StreamCoder::codeSection(size_t k, const uint16_t* prob, std::vector<SectionOutput>* out) const {
  CoderChain c;
  startSection(k, out, &c);
  runChain(c, codes_, prob, c.e);
  c.finish();                                                          // endContextBlock, WaveletCoders.cpp:62-68
}

void StreamCoder::describeChain(size_t k, const uint16_t* prob, std::vector<SectionOutput>* out, ChainDesc* d) const {
  const size_t s = sections_[k];
  const StreamPlan::Section& sec = plan_.sections[s];
  std::vector<uint8_t>& bytes = (*out)[s].bytes;
  bytes = sec.prefix;
  const uint32_t* pos = pos_ + sec.group_base;
  d->codes = codes_;
  d->prob = prob;
  d->begin = pos[0];
  d->end = pos[sec.level_first.back()];
  d->out = &bytes;
  d->cookie = 0;
}

void StreamCoder::describeSectionFused(size_t k, std::vector<SectionOutput>* out, std::vector<FusedGroup>* groups,
                                       FusedSectionDesc* d) const {
  const size_t s = sections_[k];
  const StreamPlan::Section& sec = plan_.sections[s];
  std::vector<uint8_t>& bytes = (*out)[s].bytes;
  bytes = sec.prefix;
  const uint32_t* pos = pos_ + sec.group_base;
  groups->clear();
  // the same groups, in the same order, as the model tasks of the constructor: the symbol tree's
  // nodes, then the integer levels (one model reset per level, WaveletTree.hpp:792-797)
  for (uint32_t g = 0; g < sec.n_nodes; ++g) {
    if (pos[g + 1] <= pos[g]) continue;
    FusedGroup fg;
    fg.begin = pos[g]; fg.end = pos[g + 1];
    fg.type = plan_.group_type[sec.group_base + g];
    fg.mainState = (fg.type == kInteger || fg.type == kBothLeaves) ? 0 : static_cast<uint8_t>(stateBefore(s, g));
    groups->push_back(fg);
  }
  for (size_t l = 0; l + 1 < sec.level_first.size(); ++l) {
    if (pos[sec.level_first[l + 1]] <= pos[sec.level_first[l]]) continue;
    FusedGroup fg;
    fg.begin = pos[sec.level_first[l]]; fg.end = pos[sec.level_first[l + 1]];
    fg.type = kInteger; fg.mainState = 0;
    groups->push_back(fg);
  }
  d->codes = codes_;
  d->groups = groups->data();
  d->n_groups = groups->size();
  d->out = &bytes;
  d->cookie = 0;
}

void StreamCoder::startSection(size_t k, std::vector<SectionOutput>* out, void* chain) const {
  const size_t s = sections_[k];
  const StreamPlan::Section& sec = plan_.sections[s];
  std::vector<uint8_t>& bytes = (*out)[s].bytes;
  bytes = sec.prefix;
  const uint32_t* pos = pos_ + sec.group_base;
  static_cast<CoderChain*>(chain)->start(pos[0], pos[sec.level_first.back()], &bytes);
}

// Sections taken from `cursor` (largest first) two at a time; when nothing is left to pair a
// chain with, it finishes on its own.
size_t StreamCoder::codeSectionsPaired(std::atomic<size_t>* cursor, const uint16_t* prob, std::vector<SectionOutput>* out) const {
  CoderChain lane[2];
  bool busy[2] = {false, false};
  size_t finished = 0;
  for (;;) {
    for (int l = 0; l < 2; ++l) {
      if (busy[l]) continue;
      const size_t k = cursor->fetch_add(1);
      if (k >= sections_.size()) continue;
      startSection(k, out, &lane[l]);
      busy[l] = true;
    }
    if (!busy[0] && !busy[1]) return finished;
    if (busy[0] && busy[1]) runChainPair(lane[0], lane[1], codes_, prob);
    else { CoderChain& c = busy[0] ? lane[0] : lane[1]; runChain(c, codes_, prob, c.e); }
    for (int l = 0; l < 2; ++l) {
      if (busy[l] && lane[l].i >= lane[l].e) { lane[l].finish(); busy[l] = false; ++finished; }
    }
  }
}

void StreamCoder::codeSectionW(size_t k, const uint16_t* w, std::vector<SectionOutput>* out) const {
  CoderChain c;
  startSection(k, out, &c);
  runChainW(c, w, c.e);
  c.finish();
}

void StreamCoder::describeChainW(size_t k, const uint16_t* w, std::vector<SectionOutput>* out, ChainDesc* d) const {
  describeChain(k, nullptr, out, d);
  d->codes = nullptr;
  d->w = w;
}

size_t StreamCoder::codeSectionsW(std::atomic<size_t>* cursor, const uint16_t* w, std::vector<SectionOutput>* out) const {
  size_t finished = 0;
  for (;;) {
    const size_t k = cursor->fetch_add(1);
    if (k >= sections_.size()) return finished;
    CoderChain c;
    startSection(k, out, &c);
    runChainW(c, w, c.e);
    c.finish();
    ++finished;
  }
}

void codeStreams(const StreamPlan& plan, const uint32_t* coded_pos, const uint8_t* codes, unsigned threads,
                 uint32_t* fsm8_state, std::vector<SectionOutput>* out, std::vector<uint16_t>* scratch, char model) {
  out->assign(plan.sections.size(), SectionOutput());
  StreamCoder coder(plan, coded_pos, codes, *fsm8_state, model);
  *fsm8_state = coder.endState();
  const uint64_t total = coder.elements();
  std::vector<uint16_t> local;
  std::vector<uint16_t>& prob = scratch ? *scratch : local;
  if (prob.size() < total) { std::vector<uint16_t>().swap(prob); prob.resize(total + total / 8); }
  const bool debug = std::getenv("BWTC_HIP_DEBUG") != 0;
  const auto t0 = std::chrono::steady_clock::now();
  if (model == 'B' && simdModelsAvailable()) {
    // sixteen groups per thread at a time (wavelet_simd.hpp); the engines share one cursor
    struct Source : GroupSource {
      const StreamCoder* coder; uint16_t* prob; std::mutex mu; size_t next_;
      bool next(ModelGroupDesc* d, int) {
        size_t k;
        { std::lock_guard<std::mutex> g(mu); if (next_ >= coder->modelTasks()) return false; k = next_++; }
        coder->describe(k, prob, d);
        return true;
      }
      void done(void*) {}
    } source;
    source.coder = &coder; source.prob = prob.data(); source.next_ = 0;
    const unsigned engines = std::max(1u, std::min<unsigned>(threads, static_cast<unsigned>(coder.modelTasks() / 16 + 1)));
    parallelFor(engines, engines, 0, [&](size_t) { runModelLanes(source); });
  } else {
    parallelFor(coder.modelTasks(), threads, 0, [&](size_t k) { coder.model(k, prob.data()); });
  }
  const auto t1 = std::chrono::steady_clock::now();
  {
    std::atomic<size_t> cursor(0);
    if (simdCoderAvailable() && coder.sectionTasks() >= 8) {
      // sixteen sections per thread at a time (wavelet_rc.hpp); the engines share one cursor
      struct Chains : ChainSource {
        const StreamCoder* coder; const uint16_t* prob; std::vector<SectionOutput>* out; std::atomic<size_t>* cursor;
        bool next(ChainDesc* d) {
          const size_t k = cursor->fetch_add(1);
          if (k >= coder->sectionTasks()) return false;
          coder->describeChain(k, prob, out, d);
          return true;
        }
        void done(void*) {}
      } chains;
      chains.coder = &coder; chains.prob = prob.data(); chains.out = out; chains.cursor = &cursor;
      const unsigned engines = std::max(1u, std::min<unsigned>(threads, static_cast<unsigned>(coder.sectionTasks() / 16 + 1)));
      parallelFor(engines, engines, 0, [&](size_t) { runCoderLanes(chains); });
    } else {
      const unsigned engines = std::max(1u, std::min<unsigned>(threads, static_cast<unsigned>((coder.sectionTasks() + 1) / 2)));
      parallelFor(engines, engines, 0, [&](size_t) { coder.codeSectionsPaired(&cursor, prob.data(), out); });
    }
  }
  if (debug) {
    const auto t2 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "wavelet coder: %llu elements; %zu groups modelled in %.3f s, %zu sections range coded in %.3f s\n",
                 (unsigned long long)total, coder.modelTasks(), std::chrono::duration<double>(t1 - t0).count(),
                 coder.sectionTasks(), std::chrono::duration<double>(t2 - t1).count());
  }
}

bool expandStreamsOnHost(const StreamPlan& plan, const std::vector<SectionRuns>& sections,
                         std::vector<uint32_t>* coded_pos, std::vector<uint8_t>* codes) {
  const size_t nsec = sections.size();
  const uint32_t groupMask = kMaxGroups - 1;
  std::vector<uint32_t> key;                                         // expand
  std::vector<uint32_t> runOff, runSymSteps;
  for (size_t s = 0; s < nsec; ++s) {
    for (uint64_t r = 0; r < sections[s].n_runs; ++r) {
      const uint32_t c = sections[s].symbols[r], len = sections[s].starts[r + 1] - sections[s].starts[r];
      const uint32_t off = plan.symtab[(s * 256 + c) * 2], meta = plan.symtab[(s * 256 + c) * 2 + 1];
      // gap flag of a symbol step: deeper than the common prefix with the previous run's code
      uint32_t common = 0;
      if (r > 0) common = static_cast<uint32_t>(__builtin_clzll(plan.symcode[s * 256 + c] ^ plan.symcode[s * 256 + sections[s].symbols[r - 1]]));
      runOff.push_back(static_cast<uint32_t>(key.size()));
      runSymSteps.push_back(meta & 255u);
      for (uint32_t i = 0; i < (meta & 255u); ++i) key.push_back(plan.pool[off + i] | ((i > common ? 1u : 0u) << kStepGapShift));
      if (!(meta >> 16)) continue;
      uint32_t loff = 0, lsteps = 0;
      if (len < kLenDense) {
        loff = plan.lendense[(s * kLenDense + len) * 2]; lsteps = plan.lendense[(s * kLenDense + len) * 2 + 1];
      } else {
        for (uint32_t i = plan.over_first[s]; i < plan.over_first[s + 1]; ++i)
          if (plan.over[i * 4] == len) { loff = plan.over[i * 4 + 1]; lsteps = plan.over[i * 4 + 2]; }
      }
      if (lsteps == 0) return false;
      for (uint32_t i = 0; i < lsteps; ++i) key.push_back(plan.pool[loff + i] + (((meta >> 8) & 255u) << kStepLeafShift));
    }
  }
  const size_t n = key.size();
  std::vector<uint32_t> order(n);                                    // sort
  for (size_t i = 0; i < n; ++i) order[i] = static_cast<uint32_t>(i);
  auto sortKey = [&](uint32_t i) { return key[i] >> kStepLeafShift; };
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sortKey(a) < sortKey(b); });
  const size_t groups = plan.group_type.size();
  const bool dense = !plan.id_group.empty();
  auto groupOf = [&](size_t j) {
    return dense ? plan.id_group[key[order[j]] >> kStepLeafShift] : (key[order[j]] >> kStepGroupShift) & groupMask;
  };
  // The gap flags once more the way the reference derives them -- changed = the bit differs from
  // the previous bit of the node, OR-ed down each run -- as a check of the prefix rule above.
  {
    std::vector<uint8_t> flag(n, 0);
    for (size_t j = 0; j < n; ++j) {
      const uint32_t g = groupOf(j);
      if (plan.group_type[g] == kInteger) continue;
      const bool first = j == 0 || groupOf(j - 1) != g;
      flag[order[j]] = first || ((key[order[j]] ^ key[order[j - 1]]) & 1u);
    }
    for (size_t r = 0; r < runOff.size(); ++r) {
      uint8_t g = 0;
      for (uint32_t i = 0; i < runSymSteps[r]; ++i) {
        const size_t at = runOff[r] + i;
        if (g != ((key[at] >> kStepGapShift) & 1u)) return false;
        g |= flag[at];
      }
    }
  }
  std::vector<uint32_t> groupStart(groups + 1, 0xFFFFFFFFu);
  coded_pos->assign(groups + 1, 0);                                   // select + pack
  codes->clear();
  uint64_t coded = 0;
  for (size_t j = 0; j < n; ++j) {
    const uint32_t k = key[order[j]], g = groupOf(j), bit = k & 1u;
    const bool first = j == 0 || groupOf(j - 1) != g;
    if (first) { groupStart[g] = static_cast<uint32_t>(j); (*coded_pos)[g] = static_cast<uint32_t>(coded); }
    const uint8_t type = plan.group_type[g];
    uint32_t gap = 0;
    bool keep = true;
    if (type != kRoot && type != kInteger) {
      gap = (k >> kStepGapShift) & 1u;
      const uint32_t prev = first ? (bit ^ 1u) : (key[order[j - 1]] & 1u);
      if (type == kBothLeaves) keep = gap != 0;
      else if (type == kLeftLeaf) keep = (prev | gap) != 0;
    }
    if (!keep) continue;
    if ((coded & 3) == 0) codes->push_back(0);
    codes->back() |= static_cast<uint8_t>((bit | (gap << 1)) << ((coded & 3) * 2));
    ++coded;
  }
  (*coded_pos)[groups] = static_cast<uint32_t>(coded);
  for (size_t g = groups; g-- > 0;) if (groupStart[g] == 0xFFFFFFFFu) (*coded_pos)[g] = (*coded_pos)[g + 1];
  return true;
}

}  // namespace wavelet
}  // namespace bwtc
