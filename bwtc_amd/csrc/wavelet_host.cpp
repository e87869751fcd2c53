// Host half of the wavelet coder 'B'.  See wavelet_host.hpp.  Reference quirks that decide
// output bytes are kept on purpose and marked "sic".
#include "wavelet_host.hpp"

#include <algorithm>
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

  // WaveletTree(const byte*, size_t), :294-360, from the GPU's run statistics
  void build(const SectionRuns& in) {
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
      std::map<uint32_t, Bits>::const_iterator it = integerCodes.find(len);
      if (it == integerCodes.end()) {
        lengthCode = integerCodes[0];
        fixedIntegerCode(lengthCode, len, W);
      } else {
        lengthCode = it->second;
      }
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

struct Predictor {                                                    // UnbiasedPredictor, BitPredictors.hpp:37-65
  uint16_t p, lo, initial; unsigned delay;
  void set(uint16_t lo_, unsigned delay_, uint16_t initial_) { lo = lo_; delay = delay_; initial = initial_; p = initial_; }
  void update(bool bit) {
    const uint16_t top = static_cast<uint16_t>(4096 - lo);
    if (bit) p = static_cast<uint16_t>(p + ((top - p) >> delay));
    else p = static_cast<uint16_t>(p - ((p - lo) >> delay));
  }
};

struct Fsm {                                                          // FSM<N,...> and FSM8
  bool fsm8; uint32_t n, cur; Predictor st[8];
  void reset() { for (uint32_t i = 0; i < n; ++i) st[i].p = st[i].initial; if (!fsm8) cur = n / 2; }
  uint16_t probabilityOfOne() const { return (fsm8 && cur >= 4) ? static_cast<uint16_t>(4096 - st[cur].p) : st[cur].p; }
  void updateState(bool bit) { cur = nextState(n, cur, bit); }
  void update(bool bit) { st[cur].update((fsm8 && cur >= 4) ? !bit : bit); updateState(bit); }
};

struct RangeCoder {                                                   // BitEncoder, BitCoders.cpp:59-113
  uint32_t low, high; std::vector<uint8_t>* out;
  void encode(bool bit, uint16_t p1) {
    const uint32_t size = high - low - 1;
    const uint32_t split = low + (size >> 12) * p1 + (((size & 4095u) * p1 + 2048u) >> 12);
    if (bit) high = split; else low = split + 1;
    while (((low ^ high) & 0xFF000000u) == 0) { out->push_back(static_cast<uint8_t>(low >> 24)); low <<= 8; high = (high << 8) + 255; }
  }
  void finish() { out->push_back(static_cast<uint8_t>(low >> 24)); out->insert(out->end(), 3, 255); low = 0; high = 0xFFFFFFFFu; }
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

// visitor 2: the real coder
struct Coder {
  Fsm main, ints, gaps; RangeCoder rc;
  explicit Coder(uint32_t startState, std::vector<uint8_t>* out) {
    main.fsm8 = true; main.n = 8; main.cur = startState;              // ProbabilityModel.cpp:63-75
    main.st[0].set(2, 4, 2400); main.st[1].set(2, 5, 2300); main.st[2].set(2, 5, 2200); main.st[3].set(2, 5, 2100);
    main.st[4].set(2, 5, 2100); main.st[5].set(2, 5, 2200); main.st[6].set(2, 5, 2300); main.st[7].set(2, 4, 2400);
    ints.fsm8 = false; ints.n = 3; ints.cur = 1;                       // :38-41
    for (int i = 0; i < 3; ++i) ints.st[i].set(100, 5, 2048);
    gaps.fsm8 = false; gaps.n = 4; gaps.cur = 2;                       // :43-45
    for (int i = 0; i < 4; ++i) gaps.st[i].set(2, 5, 2048);
    rc.low = 0; rc.high = 0xFFFFFFFFu; rc.out = out;
  }
  void pm(bool bit) { rc.encode(bit, main.probabilityOfOne()); main.update(bit); }
  void gap(bool bit, bool state) { rc.encode(bit, gaps.probabilityOfOne()); gaps.update(bit); if (state) main.updateState(bit); }
  void integer(bool bit) { rc.encode(bit, ints.probabilityOfOne()); ints.update(bit); }
  void resetInternal() { main.reset(); gaps.reset(); }
  void resetIntegerLevel() { ints.reset(); }
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
                    std::vector<SectionOutput>* out) {
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
    Coder coder(start[s], &(*out)[s].bytes);
    trees[s].walk(coder);
    coder.rc.finish();                                                // endContextBlock, WaveletCoders.cpp:62-68
    Tree().nodes.swap(trees[s].nodes);
  });
  if (debug) {
    const auto t2 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "wavelet host: build+state pass %.3f s, coding pass %.3f s, %u threads, %zu sections\n",
                 std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(),
                 threads, n);
  }
}

}  // namespace wavelet
}  // namespace bwtc
