// Decoding side of the host mirror: HuffmanDecoder (HuffmanCoders.cpp:88-117,323-616),
// BWTBlock::readHeader (BWTBlock.cpp:88-102), Decompressor (Decompressor.cpp:45-94).  The
// entropy decoder is serial host code (a section's streams have no length fields, so
// nothing in a block can be decoded ahead); the inverse transform runs on the GPU
// (HipInverseBWTransform).  WaveletDecoder ('B'; WaveletCoders.cpp:232-291, WaveletTree.hpp
// readShape :403-500, decodeTreeBF :857-1174, message :1277-1378, BitDecoder BitCoders.cpp:115-148)
// is serial host code for the same reason plus the adaptive models.  Coders 'H', 'B', 'b', 'u'
// and streams without precompression are accepted.
#pragma once
#include <cstring>
#include <deque>

#include "bwtc_hip.hpp"

namespace bwtc {

// MSB-first bit reader over a byte range (InStream::readBit/readByte/flushBuffer semantics)
class MemoryBitReader {
 public:
  MemoryBitReader(const byte* p, size_t n) : m_p(p), m_n(n), m_pos(0), m_bit(0) {}
  bool atEnd() const { return m_pos >= m_n; }
  size_t position() const { return m_pos; }
  bool readBit() {
    if (m_pos >= m_n) fail("unexpected end of stream");
    const bool b = (m_p[m_pos] >> (7 - m_bit)) & 1;
    if (++m_bit == 8) { m_bit = 0; ++m_pos; }
    return b;
  }
  uint64 readBits(unsigned k) { uint64 v = 0; while (k--) v = (v << 1) | (readBit() ? 1u : 0u); return v; }
  byte readByte() { return (byte)readBits(8); }
  void flushBuffer() { if (m_bit) { m_bit = 0; ++m_pos; } }
  // whole bytes from here on (after flushBuffer): where they are, how many, and a step over some of them
  const byte* here() const { return m_p + (m_pos < m_n ? m_pos : m_n); }
  size_t left() const { return m_pos < m_n ? m_n - m_pos : 0; }
  void advance(size_t bytes) { m_pos += bytes; }
  // peek up to 16 bits without consuming (zero padded past the end)
  unsigned peek16() const {
    uint32_t w = 0;
    for (int i = 0; i < 3; ++i) w = (w << 8) | (m_pos + i < m_n ? m_p[m_pos + i] : 0);
    return (w >> (8 - m_bit)) & 0xFFFF;
  }
  void skip(unsigned k) { m_bit += k; m_pos += m_bit >> 3; m_bit &= 7; }
  uint64 readPackedInteger() {                      // utils::readPackedInteger, Utils.hpp:164-178
    uint64 result = 0, read = 0x80; unsigned shift = 0;
    while (read & 0x80) { read = readBits(8); result |= (read & 0x7f) << shift; shift += 7; }
    return result;
  }
  static void fail(const char* why) { std::fprintf(stderr, "bwtc-hip: corrupt stream: %s\n", why); std::exit(1); }
 private:
  const byte* m_p; size_t m_n, m_pos; unsigned m_bit;
};

namespace detail {
inline unsigned ceilLog2(size_t n) { unsigned l = 0; while ((size_t(1) << l) < n) ++l; return l; }
// utils::binaryDecode, Utils.hpp:299-317
inline size_t binaryDecode(MemoryBitReader& in, size_t lo, size_t hi) {
  const size_t range = hi - lo + 1;
  if (range == 1) return lo;
  const unsigned k = ceilLog2(range);
  const size_t n_short = (size_t(1) << k) - range, n_long_left = (range - n_short) / 2;
  size_t v = in.readBits(k - 1);
  if (v >= n_long_left) return v + lo;
  v = (v << 1) | (in.readBit() ? 1u : 0u);
  return v < n_long_left ? v + lo : v + lo + n_short;
}
// utils::binaryInterpolativeDecode, Utils.hpp:342-360
inline void interpolativeDecode(MemoryBitReader& in, std::vector<uint32>& list, size_t lo, size_t hi,
                                size_t elements) {
  if (elements == 0) return;
  if (elements == hi - lo + 1) { for (size_t v = lo; v <= hi; ++v) list.push_back((uint32)v); return; }
  const size_t h = (elements - 1) / 2, r = elements / 2 - h;
  const size_t mid = binaryDecode(in, lo + h, hi - h - r);
  interpolativeDecode(in, list, lo, mid - 1, h);
  list.push_back((uint32)mid);
  interpolativeDecode(in, list, mid + 1, hi, elements - h - 1);
}
}  // namespace detail

class EntropyDecoder {
 public:
  virtual ~EntropyDecoder() {}
  // decodes one BWT block into block.begin() (capacity must hold it) and sets its size
  virtual void decodeBlock(BWTBlock& block, MemoryBitReader& in, size_t capacity) = 0;
};

class HuffmanDecoder : public EntropyDecoder {
 public:
  void decodeBlock(BWTBlock& block, MemoryBitReader& in, size_t capacity) {
    const uint64 compressed = in.readBits(48);                       // read48bits
    const size_t start = in.position();
    const uint32 nLF = (uint32)in.readByte() + 1;                    // BWTBlock::readHeader
    block.LFpowers().resize(nLF);
    for (uint32 i = 0; i < nLF; ++i) block.LFpowers()[i] = (uint32)in.readBits(31);
    in.flushBuffer();
    size_t sections = in.readByte();
    if (sections == 0) sections = 256;
    std::vector<uint64> lengths(sections);
    uint64 total = 0;
    for (size_t i = 0; i < sections; ++i) { lengths[i] = in.readPackedInteger(); total += lengths[i]; }
    if (total > capacity) MemoryBitReader::fail("block larger than announced");
    byte* dst = block.begin();
    for (size_t s = 0; s < sections; ++s) {
      if (lengths[s] == 0) continue;
      const uint64 nRuns = in.readPackedInteger();
      // shape: HuffmanDecoder::deserializeShape, HuffmanCoders.cpp:88-117
      const size_t maxSym = in.readByte();
      size_t nsym = in.readByte(); if (nsym == 0) nsym = 256;
      const size_t maxLen = in.readPackedInteger();
      std::vector<uint32> alphabet;
      detail::interpolativeDecode(in, alphabet, 0, maxSym, nsym);
      uint8_t clen[256] = {0};
      for (size_t i = 0; i < nsym; ++i) {
        size_t n = 1; while (!in.readBit()) ++n;
        if (n > maxLen) MemoryBitReader::fail("code length");
        clen[alphabet[i]] = (uint8_t)(maxLen - n + 1);
      }
      in.flushBuffer();
      uint32_t code[256];
      bwtc_hip_host_huffman_codes(clen, code);
      // 12-bit first-level table: entry = (symbol << 8) | length, 0 = longer code
      std::vector<uint32_t> table(1u << 12, 0);
      for (unsigned c = 0; c < 256; ++c) {
        if (!clen[c] || clen[c] > 12) continue;
        const unsigned pad = 12 - clen[c];
        for (unsigned f = 0; f < (1u << pad); ++f) table[(code[c] << pad) | f] = (c << 8) | clen[c];
      }
      std::vector<byte> runseq(nRuns);
      for (uint64 k = 0; k < nRuns; ++k) {
        const uint32_t e = table[in.peek16() >> 4];
        if (e) { runseq[k] = (byte)(e >> 8); in.skip(e & 0xff); continue; }
        uint32_t v = 0; unsigned len = 0; int sym = -1;             // long code: bit by bit
        while (sym < 0) {
          v = (v << 1) | (in.readBit() ? 1u : 0u); ++len;
          if (len > 64) MemoryBitReader::fail("huffman code");
          for (unsigned c = 0; c < 256; ++c) if (clen[c] == len && code[c] == v) { sym = (int)c; break; }
        }
        runseq[k] = (byte)sym;
      }
      in.flushBuffer();
      byte* const section_end = dst + lengths[s];
      for (uint64 k = 0; k < nRuns; ++k) {                            // gamma codes, :590-603
        unsigned zeros = 0; while (!in.readBit()) ++zeros;
        const uint64 len = (uint64(1) << zeros) | in.readBits(zeros);
        if (len > (uint64)(section_end - dst)) MemoryBitReader::fail("run length");
        std::memset(dst, runseq[k], len);
        dst += len;
      }
      in.flushBuffer();
      if (dst != section_end) MemoryBitReader::fail("section length");
    }
    if (in.position() - start != compressed) MemoryBitReader::fail("block length field");
    block.setSize((uint32)total);
  }
};

// ---- 'B' ------------------------------------------------------------------------------------
namespace detail {

// the three models of coder 'B' (probmodels/ProbabilityModel.cpp:38-75) as one table of
// fifteen predictors: main FSM8 0..7, gaps FSM<4> 8..11, integers FSM<3> 12..14.  A predictor
// moves 1/2^delay of the way towards its bound (BitPredictors.hpp:37-65); the upper four of
// the main model predict the inverted bit and are kept as 4096 - p.
struct WaveletModels {
  char model;                              // coder letter: 'B' FSM8, 'b' FSM<6, EvenIntervalPredictor<4>>, 'u' EvenIntervalPredictor<4>
  uint32 q[15];
  uint32 mc, gc, ic;
  explicit WaveletModels(char model_ = 'B') : model(model_), mc(model_ == 'B' ? 4 : model_ == 'b' ? 3 : 0), gc(2), ic(1) {
    resetMain(); resetGaps(); resetInts();
  }
  void resetMain() {                                   // FSM8::resetModel keeps the state (FSM.hpp:196-205), FSM<N> does not (:94-97)
    static const uint32 init[8] = {2400, 2300, 2200, 2100, 4096 - 2100, 4096 - 2200, 4096 - 2300, 4096 - 2400};
    for (int i = 0; i < 8; ++i) q[i] = model == 'B' ? init[i] : 2048;
    if (model == 'b') mc = 3;
  }
  void resetGaps() { for (int i = 8; i < 12; ++i) q[i] = 2048; gc = 2; }
  void resetInts() { for (int i = 12; i < 15; ++i) q[i] = 2048; ic = 1; }
  static uint32 next(uint32 states, uint32 cur, bool bit) {          // FSM.hpp:42-67
    if (states == 3) {
      if (cur == 1) return bit ? 2 : 0;
      if (cur == 2 && bit) return 2;
      if (cur == 0 && !bit) return 0;
      return 1;
    }
    if (bit) return cur >= states / 2 ? std::min(cur + 1, states - 1) : states / 2;
    return cur < states / 2 ? (cur ? cur - 1 : 0) : (states - 1) / 2;
  }
  void move(uint32 slot, bool bit, uint32 floor, uint32 delay) {
    q[slot] = bit ? q[slot] + (((4096 - floor) - q[slot]) >> delay) : q[slot] - ((q[slot] - floor) >> delay);
  }
  void moveEven(uint32 slot, bool bit) {                              // EvenIntervalPredictor<4>, BitPredictors.hpp:95-125
    if (bit) { if (q[slot] + 1024 < 4096) q[slot] += 1024; }
    else if (q[slot] > 1024) q[slot] -= 1024;
  }
  uint32 mainP() const { return q[mc]; }
  void mainUpdate(bool bit) {
    if (model == 'B') move(mc, bit, 2, (mc == 0 || mc == 7) ? 4 : 5); else moveEven(mc, bit);
    mainState(bit);
  }
  void mainState(bool bit) { if (model == 'B') mc = next(8, mc, bit); else if (model == 'b') mc = next(6, mc, bit); }
  uint32 gapP() const { return q[8 + gc]; }
  void gapUpdate(bool bit) { move(8 + gc, bit, 2, 5); gc = next(4, gc, bit); }
  uint32 intP() const { return q[12 + ic]; }
  void intUpdate(bool bit) { move(12 + ic, bit, 100, 5); ic = next(3, ic, bit); }
};

// BitDecoder, BitCoders.cpp:115-148
struct RangeDecoder {
  uint32 low, high, next;
  MemoryBitReader* in;
  void start(MemoryBitReader* r) {
    in = r; low = 0; high = 0xFFFFFFFFu; next = 0;
    for (int i = 0; i < 4; ++i) next = (next << 8) + in->readByte();
  }
  bool decode(uint32 p1) {
    const uint32 size = high - low - 1;
    const uint32 split = low + (size >> 12) * p1 + (((size & 4095u) * p1 + 2048u) >> 12);
    const bool bit = next <= split;
    if (bit) high = split; else low = split + 1;
    while (((low ^ high) & 0xFF000000u) == 0) { low <<= 8; high = (high << 8) + 255; next = (next << 8) + in->readByte(); }
    return bit;
  }
};

struct BitQueue {                                       // bits of one tree node, read back in order
  std::vector<uint64> w; size_t n, at;
  BitQueue() : n(0), at(0) {}
  void push(bool b) { if ((n & 63) == 0) w.push_back(0); w.back() |= (uint64)b << (n & 63); ++n; }
  bool operator[](size_t i) const { return (w[i >> 6] >> (i & 63)) & 1; }
  bool take() { const bool b = (*this)[at]; ++at; return b; }
};

struct WtNode {
  int left, right;
  bool hasSymbol; uint32 symbol;
  BitQueue bits;
  WtNode() : left(-1), right(-1), hasSymbol(false), symbol(0) {}
};

typedef std::pair<uint64, uint32> LenSym;

// WaveletTree::assignPrefixCodes(lengths, node, elem, bits), WaveletTree.hpp:1446-1482
inline size_t assignPrefixCodes(std::vector<WtNode>& pool, const std::vector<LenSym>& lengths, int nd, size_t elem, size_t depth) {
  if (elem >= lengths.size()) return elem;
  if (depth == lengths[elem].first - 1) {
    WtNode leaf; leaf.hasSymbol = true; leaf.symbol = lengths[elem].second;
    pool.push_back(leaf);
    const int c = (int)pool.size() - 1;
    if (pool[nd].left < 0) { pool[nd].left = c; return assignPrefixCodes(pool, lengths, nd, elem + 1, depth); }
    pool[nd].right = c;
    return elem + 1;
  }
  if (pool[nd].left < 0) {
    pool.push_back(WtNode());
    const int c = (int)pool.size() - 1;
    pool[nd].left = c;
    elem = assignPrefixCodes(pool, lengths, c, elem, depth + 1);
  }
  if (elem < lengths.size()) {
    pool.push_back(WtNode());
    const int c = (int)pool.size() - 1;
    pool[nd].right = c;
    elem = assignPrefixCodes(pool, lengths, c, elem, depth + 1);
  }
  return elem;
}

inline uint64 readPackedIntegerRev(MemoryBitReader& in) {           // utils::readPackedIntegerRev: bytes LSB first
  uint64 result = 0; unsigned shift = 0; uint32 byte_;
  do {
    byte_ = 0;
    for (int i = 0; i < 8; ++i) byte_ |= (in.readBit() ? 1u : 0u) << i;
    result |= (uint64)(byte_ & 0x7F) << shift;
    shift += 7;
  } while (byte_ & 0x80);
  return result;
}

}  // namespace detail

class WaveletDecoder : public EntropyDecoder {
 public:
  explicit WaveletDecoder(char decoder = 'B') : m_models(decoder) {}
  void decodeBlock(BWTBlock& block, MemoryBitReader& in, size_t capacity) {
    const uint64 compressed = in.readBits(48);                       // readBlockHeader, WaveletCoders.cpp:232-244
    const size_t start = in.position();
    const uint32 nLF = (uint32)in.readByte() + 1;
    block.LFpowers().resize(nLF);
    for (uint32 i = 0; i < nLF; ++i) block.LFpowers()[i] = (uint32)in.readBits(31);
    in.flushBuffer();
    size_t sections = in.readByte();
    if (sections == 0) sections = 256;
    std::vector<uint64> lengths(sections);
    uint64 total = 0;
    for (size_t i = 0; i < sections; ++i) { lengths[i] = in.readPackedInteger(); total += lengths[i]; }
    if (total > capacity) MemoryBitReader::fail("block larger than announced");
    byte* dst = block.begin();
    for (size_t s = 0; s < sections; ++s) {                          // decodeBlock, :246-291
      if (lengths[s] == 0) continue;
      const size_t rootSize = (size_t)in.readPackedInteger();
      Tree tree;
      tree.readShape(in);
      in.flushBuffer();
      m_rc.start(&in);
      tree.decodeTreeBF(rootSize, m_rc, m_models);
      const size_t got = tree.message(dst, lengths[s]);
      if (got != lengths[s]) MemoryBitReader::fail("section length");
      dst += got;
      m_models.resetMain(); m_models.resetInts(); m_models.resetGaps();   // endContextBlock, :70-76
    }
    if (in.position() - start != compressed) MemoryBitReader::fail("block length field");
    block.setSize((uint32)total);
  }

 private:
  struct Tree {
    std::vector<detail::WtNode> nodes;      // [0] = root of the symbol tree; integer data nodes hang below its leaves
    std::vector<detail::WtNode> codeTree;   // prefix code of the run lengths; [0] = root
    uint32 W;
    bool plainFixed;                        // no length has a code of its own: every run is escape coded
    Tree() : nodes(1), codeTree(1), W(0), plainFixed(false) {}

    void readShape(MemoryBitReader& in) {                            // WaveletTree.hpp:403-500
      const size_t maxSym = in.readByte();
      size_t symbols = in.readByte();
      if (symbols == 0) symbols = 256;
      size_t maxLen = (size_t)detail::readPackedIntegerRev(in);
      std::vector<uint32> alphabet;
      detail::interpolativeDecode(in, alphabet, 0, maxSym, symbols);
      std::vector<detail::LenSym> lengths;
      for (size_t i = 0; i < symbols; ++i) {
        size_t n = 1; while (!in.readBit()) ++n;                     // utils::unaryDecode
        if (n > maxLen) MemoryBitReader::fail("symbol code length");
        lengths.push_back(detail::LenSym(maxLen - n + 1, alphabet[i]));
      }
      std::sort(lengths.begin(), lengths.end());                     // assignPrefixCodes(lengths), :1440-1444
      detail::assignPrefixCodes(nodes, lengths, 0, 0, 0);
      symbolNodes = (int)nodes.size();                               // what follows are run-length nodes
      const size_t longestRun = (size_t)detail::readPackedIntegerRev(in);
      if (longestRun > 0) {
        symbols = (size_t)detail::readPackedIntegerRev(in);
        maxLen = (size_t)detail::readPackedIntegerRev(in);
        std::vector<uint32> integers;
        detail::interpolativeDecode(in, integers, 0, longestRun, symbols);
        std::vector<detail::LenSym> integerLengths;
        for (size_t i = 0; i < symbols; ++i) {
          size_t n = 1; while (!in.readBit()) ++n;
          if (n > maxLen) MemoryBitReader::fail("integer code length");
          integerLengths.push_back(detail::LenSym(maxLen + 1 - n, integers[i]));
        }
        std::sort(integerLengths.begin(), integerLengths.end());
        detail::assignPrefixCodes(codeTree, integerLengths, 0, 0, 0);
      } else {
        codeTree[0].hasSymbol = true;                                // TreeNode(0): the escape code is the empty word
        codeTree[0].symbol = 0;
        plainFixed = true;
      }
      W = (uint32)in.readBits(4);
    }

    // a node of an integer level on its way through the level-synchronised decoding, :837-853
    struct IntegerNode {
      int node, code;                       // data node; node of the length code tree or -1 once in the fixed part
      size_t bits; uint32 leadingOnes; int status;
    };

    int child(int nd, bool right) {
      int& c = right ? nodes[nd].right : nodes[nd].left;
      if (c < 0) { nodes.push_back(detail::WtNode()); const int id = (int)nodes.size() - 1; (right ? nodes[nd].right : nodes[nd].left) = id; return id; }
      return c;
    }

    void decodeTreeBF(size_t rootSize, detail::RangeDecoder& dec, detail::WaveletModels& m) {   // :857-1174
      typedef std::pair<int, detail::BitQueue> Inner;                // node, its gap vector
      std::deque<Inner> queue;
      std::deque<IntegerNode> level;
      if (rootSize == 0) MemoryBitReader::fail("empty wavelet tree");
      {
        detail::BitQueue left, right;
        bool prev = dec.decode(m.mainP());
        m.mainUpdate(prev);
        nodes[0].bits.push(prev);
        (prev ? right : left).push(true);
        for (size_t i = 1; i < rootSize; ++i) {
          const bool bit = dec.decode(m.mainP());
          m.mainUpdate(bit);
          nodes[0].bits.push(bit);
          (bit ? right : left).push(prev != bit);
          prev = bit;
        }
        if (nodes[0].left < 0) MemoryBitReader::fail("tree without symbols");
        if (nodes[nodes[0].left].hasSymbol) level.push_back(IntegerNode{nodes[0].left, 0, left.n, 0, 0});
        else queue.push_back(Inner(nodes[0].left, left));
        if (right.n > 0) {
          if (nodes[0].right < 0) MemoryBitReader::fail("bits for a missing subtree");
          if (nodes[nodes[0].right].hasSymbol) level.push_back(IntegerNode{nodes[0].right, 0, right.n, 0, 0});
          else queue.push_back(Inner(nodes[0].right, right));
        }
      }
      while (!queue.empty()) {
        m.resetMain();
        m.resetGaps();
        const int id = queue.front().first;
        detail::BitQueue gaps;
        std::swap(gaps, queue.front().second);
        queue.pop_front();
        if (nodes[id].left < 0 || nodes[id].right < 0) MemoryBitReader::fail("inner node with one child");
        const int l = nodes[id].left, r = nodes[id].right;
        detail::BitQueue left, right;
        bool prev = true;
        if (nodes[l].hasSymbol && nodes[r].hasSymbol) {
          size_t ones = 0;
          for (size_t i = 0; i < gaps.n; ++i) {
            if (!gaps[i]) prev = !prev;
            else { prev = dec.decode(m.gapP()); m.gapUpdate(prev); }
            nodes[id].bits.push(prev);
            ones += prev;
          }
          level.push_back(IntegerNode{l, 0, gaps.n - ones, 0, 0});
          level.push_back(IntegerNode{r, 0, ones, 0, 0});
        } else if (nodes[l].hasSymbol) {
          for (size_t i = 0; i < gaps.n; ++i) {
            bool bit;
            if (!gaps[i] && !prev) bit = true;
            else if (gaps[i]) { bit = dec.decode(m.gapP()); m.gapUpdate(bit); m.mainState(bit); }
            else { bit = dec.decode(m.mainP()); m.mainUpdate(bit); }
            nodes[id].bits.push(bit);
            if (bit) right.push(prev != bit || gaps[i]);
            prev = bit;
          }
          level.push_back(IntegerNode{l, 0, gaps.n - right.n, 0, 0});
          queue.push_back(Inner(r, right));
        } else if (!nodes[r].hasSymbol) {
          for (size_t i = 0; i < gaps.n; ++i) {
            bool bit;
            if (gaps[i]) { bit = dec.decode(m.gapP()); m.gapUpdate(bit); m.mainState(bit); }
            else { bit = dec.decode(m.mainP()); m.mainUpdate(bit); }
            nodes[id].bits.push(bit);
            (bit ? right : left).push(prev != bit || gaps[i]);
            prev = bit;
          }
          queue.push_back(Inner(l, left));
          queue.push_back(Inner(r, right));
        } else {
          MemoryBitReader::fail("tree shape the encoder cannot produce");
        }
      }
      // integer levels: all nodes of a level, then its left children, then its right children
      std::deque<IntegerNode> lefts, rights;
      while (!level.empty()) {
        m.resetInts();
        lefts.clear(); rights.clear();
        while (!level.empty()) {
          IntegerNode nd = level.front();
          level.pop_front();
          if (nd.code >= 0 && codeTree[nd.code].hasSymbol) {
            nodes[nd.node].hasSymbol = true;
            if (!plainFixed) nodes[nd.node].symbol = codeTree[nd.code].symbol;   // :1041-1044
            if (codeTree[nd.code].symbol != 0) continue;
          }
          size_t ones = 0;
          for (size_t i = 0; i < nd.bits; ++i) {
            const bool bit = dec.decode(m.intP());
            m.intUpdate(bit);
            ones += bit;
            nodes[nd.node].bits.push(bit);
          }
          const bool inCode = nd.code >= 0 && !codeTree[nd.code].hasSymbol;
          if (nd.bits > ones) {
            IntegerNode c = {child(nd.node, false), inCode ? codeTree[nd.code].left : -1, nd.bits - ones, nd.leadingOnes, nd.status};
            if (c.code < 0) {
              if (nd.status == 0) { c.status = 2; c.leadingOnes = W; }
              else if (nd.status == 1) { c.status = 2; c.leadingOnes += W; }
              else --c.leadingOnes;
            }
            if (c.status != 2 || c.leadingOnes > 0) lefts.push_back(c);
          }
          if (ones > 0) {
            IntegerNode c = {child(nd.node, true), inCode ? codeTree[nd.code].right : -1, ones, nd.leadingOnes, nd.status};
            if (c.code < 0) {
              if (nd.status == 0) { c.status = 1; ++c.leadingOnes; }
              else if (nd.status == 1) ++c.leadingOnes;
              else --c.leadingOnes;
            }
            if (c.status != 2 || c.leadingOnes > 0) rights.push_back(c);
          }
        }
        level.insert(level.end(), lefts.begin(), lefts.end());
        level.insert(level.end(), rights.begin(), rights.end());
      }
    }

    // message, :1277-1378: every node's bits are used in order, so a read cursor per node
    // stands in for the reference's rank bookkeeping
    size_t message(byte* out, uint64 room) {
      size_t len = 0;
      const size_t runs = nodes[0].bits.n;
      for (size_t j = 0; j < runs; ++j) {
        int nd = 0;
        do { const bool bit = take(nd); nd = bit ? nodes[nd].right : nodes[nd].left; if (nd < 0) MemoryBitReader::fail("symbol path"); }
        while (!isSymbolLeaf(nd));
        const byte symbol = (byte)nodes[nd].symbol;
        uint64 runLength = 0;
        int code = 0;
        if (!plainFixed) {
          do {
            const bool bit = take(nd);
            nd = bit ? nodes[nd].right : nodes[nd].left;
            code = bit ? codeTree[code].right : codeTree[code].left;
            if (nd < 0 || code < 0) MemoryBitReader::fail("length code path");
          } while (!codeTree[code].hasSymbol);
          runLength = codeTree[code].symbol;
        }
        if (runLength == 0) {                                        // escape: the fixed code, :514-533
          uint32 leadingOnes = 0;
          bool bit = take(nd);
          while (bit) { ++leadingOnes; nd = nodes[nd].right; if (nd < 0) MemoryBitReader::fail("fixed code"); bit = take(nd); }
          for (uint32 k = 0; k < leadingOnes + W; ++k) {
            runLength <<= 1;
            nd = bit ? nodes[nd].right : nodes[nd].left;
            if (nd < 0) MemoryBitReader::fail("fixed code");
            bit = take(nd);
            runLength |= bit ? 1u : 0u;
          }
          runLength += 1 + (((uint64(1) << leadingOnes) - 1) << W);
        }
        if (runLength > room - len) MemoryBitReader::fail("run length");
        std::memset(out + len, symbol, runLength);
        len += runLength;
      }
      return len;
    }
    bool take(int nd) {
      if (nodes[nd].bits.at >= nodes[nd].bits.n) MemoryBitReader::fail("node out of bits");
      return nodes[nd].bits.take();
    }
    bool isSymbolLeaf(int nd) const { return nd < symbolNodes && nodes[nd].hasSymbol; }
    int symbolNodes = 0;
  };
  detail::WaveletModels m_models;
  detail::RangeDecoder m_rc;
};

inline EntropyDecoder* giveEntropyDecoder(char decoder) {            // EntropyCoders.cpp:53-65
  if (decoder == 'H') return new HuffmanDecoder();
  if (decoder == 'B' || decoder == 'b' || decoder == 'u') return new WaveletDecoder(decoder);
  std::fprintf(stderr, "bwtc-hip: entropy decoder '%c' is not offered (H, B, b, u are; the reference's m / M index past "
               "their history table)\n", decoder);
  std::exit(1);
}

class Decompressor {
 public:
  Decompressor(InStream* in, OutStream* out, int device = 0)
      : m_in(in), m_out(out), m_decoder(0), m_device(device) {}
  ~Decompressor() { delete m_in; delete m_out; delete m_decoder; }

  // Decompressor::decompress, Decompressor.cpp:58-94
  size_t decompress(size_t threads) {
    if (threads != 1) { std::fprintf(stderr, "Supporting only single thread!\n"); return 0; }
    std::vector<byte> all;
    { std::vector<byte> chunk(1 << 20); size_t got;
      while ((got = m_in->readBlock(&chunk[0], chunk.size())) > 0) all.insert(all.end(), chunk.begin(), chunk.begin() + got); }
    if (all.empty()) MemoryBitReader::fail("empty input");
    MemoryBitReader in(&all[0], all.size());
    m_decoder = giveEntropyDecoder((char)in.readByte());            // readGlobalHeader, :51-56
    InverseBWTransform* ibwt = 0;
    uint32 ibwtCap = 0;
    size_t decompressedSize = 0;
    std::vector<byte> buf;
    for (;;) {
      const uint64 originalSize = in.readPackedInteger();           // PrecompressorBlock.cpp:97-108
      if (originalSize == 0) break;
      const uint64 slices = in.readPackedInteger();
      // PrecompressorBlock::readBlockHeader reads the block's grammar here (PrecompressorBlock.cpp:97-108); a
      // grammar with rules means the slices hold the PRECOMPRESSED block, which the Postprocessor expands
      Grammar grammar;
      in.flushBuffer();
      in.advance(grammar.readGrammar(in.here(), in.left()));
      const bool rules = grammar.numberOfRules() > 0;
      const size_t room = rules ? 2 * (size_t)originalSize + 64 : (size_t)originalSize;
      buf.resize(room + 1);
      size_t used = 0;
      for (uint64 i = 0; i < slices; ++i) {
        BWTBlock block(&buf[used], 0, true);
        m_decoder->decodeBlock(block, in, room - used);
        if (!ibwt || block.size() > ibwtCap) {
          delete ibwt;
          ibwtCap = (uint32)std::max<size_t>(block.size(), 1u << 20);
          ibwt = giveInverseTransformer(ibwtCap, m_device);
        }
        ibwt->doTransform(block);
        used += block.size();
      }
      if (rules) {
        Postprocessor postprocessor(false, grammar);               // Decompressor.cpp:82-88
        const size_t postSize = postprocessor.uncompress(&buf[0], used, m_out, (size_t)originalSize);
        if (postSize != originalSize) MemoryBitReader::fail("postprocessed block size");
        decompressedSize += postSize;
        continue;
      }
      if (used != originalSize) MemoryBitReader::fail("precompressor block size");
      m_out->writeBlock(&buf[0], &buf[0] + used);
      decompressedSize += used;
    }
    delete ibwt;
    m_out->flush();
    return decompressedSize;
  }
 private:
  InStream* m_in;
  OutStream* m_out;
  EntropyDecoder* m_decoder;
  int m_device;
};

}  // namespace bwtc
