// Decoding side of the host mirror: HuffmanDecoder (HuffmanCoders.cpp:88-117,323-616),
// BWTBlock::readHeader (BWTBlock.cpp:88-102), Decompressor (Decompressor.cpp:45-94).  The
// entropy decoder is serial host code (a section's streams have no length fields, so
// nothing in a block can be decoded ahead); the inverse transform runs on the GPU
// (HipInverseBWTransform).  Only coder 'H' and streams without precompression are accepted.
#pragma once
#include <cstring>

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

inline EntropyDecoder* giveEntropyDecoder(char decoder) {            // EntropyCoders.cpp:53-65
  if (decoder == 'H') return new HuffmanDecoder();
  std::fprintf(stderr, "bwtc-hip: entropy decoder '%c' is not built (only 'H')\n", decoder);
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
      if (in.readByte() != 0) MemoryBitReader::fail("grammar with rules (precompressed stream)");
      buf.resize(originalSize + 1);
      size_t used = 0;
      for (uint64 i = 0; i < slices; ++i) {
        BWTBlock block(&buf[used], 0, true);
        m_decoder->decodeBlock(block, in, originalSize - used);
        if (!ibwt || block.size() > ibwtCap) {
          delete ibwt;
          ibwtCap = (uint32)std::max<size_t>(block.size(), 1u << 20);
          ibwt = giveInverseTransformer(ibwtCap, m_device);
        }
        ibwt->doTransform(block);
        used += block.size();
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
