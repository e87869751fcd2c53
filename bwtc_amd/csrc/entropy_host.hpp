// Host-side small-table work of the 'H' coder: everything in HuffmanEncoder that is not a
// pass over the block (sectioning, code lengths, canonical codes, shape serialisation,
// packed integers, BWTBlock header).  Mirrors the reference's names so the call sites read
// like HuffmanCoders.cpp; written from the format description, not from its source.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace bwtc {
namespace utils {

// utils::packInteger (Utils.cpp:89-101): 7-bit groups, low group first, bit 7 = "more".
// Appends the bytes in stream order (low byte first, HuffmanCoders.cpp:316-322).
unsigned packInteger(uint64_t value, std::vector<uint8_t>& out);

// utils::calculateHuffmanLengths(codeLengths, freqs, 256) + calculateCodeLengths
// (Utils.cpp:408-473): minimum-redundancy code lengths; ties broken by (freq, symbol).
void calculateHuffmanLengths(const uint64_t freqs[256], uint8_t clen[256]);

// utils::computeHuffmanCodes (Utils.cpp:180-202): canonical codes, longest codes first.
void computeHuffmanCodes(const uint8_t clen[256], uint32_t code[256]);

// MSB-first bit sink.
class BitSink {
 public:
  void push(uint64_t value, unsigned nbits);
  void pushBit(bool b) { push(b ? 1u : 0u, 1); }
  size_t bits() const { return nbits_; }
  // zero-pads to a byte boundary and appends to `out`
  void flushTo(std::vector<uint8_t>& out);
 private:
  std::vector<uint8_t> bytes_;
  size_t nbits_ = 0;
};

// utils::binaryCode / binaryInterpolativeCode / unaryCode (Utils.hpp:239-297,398-402).
void binaryCode(size_t n, size_t lo, size_t hi, BitSink& bits);
void binaryInterpolativeCode(const std::vector<uint32_t>& list, size_t maxValue, BitSink& bits);
void unaryCode(BitSink& bits, size_t n);

}  // namespace utils

// HuffmanEncoder::serializeShape (HuffmanCoders.cpp:63-86), byte-padded as encodeData
// writes it (:181-192).
void serializeShape(const uint8_t clen[256], std::vector<uint8_t>& out);

// Section heuristic of {Huffman,Wavelet}Encoder::writeBlockHeader
// (HuffmanCoders.cpp:282-296, WaveletCoders.cpp:184-198).
std::vector<uint32_t> deduceSections(const uint32_t freqs[256]);

// BWTBlock::writeHeader (BWTBlock.cpp:61-86).
void writeBWTBlockHeader(const uint32_t* lf, uint32_t n_lf, std::vector<uint8_t>& out);

}  // namespace bwtc
