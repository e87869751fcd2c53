// Host-side C++ mirror of the reference's operator surface for the hot path, on top of the
// C ABI (include/bwtc_hip.h).  Same class and method names, argument meaning and error
// behaviour as pjmikkol/bwtc so that call sites and tests read like the reference's:
//
//   BWTBlock            BWTBlock.hpp:39-72, BWTBlock.cpp:61-108
//   OutStream           Streams.hpp:42-50   (+ MemoryOutStream = test/TestStreams.hpp:38-125,
//                                              RawOutStream = Streams.hpp:66-101)
//   BWTransform         bwtransforms/BWTransform.hpp:48-70, BWTransform.cpp:38-76
//   HipBWTransform      the new back-end (what INTEGRATION.md adds to the reference), 'g'
//   BWTManager          bwtransforms/BWTManager.hpp:41-58, BWTManager.cpp:36-80
//   InverseBWTransform  bwtransforms/InverseBWT.hpp:45-55, InverseBWT.cpp:42-51 (+ HipInverseBWTransform)
//   EntropyEncoder      EntropyCoders.hpp:44-66, EntropyCoders.cpp:38-51
//   HuffmanEncoder      HuffmanCoders.hpp, HuffmanCoders.cpp:46-61
//   Compressor          Compressor.hpp:76-117, Compressor.cpp:36-118
//
// There is NO CPU back-end here: 'd'/'s' are rejected, the GPU library is the only
// transformer, and every failure of the library is fatal (the reference exits on I/O errors
// too, Streams.cpp:51-55).
#pragma once
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "bwtc_hip.h"

namespace bwtc {

typedef uint8_t byte;
typedef uint32_t uint32;
typedef uint64_t uint64;

inline void hipFatal(int rc, const char* what) {
  if (rc != 0) {
    std::fprintf(stderr, "bwtc-hip: %s failed with code %d\n", what, rc);
    std::exit(1);
  }
}

// ---- streams ---------------------------------------------------------------------------
class OutStream {
 public:
  virtual ~OutStream() {}
  virtual void writeByte(byte b) = 0;
  virtual void writeBlock(const byte* begin, const byte* end) = 0;
  virtual long int getPos() = 0;
  virtual void write48bits(uint64 to_written, long int position) = 0;
  virtual void flush() = 0;
};

class MemoryOutStream : public OutStream {
 public:
  std::vector<byte> data;
  void writeByte(byte b) { data.push_back(b); }
  void writeBlock(const byte* begin, const byte* end) { data.insert(data.end(), begin, end); }
  long int getPos() { return (long int)data.size(); }
  void write48bits(uint64 v, long int position) {
    for (int i = 0; i < 6; ++i) data[position + i] = (byte)(v >> (8 * (5 - i)));
  }
  void flush() {}
};

class RawOutStream : public OutStream {
 public:
  explicit RawOutStream(const std::string& file) : m_to(0), m_own(true), m_pos(0) {
    m_to = file.empty() ? stdout : std::fopen(file.c_str(), "wb");
    m_own = !file.empty();
    if (!m_to) { std::perror(file.c_str()); std::exit(1); }     // Streams.cpp:51-55
  }
  ~RawOutStream() { if (m_to) { std::fflush(m_to); if (m_own) std::fclose(m_to); } }
  void writeByte(byte b) { std::fputc(b, m_to); ++m_pos; }
  void writeBlock(const byte* begin, const byte* end) {
    std::fwrite(begin, 1, end - begin, m_to);
    m_pos += end - begin;
  }
  long int getPos() { return m_pos; }
  void write48bits(uint64 v, long int position) {               // Streams.cpp:106-117
    std::fflush(m_to);
    std::fseek(m_to, position, SEEK_SET);
    for (int i = 0; i < 6; ++i) std::fputc((int)((v >> (8 * (5 - i))) & 0xff), m_to);
    std::fseek(m_to, 0, SEEK_END);
  }
  void flush() { std::fflush(m_to); }
 private:
  FILE* m_to;
  bool m_own;
  long int m_pos;
};

class InStream {
 public:
  virtual ~InStream() {}
  virtual size_t readBlock(byte* to, size_t max_block_size) = 0;
};

class MemoryInStream : public InStream {
 public:
  MemoryInStream(const byte* p, size_t n) : m_p(p), m_n(n), m_pos(0) {}
  size_t readBlock(byte* to, size_t max) {
    const size_t k = std::min(max, m_n - m_pos);
    std::copy(m_p + m_pos, m_p + m_pos + k, to);
    m_pos += k;
    return k;
  }
 private:
  const byte* m_p; size_t m_n, m_pos;
};

class RawInStream : public InStream {
 public:
  explicit RawInStream(const std::string& file) {
    m_from = file.empty() ? stdin : std::fopen(file.c_str(), "rb");
    m_own = !file.empty();
    if (!m_from) { std::perror(file.c_str()); std::exit(1); }
  }
  ~RawInStream() { if (m_own && m_from) std::fclose(m_from); }
  size_t readBlock(byte* to, size_t max) { return std::fread(to, 1, max, m_from); }
 private:
  FILE* m_from; bool m_own;
};

// ---- BWTBlock ----------------------------------------------------------------------------
class BWTBlock {
 public:
  BWTBlock() : m_begin(0), m_length(0), m_isTransformed(true) {}
  BWTBlock(byte* data, uint32 length, bool isTransformed)
      : m_begin(data), m_length(length), m_isTransformed(isTransformed) {}
  void setTransformed(bool t) { assert(m_isTransformed != t); m_isTransformed = t; }
  bool isTransformed() const { return m_isTransformed; }
  size_t size() const { return m_length; }
  byte* begin() { return m_begin; }
  const byte* begin() const { return m_begin; }
  byte* end() { return m_begin + m_length; }
  std::vector<uint32>& LFpowers() { return m_LFpowers; }
  const std::vector<uint32>& LFpowers() const { return m_LFpowers; }
  void setBegin(byte* b) { m_begin = b; }
  void setSize(uint32 n) { m_length = n; }
  void prepareLFpowers(uint32 startingPoints) {                 // BWTBlock.cpp:104-108
    if (m_length <= 256 || startingPoints == 0) m_LFpowers.resize(1);
    else if (startingPoints <= 256) m_LFpowers.resize(startingPoints);
    else m_LFpowers.resize(256);
  }
  size_t writeHeader(OutStream* out) const {                    // BWTBlock.cpp:61-86
    std::vector<byte> h(2048);
    const uint32 n = bwtc_hip_host_bwtblock_header(&m_LFpowers[0], (uint32)m_LFpowers.size(),
                                                   &h[0], (uint32)h.size());
    out->writeBlock(&h[0], &h[0] + n);
    return n;
  }
 private:
  byte* m_begin;
  uint32 m_length;
  std::vector<uint32> m_LFpowers;
  bool m_isTransformed;
};

// ---- transforms --------------------------------------------------------------------------
class BWTransform {
 public:
  BWTransform() {}
  virtual ~BWTransform() {}
  virtual void doTransform(byte* begin, uint32 length, std::vector<uint32>& LF) const = 0;
  virtual void doTransform(byte* begin, uint32 length, std::vector<uint32>& LF,
                           uint32 freqs[256]) const = 0;
  // block wrappers, BWTransform.cpp:38-64
  virtual void doTransform(BWTBlock& block) {
    std::reverse(block.begin(), block.end());
    byte next = *block.end();
    *block.end() = 0;
    doTransform(block.begin(), (uint32)block.size() + 1, block.LFpowers());
    block.setTransformed(true);
    *(block.begin() + block.LFpowers()[0]) = *block.end();
    *block.end() = next;
  }
  virtual void doTransform(BWTBlock& block, uint32 freqs[256]) {
    std::reverse(block.begin(), block.end());
    byte next = *block.end();
    *block.end() = 0;
    doTransform(block.begin(), (uint32)block.size() + 1, block.LFpowers(), freqs);
    block.setTransformed(true);
    *(block.begin() + block.LFpowers()[0]) = *block.end();
    *block.end() = next;
  }
  virtual uint64 maxSizeInBytes(uint64 block_size) const = 0;
  virtual uint64 maxBlockSize(uint64 memory_budget) const = 0;
  virtual uint64 suggestedBlockSize(uint64 memory_budget) const = 0;
 private:
  BWTransform(const BWTransform&);
  const BWTransform& operator=(const BWTransform&);
};

// The MI355X back-end.  One context (= one GPU, one stream, one persistent HBM workspace).
class HipBWTransform : public BWTransform {
 public:
  explicit HipBWTransform(uint32 maxBlockSize, int device = 0) : m_ctx(0) {
    hipFatal(bwtc_hip_create(device, maxBlockSize, &m_ctx), "bwtc_hip_create");
  }
  virtual ~HipBWTransform() { bwtc_hip_destroy(m_ctx); }
  void doTransform(byte* begin, uint32 length, std::vector<uint32>& LF) const {
    hipFatal(bwtc_hip_bwt(m_ctx, begin, length, &LF[0], (uint32)LF.size(), 0), "bwtc_hip_bwt");
  }
  void doTransform(byte* begin, uint32 length, std::vector<uint32>& LF, uint32 freqs[256]) const {
    hipFatal(bwtc_hip_bwt(m_ctx, begin, length, &LF[0], (uint32)LF.size(), freqs), "bwtc_hip_bwt");
  }
  // whole block on the device: reverse, sentinel, transform, end-of-block patch; the byte at
  // *block.end() is never touched, so neighbouring slices may be in flight on other GPUs
  void doTransform(BWTBlock& block) {
    hipFatal(bwtc_hip_bwt_block(m_ctx, block.begin(), (uint32)block.size(), &block.LFpowers()[0],
                                (uint32)block.LFpowers().size(), 0), "bwtc_hip_bwt_block");
    block.setTransformed(true);
  }
  void doTransform(BWTBlock& block, uint32 freqs[256]) {
    hipFatal(bwtc_hip_bwt_block(m_ctx, block.begin(), (uint32)block.size(), &block.LFpowers()[0],
                                (uint32)block.LFpowers().size(), freqs), "bwtc_hip_bwt_block");
    block.setTransformed(true);
  }
  uint64 maxSizeInBytes(uint64 n) const { return bwtc_hip_workspace_bytes((uint32)n); }
  uint64 maxBlockSize(uint64) const { return 0x7fffffffu - 2; }
  uint64 suggestedBlockSize(uint64) const { return 256u << 20; }
  bwtc_hip_ctx* context() const { return m_ctx; }
 private:
  bwtc_hip_ctx* m_ctx;
};

class BWTManager {
 public:
  BWTManager() : m_startingPoints(1), m_maxBlock(1u << 20), m_device(0) {}
  explicit BWTManager(uint32 startingPoints) : m_startingPoints(startingPoints), m_maxBlock(1u << 20), m_device(0) {}
  ~BWTManager() { for (size_t i = 0; i < m_transformers.size(); ++i) delete m_transformers[i]; }
  void doTransform(BWTBlock& block) {                            // BWTManager.cpp:46-51
    assert(!block.isTransformed());
    block.prepareLFpowers(m_startingPoints);
    m_transformers[0]->doTransform(block);
  }
  void doTransform(BWTBlock& block, uint32* freqs) {             // BWTManager.cpp:53-58
    assert(!block.isTransformed());
    block.prepareLFpowers(m_startingPoints);
    m_transformers[0]->doTransform(block, freqs);
  }
  // maxBlockSize / device are what the GPU back-end needs to size its workspace
  void setMaxBlockSize(uint32 n) { m_maxBlock = n; }
  void setDevice(int d) { m_device = d; }
  void initialize(char choice) {                                 // BWTManager.cpp:74-80
    if (!isValidChoice(choice)) {
      std::fprintf(stderr, "bwtc-hip: BWT algorithm '%c' is not available (only 'g')\n", choice);
      std::exit(1);
    }
    m_transformers.push_back(new HipBWTransform(m_maxBlock, m_device));
  }
  void setStartingPoints(uint32 sp) {                            // BWTManager.cpp:60-64
    if (sp < 1) sp = 1; else if (sp > 256) sp = 256;
    m_startingPoints = sp;
  }
  uint32 getStartingPoints() const { return m_startingPoints; }
  static bool isValidChoice(char c) { return c == 'g' || c == 'a'; }
  bwtc_hip_ctx* hipContext() const {
    return static_cast<HipBWTransform*>(m_transformers[0])->context();
  }
 private:
  std::vector<BWTransform*> m_transformers;
  uint32 m_startingPoints, m_maxBlock;
  int m_device;
};

// ---- inverse transform ---------------------------------------------------------------------
// bwtransforms/InverseBWT.hpp:45-55.  The raw virtual takes the N = size + 1 byte layout the
// reference's wrapper builds (bwt[N-1] = bwt[LF[0]], InverseBWT.cpp:47-51).
class InverseBWTransform {
 public:
  virtual ~InverseBWTransform() {}
  virtual uint64 maxBlockSize(uint64 memory_budget) const = 0;
  virtual void doTransform(byte* bwt, uint32 n, const std::vector<uint32>& LFpow) = 0;
  virtual void doTransform(BWTBlock& block) {
    byte* data = block.begin();
    *block.end() = data[block.LFpowers()[0]];
    doTransform(block.begin(), (uint32)block.size() + 1, block.LFpowers());
  }
};

class HipInverseBWTransform : public InverseBWTransform {
 public:
  explicit HipInverseBWTransform(uint32 maxBlockSize, int device = 0) : m_ctx(0) {
    hipFatal(bwtc_hip_create(device, maxBlockSize, &m_ctx), "bwtc_hip_create");
  }
  virtual ~HipInverseBWTransform() { bwtc_hip_destroy(m_ctx); }
  uint64 maxBlockSize(uint64) const { return 0x7fffffffu - 2; }
  void doTransform(byte* bwt, uint32 n, const std::vector<uint32>& LFpow) {
    // n counts the borrowed end byte; the library rebuilds it from LFpow[0] itself
    hipFatal(bwtc_hip_inverse_bwt_block(m_ctx, bwt, n - 1, &LFpow[0], (uint32)LFpow.size()),
             "bwtc_hip_inverse_bwt_block");
  }
  void doTransform(BWTBlock& block) {     // *block.end() is not needed and not touched
    hipFatal(bwtc_hip_inverse_bwt_block(m_ctx, block.begin(), (uint32)block.size(),
                                        &block.LFpowers()[0], (uint32)block.LFpowers().size()),
             "bwtc_hip_inverse_bwt_block");
  }
 private:
  bwtc_hip_ctx* m_ctx;
};

inline InverseBWTransform* giveInverseTransformer(uint32 maxBlockSize, int device = 0) {
  return new HipInverseBWTransform(maxBlockSize, device);     // InverseBWT.cpp:42-45
}

// ---- entropy coding ------------------------------------------------------------------------
class EntropyEncoder {
 public:
  virtual ~EntropyEncoder() {}
  virtual size_t transformAndEncode(BWTBlock& block, BWTManager& bwtm, OutStream* out) = 0;
  // Overlapped form for coders whose host half outlasts their device half: begin() returns
  // when the device is free for the next block, finishOldest() writes the oldest block's
  // record (waiting for it if need be).  Coders without a host half keep the defaults and
  // Compressor::compress calls transformAndEncode block by block.
  virtual bool overlapsBlocks() const { return false; }
  virtual void begin(BWTBlock&, BWTManager&) {}
  virtual size_t pending() const { return 0; }
  virtual size_t depth() const { return 1; }
  // how many blocks the caller should keep under way right now (never more than depth())
  virtual size_t inFlightLimit() const { return depth(); }
  virtual size_t finishOldest(OutStream*) { return 0; }
};

class HuffmanEncoder : public EntropyEncoder {
 public:
  HuffmanEncoder() {}
  // HuffmanCoders.cpp:51-61: transform, writeBlockHeader, encodeData, finishBlock -- all in
  // one library call; the record already carries its 48-bit length.
  size_t transformAndEncode(BWTBlock& block, BWTManager& bwtm, OutStream* out) {
    assert(!block.isTransformed());
    m_record.resize(bwtc_hip_compress_bound((uint32)block.size()));
    uint64_t n = 0;
    hipFatal(bwtc_hip_transform_and_encode(bwtm.hipContext(), block.begin(), (uint32)block.size(),
                                           bwtm.getStartingPoints(), &m_record[0], m_record.size(),
                                           &n), "bwtc_hip_transform_and_encode");
    block.setTransformed(true);
    out->writeBlock(&m_record[0], &m_record[0] + n);
    return (size_t)n;
  }
 private:
  std::vector<byte> m_record;
};

// WaveletCoders.hpp:48-77; the letter picks the main model (giveProbabilityModel).  The encoder object carries the main model's
// state from block to block (m_probModel; FSM8::resetModel keeps its state): that state
// lives in the context and a new encoder starts a new stream there.
class WaveletEncoder : public EntropyEncoder {
 public:
  explicit WaveletEncoder(char encoder = 'B') : m_letter(encoder), m_fresh(true), m_ctx(0), m_dev(0), m_devBytes(0), m_next(0) {}
  ~WaveletEncoder() {
    // blocks still under way write into the slots: let them finish before the slots go
    while (m_ctx && !m_order.empty()) {
      uint64_t n = 0;
      (void)bwtc_hip_wavelet_encode_end(m_ctx, m_slots[m_order.front()].ticket, &n);
      m_order.erase(m_order.begin());
    }
    if (m_ctx && m_dev) bwtc_hip_free(m_ctx, m_dev);
    for (size_t i = 0; i < m_slots.size(); ++i) std::free(m_slots[i].rec);
  }
  // WaveletCoders.cpp:70-86: transform, writeBlockHeader, encodeData, finishBlock
  size_t transformAndEncode(BWTBlock& block, BWTManager& bwtm, OutStream* out) {
    assert(!block.isTransformed());
    start(bwtm);
    m_record.resize(bwtc_hip_compress_bound((uint32)block.size()));
    uint64_t n = 0;
    hipFatal(bwtc_hip_transform_and_encode_wavelet(m_ctx, block.begin(), (uint32)block.size(),
                                                   bwtm.getStartingPoints(), 0, &m_record[0],
                                                   m_record.size(), &n),
             "bwtc_hip_transform_and_encode_wavelet");
    block.setTransformed(true);
    out->writeBlock(&m_record[0], &m_record[0] + n);
    return (size_t)n;
  }
  bool overlapsBlocks() const { return true; }
  // never more than the context allows (BWTC_HIP_WAVELET_DEPTH), or _begin would refuse with -6
  size_t depth() const {
    const size_t lim = m_ctx ? bwtc_hip_wavelet_depth(m_ctx) : (size_t)kDefaultDepth;
    return std::max<size_t>(1, std::min<size_t>(kMaxDepth, lim));
  }
  size_t pending() const { return m_order.size(); }
  // what the stream has shown to need (bwtc_hip_wavelet_depth_needed: block latency over block interval, with a
  // margin) once it is known, inside what the context allows: fewer blocks under way hold less page-locked memory
  size_t inFlightLimit() const {
    const size_t lim = depth();
    const size_t need = m_ctx ? bwtc_hip_wavelet_depth_needed(m_ctx) : 0;
    return need ? std::max<size_t>(4, std::min(lim, need)) : lim;
  }
  // device half: upload, transform on the device, run scanner + stream kernels; the models and
  // range coders of the block are queued on the context's worker threads.  The block's bytes
  // in host memory are left as they were.
  void begin(BWTBlock& block, BWTManager& bwtm) {
    start(bwtm);
    const size_t slots = depth();                                   // fixed once the context exists
    assert(!block.isTransformed() && m_order.size() < slots);
    const uint32 size = (uint32)block.size();
    if (m_devBytes < (uint64)size + 64) {
      if (m_dev) bwtc_hip_free(m_ctx, m_dev);
      m_dev = bwtc_hip_malloc(m_ctx, (uint64)size + 64);
      if (!m_dev) hipFatal(-2, "bwtc_hip_malloc");
      m_devBytes = (uint64)size + 64;
    }
    if (m_slots.size() < slots) m_slots.resize(slots);
    Slot& slot = m_slots[m_next % slots];
    const uint64 cap = bwtc_hip_compress_bound(size);
    if (slot.cap < cap) {                                         // not zero-filled: only the record's bytes get touched
      std::free(slot.rec);
      slot.rec = static_cast<byte*>(std::malloc(cap));
      if (!slot.rec) hipFatal(-2, "malloc");
      slot.cap = cap;
    }
    block.prepareLFpowers(bwtm.getStartingPoints());
    uint32 freqs[256];
    std::memset(freqs, 0, sizeof freqs);                          // WaveletCoders.cpp:78-79
    hipFatal(bwtc_hip_memcpy_to_device(m_ctx, m_dev, block.begin(), size), "bwtc_hip_memcpy_to_device");
    hipFatal(bwtc_hip_bwt_block_device(m_ctx, static_cast<const uint8_t*>(m_dev), static_cast<uint8_t*>(m_dev), size,
                                       &block.LFpowers()[0], (uint32)block.LFpowers().size(), freqs),
             "bwtc_hip_bwt_block_device");
    hipFatal(bwtc_hip_wavelet_encode_device_begin(m_ctx, static_cast<const uint8_t*>(m_dev), size,
                                                  &block.LFpowers()[0], (uint32)block.LFpowers().size(),
                                                  freqs, 0, slot.rec, slot.cap, &slot.ticket),
             "bwtc_hip_wavelet_encode_device_begin");
    m_order.push_back(m_next % slots);
    ++m_next;
  }
  size_t finishOldest(OutStream* out) {
    assert(!m_order.empty());
    Slot& slot = m_slots[m_order.front()];
    m_order.erase(m_order.begin());
    uint64_t n = 0;
    hipFatal(bwtc_hip_wavelet_encode_end(m_ctx, slot.ticket, &n), "bwtc_hip_wavelet_encode_end");
    out->writeBlock(slot.rec, slot.rec + n);
    return (size_t)n;
  }
 private:
  // blocks under way: what the context allows (BWTC_HIP_WAVELET_DEPTH, 16 unless the caller asks
  // for more: from 56 on the library's fused host engines are used, bwtc_hip.h)
  enum { kDefaultDepth = 12, kMaxDepth = 256 };
  struct Slot { byte* rec; uint64 cap; uint64_t ticket; Slot() : rec(0), cap(0), ticket(0) {} };
  void start(BWTManager& bwtm) {
    m_ctx = bwtm.hipContext();
    if (m_fresh) { hipFatal(bwtc_hip_wavelet_start(m_ctx, m_letter), "bwtc_hip_wavelet_start"); m_fresh = false; }
  }
  char m_letter;                     // 'B', 'b' or 'u': the main probability model, also the stream's header byte
  bool m_fresh;
  bwtc_hip_ctx* m_ctx;
  void* m_dev;
  uint64 m_devBytes;
  size_t m_next;
  std::vector<Slot> m_slots;
  std::vector<size_t> m_order;
  std::vector<byte> m_record;
};

inline EntropyEncoder* giveEntropyEncoder(char encoder) {        // EntropyCoders.cpp:38-51
  if (encoder == 'H') return new HuffmanEncoder();
  if (encoder == 'B' || encoder == 'b' || encoder == 'u') return new WaveletEncoder(encoder);
  std::fprintf(stderr, "bwtc-hip: entropy coder '%c' is not offered (H, B, b, u are; the reference's m / M index past "
               "their history table)\n", encoder);
  std::exit(1);
}

// ---- pre-stage ("--prepr p...") ---------------------------------------------------------------
// preprocessors/Grammar.hpp: the grammar of one precompressor block, behind the C ABI.
class Grammar {
 public:
  Grammar() : m_g(bwtc_hip_grammar_create()) { if (!m_g) hipFatal(-2, "bwtc_hip_grammar_create"); }
  ~Grammar() { bwtc_hip_grammar_destroy(m_g); }
  uint32 numberOfRules() const { return bwtc_hip_grammar_rules(m_g); }
  uint32 numberOfSpecialSymbols() const { return bwtc_hip_grammar_special_symbols(m_g); }
  bool isSpecial(byte symbol) const { return bwtc_hip_grammar_is_special(m_g, symbol) != 0; }
  uint32 writeGrammar(OutStream* dst) const {                     // Grammar.cpp:309-320
    // (a rule costs at most seven bytes, the freed symbols' table at most 64 KiB: forty rounds of `p` no longer overflow)
    std::vector<byte> raw(((size_t)1 << 17) + 8 * (size_t)numberOfRules());
    uint64_t n = 0;
    hipFatal(bwtc_hip_grammar_write(m_g, &raw[0], raw.size(), &n), "bwtc_hip_grammar_write");
    dst->writeBlock(&raw[0], &raw[0] + n);
    return (uint32)n;
  }
  // Grammar.cpp:198-307 from bytes in memory; returns what it consumed
  size_t readGrammar(const byte* in, size_t n) {
    uint64_t used = 0;
    hipFatal(bwtc_hip_grammar_read(m_g, in, n, &used), "bwtc_hip_grammar_read");
    return (size_t)used;
  }
  bwtc_hip_grammar* handle() const { return m_g; }
 private:
  Grammar(const Grammar&);
  Grammar& operator=(const Grammar&);
  bwtc_hip_grammar* m_g;
};

// preprocessors/Precompressor.hpp: options = one letter per round ('p' = PairReplacer).
class Precompressor {
 public:
  Precompressor() {}
  explicit Precompressor(const std::string& preprocessing) : m_preprocessingOptions(preprocessing) {}
  const std::string& options() const { return m_preprocessingOptions; }
  // Precompressor::precompress (Precompressor.cpp:62-121) over block[0..size): the new size
  size_t precompress(bwtc_hip_ctx* ctx, Grammar& grammar, byte* block, size_t size) const {
    uint64_t n = size;
    if (!m_preprocessingOptions.empty())
      hipFatal(bwtc_hip_precompress(ctx, grammar.handle(), m_preprocessingOptions.c_str(), block, size, &n), "bwtc_hip_precompress");
    return (size_t)n;
  }
 private:
  std::string m_preprocessingOptions;
};

// preprocessors/Postprocessor.hpp
class Postprocessor {
 public:
  Postprocessor(bool, const Grammar& grammar) : m_grammar(grammar) {}
  size_t uncompress(const byte* data, size_t length, OutStream* to, size_t originalSize) const {   // Postprocessor.cpp:112-133
    std::vector<byte> out(originalSize + 1);
    uint64_t n = 0;
    hipFatal(bwtc_hip_postprocess(m_grammar.handle(), data, length, &out[0], originalSize, &n), "bwtc_hip_postprocess");
    to->writeBlock(&out[0], &out[0] + n);
    return (size_t)n;
  }
 private:
  const Grammar& m_grammar;
};

// ---- Compressor ----------------------------------------------------------------------------
struct Options {                                                  // Compressor.hpp:40-48
  Options(size_t memLimit_, char entropyCoder_) : memLimit(memLimit_), entropyCoder(entropyCoder_) {}
  size_t memLimit;
  char entropyCoder;
};

namespace detail {
inline size_t writePacked(uint64 v, OutStream* out) {             // utils::packInteger
  size_t n = 0;
  do { byte b = (byte)(v & 0x7F); v >>= 7; if (v) b |= 0x80; out->writeByte(b); ++n; } while (v);
  return n;
}
}  // namespace detail

class Compressor {
 public:
  // takes ownership of the streams, like the reference (Compressor.cpp:49-53)
  Compressor(InStream* in, OutStream* out, size_t memLimit, char entropyCoder)
      : m_in(in), m_out(out), m_coder(giveEntropyEncoder(entropyCoder)),
        m_options(memLimit, entropyCoder) {}
  Compressor(InStream* in, OutStream* out, const std::string& preprocessing, size_t memLimit, char entropyCoder)   // Compressor.cpp:44-47
      : m_in(in), m_out(out), m_coder(giveEntropyEncoder(entropyCoder)), m_precompressor(preprocessing),
        m_options(memLimit, entropyCoder) {}
  ~Compressor() { delete m_in; delete m_out; delete m_coder; }

  size_t bwtBlockSize() const {                                   // Compressor.cpp:77-79
    return std::min(static_cast<size_t>(m_options.memLimit * 0.185),
                    static_cast<size_t>(0x7fffffff - 1));
  }
  void initializeBwtAlgorithm(char choice, uint32 startingPoints, int device = 0) {
    // with the pre-stage a BWT block is (memLimit - precompressed size) / 4.5 bytes (Compressor.cpp:94-97): at most memLimit / 4.5
    const size_t largest = m_precompressor.options().empty() ? bwtBlockSize()
        : std::min(static_cast<size_t>(m_options.memLimit / 4.5) + 1, static_cast<size_t>(0x7fffffff - 1));
    m_bwtmanager.setMaxBlockSize((uint32)largest);
    m_bwtmanager.setDevice(device);
    m_bwtmanager.initialize(choice);                              // Compressor.cpp:60-63
    m_bwtmanager.setStartingPoints(startingPoints);
  }
  size_t writeGlobalHeader() { m_out->writeByte((byte)m_options.entropyCoder); return 1; }
  // The blocks of the stream farmed over several contexts, one per listed device (the same device
  // may be listed more than once): bwtc_hip_farm.hpp.  Same bytes as compress().
  size_t compressFarmed(const std::vector<int>& devices, uint32 startingPoints, unsigned depth = 0);   // depth: blocks under way per context (0 = 12)

  // Compressor.cpp:65-118 with no precompression: every read of bwtBlockSize bytes is one
  // precompressor block holding one BWT block (pbBlockSize == bwtBlockSize, :81).
  size_t compress(size_t threads) {
    if (threads != 1) { std::fprintf(stderr, "Supporting only single thread!\n"); return 0; }
    if (!m_precompressor.options().empty()) return compressPrecompressed();
    size_t compressedSize = writeGlobalHeader();
    const size_t bs = bwtBlockSize();
    std::vector<byte> buf(bs + 1);
    std::vector<size_t> sizes;                                    // blocks under way (overlapping coders)
    for (;;) {
      const size_t got = m_in->readBlock(&buf[0], bs);
      if (got == 0) break;
      BWTBlock block(&buf[0], (uint32)got, false);
      if (m_coder->overlapsBlocks()) {
        // the device half of this block runs while worker threads still code earlier blocks;
        // headers and records reach the stream in block order
        while (m_coder->pending() >= m_coder->inFlightLimit()) { compressedSize += writeFinished(sizes.front()); sizes.erase(sizes.begin()); }
        m_coder->begin(block, m_bwtmanager);
        sizes.push_back(got);
        continue;
      }
      compressedSize += writeBlockPrefix(got);
      compressedSize += m_coder->transformAndEncode(block, m_bwtmanager, m_out);
    }
    while (!sizes.empty()) { compressedSize += writeFinished(sizes.front()); sizes.erase(sizes.begin()); }
    m_out->writeByte(0); ++compressedSize;                        // writeEmptyHeader, :115
    m_out->flush();
    return compressedSize;
  }
  OutStream* out() { return m_out; }
 private:
  // Compressor.cpp:65-118 with the pre-stage on: precompressor blocks of 0.74 memLimit bytes, each precompressed
  // (its grammar goes into the block's header), then cut into BWT blocks of (memLimit - its size) / 4.5 bytes.
  // Block by block: the coders' overlap is for the streams without a pre-stage.
  size_t compressPrecompressed() {
    size_t compressedSize = writeGlobalHeader();
    const size_t pbBlockSize = static_cast<size_t>(m_options.memLimit * 0.74);
    if (pbBlockSize == 0) { m_out->writeByte(0); m_out->flush(); return compressedSize + 1; }
    std::vector<byte> buf(pbBlockSize + 1);
    for (;;) {
      const size_t got = m_in->readBlock(&buf[0], pbBlockSize);
      if (got == 0) break;
      Grammar grammar;
      const size_t len = m_precompressor.precompress(m_bwtmanager.hipContext(), grammar, &buf[0], got);
      const size_t bs = std::min(static_cast<size_t>((m_options.memLimit - len) / 4.5), static_cast<size_t>(0x7fffffff - 1));
      if (bs == 0) { std::fprintf(stderr, "bwtc-hip: memory limit too small for the precompressed block\n"); std::exit(1); }
      const size_t slices = (len + bs - 1) / bs;                  // PrecompressorBlock::sliceIntoBlocks
      compressedSize += detail::writePacked(got, m_out);          // PrecompressorBlock::writeBlockHeader
      compressedSize += detail::writePacked(slices, m_out);
      compressedSize += grammar.writeGrammar(m_out);
      for (size_t s = 0; s < slices; ++s) {
        const size_t at = s * bs, size = std::min(bs, len - at);
        const byte keep = buf[at + size];                         // the slice's sentinel slot is its neighbour's first byte
        BWTBlock block(&buf[at], (uint32)size, false);
        if (m_coder->overlapsBlocks()) { m_coder->begin(block, m_bwtmanager); compressedSize += m_coder->finishOldest(m_out); }
        else compressedSize += m_coder->transformAndEncode(block, m_bwtmanager, m_out);
        buf[at + size] = keep;
      }
    }
    m_out->writeByte(0); ++compressedSize;
    m_out->flush();
    return compressedSize;
  }
  size_t writeBlockPrefix(size_t got) {                           // PrecompressorBlock.cpp:64-90
    size_t n = detail::writePacked(got, m_out);
    n += detail::writePacked(1, m_out);
    m_out->writeByte(0);                                          // empty grammar
    return n + 1;
  }
  size_t writeFinished(size_t got) {
    const size_t n = writeBlockPrefix(got);
    return n + m_coder->finishOldest(m_out);
  }
  InStream* m_in;
  OutStream* m_out;
  EntropyEncoder* m_coder;
  BWTManager m_bwtmanager;
  Precompressor m_precompressor;
  Options m_options;
};

}  // namespace bwtc
