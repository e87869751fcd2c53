// Stage 2 of the stream coder: BitEncoder (BitCoders.cpp:59-113) over a section's elements and
// their probabilities -- the one chain of the 'B' format that cannot be split.  Three ways to
// run chains, all with the same integer arithmetic per element (so the bytes are the same):
//   runChain       one chain, scalar
//   runChainPair   two chains stepped alternately by one thread (they overlap in the core)
//   runCoderLanes  sixteen chains at a time, one per AVX-512 lane, lanes refilled from a source
//                  that spans the blocks under way (the counterpart of runModelLanes)
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace bwtc {
namespace wavelet {

// The interval is kept as (low, size = high - low - 1), all modulo 2^32 like the reference's
// high/low: with t = (size * p + 2048) >> 12 (the reference's two-part product, exact in 64
// bits) a one bit gives size = t - 1 and a zero bit low += t + 1, size -= t + 1, so the
// loop-carried chain is one multiply and a subtraction.
struct CoderChain {
  uint32_t lo, size;
  uint64_t i, e;
  std::vector<uint8_t>* out;
  size_t used;
  void start(uint64_t b, uint64_t end, std::vector<uint8_t>* o) {
    lo = 0; size = 0xFFFFFFFEu; i = b; e = end; out = o; used = o->size();
    // room for a byte per seven elements at once (a text block's sections take one per ten): the 27 MB of
    // the longest section were otherwise reached through a dozen reallocations, each copying what was there
    if (end - b >= (1u << 16)) o->reserve(used + (end - b) / 7 + 4096);
  }
  uint8_t* room(uint64_t elements) {                                   // at most four bytes leave the coder per bit
    if (out->size() - used < 4 * elements + 8) out->resize(used + 4 * elements + 8 + out->size() / 2);
    return out->data() + used;
  }
  void finish() {                                                      // BitEncoder::finish
    out->resize(used + 4);
    uint8_t* o = out->data() + used;
    o[0] = static_cast<uint8_t>(lo >> 24); o[1] = 255; o[2] = 255; o[3] = 255;
  }
};

void runChain(CoderChain& c, const uint8_t* codes, const uint16_t* prob, uint64_t until);
// The same chain over elements that arrive as  w = bit << 15 | probability of the CODED bit
// (wavelet_gpu_models.hpp): with m = w & 0x7FFF the next size is  (size * m + bit - 2049) >> 12
// (arithmetic shift) for either bit value -- t - 1 = floor((size p - 2048) / 4096) for a one,
// size - t - 1 = floor((size (4096 - p) - 2049) / 4096) for a zero -- so the loop-carried chain is
// a multiply, an add and a shift.  Same bytes as runChain.
void runChainW(CoderChain& c, const uint16_t* w, uint64_t until);
void runChainPairW(CoderChain& a, const uint16_t* wa, CoderChain& b, const uint16_t* wb, uint64_t limit);
// both chains read the same block's codes/prob; runs until the shorter chain ends
void runChainPair(CoderChain& a, CoderChain& b, const uint8_t* codes, const uint16_t* prob);
// the same for chains of different blocks, at most `limit` elements of each
void runChainPair(CoderChain& a, const uint8_t* codes_a, const uint16_t* prob_a,
                  CoderChain& b, const uint8_t* codes_b, const uint16_t* prob_b, uint64_t limit);

// One section's chain for the lane engine: elements [begin, end) of a block's packed streams
// (2 bits per element, bit 0 = the coded bit) and probabilities; `out` already holds the
// section's prefix (packed bitsInRoot + shape) and receives the coded bytes and the 4-byte flush.
struct ChainDesc {
  const uint8_t* codes;
  const uint16_t* prob;
  uint64_t begin, end;
  std::vector<uint8_t>* out;
  void* cookie;              // handed back through done()
  const uint16_t* w = nullptr;   // not null: the elements are w-words (wavelet_gpu_models.hpp), codes / prob unused;
                                 //   one source hands out chains of one kind only
};

class ChainSource {
 public:
  virtual ~ChainSource() {}
  virtual bool next(ChainDesc* c) = 0;          // false: none available now
  virtual void done(void* cookie) = 0;          // the chain is coded and flushed
  virtual bool retire() { return true; }        // an engine with nothing left asks before it returns
};

// AVX-512 (F, BW, VL, DQ) present and not switched off with BWTC_HIP_SIMD=0 / BWTC_HIP_CODER_LANES=0
bool simdCoderAvailable();

// Codes chains from `src` until it is empty and every lane has drained.  max_lanes = 2: the
// scalar engine (two chains stepped alternately, no vectors), for chains so long that a lane --
// which advances a chain three to four times slower than the scalar loop does -- would keep
// their block waiting for seconds.
// busy_ns (optional): the engine's running time is added there as it goes.
void runCoderLanes(ChainSource& src, int max_lanes = 16, std::atomic<uint64_t>* busy_ns = nullptr);

}  // namespace wavelet
}  // namespace bwtc
