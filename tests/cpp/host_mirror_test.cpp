// GPU test program for the C++ host mirror (bwtc_amd/host/bwtc_hip.hpp), shaped after the
// reference's own tests:
//   raw transform + inverse on random data      test/InverseBwtTest.cpp:51-114
//   LF powers against an independently built SA  test/LFpowersTest.cpp:49-181 (via the oracle)
//   Compressor -> decompressor round trips       test/CompressorAndDecompressorTest.cpp:61-205
// TEST INFRASTRUCTURE: links the oracle (liboracle.so) as the checker / decoder.
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "bwtc_hip.hpp"
#include "bwtc_hip_decode.hpp"
#include "../../oracle/bwtc_oracle.h"

using namespace bwtc;

static int failures = 0;
#define CHECK(c, ...) do { if (!(c)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static std::vector<byte> randomData(std::mt19937& rng, size_t n, int sigma) {
  std::vector<byte> d(n);
  for (size_t i = 0; i < n; ++i) d[i] = (byte)(rng() % sigma);
  return d;
}
static std::vector<byte> repetitiveData(std::mt19937& rng, size_t n) {   // :52-59
  std::vector<byte> seed = randomData(rng, n / 40 + 1, 256), d;
  while (d.size() < n) d.insert(d.end(), seed.begin(), seed.end());
  d.resize(n);
  return d;
}

static void testRawTransformAndInverse(std::mt19937& rng) {
  HipBWTransform transform(1u << 20);
  for (int round = 0; round < 12; ++round) {
    const size_t n = 1 + rng() % 200000;
    const int sigma = (round % 3 == 0) ? 4 : 256;
    std::vector<byte> orig = randomData(rng, n, sigma);
    std::vector<byte> t(orig.rbegin(), orig.rend());              // caller reverses ...
    t.push_back(0);                                               // ... and plants the sentinel
    const uint32 sp = (uint32)(1 + rng() % std::min<size_t>(256, n));
    std::vector<uint32> LF(sp), LFo(sp);
    uint32 freqs[256] = {0}, freqso[256] = {0};
    std::vector<byte> o = t;
    transform.doTransform(&t[0], (uint32)n + 1, LF, freqs);
    orc_bwt_raw(&o[0], (uint32)n + 1, &LFo[0], sp, freqso);
    CHECK(t == o, "raw transform bytes differ (n=%zu)", n);
    CHECK(LF == LFo, "LF powers differ (n=%zu sp=%u)", n, sp);
    CHECK(std::memcmp(freqs, freqso, sizeof freqs) == 0, "freqs differ");
  }
}

static void testBlockTransformViaManager(std::mt19937& rng) {
  BWTManager manager(8);
  manager.setMaxBlockSize(1u << 20);
  manager.initialize('g');
  for (int round = 0; round < 8; ++round) {
    const size_t n = 1 + rng() % 300000;
    manager.setStartingPoints(1 + rng() % 300);
    std::vector<byte> data = round % 2 ? repetitiveData(rng, n) : randomData(rng, n, 256);
    std::vector<byte> buf(data);
    buf.push_back(0xEE);
    BWTBlock block(&buf[0], (uint32)n, false);
    uint32 freqs[256] = {0};
    manager.doTransform(block, freqs);
    CHECK(block.isTransformed(), "block not marked transformed");
    CHECK(buf[n] == 0xEE, "byte after the block was modified");
    std::vector<byte> inv(buf.begin(), buf.begin() + n);
    const int rc = orc_inverse_bwt_block(&inv[0], (uint32)n, &block.LFpowers()[0], (uint32)block.LFpowers().size());
    CHECK(rc == 0, "inverse failed rc=%d", rc);
    CHECK(inv == data, "inverse does not reproduce the input");
  }
}

// test/InverseBwtTest.cpp:51-114 with both directions on the GPU
static void testInverseTransformer(std::mt19937& rng) {
  BWTManager manager(8);
  manager.setMaxBlockSize(1u << 20);
  manager.initialize('g');
  InverseBWTransform* inverse = giveInverseTransformer(1u << 20);
  for (int round = 0; round < 8; ++round) {
    const size_t n = 1 + rng() % 500000;
    manager.setStartingPoints(1 + rng() % 256);
    std::vector<byte> data = round % 2 ? repetitiveData(rng, n) : randomData(rng, n, round % 4 ? 256 : 3);
    std::vector<byte> buf(data);
    buf.push_back(0x77);
    BWTBlock block(&buf[0], (uint32)n, false);
    manager.doTransform(block);
    inverse->doTransform(block);
    CHECK(std::equal(data.begin(), data.end(), buf.begin()), "GPU inverse(GPU forward) != input (n=%zu)", n);
    CHECK(buf[n] == 0x77, "byte after the block was modified");
  }
  delete inverse;
}

static void testCompressorRoundTrip(std::mt19937& rng) {
  struct Case { size_t size, mem; unsigned sp; bool rep; };
  const Case cases[] = {{100, 1000, 1, false},     {1000, 2000, 3, true},      {100000, 1000000, 8, false},
                        {100000, 100000, 30, true}, {700000, 1000000, 8, true}, {3000000, 100000000, 16, false}};
  for (const Case& c : cases) {
    std::vector<byte> data = c.rep ? repetitiveData(rng, c.size) : randomData(rng, c.size, 256);
    MemoryOutStream* out = new MemoryOutStream();
    Compressor compressor(new MemoryInStream(&data[0], data.size()), out, c.mem, 'H');
    compressor.initializeBwtAlgorithm('g', c.sp);
    const size_t reported = compressor.compress(1);
    CHECK(reported == out->data.size(), "compress() returned %zu, stream holds %zu", reported, out->data.size());
    // the oracle's own compressor must produce the very same stream ...
    std::vector<byte> want(orc_compress_bound(data.size()) + 100000 * 8);
    const size_t wn = orc_compress_H(&data[0], data.size(), compressor.bwtBlockSize(), c.sp, &want[0], want.size());
    CHECK(wn == out->data.size() && std::memcmp(&want[0], &out->data[0], wn) == 0,
          "stream differs from the oracle's (size %zu mem %zu)", c.size, c.mem);
    // ... and decode back to the input
    std::vector<byte> back(data.size() + 16);
    const size_t bn = orc_decompress_H(&out->data[0], out->data.size(), &back[0], back.size());
    CHECK(bn == data.size() && std::memcmp(&back[0], &data[0], bn) == 0, "round trip failed (size %zu)", c.size);
  }
}

// The 'B' coder through the mirror: WaveletEncoder block by block (transformAndEncode) and
// through Compressor::compress, which overlaps the blocks; many small blocks so that the
// model state is carried across a dozen block borders while up to eight blocks are under way.
static void testWaveletCompressor(std::mt19937& rng) {
  struct Case { size_t size, mem; unsigned sp; bool rep; };
  const Case cases[] = {{11, 1000, 1, false},        {1000, 2000, 3, true},      {100000, 1000000, 8, false},
                        {2000000, 800000, 8, true},  {3000000, 100000000, 16, false}, {5000000, 1500000, 4, true}};
  for (const Case& c : cases) {
    std::vector<byte> data = c.rep ? repetitiveData(rng, c.size) : randomData(rng, c.size, 200);
    MemoryOutStream* out = new MemoryOutStream();
    Compressor compressor(new MemoryInStream(&data[0], data.size()), out, c.mem, 'B');
    compressor.initializeBwtAlgorithm('g', c.sp);
    const size_t reported = compressor.compress(1);
    CHECK(reported == out->data.size(), "'B' compress() returned %zu, stream holds %zu", reported, out->data.size());
    std::vector<byte> want(orc_compress_bound(data.size()) + 100000 * 8);
    const size_t wn = orc_compress_B(&data[0], data.size(), compressor.bwtBlockSize(), c.sp, &want[0], want.size());
    CHECK(wn == out->data.size() && std::memcmp(&want[0], &out->data[0], wn) == 0,
          "'B' stream differs from the oracle's (size %zu mem %zu: %zu vs %zu bytes)", c.size, c.mem, out->data.size(), wn);
  }
  // block by block through the encoder object, as the reference's compress loop calls it
  {
    std::vector<byte> data = repetitiveData(rng, 900000);
    const size_t bs = 250000;
    BWTManager manager(8);
    manager.setMaxBlockSize((uint32)bs);
    manager.initialize('g');
    MemoryOutStream out;
    out.writeByte('B');
    EntropyEncoder* enc = giveEntropyEncoder('B');
    std::vector<byte> buf(bs + 1);
    for (size_t off = 0; off < data.size(); off += bs) {
      const size_t n = std::min(bs, data.size() - off);
      std::memcpy(&buf[0], &data[off], n);
      detail::writePacked(n, &out); detail::writePacked(1, &out); out.writeByte(0);
      BWTBlock block(&buf[0], (uint32)n, false);
      enc->transformAndEncode(block, manager, &out);
      CHECK(block.isTransformed(), "block not marked transformed");
    }
    out.writeByte(0);
    delete enc;
    std::vector<byte> want(orc_compress_bound(data.size()) + 100000);
    const size_t wn = orc_compress_B(&data[0], data.size(), bs, 8, &want[0], want.size());
    CHECK(wn == out.data.size() && std::memcmp(&want[0], &out.data[0], wn) == 0, "'B' block-by-block stream differs");
  }
}

// WaveletDecoder: streams written by the oracle's restatement of the reference's 'B' encoder
// (golden stream included) and by this framework's own compressor decode back to the input --
// single symbols, one-run blocks, long runs (escape codes), 256 sections, several blocks.
static void testWaveletDecompressor(std::mt19937& rng) {
  std::vector<std::vector<byte> > inputs;
  const char* golden = "abracadabra";
  inputs.push_back(std::vector<byte>(golden, golden + 11));
  inputs.push_back(std::vector<byte>(1, 'x'));
  inputs.push_back(std::vector<byte>(70001, 7));
  inputs.push_back(randomData(rng, 300000, 256));
  inputs.push_back(randomData(rng, 200000, 3));
  inputs.push_back(repetitiveData(rng, 1500000));
  { std::vector<byte> runs;                                           // long runs: lengths up to 5000
    while (runs.size() < 2000000) runs.insert(runs.end(), 1 + rng() % 5000, (byte)(rng() % 5));
    inputs.push_back(runs); }
  { std::vector<byte> skew(800000);                                  // geometric symbols, short runs
    for (size_t i = 0; i < skew.size(); ++i) { unsigned v = 0; while ((rng() & 3) && v < 200) ++v; skew[i] = (byte)v; }
    inputs.push_back(skew); }
  const size_t blockSizes[] = {1u << 30, 400000};
  for (size_t k = 0; k < inputs.size(); ++k) {
    const std::vector<byte>& data = inputs[k];
    for (size_t b = 0; b < 2; ++b) {
      const size_t bs = std::min(blockSizes[b], std::max<size_t>(data.size(), 1));
      std::vector<byte> stream(orc_compress_bound(data.size()) + 100000 * 8);
      const size_t sn = orc_compress_B(&data[0], data.size(), bs, 8, &stream[0], stream.size());
      MemoryOutStream* plain = new MemoryOutStream();
      Decompressor d(new MemoryInStream(&stream[0], sn), plain);
      const size_t n = d.decompress(1);
      CHECK(n == data.size() && plain->data == data, "'B' oracle stream -> Decompressor (input %zu, block %zu)", k, bs);
    }
  }
  for (const char* letter = "bu"; *letter; ++letter) {                // the other model letters, several blocks
    std::vector<byte> data = repetitiveData(rng, 1200000);
    MemoryOutStream* packed = new MemoryOutStream();
    Compressor compressor(new MemoryInStream(&data[0], data.size()), packed, 2000000, *letter);
    compressor.initializeBwtAlgorithm('g', 8);
    compressor.compress(1);
    std::vector<byte> stream = packed->data;
    std::vector<byte> want(orc_compress_bound(data.size()) + 100000);
    const size_t wn = orc_compress_wavelet(*letter, &data[0], data.size(), compressor.bwtBlockSize(), 8, &want[0], want.size());
    CHECK(wn == stream.size() && std::memcmp(&want[0], &stream[0], wn) == 0, "'%c' stream differs from the oracle's", *letter);
    MemoryOutStream* plain = new MemoryOutStream();
    Decompressor d(new MemoryInStream(&stream[0], stream.size()), plain);
    const size_t n = d.decompress(1);
    CHECK(n == data.size() && plain->data == data, "'%c' Compressor -> Decompressor round trip", *letter);
  }
  {                                                                   // own compressor -> own decompressor
    std::vector<byte> data = repetitiveData(rng, 3000000);
    MemoryOutStream* packed = new MemoryOutStream();
    Compressor compressor(new MemoryInStream(&data[0], data.size()), packed, 3000000, 'B');
    compressor.initializeBwtAlgorithm('g', 8);
    compressor.compress(1);
    std::vector<byte> stream = packed->data;
    MemoryOutStream* plain = new MemoryOutStream();
    Decompressor d(new MemoryInStream(&stream[0], stream.size()), plain);
    const size_t n = d.decompress(1);
    CHECK(n == data.size() && plain->data == data, "'B' Compressor -> Decompressor round trip");
  }
}

// Compressor -> Decompressor entirely through this framework (GPU forward + inverse, host
// entropy coder), and Decompressor on streams produced by the oracle's restatement of the
// reference's Compressor, including the golden stream recorded from the reference itself.
static void testDecompressor(std::mt19937& rng) {
  struct Case { size_t size, mem; unsigned sp; bool rep; };
  const Case cases[] = {{1, 1000, 1, false}, {300, 1000, 2, true}, {100000, 200000, 8, false},
                        {1500000, 2000000, 30, true}, {4000000, 100000000, 8, false}};
  for (const Case& c : cases) {
    std::vector<byte> data = c.rep ? repetitiveData(rng, c.size) : randomData(rng, c.size, c.size % 2 ? 5 : 256);
    MemoryOutStream* packed = new MemoryOutStream();
    {
      Compressor compressor(new MemoryInStream(&data[0], data.size()), packed, c.mem, 'H');
      compressor.initializeBwtAlgorithm('g', c.sp);
      compressor.compress(1);
      std::vector<byte> stream = packed->data;              // copy before the compressor dies
      MemoryOutStream* plain = new MemoryOutStream();
      Decompressor d(new MemoryInStream(&stream[0], stream.size()), plain);
      const size_t n = d.decompress(1);
      CHECK(n == data.size() && plain->data == data, "Compressor->Decompressor round trip (size %zu)", c.size);
    }
    // a stream written by the oracle's compressor
    std::vector<byte> ostream(orc_compress_bound(data.size()) + 100000 * 8);
    const size_t bs = std::min(static_cast<size_t>(c.mem * 0.185), static_cast<size_t>(0x7fffffff - 1));
    ostream.resize(orc_compress_H(&data[0], data.size(), bs, c.sp, &ostream[0], ostream.size()));
    MemoryOutStream* plain = new MemoryOutStream();
    Decompressor d(new MemoryInStream(&ostream[0], ostream.size()), plain);
    d.decompress(1);
    CHECK(plain->data == data, "Decompressor on the oracle's stream (size %zu)", c.size);
  }
  // the reference's own 31-byte stream for "abracadabra"
  const byte golden[] = {0x48, 0x0b, 0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x14, 0x00, 0x00, 0x00, 0x00, 0x0a, 0x01,
                         0x0b, 0x08, 0x72, 0x05, 0x03, 0xa1, 0x0c, 0x2d, 0x74, 0x63, 0x07, 0x40, 0xfd, 0x30, 0x00};
  MemoryOutStream* plain = new MemoryOutStream();
  Decompressor d(new MemoryInStream(golden, sizeof golden), plain);
  d.decompress(1);
  CHECK(plain->data.size() == 11 && std::memcmp(&plain->data[0], "abracadabra", 11) == 0,
        "reference golden stream does not decode to abracadabra");
}

int main() {
  std::mt19937 rng(12345);
  testRawTransformAndInverse(rng);
  testBlockTransformViaManager(rng);
  testInverseTransformer(rng);
  testCompressorRoundTrip(rng);
  testWaveletCompressor(rng);
  testWaveletDecompressor(rng);
  testDecompressor(rng);
  std::printf(failures ? "%d FAILURES\n" : "host mirror: all tests passed\n", failures);
  return failures ? 1 : 0;
}
