// CPU-only check of the host WaveletDecoder ('B'): blocks transformed and encoded by the
// oracle's restatement of the reference encoder must decode to the oracle's transformed
// bytes and LF powers.  No device call is made (the inverse transform is not part of this
// program), so it runs in the CPU test suite.
// TEST INFRASTRUCTURE: links the oracle (liboracle.so) as encoder and checker.
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

int main() {
  std::mt19937 rng(777);
  std::vector<std::vector<byte> > inputs;
  const char* golden = "abracadabra";
  inputs.push_back(std::vector<byte>(golden, golden + 11));
  inputs.push_back(std::vector<byte>(1, 'x'));
  inputs.push_back(std::vector<byte>(30000, 65));
  { std::vector<byte> d(120000); for (size_t i = 0; i < d.size(); ++i) d[i] = (byte)rng(); inputs.push_back(d); }
  { std::vector<byte> d(90000); for (size_t i = 0; i < d.size(); ++i) d[i] = (byte)("ACGT"[rng() & 3]); inputs.push_back(d); }
  { std::vector<byte> d; while (d.size() < 400000) d.insert(d.end(), 1 + rng() % 3000, (byte)(rng() % 6)); inputs.push_back(d); }
  { std::vector<byte> seed(5000), d; for (size_t i = 0; i < seed.size(); ++i) seed[i] = (byte)(rng() % 40 + 60);
    while (d.size() < 300000) d.insert(d.end(), seed.begin(), seed.end());
    inputs.push_back(d); }
  { std::vector<byte> d(150000); for (size_t i = 0; i < d.size(); ++i) { unsigned v = 0; while ((rng() & 3) && v < 250) ++v; d[i] = (byte)v; } inputs.push_back(d); }

  WaveletDecoder decoder('B');                       // one decoder: the model state runs on from block to block
  // every input as a one-block stream of its own (fresh encoder, fresh decoder)
  for (size_t k = 0; k < inputs.size(); ++k) {
    std::vector<byte> t = inputs[k];
    t.push_back(0);
    const uint32 size = (uint32)inputs[k].size();
    std::vector<uint32> lf(256);
    uint32 n_lf = 0, freqs[256];
    std::memset(freqs, 0, sizeof freqs);
    CHECK(orc_bwt_block(&t[0], size, 8, &lf[0], &n_lf, freqs) == 0, "oracle transform");
    lf.resize(n_lf);
    std::vector<byte> whole(orc_compress_bound(size) + 4096);
    const size_t sn = orc_compress_B(&inputs[k][0], size, size, 8, &whole[0], whole.size());
    CHECK(sn > 0 && whole[0] == 'B', "oracle stream");
    MemoryBitReader in(&whole[0], sn);
    CHECK(in.readByte() == 'B', "global header");
    CHECK(in.readPackedInteger() == size, "precompressor block size");
    CHECK(in.readPackedInteger() == 1, "slice count");
    CHECK(in.readByte() == 0, "grammar");
    std::vector<byte> out(size + 16);
    BWTBlock block(&out[0], 0, true);
    WaveletDecoder fresh('B');
    fresh.decodeBlock(block, in, size);
    CHECK(block.size() == size, "decoded size %u, want %u (input %zu)", (unsigned)block.size(), size, k);
    CHECK(std::memcmp(&out[0], &t[0], size) == 0, "decoded bytes differ from the oracle's BWT (input %zu)", k);
    CHECK(block.LFpowers().size() == n_lf && std::memcmp(&block.LFpowers()[0], &lf[0], n_lf * 4) == 0, "LF powers (input %zu)", k);
    CHECK(in.readByte() == 0, "end marker after the block (input %zu)", k);
  }
  // several blocks through one encoder and one decoder: the model state crosses block borders
  {
    std::vector<byte> data;
    for (int r = 0; r < 6; ++r) data.insert(data.end(), inputs[6].begin(), inputs[6].end());
    const size_t bs = 170000;
    std::vector<byte> whole(orc_compress_bound(data.size()) + 100000);
    const size_t sn = orc_compress_B(&data[0], data.size(), bs, 4, &whole[0], whole.size());
    MemoryBitReader in(&whole[0], sn);
    in.readByte();
    size_t off = 0;
    for (;;) {
      const uint64 n = in.readPackedInteger();
      if (n == 0) break;
      CHECK(in.readPackedInteger() == 1 && in.readByte() == 0, "block prefix");
      std::vector<byte> t(data.begin() + off, data.begin() + off + n);
      t.push_back(0);
      std::vector<uint32> lf(256);
      uint32 n_lf = 0, freqs[256];
      std::memset(freqs, 0, sizeof freqs);
      orc_bwt_block(&t[0], (uint32)n, 4, &lf[0], &n_lf, freqs);
      std::vector<byte> out(n + 16);
      BWTBlock block(&out[0], 0, true);
      decoder.decodeBlock(block, in, n);
      CHECK(block.size() == n && std::memcmp(&out[0], &t[0], n) == 0, "multi-block stream, block at %zu", off);
      off += n;
    }
    CHECK(off == data.size(), "multi-block stream length");
  }
  // the other model letters: 'b' and 'u' streams of the oracle through decoders of that letter
  for (const char* letter = "bu"; *letter; ++letter) {
    for (size_t k = 0; k < inputs.size(); k += 2) {
      const std::vector<byte>& data = inputs[k];
      const size_t bs = std::max<size_t>(data.size() / 2 + 1, 1);   // two blocks
      std::vector<byte> whole(orc_compress_bound(data.size()) + 100000);
      const size_t sn = orc_compress_wavelet(*letter, &data[0], data.size(), bs, 8, &whole[0], whole.size());
      MemoryBitReader in(&whole[0], sn);
      CHECK(in.readByte() == (byte)*letter, "global header letter");
      WaveletDecoder dec(*letter);
      size_t off = 0;
      for (;;) {
        const uint64 n = in.readPackedInteger();
        if (n == 0) break;
        CHECK(in.readPackedInteger() == 1 && in.readByte() == 0, "block prefix");
        std::vector<byte> t(data.begin() + off, data.begin() + off + n);
        t.push_back(0);
        std::vector<uint32> lf(256);
        uint32 n_lf = 0, freqs[256];
        std::memset(freqs, 0, sizeof freqs);
        orc_bwt_block(&t[0], (uint32)n, 8, &lf[0], &n_lf, freqs);
        std::vector<byte> out(n + 16);
        BWTBlock block(&out[0], 0, true);
        dec.decodeBlock(block, in, n);
        CHECK(block.size() == n && std::memcmp(&out[0], &t[0], n) == 0, "coder '%c', input %zu, block at %zu", *letter, k, off);
        off += n;
      }
      CHECK(off == data.size(), "coder '%c' stream length (input %zu)", *letter, k);
    }
  }
  // The matrix of test/CompressorAndDecompressorTest.cpp:61-160 across the boundary, reference
  // algorithm -> product: streams written by the oracle's literal encoder (sizes 100..100 000,
  // repetitions 0 / 2 / 50, one and several blocks, 1..30 starting points) through the product's
  // decoder; and each of them through the oracle's literal decoder as well.
  {
    struct Case { size_t length, reps, block; uint32 sp; };
    std::vector<Case> cases;
    const size_t lengths[4] = {100, 1000, 10000, 100000}, repss[3] = {0, 2, 50};
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 3; ++b) {
        Case c1 = {lengths[a], repss[b], (size_t)(lengths[a] * 100 * 0.185), 1};
        Case c2 = {lengths[a], repss[b], std::max<size_t>((size_t)(lengths[a] * (repss[b] != 2 ? 10 : 1) * 0.185), 18), 1};
        cases.push_back(c1); cases.push_back(c2);
      }
    for (uint32 sp = 1; sp <= 30; ++sp) { Case c = {10000, 0, (size_t)(100000 * 0.185), sp}; cases.push_back(c); }
    for (size_t ci = 0; ci < cases.size(); ++ci) {
      const Case& c = cases[ci];
      std::vector<byte> data;
      const size_t unit = c.reps ? c.length / c.reps : c.length;
      for (size_t i = 0; i < unit; ++i) data.push_back((byte)rng());
      for (size_t r = 1; r < c.reps; ++r) for (size_t i = 0; i < unit; ++i) data.push_back(data[i]);
      std::vector<byte> whole(orc_compress_bound(data.size()) + 8192 * (data.size() / c.block + 2));
      const size_t sn = orc_compress_B(&data[0], data.size(), c.block, c.sp, &whole[0], whole.size());
      std::vector<byte> viaOracle(data.size() + 16);
      CHECK(orc_decompress_wavelet(&whole[0], sn, &viaOracle[0], viaOracle.size(), 0) == data.size() &&
            std::memcmp(&viaOracle[0], &data[0], data.size()) == 0, "oracle decoder, case %zu", ci);
      MemoryBitReader in(&whole[0], sn);
      CHECK(in.readByte() == 'B', "global header");
      WaveletDecoder dec('B');
      size_t off = 0;
      for (;;) {
        const uint64 n = in.readPackedInteger();
        if (n == 0) break;
        CHECK(in.readPackedInteger() == 1 && in.readByte() == 0, "block prefix");
        std::vector<byte> t(data.begin() + off, data.begin() + off + n);
        t.push_back(0);
        std::vector<uint32> lf(256);
        uint32 n_lf = 0, freqs[256];
        std::memset(freqs, 0, sizeof freqs);
        orc_bwt_block(&t[0], (uint32)n, c.sp, &lf[0], &n_lf, freqs);
        std::vector<byte> out(n + 16);
        BWTBlock block(&out[0], 0, true);
        dec.decodeBlock(block, in, n);
        CHECK(block.size() == n && std::memcmp(&out[0], &t[0], n) == 0, "matrix case %zu (len %zu reps %zu block %zu sp %u), block at %zu",
              ci, c.length, c.reps, c.block, c.sp, off);
        CHECK(block.LFpowers().size() == n_lf && std::memcmp(&block.LFpowers()[0], &lf[0], n_lf * 4) == 0, "matrix case %zu LF powers", ci);
        off += n;
      }
      CHECK(off == data.size(), "matrix case %zu stream length", ci);
    }
  }
  std::printf(failures ? "%d FAILURES\n" : "wavelet decoder: all tests passed\n", failures);
  return failures ? 1 : 0;
}
