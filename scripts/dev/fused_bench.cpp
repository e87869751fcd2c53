// Development: nanoseconds per coded element of the fused model + range-coder lane engine
// (wavelet_simd.cpp, runFusedLanes) on synthetic sections, one thread.
//   g++ -O2 -std=c++17 -Ibwtc_amd/csrc scripts/dev/fused_bench.cpp bwtc_amd/csrc/wavelet_simd.cpp \
//       bwtc_amd/csrc/wavelet_rc.cpp bwtc_amd/csrc/wavelet_host.cpp bwtc_amd/csrc/entropy_host.cpp -lpthread -o /tmp/fused_bench
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "wavelet_host.hpp"
#include "wavelet_simd.hpp"

using namespace bwtc::wavelet;

struct Src : SectionSource {
  std::vector<FusedSectionDesc> secs;
  size_t at = 0, finished = 0;
  bool next(FusedSectionDesc* s) override { if (at >= secs.size()) return false; *s = secs[at++]; return true; }
  void done(void*) override { ++finished; }
};

int main(int argc, char** argv) {
  const size_t nsec = argc > 1 ? atoi(argv[1]) : 64;
  const size_t per = (argc > 2 ? atoi(argv[2]) : 4) << 20;       // elements per section
  std::mt19937_64 rng(7);
  std::vector<std::vector<uint8_t> > codes(nsec);
  std::vector<std::vector<FusedGroup> > groups(nsec);
  std::vector<std::vector<uint8_t> > outs(nsec);
  Src src;
  for (size_t s = 0; s < nsec; ++s) {
    codes[s].assign(per / 4 + 64, 0);
    for (size_t i = 0; i < per / 4; ++i) {
      uint8_t b = 0;
      for (int k = 0; k < 4; ++k) { const uint64_t r = rng(); b |= (uint8_t)((((r & 7) == 0) ? 1 : 0) | (((r >> 8) & 3) == 0 ? 2 : 0)) << (2 * k); }
      codes[s][i] = b;
    }
    // groups of mixed types and lengths, like a section's nodes and integer levels
    uint64_t at = 0;
    int t = 0;
    while (at < per) {
      uint64_t len = 1000 + rng() % (per / 6);
      if (at + len > per) len = per - at;
      FusedGroup g;
      g.begin = at; g.end = at + len;
      static const uint8_t types[] = {kRoot, kInner, kLeftLeaf, kBothLeaves, kInteger, kInner, kInteger};
      g.type = types[t++ % 7];
      g.mainState = (uint8_t)(rng() % 8);
      groups[s].push_back(g);
      at += len;
    }
    outs[s].assign(8, 0);
    FusedSectionDesc d;
    d.codes = codes[s].data(); d.groups = groups[s].data(); d.n_groups = groups[s].size(); d.out = &outs[s]; d.cookie = nullptr;
    src.secs.push_back(d);
  }
  const auto t0 = std::chrono::steady_clock::now();
  runFusedLanes(src);
  const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
  size_t bytes = 0;
  uint64_t sum = 0;
  for (size_t s = 0; s < nsec; ++s) { bytes += outs[s].size(); for (uint8_t b : outs[s]) sum = sum * 31 + b; }
  std::printf("%zu sections x %zu elements: %.3f ns per element, %.2f ns per vector step, %zu bytes out, checksum %016llx\n",
              nsec, per, ns / (double)(nsec * per), ns / (double)(nsec * per) * 16.0, bytes, (unsigned long long)sum);
  return 0;
}
