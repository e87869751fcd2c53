// Scratch micro-benchmark behind DESIGN.md's note on the range coder loop: the loop as shipped
// ("branchy") against branch-free byte output with 1-4 interleaved chains, on synthetic
// (bit, probability) streams.  clang -O3 -march=native; EPYC 9575F: 1.92 / 2.63 / 1.70 / 1.89 / 1.92
// ns per element on an easy stream, 2.06 / 2.65 / 1.69 / 1.89 / 1.91 on one with the text block's
// entropy: two chains are the sweet spot (StreamCoder::codeSectionsPaired).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64;
static inline u32 codeAt(const uint8_t* c, u64 i) { return (c[i >> 2] >> ((i & 3) * 2)) & 3u; }

// (a) current
static size_t rc_branchy(const uint8_t* codes, const uint16_t* prob, u64 b, u64 e, uint8_t* out) {
  u32 lo = 0, size = 0xFFFFFFFEu; uint8_t* o = out;
  for (u64 i = b; i < e; ++i) {
    const u32 bit = codeAt(codes, i) & 1u;
    const u32 t = (u32)(((u64)size * prob[i] + 2048u) >> 12);
    lo = bit ? lo : lo + t + 1;
    size = bit ? t - 1 : size - t - 1;
    while (((lo ^ (lo + size + 1)) & 0xFF000000u) == 0) { *o++ = (uint8_t)(lo >> 24); lo <<= 8; size = (size << 8) + 510u; }
  }
  *o++ = (uint8_t)(lo >> 24);
  return o - out;
}
// (b) branch-free first byte
#define STEP(L)                                                                           \
  {                                                                                       \
    const u32 bit = codeAt(codes, i##L) & 1u;                                             \
    const u32 t = (u32)(((u64)size##L * prob[i##L] + 2048u) >> 12);                       \
    lo##L = bit ? lo##L : lo##L + t + 1;                                                  \
    size##L = bit ? t - 1 : size##L - t - 1;                                              \
    const u32 same = ((lo##L ^ (lo##L + size##L + 1)) & 0xFF000000u) == 0;                \
    *o##L = (uint8_t)(lo##L >> 24);                                                       \
    o##L += same;                                                                         \
    lo##L = same ? lo##L << 8 : lo##L;                                                    \
    size##L = same ? (size##L << 8) + 510u : size##L;                                     \
    while (__builtin_expect(same && ((lo##L ^ (lo##L + size##L + 1)) & 0xFF000000u) == 0, 0)) { *o##L++ = (uint8_t)(lo##L >> 24); lo##L <<= 8; size##L = (size##L << 8) + 510u; } \
    ++i##L;                                                                               \
  }
static size_t rc_free1(const uint8_t* codes, const uint16_t* prob, u64 b, u64 e, uint8_t* out) {
  u32 lo0 = 0, size0 = 0xFFFFFFFEu; uint8_t* o0 = out; u64 i0 = b;
  while (i0 < e) STEP(0)
  *o0++ = (uint8_t)(lo0 >> 24);
  return o0 - out;
}
// N lanes over N equal slices (each its own output)
static size_t rc_free2(const uint8_t* codes, const uint16_t* prob, u64 b, u64 e, uint8_t* out, size_t cap) {
  const u64 n = (e - b) / 2;
  u32 lo0 = 0, size0 = 0xFFFFFFFEu, lo1 = 0, size1 = 0xFFFFFFFEu;
  uint8_t *o0 = out, *o1 = out + cap / 2; u64 i0 = b, i1 = b + n;
  for (u64 k = 0; k < n; ++k) { STEP(0) STEP(1) }
  return (o0 - out) + (o1 - (out + cap / 2));
}
static size_t rc_free3(const uint8_t* codes, const uint16_t* prob, u64 b, u64 e, uint8_t* out, size_t cap) {
  const u64 n = (e - b) / 3;
  u32 lo0 = 0, size0 = 0xFFFFFFFEu, lo1 = 0, size1 = 0xFFFFFFFEu, lo2 = 0, size2 = 0xFFFFFFFEu;
  uint8_t *o0 = out, *o1 = out + cap / 3, *o2 = out + 2 * (cap / 3); u64 i0 = b, i1 = b + n, i2 = b + 2 * n;
  for (u64 k = 0; k < n; ++k) { STEP(0) STEP(1) STEP(2) }
  return (o0 - out) + (o1 - (out + cap / 3)) + (o2 - (out + 2 * (cap / 3)));
}
static size_t rc_free4(const uint8_t* codes, const uint16_t* prob, u64 b, u64 e, uint8_t* out, size_t cap) {
  const u64 n = (e - b) / 4;
  u32 lo0 = 0, size0 = 0xFFFFFFFEu, lo1 = 0, size1 = 0xFFFFFFFEu, lo2 = 0, size2 = 0xFFFFFFFEu, lo3 = 0, size3 = 0xFFFFFFFEu;
  uint8_t *o0 = out, *o1 = out + cap / 4, *o2 = out + 2 * (cap / 4), *o3 = out + 3 * (cap / 4);
  u64 i0 = b, i1 = b + n, i2 = b + 2 * n, i3 = b + 3 * n;
  for (u64 k = 0; k < n; ++k) { STEP(0) STEP(1) STEP(2) STEP(3) }
  return (o0 - out) + (o1 - (out + cap / 4)) + (o2 - (out + 2 * (cap / 4))) + (o3 - (out + 3 * (cap / 4)));
}
int main() {
  const u64 n = 48u << 20;
  std::vector<uint8_t> codes(n / 4 + 16, 0); std::vector<uint16_t> prob(n + 16);
  std::mt19937_64 rng(1);
  // skewed probabilities like the coder's: mostly confident predictions
  for (u64 i = 0; i < n; ++i) {
    const u32 r = rng() & 1023;
    // mix tuned to the text block's streams: about 0.77 bits of information per element
    u32 p = r < 420 ? 60 + (rng() % 200) : r < 560 ? 4096 - 60 - (rng() % 200) : 300 + rng() % 3400;
    prob[i] = (uint16_t)p;
    const u32 bit = (rng() & 4095) < p;
    codes[i >> 2] |= bit << ((i & 3) * 2);
  }
  std::vector<uint8_t> out(n + 1024);
  for (int rep = 0; rep < 2; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    size_t a = rc_branchy(codes.data(), prob.data(), 0, n, out.data());
    auto t1 = std::chrono::steady_clock::now();
    size_t b1 = rc_free1(codes.data(), prob.data(), 0, n, out.data());
    auto t2 = std::chrono::steady_clock::now();
    size_t b2 = rc_free2(codes.data(), prob.data(), 0, n, out.data(), out.size());
    auto t3 = std::chrono::steady_clock::now();
    size_t b3 = rc_free3(codes.data(), prob.data(), 0, n, out.data(), out.size());
    auto t4 = std::chrono::steady_clock::now();
    size_t b4 = rc_free4(codes.data(), prob.data(), 0, n, out.data(), out.size());
    auto t5 = std::chrono::steady_clock::now();
    auto ns = [&](auto x, auto y) { return std::chrono::duration<double, std::nano>(y - x).count() / n; };
    std::printf("bytes %zu %zu %zu %zu %zu | ns/elem: branchy %.2f, free1 %.2f, free2 %.2f, free3 %.2f, free4 %.2f\n", a, b1, b2, b3, b4,
                ns(t0, t1), ns(t1, t2), ns(t2, t3), ns(t3, t4), ns(t4, t5));
  }
}
