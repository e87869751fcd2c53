#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64; typedef int64_t i64;
static inline u32 codeAt(const uint8_t* c, u64 i) { return (c[i >> 2] >> ((i & 3) * 2)) & 3u; }
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
// new: w[i] = bit<<15 | m  (m = probability of the coded bit), size' = (size*m + c) >> 12
static size_t rc_fma(const uint16_t* w, u64 b, u64 e, uint8_t* out) {
  u32 lo = 0, size = 0xFFFFFFFEu; uint8_t* o = out;
  for (u64 i = b; i < e; ++i) {
    const u32 x = w[i];
    const u32 bit = x >> 15, m = x & 0x7FFF;
    const i64 c = (i64)bit - 2049;
    const u32 ns = (u32)(((i64)((u64)size * m) + c) >> 12);
    lo = bit ? lo : lo + (size - ns);
    size = ns;
    while (((lo ^ (lo + size + 1)) & 0xFF000000u) == 0) { *o++ = (uint8_t)(lo >> 24); lo <<= 8; size = (size << 8) + 510u; }
  }
  *o++ = (uint8_t)(lo >> 24);
  return o - out;
}
// hi-form: keep lo and hi; one of them changes
static size_t rc_fma2(const uint16_t* w, u64 b, u64 e, uint8_t* out) {
  u32 lo = 0, size = 0xFFFFFFFEu; uint8_t* o = out;
  for (u64 i = b; i < e; ++i) {
    const u32 x = w[i];
    const u32 bit = x >> 15, m = x & 0x7FFF;
    const i64 c = (i64)bit - 2049;
    const u32 ns = (u32)(((i64)((u64)size * m) + c) >> 12);
    lo += (bit - 1) & (size - ns);
    size = ns;
    if (__builtin_expect(size < (1u << 24), 0)) {
      while (((lo ^ (lo + size + 1)) & 0xFF000000u) == 0) { *o++ = (uint8_t)(lo >> 24); lo <<= 8; size = (size << 8) + 510u; }
    }
  }
  *o++ = (uint8_t)(lo >> 24);
  return o - out;
}
// two chains interleaved, fma form, branchy
static size_t rc_fma_pair(const uint16_t* w, u64 b, u64 e, uint8_t* out, size_t cap) {
  const u64 n = (e - b) / 2;
  u32 lo0 = 0, size0 = 0xFFFFFFFEu, lo1 = 0, size1 = 0xFFFFFFFEu;
  uint8_t *o0 = out, *o1 = out + cap / 2; const uint16_t* w0 = w + b; const uint16_t* w1 = w + b + n;
#define ST(L) { const u32 x = w##L[k]; const u32 bit = x >> 15, m = x & 0x7FFF; const i64 c = (i64)bit - 2049; \
    const u32 ns = (u32)(((i64)((u64)size##L * m) + c) >> 12); lo##L += (bit - 1) & (size##L - ns); size##L = ns; \
    const u32 same = ((lo##L ^ (lo##L + size##L + 1)) & 0xFF000000u) == 0; *o##L = (uint8_t)(lo##L >> 24); o##L += same; \
    lo##L = same ? lo##L << 8 : lo##L; size##L = same ? (size##L << 8) + 510u : size##L; \
    while (__builtin_expect(same && ((lo##L ^ (lo##L + size##L + 1)) & 0xFF000000u) == 0, 0)) { *o##L++ = (uint8_t)(lo##L >> 24); lo##L <<= 8; size##L = (size##L << 8) + 510u; } }
  for (u64 k = 0; k < n; ++k) { ST(0) ST(1) }
  return (o0 - out) + (o1 - (out + cap / 2));
}
static size_t rc_fma_pair_br(const uint16_t* w, u64 b, u64 e, uint8_t* out, size_t cap) {
  const u64 n = (e - b) / 2;
  u32 lo0 = 0, size0 = 0xFFFFFFFEu, lo1 = 0, size1 = 0xFFFFFFFEu;
  uint8_t *o0 = out, *o1 = out + cap / 2; const uint16_t* w0 = w + b; const uint16_t* w1 = w + b + n;
#define SB(L) { const u32 x = w##L[k]; const u32 bit = x >> 15, m = x & 0x7FFF; const i64 c = (i64)bit - 2049; \
    const u32 ns = (u32)(((i64)((u64)size##L * m) + c) >> 12); lo##L += (bit - 1) & (size##L - ns); size##L = ns; \
    while (((lo##L ^ (lo##L + size##L + 1)) & 0xFF000000u) == 0) { *o##L++ = (uint8_t)(lo##L >> 24); lo##L <<= 8; size##L = (size##L << 8) + 510u; } }
  for (u64 k = 0; k < n; ++k) { SB(0) SB(1) }
  return (o0 - out) + (o1 - (out + cap / 2));
}
int main() {
  const u64 n = 48u << 20;
  std::vector<uint8_t> codes(n / 4 + 16, 0); std::vector<uint16_t> prob(n + 16), w(n + 16);
  std::mt19937_64 rng(1);
  for (u64 i = 0; i < n; ++i) {
    const u32 r = rng() & 1023;
    u32 p = r < 420 ? 60 + (rng() % 200) : r < 560 ? 4096 - 60 - (rng() % 200) : 300 + rng() % 3400;
    prob[i] = (uint16_t)p;
    const u32 bit = (rng() & 4095) < p;
    codes[i >> 2] |= bit << ((i & 3) * 2);
    w[i] = (uint16_t)((bit << 15) | (bit ? p : 4096 - p));
  }
  std::vector<uint8_t> out(n + 1024), out2(n + 1024);
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    size_t a = rc_branchy(codes.data(), prob.data(), 0, n, out.data());
    auto t1 = std::chrono::steady_clock::now();
    size_t b1 = rc_fma(w.data(), 0, n, out2.data());
    auto t2 = std::chrono::steady_clock::now();
    int same = a == b1 && !memcmp(out.data(), out2.data(), a);
    size_t b2 = rc_fma2(w.data(), 0, n, out2.data());
    auto t3 = std::chrono::steady_clock::now();
    int same2 = a == b2 && !memcmp(out.data(), out2.data(), a);
    size_t b3 = rc_fma_pair(w.data(), 0, n, out2.data(), out2.size());
    auto t4 = std::chrono::steady_clock::now();
    size_t b4 = rc_fma_pair_br(w.data(), 0, n, out2.data(), out2.size());
    auto t5 = std::chrono::steady_clock::now();
    auto ns = [&](auto x, auto y) { return std::chrono::duration<double, std::nano>(y - x).count() / n; };
    std::printf("bytes %zu %zu(%d) %zu(%d) %zu %zu | ns/elem: branchy %.2f, fma %.2f, fma2 %.2f, fma_pair_free %.2f fma_pair_br %.2f\n", a, b1, same, b2, same2, b3, b4,
                ns(t0, t1), ns(t1, t2), ns(t2, t3), ns(t3, t4), ns(t4, t5));
  }
}
