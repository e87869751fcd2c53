// prototype: 16-lane AVX-512 range coder (same arithmetic as runChain), 16 equal slices of one stream
#include <immintrin.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64;
static inline u32 codeAt(const uint8_t* c, u64 i) { return (c[i >> 2] >> ((i & 3) * 2)) & 3u; }

static size_t rc_scalar(const uint8_t* codes, const uint16_t* prob, u64 b, u64 e, uint8_t* out) {
  u32 lo = 0, size = 0xFFFFFFFEu; uint8_t* o = out;
  for (u64 i = b; i < e; ++i) {
    const u32 bit = codeAt(codes, i) & 1u;
    const u32 t = (u32)(((u64)size * prob[i] + 2048u) >> 12);
    lo = bit ? lo : lo + t + 1;
    size = bit ? t - 1 : size - t - 1;
    while (((lo ^ (lo + size + 1)) & 0xFF000000u) == 0) { *o++ = (uint8_t)(lo >> 24); lo <<= 8; size = (size << 8) + 510u; }
  }
  *o++ = (uint8_t)(lo >> 24); *o++ = 255; *o++ = 255; *o++ = 255;
  return o - out;
}

struct Lanes {
  alignas(64) u32 lo[16], size[16];
  const uint8_t* codes[16]; const uint16_t* prob[16]; u64 i[16], e[16]; uint8_t* out[16];
};

// transposes 16 rows of 16 u32
__attribute__((target("avx512f,avx512bw,avx512vl,avx512dq")))
static inline void transpose16(__m512i r[16]) {
  __m512i t[16];
  for (int i = 0; i < 16; i += 2) { t[i] = _mm512_unpacklo_epi32(r[i], r[i + 1]); t[i + 1] = _mm512_unpackhi_epi32(r[i], r[i + 1]); }
  for (int i = 0; i < 16; i += 4) {
    r[i] = _mm512_unpacklo_epi64(t[i], t[i + 2]); r[i + 1] = _mm512_unpackhi_epi64(t[i], t[i + 2]);
    r[i + 2] = _mm512_unpacklo_epi64(t[i + 1], t[i + 3]); r[i + 3] = _mm512_unpackhi_epi64(t[i + 1], t[i + 3]);
  }
  for (int i = 0; i < 16; i += 8) for (int j = 0; j < 4; ++j) {
    t[i + j] = _mm512_shuffle_i32x4(r[i + j], r[i + j + 4], 0x88);
    t[i + j + 4] = _mm512_shuffle_i32x4(r[i + j], r[i + j + 4], 0xdd);
  }
  for (int j = 0; j < 8; ++j) {
    r[j] = _mm512_shuffle_i32x4(t[j], t[j + 8], 0x88);
    r[j + 8] = _mm512_shuffle_i32x4(t[j], t[j + 8], 0xdd);
  }
}

__attribute__((target("avx512f,avx512bw,avx512vl,avx512dq")))
static void run_words(Lanes& L, u64 words) {
  __m512i lo = _mm512_load_si512(L.lo), size = _mm512_load_si512(L.size);
  const __m512i one = _mm512_set1_epi32(1), c4095 = _mm512_set1_epi32(4095), c2048 = _mm512_set1_epi32(2048);
  const __m512i top = _mm512_set1_epi32((int)0xFF000000u), c510 = _mm512_set1_epi32(510);
  alignas(64) u32 lob[16];
  for (u64 w = 0; w < words; ++w) {
    __m512i P[16];
    alignas(64) u32 cw[16];
    for (int l = 0; l < 16; ++l) {
      P[l] = _mm512_cvtepu16_epi32(_mm256_loadu_si256((const __m256i*)(L.prob[l] + L.i[l])));
      std::memcpy(&cw[l], L.codes[l] + (L.i[l] >> 2), 4);
      L.i[l] += 16;
    }
    transpose16(P);
    __m512i W = _mm512_load_si512(cw);
    for (int t = 0; t < 16; ++t) {
      const __m512i p = P[t];
      const __mmask16 kBit = _mm512_test_epi32_mask(W, one);
      W = _mm512_srli_epi32(W, 2);
      const __m512i hi = _mm512_srli_epi32(size, 12), lw = _mm512_and_si512(size, c4095);
      const __m512i tt = _mm512_add_epi32(_mm512_mullo_epi32(hi, p),
                                          _mm512_srli_epi32(_mm512_add_epi32(_mm512_mullo_epi32(lw, p), c2048), 12));
      const __m512i t1 = _mm512_add_epi32(tt, one);
      lo = _mm512_mask_add_epi32(lo, (__mmask16)~kBit, lo, t1);            // zero bit: lo += t + 1
      size = _mm512_mask_blend_epi32(kBit, _mm512_sub_epi32(size, t1), _mm512_sub_epi32(tt, one));
      // renormalise
      __m512i hiend = _mm512_add_epi32(_mm512_add_epi32(lo, size), one);
      __mmask16 m = _mm512_testn_epi32_mask(_mm512_xor_si512(lo, hiend), top);
      while (m) {
        _mm512_store_si512(lob, lo);
        unsigned mm = m;
        while (mm) { const int l = __builtin_ctz(mm); mm &= mm - 1; *L.out[l]++ = (uint8_t)(lob[l] >> 24); }
        lo = _mm512_mask_slli_epi32(lo, m, lo, 8);
        size = _mm512_mask_add_epi32(size, m, _mm512_slli_epi32(size, 8), c510);
        hiend = _mm512_add_epi32(_mm512_add_epi32(lo, size), one);
        m = _mm512_mask_testn_epi32_mask(m, _mm512_xor_si512(lo, hiend), top);
      }
    }
  }
  _mm512_store_si512(L.lo, lo); _mm512_store_si512(L.size, size);
}


// variant B: emission events appended branch-free to a linear buffer, distributed after each word
__attribute__((target("avx512f,avx512bw,avx512vl,avx512dq,avx512vbmi2")))
static void run_words_ev(Lanes& L, u64 words) {
  __m512i lo = _mm512_load_si512(L.lo), size = _mm512_load_si512(L.size);
  const __m512i one = _mm512_set1_epi32(1), c4095 = _mm512_set1_epi32(4095), c2048 = _mm512_set1_epi32(2048);
  const __m512i top = _mm512_set1_epi32((int)0xFF000000u), c510 = _mm512_set1_epi32(510);
  const __m512i laneid = _mm512_setr_epi32(0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15);
  alignas(64) u32 ev[16 * 16 * 4 + 64];
  for (u64 w = 0; w < words; ++w) {
    __m512i P[16];
    alignas(64) u32 cw[16];
    for (int l = 0; l < 16; ++l) {
      P[l] = _mm512_cvtepu16_epi32(_mm256_loadu_si256((const __m256i*)(L.prob[l] + L.i[l])));
      std::memcpy(&cw[l], L.codes[l] + (L.i[l] >> 2), 4);
      L.i[l] += 16;
    }
    transpose16(P);
    __m512i W = _mm512_load_si512(cw);
    u32 n = 0;
    for (int t = 0; t < 16; ++t) {
      const __m512i p = P[t];
      const __mmask16 kBit = _mm512_test_epi32_mask(W, one);
      W = _mm512_srli_epi32(W, 2);
      const __m512i hi = _mm512_srli_epi32(size, 12), lw = _mm512_and_si512(size, c4095);
      const __m512i tt = _mm512_add_epi32(_mm512_mullo_epi32(hi, p),
                                          _mm512_srli_epi32(_mm512_add_epi32(_mm512_mullo_epi32(lw, p), c2048), 12));
      const __m512i t1 = _mm512_add_epi32(tt, one);
      lo = _mm512_mask_add_epi32(lo, (__mmask16)~kBit, lo, t1);
      size = _mm512_mask_blend_epi32(kBit, _mm512_sub_epi32(size, t1), _mm512_sub_epi32(tt, one));
      __m512i hiend = _mm512_add_epi32(_mm512_add_epi32(lo, size), one);
      __mmask16 m = _mm512_testn_epi32_mask(_mm512_xor_si512(lo, hiend), top);
      // first byte: branch-free
      _mm512_storeu_si512(ev + n, _mm512_maskz_compress_epi32(m, _mm512_or_si512(_mm512_srli_epi32(lo, 24), _mm512_slli_epi32(laneid, 8))));
      n += (u32)__builtin_popcount(m);
      lo = _mm512_mask_slli_epi32(lo, m, lo, 8);
      size = _mm512_mask_add_epi32(size, m, _mm512_slli_epi32(size, 8), c510);
      hiend = _mm512_add_epi32(_mm512_add_epi32(lo, size), one);
      m = _mm512_mask_testn_epi32_mask(m, _mm512_xor_si512(lo, hiend), top);
      while (__builtin_expect(m != 0, 0)) {
        _mm512_storeu_si512(ev + n, _mm512_maskz_compress_epi32(m, _mm512_or_si512(_mm512_srli_epi32(lo, 24), _mm512_slli_epi32(laneid, 8))));
        n += (u32)__builtin_popcount(m);
        lo = _mm512_mask_slli_epi32(lo, m, lo, 8);
        size = _mm512_mask_add_epi32(size, m, _mm512_slli_epi32(size, 8), c510);
        hiend = _mm512_add_epi32(_mm512_add_epi32(lo, size), one);
        m = _mm512_mask_testn_epi32_mask(m, _mm512_xor_si512(lo, hiend), top);
      }
    }
    for (u32 k = 0; k < n; ++k) { const u32 e = ev[k]; *L.out[e >> 8]++ = (uint8_t)e; }
  }
  _mm512_store_si512(L.lo, lo); _mm512_store_si512(L.size, size);
}

// variant C: NV independent 16-lane vectors interleaved per thread (event buffers as in B)
template <int NV>
__attribute__((target("avx512f,avx512bw,avx512vl,avx512dq,avx512vbmi2")))
static void run_words_multi(Lanes* Ls, u64 words) {
  __m512i lo[NV], size[NV];
  for (int v = 0; v < NV; ++v) { lo[v] = _mm512_load_si512(Ls[v].lo); size[v] = _mm512_load_si512(Ls[v].size); }
  const __m512i one = _mm512_set1_epi32(1), c4095 = _mm512_set1_epi32(4095), c2048 = _mm512_set1_epi32(2048);
  const __m512i top = _mm512_set1_epi32((int)0xFF000000u), c510 = _mm512_set1_epi32(510);
  const __m512i laneid = _mm512_setr_epi32(0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15);
  alignas(64) static thread_local u32 ev[NV][16 * 16 * 4 + 64];
  alignas(64) static thread_local u32 Pm[NV][16][16];
  for (u64 w = 0; w < words; ++w) {
    __m512i W[NV]; u32 n[NV];
    for (int v = 0; v < NV; ++v) {
      __m512i P[16];
      alignas(64) u32 cw[16];
      Lanes& L = Ls[v];
      for (int l = 0; l < 16; ++l) {
        P[l] = _mm512_cvtepu16_epi32(_mm256_loadu_si256((const __m256i*)(L.prob[l] + L.i[l])));
        std::memcpy(&cw[l], L.codes[l] + (L.i[l] >> 2), 4);
        L.i[l] += 16;
      }
      transpose16(P);
      for (int t = 0; t < 16; ++t) _mm512_store_si512(Pm[v][t], P[t]);
      W[v] = _mm512_load_si512(cw); n[v] = 0;
    }
    for (int t = 0; t < 16; ++t) {
#pragma GCC unroll 4
      for (int v = 0; v < NV; ++v) {
        const __m512i p = _mm512_load_si512(Pm[v][t]);
        const __mmask16 kBit = _mm512_test_epi32_mask(W[v], one);
        W[v] = _mm512_srli_epi32(W[v], 2);
        const __m512i hi = _mm512_srli_epi32(size[v], 12), lw = _mm512_and_si512(size[v], c4095);
        const __m512i tt = _mm512_add_epi32(_mm512_mullo_epi32(hi, p),
                                            _mm512_srli_epi32(_mm512_add_epi32(_mm512_mullo_epi32(lw, p), c2048), 12));
        const __m512i t1 = _mm512_add_epi32(tt, one);
        lo[v] = _mm512_mask_add_epi32(lo[v], (__mmask16)~kBit, lo[v], t1);
        size[v] = _mm512_mask_blend_epi32(kBit, _mm512_sub_epi32(size[v], t1), _mm512_sub_epi32(tt, one));
        __m512i hiend = _mm512_add_epi32(_mm512_add_epi32(lo[v], size[v]), one);
        __mmask16 m = _mm512_testn_epi32_mask(_mm512_xor_si512(lo[v], hiend), top);
        _mm512_storeu_si512(ev[v] + n[v], _mm512_maskz_compress_epi32(m, _mm512_or_si512(_mm512_srli_epi32(lo[v], 24), _mm512_slli_epi32(laneid, 8))));
        n[v] += (u32)__builtin_popcount(m);
        lo[v] = _mm512_mask_slli_epi32(lo[v], m, lo[v], 8);
        size[v] = _mm512_mask_add_epi32(size[v], m, _mm512_slli_epi32(size[v], 8), c510);
        hiend = _mm512_add_epi32(_mm512_add_epi32(lo[v], size[v]), one);
        m = _mm512_mask_testn_epi32_mask(m, _mm512_xor_si512(lo[v], hiend), top);
        while (__builtin_expect(m != 0, 0)) {
          _mm512_storeu_si512(ev[v] + n[v], _mm512_maskz_compress_epi32(m, _mm512_or_si512(_mm512_srli_epi32(lo[v], 24), _mm512_slli_epi32(laneid, 8))));
          n[v] += (u32)__builtin_popcount(m);
          lo[v] = _mm512_mask_slli_epi32(lo[v], m, lo[v], 8);
          size[v] = _mm512_mask_add_epi32(size[v], m, _mm512_slli_epi32(size[v], 8), c510);
          hiend = _mm512_add_epi32(_mm512_add_epi32(lo[v], size[v]), one);
          m = _mm512_mask_testn_epi32_mask(m, _mm512_xor_si512(lo[v], hiend), top);
        }
      }
    }
    for (int v = 0; v < NV; ++v) for (u32 k = 0; k < n[v]; ++k) { const u32 e = ev[v][k]; *Ls[v].out[e >> 8]++ = (uint8_t)e; }
  }
  for (int v = 0; v < NV; ++v) { _mm512_store_si512(Ls[v].lo, lo[v]); _mm512_store_si512(Ls[v].size, size[v]); }
}

// variant D: as C, but the renormalisation is pure vector arithmetic (no mask register in the
// loop-carried chain): z = min((lo ^ hiend) >> 24, 1), shift = 8 * (1 - z)
template <int NV>
__attribute__((target("avx512f,avx512bw,avx512vl,avx512dq,avx512vbmi2")))
static void run_words_arith(Lanes* Ls, u64 words) {
  __m512i lo[NV], size[NV];
  for (int v = 0; v < NV; ++v) { lo[v] = _mm512_load_si512(Ls[v].lo); size[v] = _mm512_load_si512(Ls[v].size); }
  const __m512i one = _mm512_set1_epi32(1), c4095 = _mm512_set1_epi32(4095), c2048 = _mm512_set1_epi32(2048);
  const __m512i c510 = _mm512_set1_epi32(510);
  const __m512i laneid = _mm512_setr_epi32(0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15);
  alignas(64) static thread_local u32 ev[NV][16 * 16 * 4 + 64];
  alignas(64) static thread_local u32 Pm[NV][16][16];
  for (u64 w = 0; w < words; ++w) {
    __m512i W[NV]; u32 n[NV];
    for (int v = 0; v < NV; ++v) {
      __m512i P[16];
      alignas(64) u32 cw[16];
      Lanes& L = Ls[v];
      for (int l = 0; l < 16; ++l) {
        P[l] = _mm512_cvtepu16_epi32(_mm256_loadu_si256((const __m256i*)(L.prob[l] + L.i[l])));
        std::memcpy(&cw[l], L.codes[l] + (L.i[l] >> 2), 4);
        L.i[l] += 16;
      }
      transpose16(P);
      for (int t = 0; t < 16; ++t) _mm512_store_si512(Pm[v][t], P[t]);
      W[v] = _mm512_load_si512(cw); n[v] = 0;
    }
    for (int t = 0; t < 16; ++t) {
#pragma GCC unroll 4
      for (int v = 0; v < NV; ++v) {
        const __m512i p = _mm512_load_si512(Pm[v][t]);
        const __mmask16 kBit = _mm512_test_epi32_mask(W[v], one);
        W[v] = _mm512_srli_epi32(W[v], 2);
        const __m512i hi = _mm512_srli_epi32(size[v], 12), lw = _mm512_and_si512(size[v], c4095);
        const __m512i tt = _mm512_add_epi32(_mm512_mullo_epi32(hi, p),
                                            _mm512_srli_epi32(_mm512_add_epi32(_mm512_mullo_epi32(lw, p), c2048), 12));
        const __m512i t1 = _mm512_add_epi32(tt, one);
        lo[v] = _mm512_mask_add_epi32(lo[v], (__mmask16)~kBit, lo[v], t1);
        size[v] = _mm512_mask_blend_epi32(kBit, _mm512_sub_epi32(size[v], t1), _mm512_sub_epi32(tt, one));
        for (;;) {
          const __m512i hiend = _mm512_add_epi32(_mm512_add_epi32(lo[v], size[v]), one);
          const __m512i x = _mm512_srli_epi32(_mm512_xor_si512(lo[v], hiend), 24);
          const __m512i nz = _mm512_min_epu32(x, one);                    // 1: top bytes differ, 0: emit
          const __m512i emit = _mm512_sub_epi32(one, nz);                 // 1 where a byte leaves
          const __mmask16 m = _mm512_testn_epi32_mask(x, x);              // off the carried chain
          _mm512_storeu_si512(ev[v] + n[v], _mm512_maskz_compress_epi32(m, _mm512_or_si512(_mm512_srli_epi32(lo[v], 24), _mm512_slli_epi32(laneid, 8))));
          n[v] += (u32)__builtin_popcount(m);
          const __m512i sh = _mm512_slli_epi32(emit, 3);
          lo[v] = _mm512_sllv_epi32(lo[v], sh);
          size[v] = _mm512_add_epi32(_mm512_sllv_epi32(size[v], sh), _mm512_and_si512(c510, _mm512_sub_epi32(_mm512_setzero_si512(), emit)));
          if (__builtin_expect(m == 0, 1)) break;
          // a lane that emitted may emit again (rare): test once more
          const __m512i h2 = _mm512_add_epi32(_mm512_add_epi32(lo[v], size[v]), one);
          if (__builtin_expect(_mm512_mask_testn_epi32_mask(m, _mm512_srli_epi32(_mm512_xor_si512(lo[v], h2), 24), _mm512_set1_epi32(-1)) == 0, 1)) break;
        }
      }
    }
    for (int v = 0; v < NV; ++v) for (u32 k = 0; k < n[v]; ++k) { const u32 e = ev[v][k]; *Ls[v].out[e >> 8]++ = (uint8_t)e; }
  }
  for (int v = 0; v < NV; ++v) { _mm512_store_si512(Ls[v].lo, lo[v]); _mm512_store_si512(Ls[v].size, size[v]); }
}

template <int NV>
static void bench_multi(const std::vector<uint8_t>& codes, const std::vector<uint16_t>& prob, u64 n) {
  const u64 per = n / (16 * NV) / 16 * 16;
  std::vector<std::vector<uint8_t>> got(16 * NV), ref(16 * NV);
  Lanes Ls[NV];
  for (int c = 0; c < 16 * NV; ++c) {
    Lanes& L = Ls[c / 16]; const int l = c % 16;
    got[c].resize(per * 4 + 16); L.lo[l] = 0; L.size[l] = 0xFFFFFFFEu; L.codes[l] = codes.data(); L.prob[l] = prob.data();
    L.i[l] = c * per; L.e[l] = (c + 1) * per; L.out[l] = got[c].data();
  }
  auto t2 = std::chrono::steady_clock::now();
  if (getenv("ARITH")) run_words_arith<NV>(Ls, per / 16); else run_words_multi<NV>(Ls, per / 16);
  auto t3 = std::chrono::steady_clock::now();
  bool ok = true;
  for (int c = 0; c < 16 * NV; ++c) {
    Lanes& L = Ls[c / 16]; const int l = c % 16;
    uint8_t* o = L.out[l]; *o++ = (uint8_t)(L.lo[l] >> 24); *o++ = 255; *o++ = 255; *o++ = 255;
    got[c].resize(o - got[c].data());
    ref[c].resize(per * 4 + 16); ref[c].resize(rc_scalar(codes.data(), prob.data(), c * per, (c + 1) * per, ref[c].data()));
    if (got[c] != ref[c]) ok = false;
  }
  const double ns = std::chrono::duration<double, std::nano>(t3 - t2).count();
  std::printf("NV=%d: %.3f ns/el, %.2f ns per 16-lane step, chain advances one element per %.2f ns, equal %d\n", NV,
              ns / (per * 16 * NV), ns / (per * NV), ns / per, ok);
}

int main() {
  const u64 n = 64u << 20;
  std::vector<uint8_t> codes(n / 4 + 16, 0); std::vector<uint16_t> prob(n + 32);
  std::mt19937_64 rng(1);
  for (u64 i = 0; i < n; ++i) {
    const u32 r = rng() & 1023;
    u32 p; u32 bit;
    if (r < 500) { p = 3900 + (rng() % 190); bit = (rng() % 4096) < p; }
    else if (r < 900) { p = 100 + (rng() % 200); bit = (rng() % 4096) < p; }
    else { p = 1000 + (rng() % 2000); bit = (rng() % 4096) < p; }
    prob[i] = (uint16_t)p;
    codes[i >> 2] |= (uint8_t)(bit << ((i & 3) * 2));
  }
  const u64 per = n / 16;
  std::vector<std::vector<uint8_t>> ref(16), got(16);
  auto t0 = std::chrono::steady_clock::now();
  for (int l = 0; l < 16; ++l) { ref[l].resize(per * 4 + 16); ref[l].resize(rc_scalar(codes.data(), prob.data(), l * per, (l + 1) * per, ref[l].data())); }
  auto t1 = std::chrono::steady_clock::now();
  Lanes L;
  for (int l = 0; l < 16; ++l) { got[l].resize(per * 4 + 16); L.lo[l] = 0; L.size[l] = 0xFFFFFFFEu; L.codes[l] = codes.data(); L.prob[l] = prob.data(); L.i[l] = l * per; L.e[l] = (l + 1) * per; L.out[l] = got[l].data(); }
  auto t2 = std::chrono::steady_clock::now();
  if (getenv("EV")) run_words_ev(L, per / 16); else run_words(L, per / 16);
  auto t3 = std::chrono::steady_clock::now();
  size_t total = 0; bool ok = true;
  for (int l = 0; l < 16; ++l) {
    uint8_t* o = L.out[l]; *o++ = (uint8_t)(L.lo[l] >> 24); *o++ = 255; *o++ = 255; *o++ = 255;
    got[l].resize(o - got[l].data()); total += got[l].size();
    if (got[l] != ref[l]) ok = false;
  }
  bench_multi<1>(codes, prob, n); bench_multi<2>(codes, prob, n); bench_multi<3>(codes, prob, n); bench_multi<4>(codes, prob, n);
  std::printf("scalar %.2f ns/el, vector %.2f ns/el (%.2f ns/step), bytes/el %.3f, equal %d\n",
              std::chrono::duration<double, std::nano>(t1 - t0).count() / n,
              std::chrono::duration<double, std::nano>(t3 - t2).count() / n,
              std::chrono::duration<double, std::nano>(t3 - t2).count() / (n / 16), (double)total / n, ok);
}
