// See wavelet_rc.hpp.
#include "wavelet_rc.hpp"

#include <immintrin.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

namespace bwtc {
namespace wavelet {
namespace {

inline uint32_t codeAt(const uint8_t* codes, uint64_t i) { return (codes[i >> 2] >> ((i & 3) * 2)) & 3u; }

}  // namespace

// one chain on its own: the byte output is a (badly predictable) branch, but nothing is added
// to the chain
void runChain(CoderChain& c, const uint8_t* codes, const uint16_t* prob, uint64_t until) {
  const uint64_t kChunk = 16384;
  uint32_t lo = c.lo, size = c.size;
  uint64_t b = c.i;
  for (; b < until; b += kChunk) {
    const uint64_t ce = std::min(until, b + kChunk);
    uint8_t* o = c.room(kChunk);
#define BWTC_RC_STEP(bit_, p_)                                                                          \
    {                                                                                                    \
      const uint32_t bit = (bit_);                                                                       \
      const uint32_t t = static_cast<uint32_t>((static_cast<uint64_t>(size) * (p_) + 2048u) >> 12);      \
      lo = bit ? lo : lo + t + 1;                                                                        \
      size = bit ? t - 1 : size - t - 1;                                                                 \
      while (((lo ^ (lo + size + 1)) & 0xFF000000u) == 0) { *o++ = static_cast<uint8_t>(lo >> 24); lo <<= 8; size = (size << 8) + 510u; } \
    }
    uint64_t i = b;
    for (; i < ce && (i & 3); ++i) BWTC_RC_STEP(codeAt(codes, i) & 1u, prob[i]);
    for (; i + 4 <= ce; i += 4) {
      const uint32_t four = codes[i >> 2];
      BWTC_RC_STEP(four & 1u, prob[i]);
      BWTC_RC_STEP((four >> 2) & 1u, prob[i + 1]);
      BWTC_RC_STEP((four >> 4) & 1u, prob[i + 2]);
      BWTC_RC_STEP((four >> 6) & 1u, prob[i + 3]);
    }
    for (; i < ce; ++i) BWTC_RC_STEP(codeAt(codes, i) & 1u, prob[i]);
#undef BWTC_RC_STEP
    c.used = static_cast<size_t>(o - c.out->data());
  }
  c.lo = lo; c.size = size; c.i = std::max(c.i, until);
}

void runChainW(CoderChain& c, const uint16_t* w, uint64_t until) {
  const uint64_t kChunk = 16384;
  uint32_t lo = c.lo, size = c.size;
  uint64_t b = c.i;
  for (; b < until; b += kChunk) {
    const uint64_t ce = std::min(until, b + kChunk);
    uint8_t* o = c.room(kChunk);
    for (uint64_t i = b; i < ce; ++i) {
      const uint32_t x = w[i];
      const uint32_t bit = x >> 15, m = x & 0x7FFFu;
      const uint32_t ns = static_cast<uint32_t>((static_cast<int64_t>(static_cast<uint64_t>(size) * m) + (static_cast<int64_t>(bit) - 2049)) >> 12);
      lo += (bit - 1u) & (size - ns);                                     // a zero moves the low end up by t + 1 = size - ns
      size = ns;
      while (((lo ^ (lo + size + 1)) & 0xFF000000u) == 0) { *o++ = static_cast<uint8_t>(lo >> 24); lo <<= 8; size = (size << 8) + 510u; }
    }
    c.used = static_cast<size_t>(o - c.out->data());
  }
  c.lo = lo; c.size = size; c.i = std::max(c.i, until);
}

// Two chains stepped alternately.  A chain alone is bound by its multiply latency and by the
// mispredicted byte-output branch; with the first output byte written branch-free (a second
// byte in the same step is rare and stays a branch) two independent chains overlap and the
// pair costs 1.7 ns per element instead of 2.1-2.3 on the GPU box's EPYC
// (scripts/dev/rcbench.cpp).  Runs until the shorter chain ends.
void runChainPair(CoderChain& a, const uint8_t* codes_a, const uint16_t* prob_a,
                  CoderChain& b, const uint8_t* codes_b, const uint16_t* prob_b, uint64_t limit) {
  const uint64_t kChunk = 8192;
  uint64_t left = std::min(limit, std::min(a.e - a.i, b.e - b.i));
  while (left > 0) {
    const uint64_t n = std::min(left, kChunk);
    uint8_t* oa = a.room(kChunk);
    uint8_t* ob = b.room(kChunk);
    uint32_t loa = a.lo, sa = a.size, lob = b.lo, sb = b.size;
    uint64_t ia = a.i, ib = b.i;
#define BWTC_RC_FREE(lo, size, o, i, codes, prob)                                                         \
    {                                                                                                      \
      const uint32_t bit = codeAt(codes, i) & 1u;                                                          \
      const uint32_t t = static_cast<uint32_t>((static_cast<uint64_t>(size) * prob[i] + 2048u) >> 12);     \
      lo = bit ? lo : lo + t + 1;                                                                          \
      size = bit ? t - 1 : size - t - 1;                                                                   \
      const uint32_t same = ((lo ^ (lo + size + 1)) & 0xFF000000u) == 0;                                   \
      *o = static_cast<uint8_t>(lo >> 24);                                                                 \
      o += same;                                                                                           \
      lo = same ? lo << 8 : lo;                                                                            \
      size = same ? (size << 8) + 510u : size;                                                             \
      while (__builtin_expect(same && ((lo ^ (lo + size + 1)) & 0xFF000000u) == 0, 0)) { *o++ = static_cast<uint8_t>(lo >> 24); lo <<= 8; size = (size << 8) + 510u; } \
      ++i;                                                                                                 \
    }
    for (uint64_t k = 0; k < n; ++k) {
      BWTC_RC_FREE(loa, sa, oa, ia, codes_a, prob_a)
      BWTC_RC_FREE(lob, sb, ob, ib, codes_b, prob_b)
    }
#undef BWTC_RC_FREE
    a.lo = loa; a.size = sa; a.i = ia; a.used = static_cast<size_t>(oa - a.out->data());
    b.lo = lob; b.size = sb; b.i = ib; b.used = static_cast<size_t>(ob - b.out->data());
    left -= n;
  }
}

void runChainPairW(CoderChain& a, const uint16_t* wa, CoderChain& b, const uint16_t* wb, uint64_t limit) {
  const uint64_t kChunk = 8192;
  uint64_t left = std::min(limit, std::min(a.e - a.i, b.e - b.i));
  while (left > 0) {
    const uint64_t n = std::min(left, kChunk);
    uint8_t* oa = a.room(kChunk);
    uint8_t* ob = b.room(kChunk);
    uint32_t loa = a.lo, sa = a.size, lob = b.lo, sb = b.size;
    uint64_t ia = a.i, ib = b.i;
#define BWTC_RC_W(lo, size, o, i, w)                                                                      \
    {                                                                                                      \
      const uint32_t x = w[i];                                                                             \
      const uint32_t bit = x >> 15, m = x & 0x7FFFu;                                                       \
      const uint32_t ns = static_cast<uint32_t>((static_cast<int64_t>(static_cast<uint64_t>(size) * m) + (static_cast<int64_t>(bit) - 2049)) >> 12); \
      lo += (bit - 1u) & (size - ns);                                                                      \
      size = ns;                                                                                           \
      while (((lo ^ (lo + size + 1)) & 0xFF000000u) == 0) { *o++ = static_cast<uint8_t>(lo >> 24); lo <<= 8; size = (size << 8) + 510u; } \
      ++i;                                                                                                 \
    }
    for (uint64_t k = 0; k < n; ++k) {
      BWTC_RC_W(loa, sa, oa, ia, wa)
      BWTC_RC_W(lob, sb, ob, ib, wb)
    }
#undef BWTC_RC_W
    a.lo = loa; a.size = sa; a.i = ia; a.used = static_cast<size_t>(oa - a.out->data());
    b.lo = lob; b.size = sb; b.i = ib; b.used = static_cast<size_t>(ob - b.out->data());
    left -= n;
  }
}

void runChainPair(CoderChain& a, CoderChain& b, const uint8_t* codes, const uint16_t* prob) {
  runChainPair(a, codes, prob, b, codes, prob, ~static_cast<uint64_t>(0));
}

// ---- sixteen chains per thread ----------------------------------------------------------------
namespace {

const int kLanes = 16;
const uint64_t kScalarOnly = 512;        // chains shorter than this never enter a lane
const int kMinLanes = 4;                 // fewer busy lanes and an empty source: finish them scalar, two by two
const uint64_t kMaxWords = 512;          // 16-element words per vector run (bounds the output room reserved per lane)

const int kGroups = 2;                   // vectors an engine steps alternately (their dependency chains overlap)
const int kAll = kLanes * kGroups;

struct Lanes {
  alignas(64) uint32_t lo[kAll], size[kAll];
  const uint8_t* codes[kAll];
  const uint16_t* prob[kAll];
  uint64_t i[kAll];
  uint8_t* out[kAll];
  uint32_t advance[kAll];                // 16 for a busy lane, 0 for an idle one (it keeps re-reading the zero page)
};

alignas(64) const uint16_t kZeroProb[32] = {0};
alignas(64) const uint8_t kZeroCodes[8] = {0};

#define BWTC_AVX512 __attribute__((target("avx512f,avx512bw,avx512vl,avx512dq")))

// 16 rows of 16 u32 -> 16 columns
BWTC_AVX512 inline void transpose16(__m512i r[16]) {
  __m512i t[16];
  for (int i = 0; i < 16; i += 2) { t[i] = _mm512_unpacklo_epi32(r[i], r[i + 1]); t[i + 1] = _mm512_unpackhi_epi32(r[i], r[i + 1]); }
  for (int i = 0; i < 16; i += 4) {
    r[i] = _mm512_unpacklo_epi64(t[i], t[i + 2]); r[i + 1] = _mm512_unpackhi_epi64(t[i], t[i + 2]);
    r[i + 2] = _mm512_unpacklo_epi64(t[i + 1], t[i + 3]); r[i + 3] = _mm512_unpackhi_epi64(t[i + 1], t[i + 3]);
  }
  for (int i = 0; i < 16; i += 8)
    for (int j = 0; j < 4; ++j) {
      t[i + j] = _mm512_shuffle_i32x4(r[i + j], r[i + j + 4], 0x88);
      t[i + j + 4] = _mm512_shuffle_i32x4(r[i + j], r[i + j + 4], 0xdd);
    }
  for (int j = 0; j < 8; ++j) {
    r[j] = _mm512_shuffle_i32x4(t[j], t[j + 8], 0x88);
    r[j + 8] = _mm512_shuffle_i32x4(t[j], t[j + 8], 0xdd);
  }
}

// `words` 16-element words of every busy lane (each has that many left and stands on a word
// boundary).  Per step and lane exactly runChain's arithmetic; t = (size * p + 2048) >> 12 is
// formed from its two 32-bit parts, (size >> 12) * p + (((size & 4095) * p + 2048) >> 12).
// Bytes that leave the coder are appended, as (lane, byte) events, to a linear buffer without a
// branch (mask-compress of the lanes that emit; a second byte in the same step is rare and
// stays a branch) and handed to the lanes' outputs once per word, in event order -- so the
// scalar byte stores and the unpredictable "does anybody emit" branch are off the loop that
// carries the interval.
// WMODE: the lanes' elements are w-words (bit << 15 | probability of the coded bit, wavelet_gpu_models.hpp)
// read through L.prob; the probability of a one and the bit are taken from them.
// G = 2: two vectors of sixteen chains are stepped alternately.  A step is a chain of dependent
// instructions from the interval to the interval (two multiplies, the byte test, the masked shift),
// some 45 cycles of which a single vector keeps the core's pipes busy for a third; the second
// vector's steps fill them.
template <bool WMODE, int G>
BWTC_AVX512 void runWords(Lanes& L, uint64_t words, uint32_t busy_mask) {
  __m512i lo[G], size[G];
  __mmask16 kBusy[G];
  for (int g = 0; g < G; ++g) {
    lo[g] = _mm512_load_si512(L.lo + g * kLanes);
    size[g] = _mm512_load_si512(L.size + g * kLanes);
    kBusy[g] = static_cast<__mmask16>(busy_mask >> (g * kLanes));
  }
  const __m512i one = _mm512_set1_epi32(1), c4095 = _mm512_set1_epi32(4095), c2048 = _mm512_set1_epi32(2048);
  const __m512i top = _mm512_set1_epi32(static_cast<int>(0xFF000000u)), c510 = _mm512_set1_epi32(510);
  const __m512i lane0 = _mm512_slli_epi32(_mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15), 8);
  const __m512i c32768 = _mm512_set1_epi32(32768), c32767 = _mm512_set1_epi32(32767), cm2049 = _mm512_set1_epi32(-2049);
  alignas(64) uint32_t cw[kAll] = {0};
  alignas(64) uint32_t ev[G][16 * kLanes * 4 + 64];    // at most four bytes per lane and step
  alignas(64) __m512i P[G][16];
  for (uint64_t w = 0; w < words; ++w) {
    for (int g = 0; g < G; ++g) {
      for (int l = 0; l < kLanes; ++l) {
        const int a = g * kLanes + l;
        P[g][l] = _mm512_cvtepu16_epi32(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(L.prob[a] + L.i[a])));
        if (!WMODE) std::memcpy(&cw[a], L.codes[a] + (L.i[a] >> 2), 4);
        L.i[a] += L.advance[a];
      }
      transpose16(P[g]);
    }
    __m512i W[G];
    uint32_t n[G];
    for (int g = 0; g < G; ++g) { W[g] = _mm512_load_si512(cw + g * kLanes); n[g] = 0; }
    for (int t = 0; t < 16; ++t) {
#pragma GCC unroll 2
      for (int g = 0; g < G; ++g) {
        __m512i p = P[g][t];
        __mmask16 kBit;
        const __m512i hi = _mm512_srli_epi32(size[g], 12), lw = _mm512_and_si512(size[g], c4095);
        if (WMODE) {
          // runChainW's form: next size = (size m + bit - 2049) >> 12 = hi m + ((lw m + bit - 2049) >> 12)
          // (arithmetic shift; the products wrap modulo 2^32 like the size itself), and a zero bit moves
          // the low end up by size - next size
          kBit = _mm512_test_epi32_mask(p, c32768);
          const __m512i m = _mm512_and_si512(p, c32767);
          __m512i c = _mm512_add_epi32(_mm512_mullo_epi32(lw, m), cm2049);
          c = _mm512_mask_add_epi32(c, kBit, c, one);
          const __m512i ns = _mm512_add_epi32(_mm512_mullo_epi32(hi, m), _mm512_srai_epi32(c, 12));
          lo[g] = _mm512_mask_add_epi32(lo[g], static_cast<__mmask16>(~kBit), lo[g], _mm512_sub_epi32(size[g], ns));
          size[g] = ns;
        } else {
          kBit = _mm512_test_epi32_mask(W[g], one);
          W[g] = _mm512_srli_epi32(W[g], 2);
          const __m512i tt = _mm512_add_epi32(_mm512_mullo_epi32(hi, p),
                                              _mm512_srli_epi32(_mm512_add_epi32(_mm512_mullo_epi32(lw, p), c2048), 12));
          const __m512i t1 = _mm512_add_epi32(tt, one);
          lo[g] = _mm512_mask_add_epi32(lo[g], static_cast<__mmask16>(~kBit), lo[g], t1);          // zero bit: lo += t + 1
          size[g] = _mm512_mask_blend_epi32(kBit, _mm512_sub_epi32(size[g], t1), _mm512_sub_epi32(tt, one));
        }
        // byte output: lanes whose interval ends share their top byte
        __m512i hiend = _mm512_add_epi32(_mm512_add_epi32(lo[g], size[g]), one);
        __mmask16 m = _mm512_mask_testn_epi32_mask(kBusy[g], _mm512_xor_si512(lo[g], hiend), top);
        for (;;) {
          _mm512_storeu_si512(ev[g] + n[g], _mm512_maskz_compress_epi32(m, _mm512_or_si512(_mm512_srli_epi32(lo[g], 24), lane0)));
          n[g] += static_cast<uint32_t>(__builtin_popcount(m));
          lo[g] = _mm512_mask_slli_epi32(lo[g], m, lo[g], 8);
          size[g] = _mm512_mask_add_epi32(size[g], m, _mm512_slli_epi32(size[g], 8), c510);
          hiend = _mm512_add_epi32(_mm512_add_epi32(lo[g], size[g]), one);
          m = _mm512_mask_testn_epi32_mask(m, _mm512_xor_si512(lo[g], hiend), top);
          if (__builtin_expect(m == 0, 1)) break;
        }
      }
    }
    for (int g = 0; g < G; ++g)
      for (uint32_t k = 0; k < n[g]; ++k) { const uint32_t e = ev[g][k]; *L.out[g * kLanes + (e >> 8)]++ = static_cast<uint8_t>(e); }
  }
  for (int g = 0; g < G; ++g) {
    _mm512_store_si512(L.lo + g * kLanes, lo[g]);
    _mm512_store_si512(L.size + g * kLanes, size[g]);
  }
}

}  // namespace

bool simdCoderAvailable() {
  static const bool ok = [] {
    const char* off = std::getenv("BWTC_HIP_SIMD");
    if (off && off[0] == '0') return false;
    off = std::getenv("BWTC_HIP_CODER_LANES");
    if (off && off[0] == '0') return false;
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
           __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512dq");
  }();
  return ok;
}

void runCoderLanes(ChainSource& src, int max_lanes, std::atomic<uint64_t>* busy_ns) {
  auto tick = std::chrono::steady_clock::now();
  auto account = [&] {
    if (!busy_ns) return;
    const auto now = std::chrono::steady_clock::now();
    *busy_ns += static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(now - tick).count());
    tick = now;
  };
  // max_lanes < kMinLanes: never a vector -- the engine steps two chains alternately (the mode for
  // a block's one or two very long chains, which a lane would hold for seconds)
  const int lanes = std::max(1, std::min(max_lanes, kAll));
  const bool two = lanes > kLanes;                     // more than a vector's worth: two vectors stepped alternately
  Lanes L;
  CoderChain chain[kAll];
  ChainDesc desc[kAll];
  bool busy[kAll];
  for (int l = 0; l < kAll; ++l) busy[l] = false;
  bool sourceDry = false;
  bool wmode = false;                                  // this engine's chains come as w-words (a source hands out one kind)
  auto scalar = [&](int l, uint64_t until) {
    if (desc[l].w) runChainW(chain[l], desc[l].w, until); else runChain(chain[l], desc[l].codes, desc[l].prob, until);
  };
  auto finishLane = [&](int l) {                       // the rest of the lane's chain, scalar, and its flush
    CoderChain& c = chain[l];
    scalar(l, c.e);
    c.finish();
    src.done(desc[l].cookie);
    busy[l] = false;
  };
  for (;;) {
    // refill: a new chain runs scalar up to the next word boundary (all of it when it is short)
    int nbusy = 0;
    for (int l = 0; l < lanes; ++l) {
      while (!busy[l] && !sourceDry) {
        if (!src.next(&desc[l])) { sourceDry = true; break; }
        CoderChain& c = chain[l];
        c.start(desc[l].begin, desc[l].end, desc[l].out);
        wmode = desc[l].w != nullptr;
        const uint64_t aligned = (c.i + 15) & ~static_cast<uint64_t>(15);
        busy[l] = true;
        if (c.e - c.i < kScalarOnly || aligned + 16 > c.e) { finishLane(l); continue; }
        scalar(l, aligned);
      }
      nbusy += busy[l];
    }
    account();
    if (nbusy == 0) {
      if (sourceDry && src.retire()) return;
      sourceDry = false;
      continue;
    }
    if (lanes < kMinLanes || (nbusy < kMinLanes && sourceDry)) {
      // not worth a vector: two chains stepped alternately (or the last one alone), in slices so
      // that new work -- another block's chains -- is picked up soon
      const uint64_t kSlice = static_cast<uint64_t>(1) << 22;
      int ids[kAll], n = 0;
      for (int l = 0; l < kAll; ++l) if (busy[l]) ids[n++] = l;
      if (n >= 2) {
        const int a = ids[0], b = ids[1];
        if (desc[a].w && desc[b].w) runChainPairW(chain[a], desc[a].w, chain[b], desc[b].w, kSlice);
        else if (desc[a].w || desc[b].w) { scalar(a, std::min(chain[a].e, chain[a].i + kSlice)); scalar(b, std::min(chain[b].e, chain[b].i + kSlice)); }
        else runChainPair(chain[a], desc[a].codes, desc[a].prob, chain[b], desc[b].codes, desc[b].prob, kSlice);
        if (chain[a].i >= chain[a].e) finishLane(a);
        if (chain[b].i >= chain[b].e) finishLane(b);
      } else {
        CoderChain& c = chain[ids[0]];
        scalar(ids[0], std::min(c.e, c.i + kSlice));
        if (c.i >= c.e) finishLane(ids[0]);
      }
      sourceDry = false;                                 // ask again: other blocks may have arrived
      continue;
    }
    uint64_t words = kMaxWords;
    uint32_t mask = 0;
    for (int l = 0; l < kAll; ++l) if (busy[l]) { words = std::min(words, (chain[l].e - chain[l].i) >> 4); mask |= 1u << l; }
    if (words > 0) {
      for (int l = 0; l < kAll; ++l) {
        if (busy[l]) {
          CoderChain& c = chain[l];
          L.lo[l] = c.lo; L.size[l] = c.size; L.codes[l] = desc[l].codes; L.prob[l] = wmode ? desc[l].w : desc[l].prob; L.i[l] = c.i;
          L.out[l] = c.room(words * 16); L.advance[l] = 16;
        } else {
          L.lo[l] = 0; L.size[l] = 0xFFFFFFFEu; L.codes[l] = kZeroCodes; L.prob[l] = kZeroProb; L.i[l] = 0;
          L.out[l] = nullptr; L.advance[l] = 0;
        }
      }
      if (two && (mask >> kLanes)) { if (wmode) runWords<true, 2>(L, words, mask); else runWords<false, 2>(L, words, mask); }
      else { if (wmode) runWords<true, 1>(L, words, mask); else runWords<false, 1>(L, words, mask); }
      for (int l = 0; l < kAll; ++l) {
        if (!busy[l]) continue;
        CoderChain& c = chain[l];
        c.lo = L.lo[l]; c.size = L.size[l]; c.i = L.i[l];
        c.used = static_cast<size_t>(L.out[l] - c.out->data());
      }
    }
    for (int l = 0; l < kAll; ++l)                       // lanes with less than a word left: scalar tail
      if (busy[l] && chain[l].e - chain[l].i < 16) finishLane(l);
    sourceDry = false;
  }
}

}  // namespace wavelet
}  // namespace bwtc
