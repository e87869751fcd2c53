// See wavelet_simd.hpp.  The scalar statement of the same models is GroupModel<TYPE, 'B'> in
// wavelet_host.cpp (probmodels/ProbabilityModel.cpp:38-75, BitPredictors.hpp:37-65,
// FSM.hpp:42-67, :196-205); the structures here repeat its rules in a type-generic form so
// that lanes of different group types can share a vector.  (A 32-lane form on 16-bit elements --
// every value fits 12 bits -- was tried: on the GPU box's Zen 5 its step is more than twice as
// slow as the 32-bit one, so it lost on both host time and latency.)
#include "wavelet_simd.hpp"

#include <immintrin.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "wavelet_host.hpp"

namespace bwtc {
namespace wavelet {
namespace {

const int kLanes = 16;
const int kSlots = 15;                   // main 0..7, gaps 8..11, integers 12..14
const uint64_t kScalarOnly = 256;        // groups shorter than this never enter a lane
const int kMinLanes = 6;                 // fewer busy lanes and an empty source: finish them scalar (on the GPU
                                         // box's EPYC a vector step costs as much as six scalar elements)

const uint32_t kInit[16] = {2400, 2300, 2200, 2100, 4096 - 2100, 4096 - 2200, 4096 - 2300, 4096 - 2400,
                            2048, 2048, 2048, 2048, 2048, 2048, 2048, 0};
const uint8_t kNext8[8][2] = {{0, 4}, {0, 4}, {1, 4}, {2, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 7}};
const uint8_t kNext4[4][2] = {{0, 2}, {0, 2}, {1, 3}, {1, 3}};
const uint8_t kNext3[3][2] = {{0, 1}, {0, 2}, {1, 2}};

inline uint32_t codeAt(const uint8_t* codes, uint64_t i) { return (codes[i >> 2] >> ((i & 3) * 2)) & 3u; }

// One group in scalar form; what a lane holds.
struct GroupState {
  uint32_t q[16];
  uint32_t mc, gc, ic;
  uint32_t isInt, gapAlways, gapFromCode, mainMoves;   // the group's type as 0/1 switches
  const uint8_t* codes;
  uint16_t* prob;
  uint64_t i, e;
  void* cookie;
  void start(const ModelGroupDesc& g) {
    for (int k = 0; k < 16; ++k) q[k] = kInit[k];
    mc = g.mainState; gc = 2; ic = 1;
    isInt = g.type == kInteger;
    gapAlways = g.type == kBothLeaves;
    gapFromCode = g.type == kInner || g.type == kLeftLeaf;
    mainMoves = g.type == kRoot || gapFromCode;
    codes = g.codes; prob = g.prob; i = g.begin; e = g.end; cookie = g.cookie;
  }
  inline void step() {
    const uint32_t v = codeAt(codes, i), bit = v & 1u;
    const uint32_t gap = gapAlways | (gapFromCode & (v >> 1));
    const uint32_t slot = isInt ? 12u + ic : (gap ? 8u + gc : mc);
    const uint32_t delay = 5u - static_cast<uint32_t>(slot == 0 || slot == 7);
    const uint32_t floor = isInt ? 100u : 2u;
    const uint32_t pr = q[slot];
    prob[i] = static_cast<uint16_t>(pr);
    const uint32_t up = pr + (((4096u - floor) - pr) >> delay), down = pr - ((pr - floor) >> delay);
    q[slot] = bit ? up : down;
    ic = isInt ? kNext3[ic][bit] : ic;
    gc = gap ? kNext4[gc][bit] : gc;
    mc = mainMoves ? kNext8[mc][bit] : mc;
    ++i;
  }
  void run(uint64_t until) { while (i < until) step(); }
};

// Lane state between vector runs, structure of arrays
struct Lanes {
  alignas(64) uint32_t q[kSlots][kLanes];
  alignas(64) uint32_t mc[kLanes], gc[kLanes], ic[kLanes];
  alignas(64) uint32_t isInt[kLanes], gapAlways[kLanes], gapFromCode[kLanes], mainMoves[kLanes];
  const uint8_t* codes[kLanes];
  uint16_t* prob[kLanes];
  uint64_t i[kLanes], e[kLanes];
  void* cookie[kLanes];
  bool busy[kLanes];

  void put(int l, const GroupState& g) {
    for (int s = 0; s < kSlots; ++s) q[s][l] = g.q[s];
    mc[l] = g.mc; gc[l] = g.gc; ic[l] = g.ic;
    isInt[l] = g.isInt; gapAlways[l] = g.gapAlways; gapFromCode[l] = g.gapFromCode; mainMoves[l] = g.mainMoves;
    codes[l] = g.codes; prob[l] = g.prob; i[l] = g.i; e[l] = g.e; cookie[l] = g.cookie;
    busy[l] = true;
  }
  void take(int l, GroupState* g) const {
    for (int s = 0; s < kSlots; ++s) g->q[s] = q[s][l];
    g->q[15] = 0;
    g->mc = mc[l]; g->gc = gc[l]; g->ic = ic[l];
    g->isInt = isInt[l]; g->gapAlways = gapAlways[l]; g->gapFromCode = gapFromCode[l]; g->mainMoves = mainMoves[l];
    g->codes = codes[l]; g->prob = prob[l]; g->i = i[l]; g->e = e[l]; g->cookie = cookie[l];
  }
  void idle(int l) {
    busy[l] = false;
    isInt[l] = gapAlways[l] = gapFromCode[l] = mainMoves[l] = 0;
    mc[l] = gc[l] = ic[l] = 0;
  }
};

#define BWTC_AVX512 __attribute__((target("avx512f,avx512bw,avx512vl,avx512dq")))

// `words` 16-element words of every busy lane (all busy lanes have that many left, and stand
// on a word boundary).  MAIN / GAPS / INTS: which of the three models any busy lane uses; a
// vector of integer-level lanes only touches 3 of the 15 predictors.
template <bool MAIN, bool GAPS, bool INTS>
BWTC_AVX512 void runWords(Lanes& L, uint64_t words) {
  constexpr int kLo = MAIN ? 0 : GAPS ? 8 : 12, kHi = INTS ? 15 : GAPS ? 12 : 8;
  __m512i Q[kSlots];
  for (int s = kLo; s < kHi; ++s) Q[s] = _mm512_load_si512(L.q[s]);
  __m512i mc = _mm512_load_si512(L.mc), gc = _mm512_load_si512(L.gc), ic = _mm512_load_si512(L.ic);
  const __m512i zero = _mm512_setzero_si512(), one = _mm512_set1_epi32(1), two = _mm512_set1_epi32(2);
  const __mmask16 kInt = INTS ? _mm512_cmpneq_epi32_mask(_mm512_load_si512(L.isInt), zero) : static_cast<__mmask16>(0);
  const __mmask16 kGapAlways = GAPS ? _mm512_cmpneq_epi32_mask(_mm512_load_si512(L.gapAlways), zero) : static_cast<__mmask16>(0);
  const __mmask16 kGapCode = GAPS ? _mm512_cmpneq_epi32_mask(_mm512_load_si512(L.gapFromCode), zero) : static_cast<__mmask16>(0);
  const __mmask16 kMainMoves = MAIN ? _mm512_cmpneq_epi32_mask(_mm512_load_si512(L.mainMoves), zero) : static_cast<__mmask16>(0);
  const __m512i floorV = _mm512_mask_blend_epi32(kInt, two, _mm512_set1_epi32(100));
  const __m512i topV = _mm512_sub_epi32(_mm512_set1_epi32(4096), floorV);
  const __m512i five = _mm512_set1_epi32(5), c8 = _mm512_set1_epi32(8), c12 = _mm512_set1_epi32(12);
  const __m512i T8 = _mm512_setr_epi32(0, 4, 0, 4, 1, 4, 2, 4, 3, 5, 3, 6, 3, 7, 3, 7);
  const __m512i T4 = _mm512_setr_epi32(0, 2, 0, 2, 1, 3, 1, 3, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m512i T3 = _mm512_setr_epi32(0, 1, 0, 2, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  alignas(64) uint32_t in[kLanes];
  alignas(64) uint32_t out[8][kLanes];
  for (uint64_t w = 0; w < words; ++w) {
    for (int l = 0; l < kLanes; ++l) {
      uint32_t x = 0;
      if (L.busy[l]) std::memcpy(&x, L.codes[l] + (L.i[l] >> 2), 4);
      in[l] = x;
    }
    __m512i W = _mm512_load_si512(in);
    __m512i lo = zero;
    for (int step = 0; step < 16; ++step) {
      const __m512i v = _mm512_and_si512(W, _mm512_set1_epi32(3));
      W = _mm512_srli_epi32(W, 2);
      const __mmask16 kBit = _mm512_test_epi32_mask(v, one);
      const __mmask16 kGap = GAPS ? static_cast<__mmask16>(kGapAlways | (kGapCode & _mm512_test_epi32_mask(v, two))) : static_cast<__mmask16>(0);
      __m512i slot = mc;
      if (GAPS) slot = _mm512_mask_add_epi32(slot, kGap, gc, c8);
      if (INTS) slot = _mm512_mask_add_epi32(slot, kInt, ic, c12);
      // the lane's predictor: every slot's vector masked to the lanes that use it, OR-ed as a tree
      __mmask16 m[kSlots];
      __m512i part[16];
      for (int s = 0; s < 16; ++s) part[s] = zero;
      for (int s = kLo; s < kHi; ++s) {
        m[s] = _mm512_cmpeq_epi32_mask(slot, _mm512_set1_epi32(s));
        part[s - kLo] = _mm512_maskz_mov_epi32(m[s], Q[s]);
      }
      constexpr int kWidth = (kHi - kLo) > 8 ? 8 : (kHi - kLo) > 4 ? 4 : 2;
      for (int width = kWidth; width >= 1; width >>= 1)
        for (int k = 0; k < width; ++k) part[k] = _mm512_or_si512(part[k], part[k + width]);
      const __m512i pr = part[0];
      __m512i delay = five;
      if (MAIN) delay = _mm512_mask_sub_epi32(five, m[0] | m[7], five, one);
      const __m512i up = _mm512_add_epi32(pr, _mm512_srlv_epi32(_mm512_sub_epi32(topV, pr), delay));
      const __m512i down = _mm512_sub_epi32(pr, _mm512_srlv_epi32(_mm512_sub_epi32(pr, floorV), delay));
      const __m512i moved = _mm512_mask_blend_epi32(kBit, down, up);
      for (int s = kLo; s < kHi; ++s) Q[s] = _mm512_mask_mov_epi32(Q[s], m[s], moved);
      const __m512i bit = _mm512_and_si512(v, one);
      if (MAIN) mc = _mm512_mask_mov_epi32(mc, kMainMoves, _mm512_permutexvar_epi32(_mm512_add_epi32(_mm512_slli_epi32(mc, 1), bit), T8));
      if (GAPS) gc = _mm512_mask_mov_epi32(gc, kGap, _mm512_permutexvar_epi32(_mm512_add_epi32(_mm512_slli_epi32(gc, 1), bit), T4));
      if (INTS) ic = _mm512_mask_mov_epi32(ic, kInt, _mm512_permutexvar_epi32(_mm512_add_epi32(_mm512_slli_epi32(ic, 1), bit), T3));
      if (step & 1) _mm512_store_si512(out[step >> 1], _mm512_or_si512(lo, _mm512_slli_epi32(pr, 16)));
      else lo = pr;
    }
    for (int l = 0; l < kLanes; ++l) {
      if (!L.busy[l]) continue;
      uint32_t row[8];
      for (int k = 0; k < 8; ++k) row[k] = out[k][l];
      std::memcpy(L.prob[l] + L.i[l], row, 32);
      L.i[l] += 16;
    }
  }
  for (int s = kLo; s < kHi; ++s) _mm512_store_si512(L.q[s], Q[s]);
  if (MAIN) _mm512_store_si512(L.mc, mc);
  if (GAPS) _mm512_store_si512(L.gc, gc);
  if (INTS) _mm512_store_si512(L.ic, ic);
}

// picks the narrowest form that covers the busy lanes
void runWordsFor(Lanes& L, uint64_t words) {
  bool main = false, gaps = false, ints = false;
  for (int l = 0; l < kLanes; ++l) {
    if (!L.busy[l]) continue;
    if (L.isInt[l]) ints = true;
    else {
      if (L.gapAlways[l] || L.gapFromCode[l]) gaps = true;
      if (!L.gapAlways[l]) main = true;
    }
  }
  if (ints && !main && !gaps) runWords<false, false, true>(L, words);
  else if (!ints && main && !gaps) runWords<true, false, false>(L, words);
  else if (!ints) runWords<true, true, false>(L, words);
  else runWords<true, true, true>(L, words);
}

}  // namespace

bool simdModelsAvailable() {
  static const bool ok = [] {
    const char* off = std::getenv("BWTC_HIP_SIMD");
    if (off && off[0] == '0') return false;
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
           __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512dq");
  }();
  return ok;
}

void runModelLanes(GroupSource& src, int prefer) {
  Lanes L;
  std::memset(&L, 0, sizeof L);
  for (int l = 0; l < kLanes; ++l) L.idle(l);
  bool sourceDry = false;
  for (;;) {
    // refill: a new group runs scalar up to the next word boundary (all of it when it is short)
    int busy = 0;
    for (int l = 0; l < kLanes; ++l) {
      while (!L.busy[l] && !sourceDry) {
        ModelGroupDesc d;
        if (!src.next(&d, prefer)) { sourceDry = true; break; }
        GroupState g;
        g.start(d);
        const uint64_t aligned = (g.i + 15) & ~static_cast<uint64_t>(15);
        if (g.e - g.i < kScalarOnly || aligned + 16 > g.e) { g.run(g.e); src.done(g.cookie); continue; }
        g.run(aligned);
        L.put(l, g);
      }
      busy += L.busy[l];
    }
    if (busy == 0) {
      if (sourceDry && src.retire(prefer)) return;
      sourceDry = false;
      continue;
    }
    if (busy < kMinLanes && sourceDry) {                               // not worth a vector any more
      for (int l = 0; l < kLanes; ++l) {
        if (!L.busy[l]) continue;
        GroupState g;
        L.take(l, &g);
        g.run(g.e);
        src.done(g.cookie);
        L.idle(l);
      }
      sourceDry = false;                                               // the source may have been given more groups meanwhile
      continue;
    }
    uint64_t words = ~static_cast<uint64_t>(0);
    for (int l = 0; l < kLanes; ++l) if (L.busy[l]) words = std::min(words, (L.e[l] - L.i[l]) >> 4);
    if (words > 0) runWordsFor(L, std::min<uint64_t>(words, 1u << 16));
    for (int l = 0; l < kLanes; ++l) {                                 // lanes with less than a word left: scalar tail
      if (!L.busy[l] || L.e[l] - L.i[l] >= 16) continue;
      GroupState g;
      L.take(l, &g);
      g.run(g.e);
      src.done(g.cookie);
      L.idle(l);
    }
    sourceDry = false;                                                 // ask again: other blocks may have arrived
  }
}

}  // namespace wavelet
}  // namespace bwtc
