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
#include "wavelet_rc.hpp"

#include <chrono>

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
  // the same step without the store: returns the probability of a one, *bit_out = the coded bit
  inline uint32_t stepP(uint32_t* bit_out) {
    const uint32_t v = codeAt(codes, i), bit = v & 1u;
    const uint32_t gap = gapAlways | (gapFromCode & (v >> 1));
    const uint32_t slot = isInt ? 12u + ic : (gap ? 8u + gc : mc);
    const uint32_t delay = 5u - static_cast<uint32_t>(slot == 0 || slot == 7);
    const uint32_t floor = isInt ? 100u : 2u;
    const uint32_t pr = q[slot];
    const uint32_t up = pr + (((4096u - floor) - pr) >> delay), down = pr - ((pr - floor) >> delay);
    q[slot] = bit ? up : down;
    ic = isInt ? kNext3[ic][bit] : ic;
    gc = gap ? kNext4[gc][bit] : gc;
    mc = mainMoves ? kNext8[mc][bit] : mc;
    ++i;
    *bit_out = bit;
    return pr;
  }
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


// ---- fused engine ------------------------------------------------------------------------------
namespace {

const uint64_t kFusedMaxWords = 256;     // words per vector run (bounds the output room reserved per lane)

struct FusedLanes : Lanes {
  alignas(64) uint32_t lo[kLanes], size[kLanes];
  uint8_t* outp[kLanes];
};

// one lane's section between vector runs
struct FusedChain {
  FusedSectionDesc d;
  size_t g;                 // current group
  GroupState m;             // its model (m.i = the chain's position)
  CoderChain c;
  void startGroup() {
    ModelGroupDesc gd;
    gd.codes = d.codes; gd.prob = nullptr; gd.begin = d.groups[g].begin; gd.end = d.groups[g].end;
    gd.type = d.groups[g].type; gd.mainState = d.groups[g].mainState; gd.cookie = nullptr;
    m.start(gd);
  }
  // scalar: model + coder, elements [m.i, until)
  void runScalar(uint64_t until) {
    while (m.i < until) {
      const uint64_t ce = std::min<uint64_t>(until, m.i + 4096);
      uint8_t* o = c.room(ce - m.i);
      uint32_t lo = c.lo, size = c.size;
      while (m.i < ce) {
        uint32_t bit;
        const uint32_t p = m.stepP(&bit);
        const uint32_t t = static_cast<uint32_t>((static_cast<uint64_t>(size) * p + 2048u) >> 12);
        lo = bit ? lo : lo + t + 1;
        size = bit ? t - 1 : size - t - 1;
        while (((lo ^ (lo + size + 1)) & 0xFF000000u) == 0) { *o++ = static_cast<uint8_t>(lo >> 24); lo <<= 8; size = (size << 8) + 510u; }
      }
      c.lo = lo; c.size = size;
      c.used = static_cast<size_t>(o - c.out->data());
    }
  }
  // Brings the lane to the next place where a vector run can start: at least one whole word of
  // the current group ahead and the position on a word boundary.  Groups too short for that are
  // coded here.  false: the section is finished (flushed).
  bool settle() {
    for (;;) {
      if (m.i >= m.e) {
        if (++g >= d.n_groups) { c.finish(); return false; }
        startGroup();
      }
      const uint64_t aligned = (m.i + 15) & ~static_cast<uint64_t>(15);
      if (m.e - m.i < kScalarOnly || aligned + 16 > m.e) { runScalar(m.e); continue; }
      if (m.i < aligned) runScalar(aligned);
      return true;
    }
  }
};

// `words` 16-element words of every busy lane; each stands on a word boundary with that many
// words of its current group ahead.  The coder half is wavelet_rc.cpp's runWords; the probability
// goes from the model to it in a register.
//
// The model half keeps the fifteen predictors of the sixteen lanes as 16-bit values, two slots to
// a register (R[k]: slot 2k in elements 0..15, slot 2k+1 in 16..31).  A lane's predictor is then
// picked by four two-register word permutes (index = slot's low two bits and the lane) and three
// blends on the slot's upper two bits, and put back by eight word compares + blends -- 30
// operations where the form of runWords (a compare, a masked move and an OR per slot to select,
// a masked move per slot to update) takes 60; the loop is bound by the number of vector
// operations, not by a dependency chain.
BWTC_AVX512 void runFusedWords(FusedLanes& L, uint64_t words, uint32_t busy_mask) {
  __m512i R[8];
  for (int k = 0; k < 8; ++k) {
    const __m256i a = _mm512_cvtepi32_epi16(_mm512_load_si512(L.q[2 * k]));
    const __m256i b = 2 * k + 1 < kSlots ? _mm512_cvtepi32_epi16(_mm512_load_si512(L.q[2 * k + 1])) : _mm256_setzero_si256();
    R[k] = _mm512_inserti64x4(_mm512_castsi256_si512(a), b, 1);
  }
  __m512i mc = _mm512_load_si512(L.mc), gc = _mm512_load_si512(L.gc), ic = _mm512_load_si512(L.ic);
  __m512i lo = _mm512_load_si512(L.lo), size = _mm512_load_si512(L.size);
  const __m512i zero = _mm512_setzero_si512(), one = _mm512_set1_epi32(1), two = _mm512_set1_epi32(2);
  const __mmask16 kInt = _mm512_cmpneq_epi32_mask(_mm512_load_si512(L.isInt), zero);
  const __mmask16 kGapAlways = _mm512_cmpneq_epi32_mask(_mm512_load_si512(L.gapAlways), zero);
  const __mmask16 kGapCode = _mm512_cmpneq_epi32_mask(_mm512_load_si512(L.gapFromCode), zero);
  const __mmask16 kMainMoves = _mm512_cmpneq_epi32_mask(_mm512_load_si512(L.mainMoves), zero);
  const __mmask16 kBusy = static_cast<__mmask16>(busy_mask);
  const __m512i floorV = _mm512_mask_blend_epi32(kInt, two, _mm512_set1_epi32(100));
  const __m512i topV = _mm512_sub_epi32(_mm512_set1_epi32(4096), floorV);
  const __m512i c8 = _mm512_set1_epi32(8), c12 = _mm512_set1_epi32(12);
  const __m512i four = _mm512_set1_epi32(4);
  const __m512i Tdelay = _mm512_setr_epi32(4, 5, 5, 5, 5, 5, 5, 4, 5, 5, 5, 5, 5, 5, 5, 5);
  const __m512i T8 = _mm512_setr_epi32(0, 4, 0, 4, 1, 4, 2, 4, 3, 5, 3, 6, 3, 7, 3, 7);
  const __m512i T4 = _mm512_setr_epi32(0, 2, 0, 2, 1, 3, 1, 3, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m512i T3 = _mm512_setr_epi32(0, 1, 0, 2, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m512i c4095 = _mm512_set1_epi32(4095), c2048 = _mm512_set1_epi32(2048), c510 = _mm512_set1_epi32(510);
  const __m512i top = _mm512_set1_epi32(static_cast<int>(0xFF000000u));
  const __m512i lane_id = _mm512_slli_epi32(_mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15), 8);
  // 16-bit constants: the lane number of every element, and per register the two slots it holds
  alignas(64) uint16_t iota_w[32], slots_w[8][32];
  for (int i = 0; i < 32; ++i) {
    iota_w[i] = static_cast<uint16_t>(i & 15);
    for (int k = 0; k < 8; ++k) slots_w[k][i] = static_cast<uint16_t>(2 * k + (i >> 4));
  }
  const __m512i iotaW = _mm512_load_si512(iota_w), c48w = _mm512_set1_epi16(0x30);
  __m512i slotsW[8];
  for (int k = 0; k < 8; ++k) slotsW[k] = _mm512_load_si512(slots_w[k]);
  alignas(64) uint32_t in[kLanes];
  alignas(64) uint32_t ev[16 * kLanes * 4 + 64];
  for (uint64_t w = 0; w < words; ++w) {
    for (int l = 0; l < kLanes; ++l) {
      uint32_t x = 0;
      if (L.busy[l]) { std::memcpy(&x, L.codes[l] + (L.i[l] >> 2), 4); L.i[l] += 16; }
      in[l] = x;
    }
    __m512i W = _mm512_load_si512(in);
    uint32_t n = 0;
    for (int step = 0; step < 16; ++step) {
      // ---- model: the lane's predictor and its move
      const __m512i v = _mm512_and_si512(W, _mm512_set1_epi32(3));
      W = _mm512_srli_epi32(W, 2);
      const __mmask16 kBit = _mm512_test_epi32_mask(v, one);
      const __mmask16 kGap = static_cast<__mmask16>(kGapAlways | (kGapCode & _mm512_test_epi32_mask(v, two)));
      __m512i slot = mc;
      slot = _mm512_mask_add_epi32(slot, kGap, gc, c8);
      slot = _mm512_mask_add_epi32(slot, kInt, ic, c12);
      const __m512i slotW = _mm512_broadcast_i64x4(_mm512_cvtepi32_epi16(slot));         // the lanes' slots, in both halves
      const __m512i idx = _mm512_or_si512(_mm512_and_si512(_mm512_slli_epi16(slotW, 4), c48w), iotaW);
      const __m512i p0 = _mm512_permutex2var_epi16(R[0], idx, R[1]), p1 = _mm512_permutex2var_epi16(R[2], idx, R[3]);
      const __m512i p2 = _mm512_permutex2var_epi16(R[4], idx, R[5]), p3 = _mm512_permutex2var_epi16(R[6], idx, R[7]);
      const __mmask32 k4 = _mm512_test_epi32_mask(slot, four), k8 = _mm512_test_epi32_mask(slot, c8);
      const __m512i prW = _mm512_mask_blend_epi16(k8, _mm512_mask_blend_epi16(k4, p0, p1), _mm512_mask_blend_epi16(k4, p2, p3));
      const __m512i pr = _mm512_cvtepu16_epi32(_mm512_castsi512_si256(prW));
      const __m512i delay = _mm512_permutexvar_epi32(slot, Tdelay);          // 4 for the main model's two end states, else 5
      const __m512i up = _mm512_add_epi32(pr, _mm512_srlv_epi32(_mm512_sub_epi32(topV, pr), delay));
      const __m512i down = _mm512_sub_epi32(pr, _mm512_srlv_epi32(_mm512_sub_epi32(pr, floorV), delay));
      const __m512i moved = _mm512_mask_blend_epi32(kBit, down, up);
      const __m512i movedW = _mm512_broadcast_i64x4(_mm512_cvtepi32_epi16(moved));
      for (int k = 0; k < 8; ++k) R[k] = _mm512_mask_blend_epi16(_mm512_cmpeq_epi16_mask(slotW, slotsW[k]), R[k], movedW);
      const __m512i bit = _mm512_and_si512(v, one);
      mc = _mm512_mask_mov_epi32(mc, kMainMoves, _mm512_permutexvar_epi32(_mm512_add_epi32(_mm512_slli_epi32(mc, 1), bit), T8));
      gc = _mm512_mask_mov_epi32(gc, kGap, _mm512_permutexvar_epi32(_mm512_add_epi32(_mm512_slli_epi32(gc, 1), bit), T4));
      ic = _mm512_mask_mov_epi32(ic, kInt, _mm512_permutexvar_epi32(_mm512_add_epi32(_mm512_slli_epi32(ic, 1), bit), T3));
      // ---- coder: BitEncoder's step with that probability (wavelet_rc.cpp runWords)
      const __m512i hi = _mm512_srli_epi32(size, 12), lw = _mm512_and_si512(size, c4095);
      const __m512i tt = _mm512_add_epi32(_mm512_mullo_epi32(hi, pr),
                                          _mm512_srli_epi32(_mm512_add_epi32(_mm512_mullo_epi32(lw, pr), c2048), 12));
      const __m512i t1 = _mm512_add_epi32(tt, one);
      lo = _mm512_mask_add_epi32(lo, static_cast<__mmask16>(~kBit), lo, t1);
      size = _mm512_mask_blend_epi32(kBit, _mm512_sub_epi32(size, t1), _mm512_sub_epi32(tt, one));
      __m512i hiend = _mm512_add_epi32(_mm512_add_epi32(lo, size), one);
      __mmask16 em = _mm512_mask_testn_epi32_mask(kBusy, _mm512_xor_si512(lo, hiend), top);
      for (;;) {
        _mm512_storeu_si512(ev + n, _mm512_maskz_compress_epi32(em, _mm512_or_si512(_mm512_srli_epi32(lo, 24), lane_id)));
        n += static_cast<uint32_t>(__builtin_popcount(em));
        lo = _mm512_mask_slli_epi32(lo, em, lo, 8);
        size = _mm512_mask_add_epi32(size, em, _mm512_slli_epi32(size, 8), c510);
        hiend = _mm512_add_epi32(_mm512_add_epi32(lo, size), one);
        em = _mm512_mask_testn_epi32_mask(em, _mm512_xor_si512(lo, hiend), top);
        if (__builtin_expect(em == 0, 1)) break;
      }
    }
    for (uint32_t k = 0; k < n; ++k) { const uint32_t e = ev[k]; *L.outp[e >> 8]++ = static_cast<uint8_t>(e); }
  }
  for (int k = 0; k < 8; ++k) {
    _mm512_store_si512(L.q[2 * k], _mm512_cvtepu16_epi32(_mm512_castsi512_si256(R[k])));
    if (2 * k + 1 < kSlots) _mm512_store_si512(L.q[2 * k + 1], _mm512_cvtepu16_epi32(_mm512_extracti64x4_epi64(R[k], 1)));
  }
  _mm512_store_si512(L.mc, mc);
  _mm512_store_si512(L.gc, gc);
  _mm512_store_si512(L.ic, ic);
  _mm512_store_si512(L.lo, lo);
  _mm512_store_si512(L.size, size);
}

}  // namespace

void runFusedLanes(SectionSource& src, std::atomic<uint64_t>* busy_ns) {
  auto tick = std::chrono::steady_clock::now();
  auto account = [&] {
    if (!busy_ns) return;
    const auto now = std::chrono::steady_clock::now();
    *busy_ns += static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(now - tick).count());
    tick = now;
  };
  static FusedLanes* dummy = nullptr; (void)dummy;
  std::vector<FusedChain> chain(kLanes);
  FusedLanes L;
  std::memset(static_cast<Lanes*>(&L), 0, sizeof(Lanes));
  for (int l = 0; l < kLanes; ++l) { L.idle(l); L.lo[l] = 0; L.size[l] = 0xFFFFFFFEu; L.outp[l] = nullptr; }
  bool sourceDry = false;
  for (;;) {
    int busy = 0;
    for (int l = 0; l < kLanes; ++l) {
      while (!L.busy[l] && !sourceDry) {
        FusedChain& ch = chain[l];
        if (!src.next(&ch.d)) { sourceDry = true; break; }
        ch.c.start(ch.d.n_groups ? ch.d.groups[0].begin : 0, ch.d.n_groups ? ch.d.groups[ch.d.n_groups - 1].end : 0, ch.d.out);
        if (ch.d.n_groups == 0) { ch.c.finish(); src.done(ch.d.cookie); continue; }
        ch.g = 0;
        ch.startGroup();
        if (!ch.settle()) { src.done(ch.d.cookie); continue; }
        L.busy[l] = true;
      }
      busy += L.busy[l];
    }
    account();
    if (busy == 0) {
      if (sourceDry && src.retire()) return;
      sourceDry = false;
      continue;
    }
    if (busy < kMinLanes && sourceDry) {
      // too few chains for a vector: a slice of each, scalar; then ask the source again
      for (int l = 0; l < kLanes; ++l) {
        if (!L.busy[l]) continue;
        FusedChain& ch = chain[l];
        ch.runScalar(std::min<uint64_t>(ch.m.e, ch.m.i + (static_cast<uint64_t>(1) << 20)));
        if (!ch.settle()) { src.done(ch.d.cookie); L.idle(l); }
      }
      sourceDry = false;
      continue;
    }
    uint64_t words = kFusedMaxWords;
    uint32_t mask = 0;
    for (int l = 0; l < kLanes; ++l) if (L.busy[l]) { words = std::min(words, (chain[l].m.e - chain[l].m.i) >> 4); mask |= 1u << l; }
    for (int l = 0; l < kLanes; ++l) {
      if (!L.busy[l]) continue;
      FusedChain& ch = chain[l];
      L.put(l, ch.m);                              // predictors, state machines, type switches, position
      L.lo[l] = ch.c.lo; L.size[l] = ch.c.size;
      L.outp[l] = ch.c.room(words * 16);
    }
    runFusedWords(L, words, mask);
    for (int l = 0; l < kLanes; ++l) {
      if (!L.busy[l]) continue;
      FusedChain& ch = chain[l];
      L.take(l, &ch.m);
      ch.c.lo = L.lo[l]; ch.c.size = L.size[l];
      ch.c.i = ch.m.i;
      ch.c.used = static_cast<size_t>(L.outp[l] - ch.c.out->data());
      if (ch.m.e - ch.m.i < 16 && !ch.settle()) { src.done(ch.d.cookie); L.idle(l); }
    }
    sourceDry = false;
  }
}

}  // namespace wavelet
}  // namespace bwtc
