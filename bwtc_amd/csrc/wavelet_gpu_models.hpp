// The adaptive models of the 'B' coder as data-parallel passes (device: wavelet_gpu_models.hip;
// the same lane functions run on the host for the CPU tests, modelsOnHostLanes below).
//
// What the reference does (WaveletTree.hpp:637-809 with the models of
// probmodels/ProbabilityModel.cpp:38-75): every coded element is coded with the probability of
// ONE of fifteen small predictors -- the eight UnbiasedPredictors of FSM8 (main model), the four
// of FSM<4> (gaps), the three of FSM<3> (integer codes) -- picked by three tiny state machines
// that only look at the coded bits, and that predictor then moves 1/2^delay of the way towards
// the coded bit (BitPredictors.hpp:37-65).  A model task (a symbol-tree node, or one integer
// level) starts with fresh predictors; the main machine's state runs on from task to task, section
// to section and block to block (FSM8::resetModel keeps it, FSM.hpp:196-205).
//
// Serial as written, but the structure is: (1) state machines = finite automata over the bits:
// a composition scan gives every chunk its exact start states; (2) given the states, every
// element names its predictor ("slot"), and the fifteen predictors of a task are INDEPENDENT
// one-dimensional chains  x <- x +- ((bound - x) >> d)  over the subsequence of bits that name
// them.  Those subsequences are gathered, stably, into one bit array ("slot space": slot-major,
// then task, then order), where every chain is contiguous.  (3) A chain is cut into slot-chunks.
// Both update maps are monotone and non-expanding on integers, so the images of the two extreme
// start values bracket every start value; after kWarm updates the bracket is narrower than 2^d
// (checked, not assumed).  Inside a bracket that narrow a step merges AT MOST ONE pair of
// neighbouring candidates -- the pair (v, v+1) with (bound - v) = 0 mod 2^d -- so the whole map
// "value at the chunk's start -> value at its end" is carried as (image of the lowest candidate,
// bit mask of the surviving increments): O(1) per step.  A short serial walk over a chain's
// chunks (one table look-up each) then gives every slot-chunk its TRUE start value; the bracket
// pass has also left the bracket as it stood at every 32nd position, which that start value
// resolves to the true value there; and the last pass walks the elements in coding order with
// exact predictors (each started from the nearest such position) and writes, per element,
//     w = bit << 15 | probability of the coded bit (12 bits),
// which is all the range coder needs (wavelet_rc.hpp, runChainW).  Nothing is approximated: the
// bytes are those of the sequential encoder, and an input the scheme cannot bracket raises an
// error flag instead of a wrong byte.
#pragma once
#include <cstdint>
#include <vector>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BWTC_GM_HD __host__ __device__ __forceinline__
#else
#define BWTC_GM_HD inline
#endif

namespace bwtc {
namespace wavelet {
struct StreamPlan;
namespace gm {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr u32 kChunk = 2048;       // elements per chunk: tasks cut along an aligned grid of the packed stream
constexpr u32 kSlotChunk = 2048;   // slot-space positions per slot-chunk
constexpr u32 kWarm = 256;         // updates from the extreme values before a slot-chunk (bracket < 2^d after ~175)
constexpr u32 kSample = 32;        // the bracket is kept at every kSample-th slot-space position
constexpr u32 kSlots = 15;
constexpr u32 kSlotStride = 16;    // rows of the per-lane slot tables
enum TaskType { kTRoot = 0, kTGaps = 1, kTInner = 2, kTInts = 3 };

struct Task { u32 begin, end, type, first_chunk; };      // elements [begin, end) of the packed stream
struct Chunk { u32 begin, end, task_first; };            // task | (first chunk of its task) << 31

// error bits (a set bit means: this block is not coded from these probabilities)
constexpr u32 kErrBracket = 1, kErrChain = 2, kErrTotal = 4;

// ---- the fifteen predictors -------------------------------------------------------------------
// slot 0..7 main (kept as the probability of a ONE: the upper four are InversePredictors,
// 4096 - p), 8..11 gaps, 12..14 integer codes
BWTC_GM_HD u32 slotFloor(u32 k) { return k >= 12 ? 100u : 2u; }
BWTC_GM_HD u32 slotDelay(u32 k) { return (k == 0 || k == 7) ? 4u : 5u; }
BWTC_GM_HD u32 slotInit(u32 k) { return k < 4 ? 2400u - 100u * k : k < 8 ? 1996u - 100u * (k - 4u) : 2048u; }
BWTC_GM_HD u32 moved(u32 x, u32 bit, u32 floor, u32 d) {
  return bit ? x + (((4096u - floor) - x) >> d) : x - ((x - floor) >> d);
}

// ---- the three state machines (FSM.hpp:42-67) ---------------------------------------------------
BWTC_GM_HD u32 next8(u32 s, u32 bit) { return bit ? (s >= 4 ? (s < 7 ? s + 1 : 7u) : 4u) : (s < 4 ? (s ? s - 1 : 0u) : 3u); }
BWTC_GM_HD u32 next4(u32 s, u32 bit) { return (bit << 1) | (s >> 1); }
BWTC_GM_HD u32 next3(u32 s, u32 bit) { return bit ? (s < 2 ? s + 1 : 2u) : (s ? s - 1 : 0u); }

// packed states of a chunk's start: mc | gc << 3 | ic << 5
BWTC_GM_HD u32 packState(u32 mc, u32 gc, u32 ic) { return mc | (gc << 3) | (ic << 5); }

// A chunk's effect on the three machines as maps: bits 0..23 the images of main states 0..7
// (3 bits each), 24..31 of gap states 0..3 (2 bits each), 32..37 of integer states 0..2.
constexpr u64 kMapIdentity = 0xFAC688ull | (0xE4ull << 24) | (0x24ull << 32);
BWTC_GM_HD u64 mapConstGapsInts(u64 m, u32 gc, u32 ic) {           // the machines a task start resets
  const u64 g = gc * 0x55u, i = ic * 0x15u;
  return (m & 0xFFFFFFull) | (g << 24) | (i << 32);
}
// g after f
BWTC_GM_HD u64 mapCompose(u64 f, u64 g) {
  u64 h = 0;
  for (u32 i = 0; i < 8; ++i) { const u32 fi = (u32)(f >> (3 * i)) & 7u; h |= (u64)((u32)(g >> (3 * fi)) & 7u) << (3 * i); }
  for (u32 i = 0; i < 4; ++i) { const u32 fi = (u32)(f >> (24 + 2 * i)) & 3u; h |= (u64)((u32)(g >> (24 + 2 * fi)) & 3u) << (24 + 2 * i); }
  for (u32 i = 0; i < 3; ++i) { const u32 fi = (u32)(f >> (32 + 2 * i)) & 3u; h |= (u64)((u32)(g >> (32 + 2 * fi)) & 3u) << (32 + 2 * i); }
  return h;
}
BWTC_GM_HD u32 mapApply(u64 m, u32 st) {
  const u32 mc = st & 7u, gc = (st >> 3) & 3u, ic = (st >> 5) & 3u;
  return packState((u32)(m >> (3 * mc)) & 7u, (u32)(m >> (24 + 2 * gc)) & 3u, (u32)(m >> (32 + 2 * ic)) & 3u);
}

BWTC_GM_HD u32 codeAt(const u32* packed, u32 i) { return (packed[i >> 4] >> ((i & 15u) * 2u)) & 3u; }
BWTC_GM_HD u32 bitAt(const u32* bits, u32 p) { return (bits[p >> 5] >> (p & 31u)) & 1u; }
BWTC_GM_HD u32 popc(u32 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (u32)__popc(v);
#else
  return (u32)__builtin_popcount(v);
#endif
}

// f(i, v) for the elements [begin, end) of the packed stream: 64 elements (16 bytes) per load, the
// next load issued before the current piece is walked (a lane's loads are a serial chain of HBM /
// L2 latencies otherwise); the packed array is padded to whole 16-byte pieces.
struct Piece { u32 w[4]; };
BWTC_GM_HD Piece loadPiece(const u32* packed, u32 q) {
  Piece p;
#if defined(__HIP_DEVICE_COMPILE__)
  const uint4 v = reinterpret_cast<const uint4*>(packed)[q];
  p.w[0] = v.x; p.w[1] = v.y; p.w[2] = v.z; p.w[3] = v.w;
#else
  for (u32 k = 0; k < 4; ++k) p.w[k] = packed[q * 4u + k];
#endif
  return p;
}
BWTC_GM_HD u32 pieceWord(const Piece& p, u32 k) {           // selects, not an indexed (scratch) access
  const u32 a = k & 1u ? p.w[1] : p.w[0], b = k & 1u ? p.w[3] : p.w[2];
  return k & 2u ? b : a;
}
template <class F>
BWTC_GM_HD void forElements(const u32* packed, u32 begin, u32 end, F f) {
  if (begin >= end) return;
  const u32 q0 = begin >> 6, q1 = (end - 1u) >> 6;
  Piece cur = loadPiece(packed, q0);
  for (u32 q = q0; q <= q1; ++q) {
    const Piece nxt = loadPiece(packed, q < q1 ? q + 1u : q);
    const u32 base = q << 6;
    if (base >= begin && base + 64u <= end) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
      for (u32 k = 0; k < 4; ++k) {
        u32 word = pieceWord(cur, k);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 4
#endif
        for (u32 e = 0; e < 16; ++e, word >>= 2) f(base + k * 16u + e, word & 3u);
      }
    } else {
      for (u32 k = 0; k < 4; ++k) {
        const u32 wb = base + k * 16u;
        const u32 lo = wb > begin ? wb : begin, hi = wb + 16u < end ? wb + 16u : end;
        if (lo >= hi) continue;
        u32 word = pieceWord(cur, k) >> ((lo & 15u) * 2u);
        for (u32 i = lo; i < hi; ++i, word >>= 2) f(i, word & 3u);
      }
    }
    cur = nxt;
  }
}
// f(p, bit) for the positions [p0, p1) of a bit array, one load per thirty-two
template <class F>
BWTC_GM_HD void forBits(const u32* bits, u32 p0, u32 p1, F f) {
  if (p0 >= p1) return;
  for (u32 wi = p0 >> 5; wi <= (p1 - 1u) >> 5; ++wi) {
    const u32 lo = wi << 5 > p0 ? wi << 5 : p0, hi = (wi + 1u) << 5 < p1 ? (wi + 1u) << 5 : p1;
    u32 word = bits[wi] >> (lo & 31u);
    for (u32 p = lo; p < hi; ++p, word >>= 1) f(p, word & 1u);
  }
}

// One element's slot and the machines' next states.
struct Machines { u32 mc, gc, ic; };
BWTC_GM_HD u32 stepMachines(u32 type, u32 v, Machines& m) {
  const u32 bit = v & 1u, flag = v >> 1;
  u32 slot;
  if (type == kTRoot) { slot = m.mc; m.mc = next8(m.mc, bit); }
  else if (type == kTGaps) { slot = 8u + m.gc; m.gc = next4(m.gc, bit); }
  else if (type == kTInts) { slot = 12u + m.ic; m.ic = next3(m.ic, bit); }
  else {
    slot = flag ? 8u + m.gc : m.mc;
    m.mc = next8(m.mc, bit);                     // pm.updateState also under a gap (WaveletTree.hpp:724,751)
    if (flag) m.gc = next4(m.gc, bit);
  }
  return slot;
}

template <u32 TYPE>
BWTC_GM_HD u32 stepMachinesT(u32 v, Machines& m) { return stepMachines(TYPE, v, m); }   // TYPE is a constant: one branch survives

// The walks are compiled once per task type (the inner loops then carry no type tests); a wave's
// lanes nearly always share their type (a task's chunks are consecutive).
#define BWTC_GM_BY_TYPE(type, CALL)                                                        \
  do {                                                                                     \
    if ((type) == kTRoot) { constexpr u32 TYPE = kTRoot; CALL; }                           \
    else if ((type) == kTGaps) { constexpr u32 TYPE = kTGaps; CALL; }                      \
    else if ((type) == kTInts) { constexpr u32 TYPE = kTInts; CALL; }                      \
    else { constexpr u32 TYPE = kTInner; CALL; }                                           \
  } while (0)

// ---- pass 1: a chunk's state maps ------------------------------------------------------------------
template <u32 type>
BWTC_GM_HD u64 laneMapT(const u32* packed, u32 begin, u32 end) {
  const bool adv = type == kTRoot || type == kTInner;
  bool changed = false;
  u32 first = 0, run = 0, s = 0;                  // main: leading run of equal bits, then one state
  u32 gk = 0, gb1 = 0, gb2 = 0;                   // gaps: number of updates (capped at 2), last and last-but-one bit
  u32 i0 = 0, i1 = 1, i2 = 2;                     // integers: the images themselves
  forElements(packed, begin, end, [&](u32, u32 v) {
    const u32 bit = v & 1u, flag = v >> 1;
    if (adv) {
      if (changed) s = next8(s, bit);
      else if (run == 0) { first = bit; run = 1; }
      else if (bit == first) run = run < 4 ? run + 1 : 4;
      else { changed = true; s = bit ? 4u : 3u; }  // a change of bit value fixes the state (FSM.hpp:42-67)
    }
    if (type == kTGaps || (type == kTInner && flag)) { gb2 = gb1; gb1 = bit; gk = gk < 2 ? gk + 1 : 2; }
    if (type == kTInts) { i0 = next3(i0, bit); i1 = next3(i1, bit); i2 = next3(i2, bit); }
  });
  u64 m = 0;
  for (u32 st = 0; st < 8; ++st) {
    u32 x = st;
    if (changed) x = s; else for (u32 r = 0; r < run; ++r) x = next8(x, first);
    m |= (u64)x << (3 * st);
  }
  for (u32 st = 0; st < 4; ++st) {
    const u32 x = gk == 0 ? st : gk == 1 ? ((gb1 << 1) | (st >> 1)) : ((gb1 << 1) | gb2);
    m |= (u64)x << (24 + 2 * st);
  }
  m |= (u64)i0 << 32; m |= (u64)i1 << 34; m |= (u64)i2 << 36;
  return m;
}
// The same map from the chunk's TAIL alone, where that is enough (nearly always):
//   main      four advancing elements fix the state whatever it was -- a change of bit value among
//             them does, and so do four equal bits (FSM.hpp:42-67 saturates after four steps);
//   gaps      the state is (last bit, bit before it) of the gap-coded elements;
//   integers  two equal bits in a row fix the saturating counter; what follows is replayed.
// Falls back to the full walk when the tail does not decide (short chunks, an INNER chunk with
// fewer than two gap-coded elements in its last 32, integer bits that alternate over 64 elements).
BWTC_GM_HD u64 laneMap(const u32* packed, u32 begin, u32 end, u32 type) {
  const u32 n = end - begin;
  bool quick = n >= 4;
  u64 m = 0;
  if (quick) {
    const u32 want = type == kTInts ? 64u : type == kTInner ? 32u : 4u;
    const u32 tail = n < want ? n : want;          // elements end - tail .. end - 1
    u32 bits = 0, flags = 0;                       // bit e = element end - 1 - e (the LAST element is bit 0), e < 32
    u64 bits64 = 0;                                // the same for all 64 (integers)
    {
      // the last 64 elements lie in at most five packed words: loaded together, then unpacked
      const u32 w_hi = (end - 1u) >> 4, w_lo = (end - tail) >> 4;
      u32 wd[5];
      for (u32 k = 0; k < 5; ++k) wd[k] = w_hi >= k && w_hi - k >= w_lo ? packed[w_hi - k] : 0u;
      for (u32 e = 0; e < tail; ++e) {
        const u32 i = end - 1u - e, k = w_hi - (i >> 4);
        const u32 word = k == 0 ? wd[0] : k == 1 ? wd[1] : k == 2 ? wd[2] : k == 3 ? wd[3] : wd[4];
        const u32 v = (word >> ((i & 15u) * 2u)) & 3u;
        if (e < 32u) { bits |= (v & 1u) << e; flags |= (v >> 1) << e; }
        bits64 |= (u64)(v & 1u) << e;
      }
    }
    if (type == kTRoot || type == kTInner) {
      const u32 b4 = bits & 1u;
      const u32 x = (bits ^ (0u - b4)) & 15u;      // set where one of the last four bits differs from the last one
      const u32 r = x & 2u ? 1u : x & 4u ? 2u : x & 8u ? 3u : 4u;
      const u32 st = r == 4u ? (b4 ? 7u : 0u) : (b4 ? 3u + r : 4u - r);
      m |= (u64)(st * 0x249249u);                  // the same three bits eight times
    } else {
      m |= kMapIdentity & 0xFFFFFFull;
    }
    if (type == kTGaps) {
      m |= (u64)((((bits & 1u) << 1) | ((bits >> 1) & 1u)) * 0x55u) << 24;
    } else if (type == kTInner) {
      const u32 f = flags & ((tail >= 32u) ? 0xFFFFFFFFu : ((1u << tail) - 1u));
      if (popc(f) >= 2u) {
        const u32 e1 = (u32)__builtin_ctz(f), e2 = (u32)__builtin_ctz(f & (f - 1u));
        m |= (u64)(((((bits >> e1) & 1u) << 1) | ((bits >> e2) & 1u)) * 0x55u) << 24;
      } else quick = false;
    } else {
      m |= kMapIdentity & (0xFFull << 24);
    }
    if (type == kTInts) {
      const u64 eq = ~(bits64 ^ (bits64 >> 1)) & (tail >= 64u ? ~0ull >> 1 : ((1ull << (tail - 1u)) - 1ull));   // bit e: elements e and e+1 (from the end) are equal
      if (eq) {
        const u32 e = (u32)__builtin_ctzll(eq);    // the latest such pair: element end - 1 - e is its second bit
        u32 st = (bits64 >> e) & 1u ? 2u : 0u;
        for (u32 k = e; k-- > 0;) st = next3(st, (u32)(bits64 >> k) & 1u);
        m |= (u64)(st * 0x15u) << 32;
      } else quick = false;
    } else {
      m |= kMapIdentity & (0x3Full << 32);
    }
  }
  if (quick) return m;
  BWTC_GM_BY_TYPE(type, m = laneMapT<TYPE>(packed, begin, end));
  return m;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define BWTC_GM_LOCAL_ADD(ptr, val) __hip_atomic_fetch_add((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)   /* ds_add_u32, no round trip */
#else
#define BWTC_GM_LOCAL_ADD(ptr, val) (*(ptr) += (val))
#endif
// ---- pass 2: updates per slot ------------------------------------------------------------------------
// tab: kSlotStride rows, this lane's column (row k at tab[k * stride]); zeroed here
template <u32 TYPE>
BWTC_GM_HD void laneCountT(const u32* packed, u32 begin, u32 end, u32 state, u32* tab, u32 stride) {
  Machines m = {state & 7u, (state >> 3) & 3u, (state >> 5) & 3u};
  forElements(packed, begin, end, [&](u32, u32 v) {
    const u32 slot = stepMachinesT<TYPE>(v, m);
    BWTC_GM_LOCAL_ADD(&tab[slot * stride], 1u);
  });
}
BWTC_GM_HD void laneCount(const u32* packed, u32 begin, u32 end, u32 type, u32 state, u32* tab, u32 stride) {
  for (u32 k = 0; k < kSlots; ++k) tab[k * stride] = 0;
  BWTC_GM_BY_TYPE(type, laneCountT<TYPE>(packed, begin, end, state, tab, stride));
}

// ---- pass 3: the bits, gathered by slot --------------------------------------------------------------
// pos/acc: two tables as above.  pos[k] starts at base[k] (the chunk's first position of slot k in
// slot space).  The words of `sbits` are shared with neighbouring chunks: OR-ed in.
#if defined(__HIP_DEVICE_COMPILE__)
#define BWTC_GM_OR(ptr, val) atomicOr((ptr), (val))
#else
#define BWTC_GM_OR(ptr, val) (*(ptr) |= (val))
#endif
// `shared`: bit k set = the word slot k is filling also holds bits of the chunk before (it did not
// start on a word border): that one is OR-ed in; the words after it are this chunk's alone and are
// stored (an atomic costs a 64-byte round trip to memory, and there would be one per 32 elements).
template <u32 TYPE>
BWTC_GM_HD void lanePartitionT(const u32* packed, u32 begin, u32 end, u32 state, u32* pos, u32* acc, u32 stride, u32* sbits,
                               u32& shared) {
  Machines m = {state & 7u, (state >> 3) & 3u, (state >> 5) & 3u};
  forElements(packed, begin, end, [&](u32, u32 v) {
    const u32 slot = stepMachinesT<TYPE>(v, m);
    const u32 p = pos[slot * stride];
    u32 a = acc[slot * stride] | ((v & 1u) << (p & 31u));
    if ((p & 31u) == 31u) {
      if ((shared >> slot) & 1u) { if (a) BWTC_GM_OR(&sbits[p >> 5], a); shared &= ~(1u << slot); }
      else sbits[p >> 5] = a;
      a = 0;
    }
    acc[slot * stride] = a;
    pos[slot * stride] = p + 1;
  });
}
BWTC_GM_HD void lanePartition(const u32* packed, u32 begin, u32 end, u32 type, u32 state, u32* pos, u32* acc,
                              u32 stride, u32* sbits) {
  u32 shared = 0;
  for (u32 k = 0; k < kSlots; ++k) shared |= ((pos[k * stride] & 31u) != 0u ? 1u : 0u) << k;
  BWTC_GM_BY_TYPE(type, lanePartitionT<TYPE>(packed, begin, end, state, pos, acc, stride, sbits, shared));
  for (u32 k = 0; k < kSlots; ++k) {
    const u32 a = acc[k * stride];
    if (a) BWTC_GM_OR(&sbits[(pos[k * stride] - 1u) >> 5], a);       // a != 0: at least one bit since the last flush
  }
}

// ---- slot space ----------------------------------------------------------------------------------------
// sb[0..ns]: start of every stream (slot-major, then task), sb[ns] = total; the last stream that
// starts at or before p
BWTC_GM_HD u32 streamAt(const u32* sb, u32 ns, u32 p) {
  u32 lo = 0, hi = ns;                             // sb[lo] <= p < sb[hi] (sb[0] = 0; p < total = sb[ns])
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (sb[mid] <= p) lo = mid; else hi = mid; }
  return lo;
}

// One step of a bracket: x0 = image of the lowest candidate, mask bit j = candidates j and j+1
// still differ (by exactly one).  Fewer than 2^d candidates' values: at most one pair merges.
BWTC_GM_HD void bracketStep(u32& x0, u32& mask, u32 bit, u32 floor, u32 d) {
  if (mask) {
    const u32 mod = (1u << d) - 1u;
    const u32 t = (bit ? (4096u - floor) - x0 : floor - 1u - x0) & mod;   // the merging pair: values x0 + t, x0 + t + 1
    if (t < popc(mask)) {
      u32 m = mask;
      for (u32 i = 0; i < t; ++i) m &= m - 1u;
      mask &= ~(m & (0u - m));
    }
  }
  x0 = moved(x0, bit, floor, d);
}

struct SlotMap { u32 lo_x0; u32 mask; };           // lo_x0 = bracket base L | x0 at the chunk's end << 16

// pass 4: slot-chunk j = positions [j * kSlotChunk, ...): its map from start value to end value
// Also leaves, at every kSample-th position, the bracket as it stands there: snaps[p / kSample] =
// {image of the lowest candidate, mask}.  With the chunk's true start value t (pass 5) the true
// value at that position is  x0 + popcount(mask & ((1 << (t - L)) - 1))  -- no second walk over slot
// space is needed for the values the last pass starts from.
BWTC_GM_HD SlotMap laneBracket(const u32* sbits, const u32* sb, u32 ns, u32 nt, u32 total, u32 j, SlotMap* snaps, u32* err) {
  const u32 p0 = j * kSlotChunk, p1 = p0 + kSlotChunk < total ? p0 + kSlotChunk : total;
  u32 sg = streamAt(sb, ns, p0);
  u32 k = sg / nt, floor = slotFloor(k), d = slotDelay(k);
  const u32 s0 = sb[sg];
  u32 x0, mask, L;
  if (p0 - s0 <= kWarm) {                          // the chain starts close by: exact
    x0 = slotInit(k);
    forBits(sbits, s0, p0, [&](u32, u32 b) { x0 = moved(x0, b, floor, d); });
    L = x0; mask = 0;
  } else {
    u32 lo = floor, hi = 4096u - floor;
    forBits(sbits, p0 - kWarm, p0, [&](u32, u32 b) { lo = moved(lo, b, floor, d); hi = moved(hi, b, floor, d); });
    const u32 w = hi - lo;
    if (w >= (1u << d)) { *err |= kErrBracket; return SlotMap{0, 0}; }
    L = lo; x0 = lo; mask = (1u << w) - 1u;
  }
  u32 next = sb[sg + 1];
  Piece cur = loadPiece(sbits, p0 >> 7);           // p0 is a multiple of 128: four words per load, the next one under way
  for (u32 gp = p0; gp < p1; gp += 64u) {          // two words = two snapshots = one 16-byte store
    SlotMap sn[2] = {SlotMap{0, 0}, SlotMap{0, 0}};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (u32 wk = 0; wk < 2; ++wk) {
      const u32 wp = gp + wk * 32u;
      if (wp >= p1) break;
      const u32 hi = wp + 32u < p1 ? wp + 32u : p1;
      u32 word = pieceWord(cur, (wp >> 5) & 3u);
      if (((wp >> 5) & 3u) == 3u) cur = loadPiece(sbits, (wp >> 7) + 1u);
      if (next >= hi || sg + 1u >= ns) {           // no chain starts inside this word
        sn[wk] = SlotMap{x0, mask};
        // equal bits that move no candidate any more: nothing to do for the whole word
        if (!(hi - wp == 32u && ((word == 0u && ((x0 + popc(mask) - floor) >> d) == 0u) ||
                                 (word == 0xFFFFFFFFu && (((4096u - floor) - x0) >> d) == 0u))))
          for (u32 p = wp; p < hi; ++p, word >>= 1) bracketStep(x0, mask, word & 1u, floor, d);
      } else {
        for (u32 p = wp; p < hi; ++p, word >>= 1) {
          while (p >= next && sg + 1u < ns) {      // another chain starts here: fresh predictor, known exactly
            ++sg; next = sb[sg + 1];
            k = sg / nt; floor = slotFloor(k); d = slotDelay(k);
            x0 = slotInit(k); mask = 0;
          }
          if (p == wp) sn[wk] = SlotMap{x0, mask};
          bracketStep(x0, mask, word & 1u, floor, d);
        }
      }
    }
    if (gp + 64u <= p1) {
#if defined(__HIP_DEVICE_COMPILE__)
      *reinterpret_cast<uint4*>(snaps + gp / kSample) = make_uint4(sn[0].lo_x0, sn[0].mask, sn[1].lo_x0, sn[1].mask);
#else
      snaps[gp / kSample] = sn[0]; snaps[gp / kSample + 1] = sn[1];
#endif
    } else {
      snaps[gp / kSample] = sn[0];
      if (gp + 32u < p1) snaps[gp / kSample + 1] = sn[1];
    }
  }
  return SlotMap{L | (x0 << 16), mask};
}

// pass 5: the true value at every slot-chunk border.  Slot space is ONE chain of maps for this
// purpose: a chunk in which a predictor chain starts ends exactly (mask 0), so its map is a
// constant and nothing runs across.  Three levels: (a) per group of `gsize` slot-chunks the
// images of the group's first bracket (one lane per candidate, laneChainGroup), (b) a serial
// walk over the groups (a table look-up each, laneChainTop), (c) per group the chunks' values
// from the group's true start (laneChainFill).
// slot-chunks per group: at most kChainGroups groups, so that level (b) stays a walk over one LDS table
constexpr u32 kChainGroups = 1024;
BWTC_GM_HD u32 chainGroupSize(u32 nsc) { const u32 g = (nsc + kChainGroups - 1u) / kChainGroups; return g < 16u ? 16u : g; }
BWTC_GM_HD u32 slotMapApply(const SlotMap m, u32 t, u32* err) {
  const u32 off = t - (m.lo_x0 & 0xFFFFu);
  if (off > 31u) { if (m.mask && err) *err |= kErrChain; return m.lo_x0 >> 16; }   // a constant map takes any value
  return (m.lo_x0 >> 16) + popc(m.mask & ((1u << off) - 1u));
}
// candidate `cand` (0..31) of group g's first bracket -> its value after the group; gL[g] = the bracket's base
BWTC_GM_HD u32 laneChainGroup(const SlotMap* smap, u32 nsc, u32 gsize, u32 g, u32 cand) {
  const u32 j0 = g * gsize, j1 = j0 + gsize < nsc ? j0 + gsize : nsc;
  u32 t = (smap[j0].lo_x0 & 0xFFFFu) + cand;
  for (u32 j = j0; j < j1; ++j) t = slotMapApply(smap[j], t, nullptr);
  return t;
}
// gmap[g * 32 + cand], gL[g] -> tg[g] = the true value at group g's first border
BWTC_GM_HD void laneChainTop(const unsigned short* gmap, const unsigned short* gL, u32 ng, unsigned short* tg, u32* err) {
  u32 t = gL[0];                                   // slot space starts with a chain's start: exact
  for (u32 g = 0; g < ng; ++g) {
    tg[g] = (unsigned short)t;
    u32 off = t - gL[g];
    if (off > 31u) off = 0;                        // only a constant group map may be entered off its bracket (checked in Fill)
    t = gmap[g * 32u + off];
  }
}
BWTC_GM_HD void laneChainFill(const SlotMap* smap, u32 nsc, u32 gsize, u32 g, const unsigned short* tg,
                              unsigned short* sstart, u32* err) {
  const u32 j0 = g * gsize, j1 = j0 + gsize < nsc ? j0 + gsize : nsc;
  u32 t = tg[g];
  for (u32 j = j0; j < j1; ++j) { sstart[j] = (unsigned short)t; t = slotMapApply(smap[j], t, err); }
}

// pass 7: the elements of a chunk in coding order with exact predictors.  q: table row k = slot k.
// out[i] = bit << 15 | probability of the coded bit
template <u32 TYPE>
BWTC_GM_HD u32 emitStep(u32 v, Machines& m, u32* q, u32 stride) {
  const u32 bit = v & 1u;
  const u32 slot = stepMachinesT<TYPE>(v, m);
  const u32 pr = q[slot * stride];
  q[slot * stride] = moved(pr, bit, TYPE == kTInts ? 100u : 2u, (TYPE == kTGaps || TYPE == kTInts) ? 5u : slotDelay(slot));
  return (bit << 15) | (bit ? pr : 4096u - pr);
}
// `stage` (stride `stride` between its 32 words): room for the 64 elements = 128 bytes = one line of
// output that a piece of 64 elements makes.  A piece that lies wholly inside the chunk is collected there
// and leaves with eight 16-byte stores back to back: written 32 bytes at a time, sixteen elements' worth
// of work apart, the lines of 130 000 lanes were open at once -- half of the L2 -- and part of them left
// it half-written (1.6 x the bytes written, and fetched again to be completed).
template <u32 TYPE>
BWTC_GM_HD void laneEmitT(const u32* packed, u32 begin, u32 end, u32 state, u32* q, u32 stride, unsigned short* out, u32* stage) {
  Machines m = {state & 7u, (state >> 3) & 3u, (state >> 5) & 3u};
  if (begin >= end) return;
  const u32 q0 = begin >> 6, q1 = (end - 1u) >> 6;
  Piece cur = loadPiece(packed, q0);
  for (u32 qi = q0; qi <= q1; ++qi) {
    const Piece nxt = loadPiece(packed, qi < q1 ? qi + 1u : qi);
    const bool whole = (qi << 6) >= begin && (qi << 6) + 64u <= end;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (u32 k = 0; k < 4; ++k) {
      const u32 wb = (qi << 6) + k * 16u;
      u32 word = pieceWord(cur, k);
      if (wb >= begin && wb + 16u <= end) {                      // sixteen elements: 32 bytes of output
        u32 o[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (u32 e = 0; e < 8; ++e) {
          const u32 a = emitStep<TYPE>(word & 3u, m, q, stride);
          const u32 b = emitStep<TYPE>((word >> 2) & 3u, m, q, stride);
          word >>= 4;
          o[e] = a | (b << 16);
        }
        if (whole) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
          for (u32 e = 0; e < 8; ++e) stage[(k * 8u + e) * stride] = o[e];
          if (k == 3u) {
            u32* dst = reinterpret_cast<u32*>(out + (qi << 6));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (u32 v = 0; v < 8; ++v)
              reinterpret_cast<uint4*>(dst)[v] = make_uint4(stage[(4u * v) * stride], stage[(4u * v + 1u) * stride],
                                                            stage[(4u * v + 2u) * stride], stage[(4u * v + 3u) * stride]);
#else
            for (u32 v = 0; v < 32; ++v) dst[v] = stage[v * stride];
#endif
          }
          continue;
        }
        u32* dst = reinterpret_cast<u32*>(out + wb);
#if defined(__HIP_DEVICE_COMPILE__)
        reinterpret_cast<uint4*>(dst)[0] = make_uint4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<uint4*>(dst)[1] = make_uint4(o[4], o[5], o[6], o[7]);
#else
        for (u32 e = 0; e < 8; ++e) dst[e] = o[e];
#endif
      } else {
        const u32 lo = wb > begin ? wb : begin, hi = wb + 16u < end ? wb + 16u : end;
        if (lo >= hi) continue;
        word >>= (lo & 15u) * 2u;
        for (u32 i = lo; i < hi; ++i, word >>= 2) out[i] = (unsigned short)emitStep<TYPE>(word & 3u, m, q, stride);
      }
    }
    cur = nxt;
  }
}
BWTC_GM_HD void laneEmit(const u32* packed, u32 begin, u32 end, u32 type, u32 state, u32 task, u32 chunk, u32 nc, u32 nt,
                         const u32* base, const u32* sb, const u32* sbits, const SlotMap* snaps, const SlotMap* smap,
                         const unsigned short* sstart, u32* q, u32 stride, unsigned short* out, u32* err, u32* stage) {
  const u32 k_lo = type == kTGaps ? 8u : type == kTInts ? 12u : 0u;
  const u32 k_hi = type == kTRoot ? 8u : type == kTInts ? 15u : 12u;
  for (u32 k = k_lo; k < k_hi; ++k) {
    const u32 P = base[k * nc + chunk], s0 = sb[k * nt + task];
    const u32 floor = slotFloor(k), d = slotDelay(k);
    const u32 a0 = P & ~(kSample - 1u);
    u32 a, x;
    if (a0 > s0) {                                 // the bracket as it stood at a0, resolved with its slot-chunk's true start value
      const u32 jc = a0 / kSlotChunk;
      const SlotMap sn = snaps[a0 / kSample];
      const u32 off = (u32)sstart[jc] - (smap[jc].lo_x0 & 0xFFFFu);
      if (off > 31u && sn.mask) *err |= kErrChain;
      a = a0; x = sn.lo_x0 + popc(sn.mask & ((1u << (off > 31u ? 31u : off)) - 1u));
    } else { a = s0; x = slotInit(k); }
    forBits(sbits, a, P, [&](u32, u32 b) { x = moved(x, b, floor, d); });
    q[k * stride] = x;
  }
  BWTC_GM_BY_TYPE(type, laneEmitT<TYPE>(packed, begin, end, state, q, stride, out, stage));
}

// ---- host side ---------------------------------------------------------------------------------------------
// Tasks (in coding order) and chunks of a block from its plan and the groups' coded positions.
void buildTasks(const ::bwtc::wavelet::StreamPlan& plan, const u32* coded_pos, std::vector<Task>* tasks,
                std::vector<Chunk>* chunks);

// The passes above run lane by lane on the host (tests; the product runs them on the GPU).
// state: the main machine's carried state, updated.  false: an error flag was raised.
bool modelsOnHostLanes(const u32* packed, u32 total, const std::vector<Task>& tasks, const std::vector<Chunk>& chunks,
                       u32* state, unsigned short* out);

}  // namespace gm
}  // namespace wavelet
}  // namespace bwtc
