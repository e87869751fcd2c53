// LSD radix sort of (key, u32 value) pairs for gfx950, 8-bit digits, stable.
//
// One pass = three steps, none of which spins on another workgroup:
//   k_radix_hist    per-tile 256-bin digit histogram in LDS (ds_add), written bin-major
//                   to table[bin * ntiles + tile]
//   exclusive_scan  over the flattened table -> global base of every (bin, tile)
//   k_radix_scatter re-reads the tile, ranks every key among equal digits of its wave with
//                   8 wave ballots (no atomics, data-independent cost, stable), turns the
//                   per-wave counts into tile offsets, reorders the tile through LDS and
//                   writes each digit's run to consecutive addresses.
//
// HBM traffic per pass and element: hist reads the key once; scatter reads key+value and
// writes key+value.  The table adds 2 * 256 * 4 B per tile (< 3 % at these tile sizes).
#pragma once
#include "common.hpp"
#include "scan.hpp"
#include <algorithm>

namespace bwtc_hip {

constexpr int kRadixBits = 8;
constexpr int kRadixBins = 1 << kRadixBits;
#ifndef BWTC_RADIX_TPB
#define BWTC_RADIX_TPB 512
#endif
#ifndef BWTC_RADIX_E64
#define BWTC_RADIX_E64 8
#endif
#ifndef BWTC_RADIX_E32
#define BWTC_RADIX_E32 16
#endif
#ifndef BWTC_HIST_COPIES
#define BWTC_HIST_COPIES 16
#endif
constexpr int kRadixTPB = BWTC_RADIX_TPB;         // 8 waves
constexpr int kRadixWaves = kRadixTPB / kWave;

// Checked builds (make EXTRA=-DBWTC_RADIX_CHECKED): the scatter trusts a table of bases that another kernel built from
// another copy of the digit (a digit plane, a key maker's first plane); a disagreement between the two makes it store
// past its bases -- in round 4 past the array, a memory fault.  Checked, an item whose destination lies outside the
// output is not stored and raises this word instead; the transform's entry points then return -3.
#ifdef BWTC_RADIX_CHECKED
static __device__ u32 g_radix_fault;
static inline int radix_fault_take() {
  u32 v = 0, z = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_radix_fault), 4) != hipSuccess) return 1;
  if (v) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_radix_fault), &z, 4);
  return v ? 1 : 0;
}
#else
static inline int radix_fault_take() { return 0; }
#endif

template <typename K> struct RadixCfg;
template <> struct RadixCfg<u32> { static constexpr int E = BWTC_RADIX_E32; };   // 8192 pairs, 64 KiB LDS
template <> struct RadixCfg<u64> { static constexpr int E = BWTC_RADIX_E64; };    // 4096 pairs, 48 KiB LDS

template <typename K> constexpr int radix_tile() { return kRadixTPB * RadixCfg<K>::E; }

template <typename K>
__device__ __forceinline__ u32 radix_digit(K k, int shift) {
  return (u32)(k >> shift) & (kRadixBins - 1);
}

// Long items (the suffix sorter's long-key initial sort): every item carries a second 32-bit key word w
// beside (key, value).  The early passes take their digits from w, the later ones from the key; dmask
// cuts a field's top digit to the field's bits (whatever sits above them is payload, not order).
// When w's bits are no whole number of digits, one "bridge" digit takes w's top `bridge` bits and the key's lowest
// ones above them, so that the passes number ceil((w bits + key bits) / 8) and not one more.
struct LongArgs {
  const u32* win; u32* wout;
  u32 dmask;                     // this pass's digit mask
  int bridge;                    // bridge digit: how many of its bits are w's (the pass's shift is into w)
  int nshift, nfrom_w; u32 ndmask;   // the NEXT pass's digit (for the plane this pass leaves): 0 key, 1 w, 2 bridge
  int nbridge;
};
template <typename K, int LONG>
__device__ __forceinline__ u32 long_digit(K k, u32 w, int shift, u32 dmask, int bridge = 0) {
  if (LONG == 3) return ((w >> shift) | ((u32)k << bridge)) & dmask;
  if (LONG == 2) return (w >> shift) & dmask;
  if (LONG == 1) return (u32)(k >> shift) & dmask;
  return radix_digit(k, shift);
}

// Segmented sorts (the wavelet coder's step sort): the items are a row of segments, each a whole number
// of tiles, and every segment is sorted on its own -- its items never leave it.  A pass is the same three
// steps; only the table is laid out segment by segment (bin-major inside a segment), so that the one
// exclusive scan over it yields every segment's bases.  A segment's padding carries the all-ones key: the
// last digit in every pass, so it stays at the segment's end.
//   tile_first[nseg + 1]  first tile of every segment (tile_first[nseg] = ntiles; empty segments allowed)
struct SegArgs {
  const u32* tile_first = nullptr;
  u32 nseg = 0;
};
struct SegTile { u32 seg, first, count; };       // the tile's segment, the segment's first tile and number of tiles
__device__ __forceinline__ SegTile seg_of_tile(const SegArgs& sa, u32 tile) {
  u32 lo = 0, hi = sa.nseg;                       // the last segment that starts at or before the tile
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (sa.tile_first[mid] <= tile) lo = mid; else hi = mid; }
  const u32 f = sa.tile_first[lo];
  return SegTile{lo, f, sa.tile_first[lo + 1] - f};
}
__device__ __forceinline__ u64 seg_table_at(const SegTile& st, u32 bin, u32 tile) {
  return (u64)kRadixBins * st.first + (u64)bin * st.count + (tile - st.first);
}

// skip: items whose key is all ones do not exist (first pass of a sort whose input was written
// with holes, see k_gather_dense): they are neither counted nor moved, so the pass compacts.
template <typename K, bool SKIP, int E_ = RadixCfg<K>::E>
__global__ __launch_bounds__(kRadixTPB) void k_radix_hist(const K* __restrict__ keys,
                                                          u32* __restrict__ table, u64 n,
                                                          int shift, u32 ntiles) {
  constexpr int E = E_;
  // 16 interleaved copies per bin (copy = lane & 15): equal digits of neighbouring lanes land
  // on different LDS addresses and banks, so skewed digit distributions (text, DNA, all-equal
  // blocks) do not serialise the ds_add.
  __shared__ u32 hist[kRadixBins * 16];
  // tiles dealt out as in k_radix_scatter (a contiguous range per XCD; the grid is a whole number of rounds of 8)
  const u32 tile = (blockIdx.x & 7u) * ((ntiles + 7u) / 8u) + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  for (u32 i = threadIdx.x; i < kRadixBins * 16; i += kRadixTPB) hist[i] = 0;
  __syncthreads();
  const u64 tile_base = (u64)tile * (kRadixTPB * E);
  const u32 wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const u64 wbase = tile_base + (u64)wave * (kWave * E) + lane;
  const u32 copy = lane & 15u;
  K k[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    u64 i = wbase + (u64)e * kWave;
    k[e] = (i < n) ? keys[i] : (K)0;
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    u64 i = wbase + (u64)e * kWave;
    if (i < n && !(SKIP && k[e] == ~(K)0)) atomicAdd(&hist[radix_digit(k[e], shift) * 16u + copy], 1u);
  }
  __syncthreads();
  if (threadIdx.x < kRadixBins) {
    u32 c = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) c += hist[threadIdx.x * 16u + r];
    table[(u64)threadIdx.x * ntiles + tile] = c;
  }
}

// PLANE: the pass also writes plane[dst] = the item's NEXT digit, one byte per item in output
// order, so that the next pass's histogram kernel reads 1 byte per item instead of the key.
// KEYS: keys only (no values staged: half the LDS of a 32-bit pair tile, twice the workgroups per CU).
// V: the values' type in memory, u32 or u16 (the suffix sorter's initial sort keeps the upper bits
// of a suffix number in spare bits of its key, so that a pass moves 10 bytes per item instead of 12).
// LONG (see LongArgs): 0 plain items; 1 long items, digit from the key; 2 long items, digit from w; 3 the bridge digit.
// SEG (see SegArgs): a segmented pass (32-bit keys only, no long items).
template <typename K, bool SKIP, bool PLANE, bool KEYS, typename V = u32, int LONG = 0, int E_ = RadixCfg<K>::E, bool SEG = false>
__global__ __launch_bounds__(kRadixTPB) void k_radix_scatter(
    const K* __restrict__ kin, const V* __restrict__ vin, K* __restrict__ kout,
    V* __restrict__ vout, const u32* __restrict__ table, u64 n, int shift, u32 ntiles, int values_mode,
    u8* __restrict__ plane, LongArgs la, SegArgs sa) {
  static_assert(!SEG || (sizeof(K) == 4 && LONG == 0 && !SKIP), "segmented passes sort plain 32-bit keys");
  constexpr int E = E_;
  constexpr int TILE = kRadixTPB * E;
  // LDS: the per-wave digit counters are dead once every thread has turned them into its
  // tile slots, so the reorder staging (keys, then values) reuses their bytes; that keeps
  // the (u64,u32) kernel at 50 KiB = three workgroups per CU.
  constexpr int kStageBytes = TILE * (int)(sizeof(K) + (KEYS ? 0 : sizeof(V)) + (LONG ? sizeof(u32) : 0));
  constexpr int kCntBytes = kRadixWaves * kRadixBins * (int)sizeof(u32);
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[kStageBytes > kCntBytes ? kStageBytes : kCntBytes];
  K* s_key = reinterpret_cast<K*>(s_raw);
  u32* s_w = reinterpret_cast<u32*>(s_raw + TILE * sizeof(K));                       // LONG only
  V* s_val = reinterpret_cast<V*>(s_raw + TILE * (sizeof(K) + (LONG ? sizeof(u32) : 0)));
  u32 (*s_cnt)[kRadixBins] = reinterpret_cast<u32 (*)[kRadixBins]>(s_raw);
  __shared__ u32 s_base[kRadixBins];               // first tile slot of each digit
  __shared__ u32 s_gofs[kRadixBins];               // global base minus tile slot
  __shared__ u32 s_scr[kRadixTPB / kWave + 1];

  const u32 tid = threadIdx.x;
  const u32 wave = tid / kWave, lane = tid % kWave;
  // Workgroups are dealt round-robin over the 8 XCDs (speed only, never correctness): give
  // every XCD a contiguous range of tiles so that the partial cache lines at the seams of
  // neighbouring tiles' output runs meet in ONE L2 instead of being written back twice.
  const u32 per_xcd = (ntiles + 7u) / 8u;
  const u32 tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const u64 tile_base = (u64)tile * TILE;
  const u64 left = n - tile_base;
  const u32 tile_n = left < (u64)TILE ? (u32)left : (u32)TILE;
  SegTile sg = {0, 0, 0};
  if (SEG) sg = seg_of_tile(sa, tile);

  for (u32 i = tid; i < kRadixWaves * kRadixBins; i += kRadixTPB) (&s_cnt[0][0])[i] = 0;

  // wave w owns tile slots [w*64*E, (w+1)*64*E); iteration e covers 64 consecutive slots
  const u32 wslot = wave * (kWave * E) + lane;
  K k[E];
  u32 v[E];
  u32 r[E];
  u32 okm = 0;                                     // SKIP: bit e = this thread's item e exists
  // does item e of this thread exist?  (without holes that is a compare, no register)
#define BWTC_EXISTS(e) (SKIP ? (bool)((okm >> (e)) & 1u) : (wslot + (e) * kWave) < tile_n)
  // All loads of the tile back to back.  (With the values' mode switched on per item inside one
  // loop the compiler put a full wait behind every item's loads: sixteen memory latencies in a
  // row at the head of the kernel that runs twenty times per block.)  A hole's value is loaded
  // like any other and never used.
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const u32 slot = wslot + e * kWave;
    k[e] = slot < tile_n ? kin[tile_base + slot] : (K)0;
  }
  u32 w[LONG ? E : 1] = {};
  if (LONG) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const u32 slot = wslot + e * kWave;
      w[LONG ? e : 0] = slot < tile_n ? la.win[tile_base + slot] : 0u;
    }
  }
  // values_mode 1 / 3: first pass of a sort whose values are the items' own positions (3: counted
  // down from n - 1, the suffix sorter's descending slots), nothing to read; 2: keys only.
  if (KEYS) {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = 0u;
  } else if (values_mode == 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const u32 slot = wslot + e * kWave;
      v[e] = slot < tile_n ? (u32)vin[tile_base + slot] : 0u;
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const u32 pos = (u32)tile_base + wslot + e * kWave;
      v[e] = values_mode == 1 ? pos : (u32)n - 1u - pos;
    }
  }
  if (SKIP) {
#pragma unroll
    for (int e = 0; e < E; ++e) okm |= ((wslot + e * kWave < tile_n && k[e] != ~(K)0) ? 1u : 0u) << e;
  }
  __syncthreads();

  volatile lds_u32* my_cnt = lds_volatile(&s_cnt[wave][0]);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const bool ok = BWTC_EXISTS(e);
    const u32 d = long_digit<K, LONG>(k[e], w[LONG ? e : 0], shift, la.dmask, la.bridge);
    u32 below, peers;
    wave_digit_rank<kRadixBits>(d, ok, &below, &peers);
    u32 prev = 0;
    if (ok) prev = my_cnt[d];
    __builtin_amdgcn_wave_barrier();
    r[e] = prev + below;
    // the highest lane of each group publishes the new count
    if (ok && below + 1u == peers) my_cnt[d] = prev + peers;
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // per digit: exclusive prefix over waves, then exclusive prefix over digits
  u32 dig_total = 0;
  if (tid < kRadixBins) {
#pragma unroll
    for (int w = 0; w < kRadixWaves; ++w) {
      u32 c = s_cnt[w][tid];
      s_cnt[w][tid] = dig_total;
      dig_total += c;
    }
  }
  u32 tile_valid;                                   // items of the tile that exist
  u32 dig_base = block_scan_excl_add<kRadixTPB>(tid < kRadixBins ? dig_total : 0u, s_scr, &tile_valid);
  if (tid < kRadixBins) {
    s_base[tid] = dig_base;
    s_gofs[tid] = table[SEG ? seg_table_at(sg, tid, tile) : (u64)tid * ntiles + tile] - dig_base;
  }
  __syncthreads();

#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (BWTC_EXISTS(e)) {
      const u32 d = long_digit<K, LONG>(k[e], w[LONG ? e : 0], shift, la.dmask, la.bridge);
      r[e] += s_base[d] + s_cnt[wave][d];          // final tile slot
    }
  }
  __syncthreads();                                  // counters are dead from here on
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (BWTC_EXISTS(e)) {
      s_key[r[e]] = k[e];
      if (!KEYS) s_val[r[e]] = (V)v[e];
      if (LONG) s_w[r[e]] = w[LONG ? e : 0];
    }
  }
#undef BWTC_EXISTS
  __syncthreads();

  const u32 tile_out = SKIP ? tile_valid : tile_n;
  // all LDS reads of a thread's E items first, then the offsets, then the stores (a loop over i
  // ran two dependent LDS latencies per item, one item at a time)
  K kk[E];
  u32 vv[E], dst[E], ww[LONG ? E : 1] = {};
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const u32 i = tid + (u32)e * kRadixTPB;
    kk[e] = i < tile_out ? s_key[i] : (K)0;
    vv[e] = (i < tile_out && !KEYS) ? s_val[i] : 0u;
    if (LONG) ww[LONG ? e : 0] = i < tile_out ? s_w[i] : 0u;
  }
#pragma unroll
  for (int e = 0; e < E; ++e)
    dst[e] = s_gofs[long_digit<K, LONG>(kk[e], ww[LONG ? e : 0], shift, la.dmask, la.bridge)] + tid + (u32)e * kRadixTPB;
#pragma unroll
  for (int e = 0; e < E; ++e) {
#ifdef BWTC_RADIX_CHECKED
    if (tid + (u32)e * kRadixTPB < tile_out && (u64)dst[e] >= n) { atomicOr(&g_radix_fault, 1u); continue; }
#endif
    if (tid + (u32)e * kRadixTPB < tile_out) {
      // plain stores on purpose: the runs of neighbouring tiles are merged into full lines in
      // L2; non-temporal stores measured 30 % slower here, non-temporal loads no better
      kout[dst[e]] = kk[e];
      if (!KEYS) vout[dst[e]] = (V)vv[e];
      if (LONG) la.wout[dst[e]] = ww[LONG ? e : 0];
      if (PLANE) {
        if (LONG) plane[dst[e]] = (u8)((la.nfrom_w == 2 ? (ww[LONG ? e : 0] >> la.nshift) | ((u32)kk[e] << la.nbridge)
                                         : la.nfrom_w ? ww[LONG ? e : 0] >> la.nshift : (u32)(kk[e] >> la.nshift)) & la.ndmask);
        else plane[dst[e]] = (u8)radix_digit(kk[e], shift + kRadixBits);
      }
    }
  }
}

// Tile histogram from a digit plane (see PLANE above): same table as k_radix_hist<K, false>.
// A workgroup takes TWO tiles, half its threads each: the kernel is a chain of latencies (clear
// LDS, load, count, add up the copies, store) with little to do in between, so twice the bytes per
// workgroup is close to twice the speed.
template <typename K, int E_ = RadixCfg<K>::E, bool SEG = false>
__global__ __launch_bounds__(kRadixTPB) void k_radix_hist_plane(const u8* __restrict__ plane,
                                                                u32* __restrict__ table, u64 n,
                                                                u32 ntiles, SegArgs sa) {
  constexpr int E = E_;
  constexpr int HALF = kRadixTPB / 2;
  constexpr int B = 2 * E;                          // bytes of the tile per thread (8, 12, 16 or 32)
  static_assert(B == 8 || B == 12 || B % 16 == 0, "8-byte, 12-byte or whole 16-byte loads");
  constexpr u32 C = BWTC_HIST_COPIES;                // interleaved copies per bin (copy = thread & (C - 1))
  __shared__ __attribute__((aligned(16))) u32 hist[2][kRadixBins * C];
  {
    uint4* z = reinterpret_cast<uint4*>(&hist[0][0]);
    for (u32 i = threadIdx.x; i < 2u * kRadixBins * C / 4u; i += kRadixTPB) z[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  const u32 half = threadIdx.x / HALF, t = threadIdx.x % HALF;
  // tile pairs are dealt out as k_radix_scatter deals its tiles: a contiguous range per XCD (workgroup b runs on XCD
  // b % 8), so the 4-byte words that neighbouring tiles store into a bin's row meet in ONE L2 and leave as whole lines
  // (155 -> 115 us for 2^28 items; scripts/dev/hist_probe.cpp) -- and the scatter that reads them runs on the same XCD
  const u32 npairs = (ntiles + 1u) / 2u, per_xcd = (npairs + 7u) / 8u;
  const u32 pair = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const u32 tile = pair * 2u + half;
  if (pair < npairs && tile < ntiles) {
    const u64 base = (u64)tile * (kRadixTPB * E) + (u64)t * B;
    const u32 copy = t & (C - 1u);
    u32* h = hist[half];
    if (B == 12 && base + B <= n) {
      const u32* q = reinterpret_cast<const u32*>(plane + base);     // 12 t: a multiple of four
      const u32 w[3] = {q[0], q[1], q[2]};
#pragma unroll
      for (int b = 0; b < 12; ++b) atomicAdd(&h[((w[b >> 2] >> (8 * (b & 3))) & 255u) * C + copy], 1u);
    } else if (B == 8 && base + B <= n) {
      const uint2 q = *reinterpret_cast<const uint2*>(plane + base);
      const u32 w[2] = {q.x, q.y};
#pragma unroll
      for (int b = 0; b < 8; ++b) atomicAdd(&h[((w[b >> 2] >> (8 * (b & 3))) & 255u) * C + copy], 1u);
    } else if (B % 16 == 0 && base + B <= n) {
      constexpr int Q = B >= 16 ? B / 16 : 1;
      uint4 q[Q];
#pragma unroll
      for (int i = 0; i < Q; ++i) q[i] = *reinterpret_cast<const uint4*>(plane + base + 16 * i);
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const u32 w[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
        for (int b = 0; b < 16; ++b) atomicAdd(&h[((w[b >> 2] >> (8 * (b & 3))) & 255u) * C + copy], 1u);
      }
    } else {
      for (int b = 0; b < B; ++b) if (base + b < n) atomicAdd(&h[(u32)plane[base + b] * C + copy], 1u);
    }
  }
  __syncthreads();
  if (pair < npairs && tile < ntiles) {
    SegTile sg = {0, 0, 0};
    if (SEG) sg = seg_of_tile(sa, tile);
    for (u32 bin = t; bin < (u32)kRadixBins; bin += HALF) {
      const uint4* r4 = reinterpret_cast<const uint4*>(&hist[half][bin * C]);
      u32 c = 0;
#pragma unroll
      for (u32 r = 0; r < C / 4u; ++r) { const uint4 v = r4[r]; c += v.x + v.y + v.z + v.w; }
      table[SEG ? seg_table_at(sg, bin, tile) : (u64)bin * ntiles + tile] = c;
    }
  }
}

// Tile histogram of digit (w >> shift) & 255 of 32-bit words (the long-key sort's second word): k_radix_hist_plane's
// scheme -- two tiles per workgroup, sixteen interleaved copies per bin, tile pairs dealt out per XCD -- with 16-byte loads
// of four words.  Same table as k_radix_hist<u32> over tiles of kRadixTPB * E_ words.
template <int E_>
__global__ __launch_bounds__(kRadixTPB) void k_radix_hist_words(const u32* __restrict__ words, u32* __restrict__ table, u64 n,
                                                                int shift, u32 ntiles) {
  constexpr int HALF = kRadixTPB / 2;
  constexpr int TILE = kRadixTPB * E_;                 // words of a tile
  constexpr int Q = TILE / (HALF * 4);                 // 16-byte loads per thread
  static_assert(TILE % (HALF * 4) == 0, "whole 16-byte loads");
  constexpr u32 C = BWTC_HIST_COPIES;
  __shared__ __attribute__((aligned(16))) u32 hist[2][kRadixBins * C];
  {
    uint4* z = reinterpret_cast<uint4*>(&hist[0][0]);
    for (u32 i = threadIdx.x; i < 2u * kRadixBins * C / 4u; i += kRadixTPB) z[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  const u32 half = threadIdx.x / HALF, t = threadIdx.x % HALF;
  const u32 npairs = (ntiles + 1u) / 2u, per_xcd = (npairs + 7u) / 8u;
  const u32 pair = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const u32 tile = pair * 2u + half;
  if (pair < npairs && tile < ntiles) {
    const u64 base = (u64)tile * TILE;
    const u32 copy = t & (C - 1u);
    u32* h = hist[half];
    if (base + TILE <= n) {
      const uint4* src = reinterpret_cast<const uint4*>(words + base);     // tiles start at multiples of 16 bytes
      uint4 q[Q];
#pragma unroll
      for (int i = 0; i < Q; ++i) q[i] = src[t + (u32)i * HALF];
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const u32 w[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
        for (int b = 0; b < 4; ++b) atomicAdd(&h[((w[b] >> shift) & 255u) * C + copy], 1u);
      }
    } else {
      for (u64 i = base + t; i < n; i += HALF) atomicAdd(&h[((words[i] >> shift) & 255u) * C + copy], 1u);
    }
  }
  __syncthreads();
  if (pair < npairs && tile < ntiles) {
    for (u32 bin = t; bin < (u32)kRadixBins; bin += HALF) {
      const uint4* r4 = reinterpret_cast<const uint4*>(&hist[half][bin * C]);
      u32 c = 0;
#pragma unroll
      for (u32 r = 0; r < C / 4u; ++r) { const uint4 v = r4[r]; c += v.x + v.y + v.z + v.w; }
      table[(u64)bin * ntiles + tile] = c;
    }
  }
}

// k_radix_hist_plane's grid: two tiles per workgroup, rounded up to whole rounds of the 8 XCDs
static inline u32 hist_plane_grid(u32 ntiles) { return (((ntiles + 1u) / 2u + 7u) / 8u) * 8u; }
static inline u64 radix_table_words(u64 max_n) {
  // sized for the smallest tile in use (six items per thread, the long-key sort's), which gives the largest tile count
  const u64 tile = (u64)kRadixTPB * 6;
  u64 ntiles = (max_n + tile - 1) / tile;
  return ntiles * kRadixBins;
}
static inline u64 radix_partial_words(u64 max_n) {
  return (radix_table_words(max_n) + kScanTile - 1) / kScanTile + 1;
}

// Optional HIP-event probe around every scatter launch (bench.py's roofline figure).
struct ScatterProbe {
  static constexpr int kMax = 1024;
  bool enabled = false;
  hipEvent_t ev[2 * kMax];
  u64 bytes[kMax];
  int used = 0;
  bool created = false;
  u64 total_launches = 0, total_bytes = 0;
  double total_ms = 0.0;
  bool begin(hipStream_t st) {
    if (!enabled || used >= kMax) return false;
    if (!created) { for (int i = 0; i < 2 * kMax; ++i) note_error(hipEventCreate(&ev[i])); created = true; }
    note_error(hipEventRecord(ev[2 * used], st));
    return true;
  }
  void end(hipStream_t st, u64 nbytes) { note_error(hipEventRecord(ev[2 * used + 1], st)); bytes[used++] = nbytes; }
  // call after the stream has been synchronised
  void harvest() {
    for (int i = 0; i < used; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) == hipSuccess) {
        total_ms += ms; total_bytes += bytes[i]; ++total_launches;
      }
    }
    used = 0;
  }
  void destroy() { if (created) for (int i = 0; i < 2 * kMax; ++i) (void)hipEventDestroy(ev[i]); created = false; }
};

// Sorts n pairs by key bits [bit_lo, nbits) (stable, so lower bits keep their order).  Buffers ping-pong; on return *k_sorted/*v_sorted
// point at whichever of (k0,v0)/(k1,v1) holds the result.  values_are_positions: v0 need not be
// filled, item i's value is i (the first pass makes them up instead of reading them).  keys_only:
// v0/v1 are not touched at all.
// n_holes > 0: the input holds n + n_holes slots of which n_holes carry the all-ones key and do
// not exist; the first pass drops them (at least one pass is then made, even for nbits == 0).
// plane0 / plane1 (optional, n bytes each, 16-byte aligned): digit planes; every pass but the
// last leaves the next pass's digits there and the next histogram is taken from them.
// plane0_ready: the producer of the keys already left the first pass's digits in plane0.
template <typename K, typename V = u32>
static inline void radix_sort_pairs(K* k0, K* k1, V* v0, V* v1, u64 n, int nbits,
                                    u32* table, u32* partial, hipStream_t st,
                                    K** k_sorted, V** v_sorted, ScatterProbe* probe = nullptr,
                                    int bit_lo = 0, bool values_are_positions = false, bool keys_only = false,
                                    u64 n_holes = 0, u8* plane0 = nullptr, u8* plane1 = nullptr,
                                    bool plane0_ready = false, bool values_descend = false) {
  K* kin = k0; K* kout = k1;
  V* vin = v0; V* vout = v1;
  if (n > 1 || n_holes) {
    bool first = true;
    for (int shift = bit_lo; shift < nbits || (first && n_holes); shift += kRadixBits) {
      const u64 n_in = first ? n + n_holes : n;
      const int skip = first && n_holes ? 1 : 0;
      const u32 ntiles = ceil_div(n_in, radix_tile<K>());
      // values_descend: item i's value is n - 1 - i (v0 need not be filled either)
      const int vmode = keys_only ? 2 : (values_are_positions && shift == bit_lo) ? 1 : (values_descend && shift == bit_lo) ? 3 : 0;
      const dim3 sgrid(((ntiles + 7u) / 8u) * 8u);
      const bool have_plane = plane0 && (!first || (plane0_ready && !skip));   // the previous pass (or the producer) left this pass's digits
      const bool make_plane = plane0 && shift + kRadixBits < nbits;         // and this one leaves the next pass's
      if (have_plane) hipLaunchKernelGGL(k_radix_hist_plane<K>, dim3(hist_plane_grid(ntiles)), dim3(kRadixTPB), 0, st, (const u8*)plane0, table, n_in, ntiles, SegArgs());
      else if (skip) hipLaunchKernelGGL((k_radix_hist<K, true>), sgrid, dim3(kRadixTPB), 0, st, kin, table, n_in, shift, ntiles);
      else hipLaunchKernelGGL((k_radix_hist<K, false>), sgrid, dim3(kRadixTPB), 0, st, kin, table, n_in, shift, ntiles);
      exclusive_scan_u32(table, (u64)ntiles * kRadixBins, partial, st);
      const bool timed = probe && probe->begin(st);
#define BWTC_SCATTER(S, P, KO) hipLaunchKernelGGL((k_radix_scatter<K, S, P, KO, V>), sgrid, dim3(kRadixTPB), 0, st, kin, vin, kout, vout, table, n_in, shift, ntiles, vmode, make_plane ? plane1 : (u8*)nullptr, LongArgs(), SegArgs())
      if (keys_only) { if (make_plane) BWTC_SCATTER(false, true, true); else BWTC_SCATTER(false, false, true); }   // never with holes
      else if (skip) { if (make_plane) BWTC_SCATTER(true, true, false); else BWTC_SCATTER(true, false, false); }
      else { if (make_plane) BWTC_SCATTER(false, true, false); else BWTC_SCATTER(false, false, false); }
#undef BWTC_SCATTER
      { u8* tp = plane0; plane0 = plane1; plane1 = tp; }
      // algorithmic bytes of the pass: every item read once and written once, plus the digit plane it leaves
      if (timed) probe->end(st, (n_in + n) * (sizeof(K) + (keys_only ? 0 : sizeof(V))) + (make_plane ? n : 0));
      K* tk = kin; kin = kout; kout = tk;
      V* tv = vin; vin = vout; vout = tv;
      first = false;
    }
  }
  *k_sorted = kin;
  *v_sorted = vin;
}

// Segmented sort of 32-bit keys (see SegArgs) by bits [bit_lo, bit_lo + 16): two passes, stable; n = whole
// tiles.  The producer of the keys left the first pass's digits in plane0; after the two passes the keys are in k0 again.
static inline void radix_sort_keys_segmented(u32* k0, u32* k1, u64 n, int bit_lo, const u32* tile_first, u32 nseg,
                                             u32* table, u32* partial, hipStream_t st, u32** k_sorted,
                                             u8* plane0, u8* plane1) {
  constexpr int E = RadixCfg<u32>::E;
  const u32 ntiles = (u32)(n / radix_tile<u32>());
  const dim3 sgrid(((ntiles + 7u) / 8u) * 8u), hgrid(hist_plane_grid(ntiles)), tpb(kRadixTPB);
  SegArgs sa; sa.tile_first = tile_first; sa.nseg = nseg;
  hipLaunchKernelGGL((k_radix_hist_plane<u32, E, true>), hgrid, tpb, 0, st, (const u8*)plane0, table, n, ntiles, sa);
  exclusive_scan_u32(table, (u64)ntiles * kRadixBins, partial, st);
  hipLaunchKernelGGL((k_radix_scatter<u32, false, true, true, u32, 0, E, true>), sgrid, tpb, 0, st, (const u32*)k0, (const u32*)nullptr, k1,
                     (u32*)nullptr, (const u32*)table, n, bit_lo, ntiles, 2, plane1, LongArgs(), sa);
  hipLaunchKernelGGL((k_radix_hist_plane<u32, E, true>), hgrid, tpb, 0, st, (const u8*)plane1, table, n, ntiles, sa);
  exclusive_scan_u32(table, (u64)ntiles * kRadixBins, partial, st);
  hipLaunchKernelGGL((k_radix_scatter<u32, false, false, true, u32, 0, E, true>), sgrid, tpb, 0, st, (const u32*)k1, (const u32*)nullptr, k0,
                     (u32*)nullptr, (const u32*)table, n, bit_lo + kRadixBits, ntiles, 2, (u8*)nullptr, LongArgs(), sa);
  *k_sorted = k0;
}

// Long-key sort of the suffix sorter: items (u64 key, V value, u32 w) ordered by (key bits [0, kbits),
// w bits [0, wbits)) -- w is the LESS significant word, so its passes come first.  Stable.  The
// producer of the items left the first pass's digits (of w) in plane0; both planes are required.
// The first pass makes the values up (item i's value is n - 1 - i).
// EL: items per thread (a tile is kRadixTPB * EL items; 8 for 14-byte items = 56 KiB of LDS staging, two workgroups
// per CU; 6 keeps three: 42 KiB for 14-byte, 48 KiB for 16-byte items).
template <typename V, int EL = 8>
static inline void radix_sort_long(u64* k0, u64* k1, V* v0, V* v1, u32* w0, u32* w1, u64 n, int kbits, int wbits,
                                   u32* table, u32* partial, hipStream_t st, u64** k_sorted, V** v_sorted,
                                   u32** w_sorted, ScatterProbe* probe, u8* plane0, u8* plane1, int wlo = 0,
                                   bool values_given = false, bool direct_w = true) {
  // direct_w: the passes over w's digits (but the first, whose plane the producer made) take their histograms from the w
  // words themselves and the pass before them leaves no plane.  A plane costs its scattered byte stores (runs of 16 bytes:
  // 0.33 ms of a 1.74 ms pass over 2^28 items) and the histogram's read of it (0.13 ms); reading 4 bytes per item costs
  // 0.2 ms.  (A key digit's 8 bytes per item cost what the plane does: those passes keep theirs.)
  // wlo: w's lowest bits that do NOT take part in the order (the sorted bits are [wlo, wlo + wbits))
  u64 *kin = k0, *kout = k1;
  V *vin = v0, *vout = v1;
  u32 *win = w0, *wout = w1;
  struct Pass { int from_w, shift; u32 dmask; int bridge; };    // from_w: 0 key, 1 w, 2 bridge (w's top bits under the key's lowest)
  Pass ps[24];
  int np = 0;
  int s = 0;
  for (; s + kRadixBits <= wbits; s += kRadixBits) ps[np++] = Pass{1, wlo + s, (1u << kRadixBits) - 1u, 0};
  int ks = 0;
  if (s < wbits) {
    const int r = wbits - s;                                     // w bits left: the bridge digit's low bits
    ps[np++] = Pass{2, wlo + s, (1u << std::min(kRadixBits, r + kbits)) - 1u, r};
    ks = std::min(kbits, kRadixBits - r);
  }
  for (; ks < kbits; ks += kRadixBits) ps[np++] = Pass{0, ks, (1u << std::min(kRadixBits, kbits - ks)) - 1u, 0};
  if (n > 1) {
    const u32 ntiles = ceil_div(n, (u64)kRadixTPB * EL);
    const dim3 sgrid(((ntiles + 7u) / 8u) * 8u);
    for (int p = 0; p < np; ++p) {
      const bool from_words = direct_w && p > 0 && ps[p].from_w == 1 && ps[p].dmask == (1u << kRadixBits) - 1u;
      if (from_words) hipLaunchKernelGGL((k_radix_hist_words<EL>), dim3(hist_plane_grid(ntiles)), dim3(kRadixTPB), 0, st, (const u32*)win, table, n, ps[p].shift, ntiles);
      else hipLaunchKernelGGL((k_radix_hist_plane<u64, EL>), dim3(hist_plane_grid(ntiles)), dim3(kRadixTPB), 0, st, (const u8*)plane0, table, n, ntiles, SegArgs());
      exclusive_scan_u32(table, (u64)ntiles * kRadixBins, partial, st);
      const bool last = p + 1 == np;
      const bool no_plane = last || (direct_w && ps[p + 1].from_w == 1 && ps[p + 1].dmask == (1u << kRadixBits) - 1u);   // the next pass reads none
      LongArgs la;
      la.win = win; la.wout = wout; la.dmask = ps[p].dmask; la.bridge = ps[p].bridge;
      la.nshift = last ? 0 : ps[p + 1].shift; la.nfrom_w = last ? 0 : ps[p + 1].from_w; la.ndmask = last ? 0u : ps[p + 1].dmask;
      la.nbridge = last ? 0 : ps[p + 1].bridge;
      const int vmode = (p == 0 && !values_given) ? 3 : 0;     // values_given: v0 holds the items' values (else the first pass makes them up)
      const bool timed = probe && probe->begin(st);
#define BWTC_SCATTER_L(P, L) hipLaunchKernelGGL((k_radix_scatter<u64, false, P, false, V, L, EL>), sgrid, dim3(kRadixTPB), 0, st, kin, vin, kout, vout, table, n, ps[p].shift, ntiles, vmode, no_plane ? (u8*)nullptr : plane1, la, SegArgs())
      if (ps[p].from_w == 2) { if (no_plane) BWTC_SCATTER_L(false, 3); else BWTC_SCATTER_L(true, 3); }
      else if (ps[p].from_w) { if (no_plane) BWTC_SCATTER_L(false, 2); else BWTC_SCATTER_L(true, 2); }
      else { if (no_plane) BWTC_SCATTER_L(false, 1); else BWTC_SCATTER_L(true, 1); }
#undef BWTC_SCATTER_L
      if (timed) probe->end(st, 2 * n * (sizeof(u64) + sizeof(V) + sizeof(u32)) + (no_plane ? 0 : n));
      { u8* tp = plane0; plane0 = plane1; plane1 = tp; }
      { u64* t = kin; kin = kout; kout = t; }
      { V* t = vin; vin = vout; vout = t; }
      { u32* t = win; win = wout; wout = t; }
    }
  }
  *k_sorted = kin; *v_sorted = vin; *w_sorted = win;
}


// ---------------------------------------------------------------------------------------
// Chained single-read variant ("sweep"): no per-pass histogram kernel.
//
// One kernel reads the keys once and builds the global digit histogram of every pass (the
// multiset of keys does not change between passes); k_radix_bases turns them into the
// global base of every digit.  The pass kernel then ranks a tile exactly like
// k_radix_scatter, publishes its 256 digit counts, and gets the counts of all earlier tiles
// by decoupled look-back over per-(tile, digit) status words (2 flag bits + 30-bit value,
// relaxed agent-scope atomics, one word = one granule).
//
// Progress: tiles are handed out by a ticket counter, so every tile a workgroup waits for
// was claimed by a workgroup that is already resident, and the lowest unfinished tile never
// waits; nothing depends on dispatch order or XCD placement.  (Per-XCD chains would keep the
// output seams in one L2, but the per-chain digit bases change with every pass and would
// need the histogram pass back.)  Every spin is bounded: on timeout the tile publishes an
// error word, raises *error and writes nothing; waiters that read the error word do the same.
// ---------------------------------------------------------------------------------------
constexpr int kChains = 1;
constexpr int kSweepMaxPasses = 8;
constexpr u32 kStAgg = 1u << 30, kStIncl = 2u << 30, kStErr = 3u << 30, kStMask = (1u << 30) - 1u;
constexpr u32 kSweepSpinLimit = 1u << 22;

struct SweepWs {
  u32* hist_all;   // [kSweepMaxPasses][kChains][256]
  u32* bases;      // [kSweepMaxPasses][kChains][256]
  u32* tickets;    // [kSweepMaxPasses][kChains]
  u32* status;     // [ntiles][256]
  u32* error;      // 1 word, sticky
  static constexpr u64 small_words() { return 2ull * kSweepMaxPasses * kChains * 256 + kSweepMaxPasses * kChains + 64; }
};

template <typename K>
__global__ __launch_bounds__(kRadixTPB) void k_radix_hist_all(const K* __restrict__ keys, u64 n,
                                                              int bit_lo, int npasses,
                                                              u32 tiles_per_chain, u32 ntiles,
                                                              u32* __restrict__ hist_all) {
  constexpr int E = RadixCfg<K>::E;
  constexpr int TILE = kRadixTPB * E;
  __shared__ u32 hist[kSweepMaxPasses][kRadixBins][4];      // 4 interleaved copies per bin
  for (u32 i = threadIdx.x; i < kSweepMaxPasses * kRadixBins * 4; i += kRadixTPB) (&hist[0][0][0])[i] = 0;
  __syncthreads();
  const u32 chain = blockIdx.x % kChains;
  const u32 nper = gridDim.x / kChains;
  const u32 first = chain * tiles_per_chain;
  const u32 last = min(first + tiles_per_chain, ntiles);
  const u32 copy = threadIdx.x & 3u;
  for (u32 tile = first + blockIdx.x / kChains; tile < last; tile += nper) {
    const u64 base = (u64)tile * TILE + threadIdx.x;
    K k[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { const u64 i = base + (u64)e * kRadixTPB; k[e] = i < n ? keys[i] : (K)0; }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (base + (u64)e * kRadixTPB < n) {
        for (int p = 0; p < npasses; ++p)
          atomicAdd(&hist[p][radix_digit(k[e], bit_lo + p * kRadixBits)][copy], 1u);
      }
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < (u32)npasses * kRadixBins; i += kRadixTPB) {
    const u32 p = i / kRadixBins, d = i % kRadixBins;
    const u32 c = hist[p][d][0] + hist[p][d][1] + hist[p][d][2] + hist[p][d][3];
    if (c) atomicAdd(&hist_all[((u64)p * kChains + chain) * kRadixBins + d], c);
  }
}

// grid = npasses blocks of 256: bases[p][c][d] = #keys with smaller digit + #keys with digit d
// in earlier chains.
static __global__ __launch_bounds__(kRadixBins) void k_radix_bases(const u32* __restrict__ hist_all,
                                                                   u32* __restrict__ bases) {
  __shared__ u32 scr[kRadixBins / kWave + 1];
  const u32 p = blockIdx.x, d = threadIdx.x;
  u32 total = 0;
  for (int c = 0; c < kChains; ++c) total += hist_all[((u64)p * kChains + c) * kRadixBins + d];
  u32 unused;
  u32 run = block_scan_excl_add<kRadixBins>(total, scr, &unused);
  for (int c = 0; c < kChains; ++c) {
    bases[((u64)p * kChains + c) * kRadixBins + d] = run;
    run += hist_all[((u64)p * kChains + c) * kRadixBins + d];
  }
}

template <typename K>
__global__ __launch_bounds__(kRadixTPB) void k_radix_sweep(
    const K* __restrict__ kin, const u32* __restrict__ vin, K* __restrict__ kout,
    u32* __restrict__ vout, u64 n, int shift, u32 ntiles, u32 tiles_per_chain,
    const u32* __restrict__ bases, u32* __restrict__ status, u32* __restrict__ tickets,
    u32* __restrict__ error) {
  constexpr int E = RadixCfg<K>::E;
  constexpr int TILE = kRadixTPB * E;
  constexpr int kStageBytes = TILE * (int)(sizeof(K) + sizeof(u32));
  constexpr int kCntBytes = kRadixWaves * kRadixBins * (int)sizeof(u32);
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[kStageBytes > kCntBytes ? kStageBytes : kCntBytes];
  K* s_key = reinterpret_cast<K*>(s_raw);
  u32* s_val = reinterpret_cast<u32*>(s_raw + TILE * sizeof(K));
  u32 (*s_cnt)[kRadixBins] = reinterpret_cast<u32 (*)[kRadixBins]>(s_raw);
  __shared__ u32 s_base[kRadixBins];
  __shared__ u32 s_gofs[kRadixBins];
  __shared__ u32 s_scr[kRadixTPB / kWave + 1];
  __shared__ u32 s_claim[3];                       // chain (always 0 with one chain), ticket, abort

  const u32 tid = threadIdx.x;
  const u32 wave = tid / kWave, lane = tid % kWave;

  if (tid == 0) {
    u32 chain = 0xFFFFFFFFu, ticket = 0;
    for (u32 k = 0; k < (u32)kChains; ++k) {
      const u32 c = (blockIdx.x + k) % (u32)kChains;
      const u32 cfirst = c * tiles_per_chain;
      if (cfirst >= ntiles) continue;
      const u32 ccount = min(tiles_per_chain, ntiles - cfirst);
      const u32 t = atomicAdd(&tickets[c], 1u);
      if (t < ccount) { chain = c; ticket = t; break; }
    }
    s_claim[0] = chain; s_claim[1] = ticket; s_claim[2] = 0;
  }
  for (u32 i = tid; i < kRadixWaves * kRadixBins; i += kRadixTPB) (&s_cnt[0][0])[i] = 0;
  __syncthreads();
  const u32 chain = s_claim[0], ticket = s_claim[1];
  if (chain == 0xFFFFFFFFu) return;                // every tile already claimed
  const u32 chain_first = chain * tiles_per_chain;
  const u32 tile = chain_first + ticket;
  const u64 tile_base = (u64)tile * TILE;
  const u64 left = n - tile_base;
  const u32 tile_n = left < (u64)TILE ? (u32)left : (u32)TILE;

  const u32 wslot = wave * (kWave * E) + lane;
  K k[E];
  u32 v[E];
  u32 r[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const u32 slot = wslot + e * kWave;
    const bool ok = slot < tile_n;
    k[e] = ok ? kin[tile_base + slot] : (K)0;
    v[e] = ok ? vin[tile_base + slot] : 0u;
  }

  volatile lds_u32* my_cnt = lds_volatile(&s_cnt[wave][0]);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const bool ok = (wslot + e * kWave) < tile_n;
    const u32 d = radix_digit(k[e], shift);
    u32 below, peers;
    wave_digit_rank<kRadixBits>(d, ok, &below, &peers);
    u32 prev = 0;
    if (ok) prev = my_cnt[d];
    __builtin_amdgcn_wave_barrier();
    r[e] = prev + below;
    if (ok && below + 1u == peers) my_cnt[d] = prev + peers;
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  u32 dig_total = 0;
  if (tid < kRadixBins) {
#pragma unroll
    for (int w = 0; w < kRadixWaves; ++w) {
      const u32 c = s_cnt[w][tid];
      s_cnt[w][tid] = dig_total;
      dig_total += c;
    }
  }
  // publish this tile's digit counts, then look back along the chain
  u32 excl = 0;
  if (tid < kRadixBins) {
    u32* my_status = status + (u64)tile * kRadixBins + tid;
    if (ticket == 0) {
      __hip_atomic_store(my_status, kStIncl | dig_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __hip_atomic_store(my_status, kStAgg | dig_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool bad = false;
      u32 t = ticket;
      while (t > 0) {
        --t;
        const u32* st = status + (u64)(chain_first + t) * kRadixBins + tid;
        u32 val = 0, spins = 0;
        for (;;) {
          val = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (val >> 30) break;
          if (++spins > kSweepSpinLimit) break;
          __builtin_amdgcn_s_sleep(2);
        }
        const u32 flag = val >> 30;
        if (flag == 0 || flag == 3) { bad = true; break; }
        excl += val & kStMask;
        if (flag == 2) break;
      }
      if (bad) {
        s_claim[2] = 1;
        atomicOr(error, 1u);
        __hip_atomic_store(my_status, kStErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_store(my_status, kStIncl | (excl + dig_total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  u32 unused;
  const u32 dig_base = block_scan_excl_add<kRadixTPB>(tid < kRadixBins ? dig_total : 0u, s_scr, &unused);
  if (tid < kRadixBins) {
    s_base[tid] = dig_base;
    s_gofs[tid] = bases[(u64)chain * kRadixBins + tid] + excl - dig_base;
  }
  __syncthreads();
  if (s_claim[2]) return;                          // a look-back timed out: write nothing

#pragma unroll
  for (int e = 0; e < E; ++e) {
    if ((wslot + e * kWave) < tile_n) {
      const u32 d = radix_digit(k[e], shift);
      r[e] += s_base[d] + s_cnt[wave][d];
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if ((wslot + e * kWave) < tile_n) {
      s_key[r[e]] = k[e];
      s_val[r[e]] = v[e];
    }
  }
  __syncthreads();
  for (u32 i = tid; i < tile_n; i += kRadixTPB) {
    const K kk = s_key[i];
    const u32 dst = s_gofs[radix_digit(kk, shift)] + i;
    kout[dst] = kk;
    vout[dst] = s_val[i];
  }
}

// Same contract as radix_sort_pairs; ws.status needs ntiles * 256 words.
template <typename K>
static inline void radix_sort_pairs_sweep(K* k0, K* k1, u32* v0, u32* v1, u64 n, int nbits,
                                          const SweepWs& ws, hipStream_t st, K** k_sorted,
                                          u32** v_sorted, ScatterProbe* probe = nullptr,
                                          int bit_lo = 0) {
  K* kin = k0; K* kout = k1;
  u32* vin = v0; u32* vout = v1;
  const int npasses = (nbits - bit_lo + kRadixBits - 1) / kRadixBits;
  if (n > 1 && npasses > 0) {
    const u32 ntiles = ceil_div(n, radix_tile<K>());
    const u32 per_chain = (ntiles + kChains - 1) / kChains;
    note_error(hipMemsetAsync(ws.hist_all, 0, (size_t)npasses * kChains * kRadixBins * 4, st));
    note_error(hipMemsetAsync(ws.tickets, 0, (size_t)kSweepMaxPasses * kChains * 4, st));
    const u32 hist_blocks = kChains * (per_chain < 768u ? per_chain : 768u);
    hipLaunchKernelGGL(k_radix_hist_all<K>, dim3(hist_blocks), dim3(kRadixTPB), 0, st, kin, n, bit_lo,
                       npasses, per_chain, ntiles, ws.hist_all);
    hipLaunchKernelGGL(k_radix_bases, dim3(npasses), dim3(kRadixBins), 0, st, ws.hist_all, ws.bases);
    for (int p = 0; p < npasses; ++p) {
      note_error(hipMemsetAsync(ws.status, 0, (size_t)ntiles * kRadixBins * 4, st));
      const bool timed = probe && probe->begin(st);
      hipLaunchKernelGGL(k_radix_sweep<K>, dim3(((ntiles + 7u) / 8u) * 8u), dim3(kRadixTPB), 0, st,
                         kin, vin, kout, vout, n, bit_lo + p * kRadixBits, ntiles, per_chain,
                         ws.bases + (size_t)p * kChains * kRadixBins, ws.status,
                         ws.tickets + p * kChains, ws.error);
      if (timed) probe->end(st, n * 2 * (sizeof(K) + sizeof(u32)));
      K* tk = kin; kin = kout; kout = tk;
      u32* tv = vin; vin = vout; vout = tv;
    }
  }
  *k_sorted = kin;
  *v_sorted = vin;
}

}  // namespace bwtc_hip
