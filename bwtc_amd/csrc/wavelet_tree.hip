// Device half of the wavelet coder 'B': the bit vectors of the WaveletTree
// (WaveletTree.hpp:294-360, pushRun :1251-1275) and the traversal of encodeTreeBF
// (:637-809), done as whole-block passes instead of per-bit pushes.
//
// The host numbers every coded place of the block ("group": a visited node of a section's
// symbol tree, or one node position of an integer level; wavelet_host.cpp planStreams) and
// hands over lookup tables.  Then
//   expand    every run -> its steps (group, leaf rank, bit, gap flag): the code of its symbol
//             along the symbol tree, then the code of its length below the symbol's leaf.
//             The reference derives a step's gap flag from gap vectors handed down the tree
//             (WaveletTree.hpp:672-676, :735-741, :771-777): it is the OR, over the nodes above
//             the step on the run's path, of "this run's bit differs from the bit of the previous
//             run through that node".  Above the point where the codes of this run's and the
//             previous run's symbols part, the previous run through every node is the previous
//             run itself, with the same bit; at the parting node the bits differ; so the flag
//             is simply  depth > common prefix of the two codes  (0 for a section's first run,
//             whose every node sees it first), and is written into the step word here, below
//             the bits the sort looks at.
//   sort      stable LSD radix sort of the step words by (group, leaf rank), keys only: the bits
//             of every node, in run order, nodes in coding order (radix_sort.hpp, the suffix
//             sorter's passes)
//   select    encodeTreeBF codes every bit of some nodes and only part of others (:716-787,
//             "previous bit of the node" is the neighbour in sorted order); the survivors are
//             compacted to 2 bits each (bit, gap flag) for the host's models and range coder
// Everything is HBM-streaming integer work; the sort dominates.
#include "bwt_engine.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"
#include "wavelet_host.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace bwtc_hip {

using bwtc::wavelet::kStepGroupShift;
using bwtc::wavelet::kStepLeafShift;
using bwtc::wavelet::kStepGapShift;
constexpr u32 kStepGroupMask = bwtc::wavelet::kMaxGroups - 1;
constexpr u32 kWtLenDense = bwtc::wavelet::kLenDense;

constexpr int kWtTPB = 256;
constexpr int kWtE = 16;
constexpr int kWtTile = kWtTPB * kWtE;

struct WtTables {
  const u32* first_run;     // [nsec + 1]
  u32 nsec;
  const uint2* symtab;      // [nsec][256]
  const u64* symcode;       // [nsec][256] codes, first bit in bit 63
  const uint2* lendense;    // [nsec][kWtLenDense]
  const u32* over_first;    // [nsec + 1]
  const uint4* over;
  const u32* pool;
};

__device__ __forceinline__ u32 wt_section_of(const u32* s_first, u32 nsec, u32 r) {
  u32 lo = 0, hi = nsec;
  while (hi - lo > 1) { u32 mid = (lo + hi) >> 1; if (s_first[mid] <= r) lo = mid; else hi = mid; }
  return lo;
}

// The section of a workgroup's first run, searched with uniform (scalar) loads from the table in
// global memory; *uniform = all `cnt` runs from r0 on lie in that section (nearly always: a block
// has some eighty sections and a hundred million runs), so that the lanes need no search of their own.
__device__ __forceinline__ u32 wt_section_uniform(const u32* __restrict__ first_run, u32 nsec, u32 r0,
                                                  u32 cnt, bool* uniform, u32* sec_first) {
  u32 lo = 0, hi = nsec;
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (first_run[mid] <= r0) lo = mid; else hi = mid; }
  *uniform = r0 + cnt <= first_run[lo + 1];
  *sec_first = first_run[lo];
  return lo;
}

__device__ __forceinline__ uint2 wt_length_entry(const WtTables& t, u32 s, u32 len) {
  if (len < kWtLenDense) return t.lendense[(u64)s * kWtLenDense + len];
  u32 lo = t.over_first[s], hi = t.over_first[s + 1];
  while (lo < hi) { u32 mid = (lo + hi) >> 1; if (t.over[mid].x < len) lo = mid + 1; else hi = mid; }
  const uint4 e = t.over[lo];
  return make_uint2(e.y, e.z);
}

// steps per run -> cnt[r]; cnt[n_runs] = 0 so that the exclusive scan leaves the total there.
// Four runs per thread, and the table lookups unconditional, so that a thread has its eight
// dependent lookups in flight together (one run per thread was a chain of three latencies).
constexpr int kWtCountE = 4;

__global__ __launch_bounds__(kWtTPB) void k_wt_count(const u32* __restrict__ run_start,
                                                     const u8* __restrict__ run_sym, u32 n_runs,
                                                     WtTables t, u32* __restrict__ cnt) {
  __shared__ u32 s_first[257];
  const u32 r0 = blockIdx.x * (kWtTPB * kWtCountE);
  const u32 rs = min(r0, n_runs - 1u);                     // the last workgroup may hold only slot n_runs
  bool uniform; u32 sec_first;
  const u32 sA = wt_section_uniform(t.first_run, t.nsec, rs, min((u32)(kWtTPB * kWtCountE), n_runs - rs), &uniform, &sec_first);
  if (!uniform) {                                          // a section border inside the workgroup (same for all its threads)
    for (u32 i = threadIdx.x; i <= t.nsec; i += kWtTPB) s_first[i] = t.first_run[i];
    __syncthreads();
  }
  u32 sec[kWtCountE], len[kWtCountE];
  uint2 sy[kWtCountE], led[kWtCountE];
#pragma unroll
  for (int u = 0; u < kWtCountE; ++u) {
    const u32 r = min(r0 + (u32)u * kWtTPB + threadIdx.x, n_runs - 1u);
    sec[u] = uniform ? sA : wt_section_of(s_first, t.nsec, r);
    len[u] = run_start[r + 1] - run_start[r];
    sy[u] = t.symtab[sec[u] * 256u + run_sym[r]];
    led[u] = t.lendense[(u64)sec[u] * kWtLenDense + min(len[u], kWtLenDense - 1u)];
  }
#pragma unroll
  for (int u = 0; u < kWtCountE; ++u) {
    const u32 r = r0 + (u32)u * kWtTPB + threadIdx.x;
    if (r > n_runs) continue;
    u32 c = 0;
    if (r < n_runs) {
      c = sy[u].y & 255u;
      if (sy[u].y >> 16) c += len[u] < kWtLenDense ? led[u].y : wt_length_entry(t, sec[u], len[u]).y;
    }
    cnt[r] = c;
  }
}

// Two phases per workgroup of 512 runs (two per thread).  One thread per RUN looks the run up (symbol entry, gap
// prefix, length entry), parks what its steps need in LDS and writes its number into the owner
// byte of each of its steps; then one thread per STEP of the workgroup's consecutive stretch of
// output (off[] is the prefix of the step counts) fetches its word from the pool, so the lookups
// of all steps are independent and consecutive lanes store consecutive words.  (A thread walking
// its own run's steps ran the pool lookups one after the other, as many as the longest code in its
// wave, and stored a word per lane and cache line.)  Stretches longer than the owner window (very
// long codes) find the run by a search over the 257 offsets instead.  `plane` receives the sort's
// first digit of every step word (radix_sort.hpp, digit planes).
constexpr u32 kWtOwnerWin = 8192;     // steps of a workgroup's runs that the owner table covers: 16 per run on average
constexpr int kWtExpandR = 2;         // runs per thread (their lookups are in flight together)
constexpr u32 kWtExpandRuns = kWtTPB * kWtExpandR;

__global__ __launch_bounds__(kWtTPB) void k_wt_expand(const u32* __restrict__ run_start,
                                                      const u8* __restrict__ run_sym, u32 n_runs,
                                                      WtTables t, const u32* __restrict__ off,
                                                      u32* __restrict__ key, u8* __restrict__ plane,
                                                      const uint4* __restrict__ segtab) {
  __shared__ u32 s_off[kWtExpandRuns + 1];
  __shared__ u8 s_sec[kWtExpandRuns];         // segmented layout (k_wt_segments): the run's section
  __shared__ u32 s_sym[kWtExpandRuns];        // pool offset of the run's symbol steps
  __shared__ u32 s_len[kWtExpandRuns];        // pool offset of its length steps
  __shared__ u32 s_meta[kWtExpandRuns];       // symbol steps | common prefix << 8 | leaf rank << 16
  __shared__ unsigned short s_owner[kWtOwnerWin];
  const u32 r0 = blockIdx.x * kWtExpandRuns;
  const u32 cnt = min(kWtExpandRuns, n_runs - r0);
  bool uniform; u32 sec_first0;
  const u32 s_uniform = wt_section_uniform(t.first_run, t.nsec, r0, cnt, &uniform, &sec_first0);
  // Everything a run needs is looked up before the first barrier, in two rounds of independent
  // loads (the runs' own words and offsets, then the table entries they select).
  u32 my_off[kWtExpandR], c[kWtExpandR], cp[kWtExpandR], len[kWtExpandR], sec[kWtExpandR], sec_first[kWtExpandR];
#pragma unroll
  for (int q = 0; q < kWtExpandR; ++q) {
    const u32 r = r0 + (u32)q * kWtTPB + threadIdx.x;
    const u32 rr = min(r, n_runs - 1u);
    my_off[q] = off[min(r, n_runs)];                        // off[n_runs] = total
    c[q] = run_sym[rr];
    cp[q] = run_sym[rr ? rr - 1u : 0u];
    len[q] = run_start[rr + 1] - run_start[rr];
    sec[q] = s_uniform; sec_first[q] = sec_first0;
  }
  const u32 end_off = threadIdx.x == 0 ? off[min(r0 + kWtExpandRuns, n_runs)] : 0u;
  if (!uniform) {                                           // a section border inside the workgroup: the lanes search for themselves
#pragma unroll
    for (int q = 0; q < kWtExpandR; ++q) {
      const u32 rr = min(r0 + (u32)q * kWtTPB + threadIdx.x, n_runs - 1u);
      u32 lo = 0, hi = t.nsec;
      while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (t.first_run[mid] <= rr) lo = mid; else hi = mid; }
      sec[q] = lo;
      sec_first[q] = t.first_run[lo];
    }
  }
  uint2 sy[kWtExpandR], led[kWtExpandR];
  u64 code_c[kWtExpandR], code_p[kWtExpandR];
#pragma unroll
  for (int q = 0; q < kWtExpandR; ++q) {
    sy[q] = t.symtab[sec[q] * 256u + c[q]];
    code_c[q] = t.symcode[sec[q] * 256u + c[q]];
    code_p[q] = t.symcode[sec[q] * 256u + cp[q]];
    led[q] = t.lendense[(u64)sec[q] * kWtLenDense + min(len[q], kWtLenDense - 1u)];
  }
#pragma unroll
  for (int q = 0; q < kWtExpandR; ++q) {
    const u32 x = (u32)q * kWtTPB + threadIdx.x;            // the run's number in the workgroup
    const u32 r = r0 + x;
    s_off[x] = my_off[q];
    if (r < n_runs) {
      // steps at depth > (common prefix with the previous run's code) carry the gap flag
      const u32 common = r > sec_first[q] ? (u32)__clzll(code_c[q] ^ code_p[q]) : 0u;
      u32 len_off = 0;
      if (sy[q].y >> 16) len_off = len[q] < kWtLenDense ? led[q].x : wt_length_entry(t, sec[q], len[q]).x;
      s_sym[x] = sy[q].x;
      s_len[x] = len_off;
      s_sec[x] = (u8)sec[q];
      s_meta[x] = (sy[q].y & 255u) | (min(common, 64u) << 8) | (((sy[q].y >> 8) & 255u) << 16);
    }
  }
  if (threadIdx.x == 0) s_off[kWtExpandRuns] = end_off;
  __syncthreads();
  const u32 o0 = s_off[0], o1 = s_off[cnt];
  const bool owned = o1 - o0 <= kWtOwnerWin;
  if (owned) {
#pragma unroll
    for (int q = 0; q < kWtExpandR; ++q) {
      const u32 x = (u32)q * kWtTPB + threadIdx.x;
      if (r0 + x < n_runs) {
        const u32 b = s_off[x] - o0, e = s_off[x + 1] - o0;
        for (u32 i = b; i < e; ++i) s_owner[i] = (unsigned short)x;
      }
    }
  }
  __syncthreads();
  // four steps per thread and round, every pool lookup of the round in flight before the first is used
  constexpr int kU = 4;
  for (u32 jb = o0 + threadIdx.x; jb < o1; jb += kWtTPB * kU) {
    u32 addr[kU], extra[kU], w[kU];
    uint4 sg[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const u32 j = jb + (u32)u * kWtTPB;
      addr[u] = 0; extra[u] = 0; sg[u] = make_uint4(0, 0, 0, 0);
      if (j < o1) {
        u32 lo;
        if (owned) lo = s_owner[j - o0];
        else {                                              // the last run whose first step is <= j owns j
          lo = 0;
          u32 hi = cnt;
          while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_off[mid] <= j) lo = mid; else hi = mid; }
        }
        const u32 i = j - s_off[lo];
        const u32 meta = s_meta[lo];
        const u32 ns = meta & 255u;
        const bool sym = i < ns;
        addr[u] = (sym ? s_sym[lo] : s_len[lo] - ns) + i;
        extra[u] = sym ? (i > ((meta >> 8) & 255u) ? 1u : 0u) << kStepGapShift : (meta >> 16) << kStepLeafShift;
        if (segtab) sg[u] = segtab[s_sec[lo]];
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) w[u] = t.pool[addr[u]] + extra[u];       // past the end: pool[0], not stored (+: a dense id takes the leaf rank as a summand)
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const u32 j = jb + (u32)u * kWtTPB;
      if (j < o1) {
        u32 word = w[u];
        if (segtab) {
          // the section's own dense id instead of (block-wide group, leaf rank): one per symbol-tree node, then
          // one per (integer node, symbol leaf)
          const u32 g = (word >> kStepGroupShift) - sg[u].y;
          const u32 id = g < sg[u].z ? g : sg[u].z + (g - sg[u].z) * sg[u].w + ((word >> kStepLeafShift) & 255u);
          word = (id << kStepLeafShift) | (word & 3u);
        }
        key[j + sg[u].x] = word; plane[j + sg[u].x] = (u8)(word >> kStepLeafShift);
      }
    }
  }
}

// Segmented layout of the steps (radix_sort.hpp, SegArgs): a section's steps only ever meet steps of the same
// section in sorted order (its groups are its own), so every section is a segment, padded to whole sort tiles, and
// the sort needs the digits that order ONE section's groups: two passes (leaf rank, group within the section)
// instead of three over the block-wide group number.
//   pstart[s]      where section s starts in the padded row (a multiple of the tile), pstart[nsec] = padded total
//   tile_first[s]  the same in tiles
//   segtab[s]      {pstart[s] - (steps before the section), first group, symbol-tree groups, symbol leaves}: the
//                  first is filled in here, the others come from the host; a step's key is its section's dense id
//                  (k_wt_expand), at most 65 536 per section
//   totals[0] = steps, totals[1] = padded total
__global__ void k_wt_segments(const u32* __restrict__ first_run, u32 nsec, const u32* __restrict__ off, u32 tile,
                              u32* __restrict__ pstart, u32* __restrict__ tile_first, uint4* __restrict__ segtab,
                              u32* __restrict__ totals) {
  if (blockIdx.x || threadIdx.x) return;
  u32 at = 0;
  for (u32 s = 0; s < nsec; ++s) {
    const u32 b = off[first_run[s]], e = off[first_run[s + 1]];
    pstart[s] = at; tile_first[s] = at / tile; segtab[s].x = at - b;
    at += (e - b + tile - 1u) / tile * tile;
  }
  pstart[nsec] = at; tile_first[nsec] = at / tile;
  totals[0] = off[first_run[nsec]]; totals[1] = at;
}
// the padding behind every section's steps: the all-ones key (and its digit in the first plane)
__global__ __launch_bounds__(kWtTPB) void k_wt_pad_fill(const u32* __restrict__ first_run, const u32* __restrict__ off,
                                                        const u32* __restrict__ pstart, u32* __restrict__ key,
                                                        u8* __restrict__ plane) {
  const u32 s = blockIdx.x;
  const u32 b = pstart[s] + (off[first_run[s + 1]] - off[first_run[s]]), e = pstart[s + 1];
  for (u32 j = b + threadIdx.x; j < e; j += kWtTPB) { key[j] = 0xFFFFFFFFu; plane[j] = 0xFF; }
}

constexpr u8 kWtSkip = 0xFF;

// sorted order: what encodeTreeBF does with the bit -> code[j] = bit | gap << 1, or kWtSkip;
// coded elements per tile -> tile_count; first element of every group -> group_start
template <bool DENSE>
__device__ __forceinline__ u32 wt_group_of(u32 k, const u32* __restrict__ id_group) {
  return DENSE ? id_group[k >> kStepLeafShift] : (k >> kStepGroupShift) & kStepGroupMask;
}

// SEG: the segmented layout -- the tile lies in one section, whose table entry turns a dense id back into the group
template <bool DENSE, bool SEG>
__global__ __launch_bounds__(kWtTPB) void k_wt_select(const u32* __restrict__ key, u32 n,
                                                      const u8* __restrict__ group_type, const u32* __restrict__ id_group,
                                                      u8* __restrict__ code,
                                                      u32* __restrict__ tile_count,
                                                      u32* __restrict__ group_start,
                                                      const u32* __restrict__ seg_tile_first, u32 nseg,
                                                      const uint4* __restrict__ segtab, const u32* __restrict__ seg_magic) {
  __shared__ u32 scratch[kWtTPB / kWave + 1];
  const u32 base = blockIdx.x * kWtTile;
  u32 sg_base = 0, sg_nodes = 0, sg_magic = 0, sg_start = 0;
  if (SEG) {
    const u32 tile = base / (u32)radix_tile<u32>();          // the sort's tiles are whole multiples of this kernel's
    u32 lo = 0, hi = nseg;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (seg_tile_first[mid] <= tile) lo = mid; else hi = mid; }
    const uint4 t = segtab[lo];
    sg_base = t.y; sg_nodes = t.z; sg_magic = seg_magic[lo];
    sg_start = seg_tile_first[lo] * (u32)radix_tile<u32>();
  }
  // group of a key of this tile's section: ids below sg_nodes are the groups themselves, the others hold one group per
  // `leaves` ids (the division as a multiplication by the host's 2^32 / leaves rounded up; 0: one leaf)
  auto seg_group = [&](u32 k) {
    const u32 id = k >> kStepLeafShift;
    if (id < sg_nodes) return sg_base + id;
    const u32 x = id - sg_nodes;
    return sg_base + sg_nodes + (sg_magic ? __umulhi(x, sg_magic) : x);
  };
  // all loads first: sixteen independent pairs in flight per thread instead of one at a time
  u32 k[kWtE], kp[kWtE];
#pragma unroll
  for (int e = 0; e < kWtE; ++e) {
    const u32 j = base + e * kWtTPB + threadIdx.x;
    k[e] = j < n ? key[j] : 0u;
    kp[e] = (j < n && j) ? key[j - 1] : 0u;
  }
  // (segmented layout: padding carries the all-ones key, in no group and never coded; its group number, kMaxGroups - 1,
  // is no section's, so the first step behind it starts its group)
  u32 grp[kWtE], grp_prev[kWtE];
#pragma unroll
  for (int e = 0; e < kWtE; ++e) {
    if (SEG) {
      grp[e] = k[e] == 0xFFFFFFFFu ? 0u : seg_group(k[e]);
      grp_prev[e] = kp[e] == 0xFFFFFFFFu ? 0xFFFFFFFFu : seg_group(kp[e]);
    } else {
      grp[e] = wt_group_of<DENSE>(k[e], id_group);         // k = 0 past the end: id / group 0 exists
      grp_prev[e] = wt_group_of<DENSE>(kp[e], id_group);
    }
  }
  u8 type[kWtE];
#pragma unroll
  for (int e = 0; e < kWtE; ++e) type[e] = group_type[grp[e]];
  u32 coded = 0;
#pragma unroll
  for (int e = 0; e < kWtE; ++e) {
    const u32 j = base + e * kWtTPB + threadIdx.x;
    if (SEG && j < n && k[e] == 0xFFFFFFFFu) {
      code[j] = kWtSkip;
    } else if (j < n) {
      const u32 g = grp[e];
      // (SEG: the step before a section's first one is the last of the section before, read with the wrong table)
      const bool first = j == 0 || grp_prev[e] != g || (SEG && j == sg_start);
      if (first) group_start[g] = j;
      const u32 bit = k[e] & 1u;
      u8 c;
      if (type[e] == bwtc::wavelet::kRoot || type[e] == bwtc::wavelet::kInteger) {
        c = (u8)bit;
      } else {
        const u32 gap = (k[e] >> kStepGapShift) & 1u;
        const u32 prev = first ? (bit ^ 1u) : (kp[e] & 1u);
        bool keep = true;
        if (type[e] == bwtc::wavelet::kBothLeaves) keep = gap != 0;
        else if (type[e] == bwtc::wavelet::kLeftLeaf) keep = (prev | gap) != 0;
        c = keep ? (u8)(bit | (gap << 1)) : kWtSkip;
      }
      code[j] = c;
      coded += c != kWtSkip;
    }
  }
  u32 total;
  block_scan_excl_add<kWtTPB>(coded, scratch, &total);
  if (threadIdx.x == 0) tile_count[blockIdx.x] = total;
}

// Keeps the coded elements, in order, and packs them: element i of the block's coded sequence in
// bits 2(i % 16).. of word i / 16.  A tile's survivors are gathered in LDS and the words they fill
// are assembled from there; the words a tile shares with its neighbours (its first and last) are
// OR-ed into the zeroed output, the others stored.
__global__ __launch_bounds__(kWtTPB) void k_wt_compact_pack(const u8* __restrict__ code, u32 n,
                                                            const u32* __restrict__ tile_off,
                                                            u32* __restrict__ packed) {
  __shared__ u32 scratch[kWtTPB / kWave + 1];
  __shared__ u8 s_val[kWtTile];
  const u32 base = blockIdx.x * kWtTile + threadIdx.x * kWtE;
  u8 c[kWtE];
  u32 mine = 0;
  if (base + kWtE <= n) {
    const uint4 v = *reinterpret_cast<const uint4*>(code + base);
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < kWtE; ++e) c[e] = (u8)(w[e >> 2] >> (8 * (e & 3)));
  } else {
#pragma unroll
    for (int e = 0; e < kWtE; ++e) c[e] = base + e < n ? code[base + e] : kWtSkip;
  }
#pragma unroll
  for (int e = 0; e < kWtE; ++e) mine += c[e] != kWtSkip;
  u32 total;
  u32 o = block_scan_excl_add<kWtTPB>(mine, scratch, &total);
#pragma unroll
  for (int e = 0; e < kWtE; ++e) if (c[e] != kWtSkip) s_val[o++] = c[e];
  __syncthreads();
  if (total == 0) return;
  const u32 G = tile_off[blockIdx.x];                   // coded elements before the tile
  const u32 w0 = G >> 4, w1 = (G + total - 1u) >> 4;
  for (u32 w = w0 + threadIdx.x; w <= w1; w += kWtTPB) {
    u32 word = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const u32 i = (w << 4) + (u32)e - G;              // wraps below zero for elements of the previous tile
      if (i < total) word |= (u32)(s_val[i] & 3u) << (2 * e);
    }
    if (w == w0 || w == w1) atomicOr(&packed[w], word);
    else packed[w] = word;
  }
}

// coded elements before the first element of every group: the tile's offset plus the coded
// elements of the tile before the group's first one, counted by a wave (one lane walking up to
// 4096 bytes on its own was a quarter of a millisecond of latencies)
__global__ __launch_bounds__(kWtTPB) void k_wt_group_pos(const u32* __restrict__ group_start,
                                                         u32 n_groups, u32 n,
                                                         const u8* __restrict__ code,
                                                         const u32* __restrict__ tile_off,
                                                         u32* __restrict__ coded_pos) {
  const u32 g = blockIdx.x * (kWtTPB / kWave) + threadIdx.x / kWave;
  const u32 lane = lane_id();
  if (g >= n_groups) return;
  const u32 j = group_start[g];
  if (j >= n) { if (lane == 0) coded_pos[g] = 0xFFFFFFFFu; return; }     // empty group, the host fills it in
  const u32 tile = j / kWtTile;
  u32 c = 0;
  for (u32 i = tile * kWtTile + lane; i < j; i += kWave) c += code[i] != kWtSkip;
  c = wave_scan_add(c, lane);
  if (lane == kWave - 1) coded_pos[g] = tile_off[tile] + c;
}

static inline u64 align_up(u64 v, u64 a) { return (v + a - 1) / a * a; }

int BwtEngine::reserve_wavelet(u64 device_bytes, u64 host_bytes) {
  if (device_bytes > wt_bytes) {
    if (d_wt) (void)hipFree(d_wt);
    d_wt = nullptr; wt_bytes = 0;
    const u64 want = align_up(device_bytes + device_bytes / 8, 1u << 20);
    BWTC_HIP_TRY(hipMalloc(&d_wt, want));
    wt_bytes = want;
  }
  if (host_bytes > h_wt_bytes) {
    if (h_wt) (void)hipHostFree(h_wt);
    h_wt = nullptr; h_wt_bytes = 0;
    const u64 want = align_up(host_bytes + host_bytes / 8, 1u << 20);
    BWTC_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_wt), want, hipHostMallocDefault));
    h_wt_bytes = want;
  }
  return 0;
}

int BwtEngine::reserve_stats(u64 bytes) {
  if (bytes <= h_stats_bytes) return 0;
  if (h_stats) (void)hipHostFree(h_stats);
  h_stats = nullptr; h_stats_bytes = 0;
  const u64 want = align_up(bytes + bytes / 8, 1u << 16);
  BWTC_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_stats), want, hipHostMallocDefault));
  h_stats_bytes = want;
  return 0;
}

// Runs of the block are in d_run_start / d_run_sym as wavelet_section_stats_device left them.  On success coded_pos has plan.group_type.size() + 1 entries and `codes` holds the
// packed elements.
int wavelet_streams_device(BwtEngine& e, const u32* d_run_start, const u8* d_run_sym, u32 n_runs, const std::vector<u32>& first_run,
                           const bwtc::wavelet::StreamPlan& plan, std::vector<u32>* coded_pos,
                           PinnedBytes* codes, bool async_copy, const u32** d_packed) {
  BwtEngine::ScanScope scan_scope(e);
  hipStream_t st = e.stream;
  const auto t_entry = std::chrono::steady_clock::now();
  const u32 nsec = (u32)plan.sections.size();
  const u32 n_groups = (u32)plan.group_type.size();
  if (nsec == 0 || nsec > 256 || first_run.size() != nsec + 1 || n_runs == 0) return -1;
  BWTC_HIP_TRY(e.codes_wait());                       // the previous block's packed streams have left the device
  BWTC_HIP_TRY(hipSetDevice(e.device));
  // ... and its model passes (on their own stream, beside this block's transform) have read them: no host wait, the
  // stream kernels below queue up behind the event
  if (e.gm_done_pending) { BWTC_HIP_TRY(hipStreamWaitEvent(st, e.ev_gm_done, 0)); e.gm_done_pending = false; }
  const bool dense_ids = !plan.id_group.empty();
  // segmented layout (k_wt_segments): when every section's dense ids fit two digits
  bool segmented = !dense_ids && e.wt_segmented;
  for (u32 s = 0; s < nsec && segmented; ++s) {
    const bwtc::wavelet::StreamPlan::Section& sec = plan.sections[s];
    const u32 groups = (s + 1 < nsec ? plan.sections[s + 1].group_base : n_groups) - sec.group_base;
    if (groups < sec.n_nodes || sec.n_leaves > 256u ||
        sec.n_nodes + (u64)(groups - sec.n_nodes) * std::max(1u, sec.n_leaves) > 65536ull) segmented = false;
    if (std::getenv("BWTC_HIP_DEBUG_SEG"))
      std::fprintf(stderr, "section %u: %u groups, %u symbol nodes, %u leaves -> %llu ids\n", s, groups, sec.n_nodes, sec.n_leaves,
                   (unsigned long long)(sec.n_nodes + (u64)(groups - sec.n_nodes) * std::max(1u, sec.n_leaves)));
  }
  const u64 cap = plan.max_elements + (segmented ? (u64)nsec * radix_tile<u32>() : 0ull);
  if (cap + kWtTile >= (1ull << 32)) return -5;             // step indices are 32 bit
  const int key_bits = dense_ids ? (int)kStepLeafShift + bit_width_u64(plan.id_group.size() - 1)
                                 : (int)kStepGroupShift + bit_width_u64(n_groups ? n_groups - 1 : 0);

  // workspace layout
  u64 at = 0;
  auto take = [&](u64 bytes) { const u64 o = at; at = align_up(at + bytes, 256); return o; };
  const u64 o_first = take((nsec + 1) * 4ull);
  const u64 o_symtab = take(plan.symtab.size() * 4ull);
  const u64 o_symcode = take(plan.symcode.size() * 8ull);
  const u64 o_lendense = take(plan.lendense.size() * 4ull);
  const u64 o_overfirst = take(plan.over_first.size() * 4ull);
  const u64 o_over = take(plan.over.size() * 4ull + 16);
  const u64 o_pool = take(plan.pool.size() * 4ull + 16);
  const u64 o_gtype = take(n_groups + 16);
  const u64 o_idgroup = take(plan.id_group.size() * 4ull + 16);
  const u64 o_segtab = take(nsec * 16ull), o_segmagic = take(nsec * 4ull);
  const u64 tables_end = at;
  const u64 o_pstart = take((nsec + 1) * 4ull), o_tilefirst = take((nsec + 1) * 4ull), o_totals = take(16);
  const u64 o_gstart = take((n_groups + 1) * 4ull);
  const u64 o_gpos = take((n_groups + 1) * 4ull);
  const u64 o_cnt = take(((u64)n_runs + 1) * 4);
  const u64 o_cnt_partial = take(((u64)ceil_div((u64)n_runs + 1, kScanTile) + 1) * 4);
  const u64 o_k0 = take(cap * 4 + 64), o_k1 = take(cap * 4 + 64);
  const u64 table_words = (u64)ceil_div(cap, radix_tile<u32>()) * kRadixBins + kRadixBins;
  const u64 o_table = take(table_words * 4);
  const u64 o_partial = take(((u64)ceil_div(table_words, kScanTile) + 1) * 4);
  const u64 o_code = take(cap + 64);
  const u64 o_compact = take(cap + 64);
  const u64 o_packed = take(cap / 4 + 64);
  const u64 n_tiles_cap = (u64)ceil_div(cap, kWtTile) + 1;
  const u64 o_tile = take(n_tiles_cap * 4);
  const u64 o_tile_partial = take(((u64)ceil_div(n_tiles_cap, kScanTile) + 1) * 4);
  const u64 h_gpos = align_up(tables_end, 256);            // the groups' coded positions come down into the pinned buffer too
  int rc = e.reserve_wavelet(at, h_gpos + (n_groups + 1) * 4ull);
  if (rc) return rc;
  u8* base = static_cast<u8*>(e.d_wt);
  auto ptr32 = [&](u64 o) { return reinterpret_cast<u32*>(base + o); };

  // tables: staged in the pinned buffer, one copy
  {
    u8* h = e.h_wt;
    std::memcpy(h + o_first, first_run.data(), (nsec + 1) * 4ull);
    std::memcpy(h + o_symtab, plan.symtab.data(), plan.symtab.size() * 4ull);
    std::memcpy(h + o_symcode, plan.symcode.data(), plan.symcode.size() * 8ull);
    std::memcpy(h + o_lendense, plan.lendense.data(), plan.lendense.size() * 4ull);
    std::memcpy(h + o_overfirst, plan.over_first.data(), plan.over_first.size() * 4ull);
    if (!plan.over.empty()) std::memcpy(h + o_over, plan.over.data(), plan.over.size() * 4ull);
    if (!plan.pool.empty()) std::memcpy(h + o_pool, plan.pool.data(), plan.pool.size() * 4ull);
    std::memcpy(h + o_gtype, plan.group_type.data(), n_groups);
    if (dense_ids) std::memcpy(h + o_idgroup, plan.id_group.data(), plan.id_group.size() * 4ull);
    for (u32 s2 = 0; s2 < nsec; ++s2) {
      const u32 leaves = std::max(1u, plan.sections[s2].n_leaves);
      u32* row = reinterpret_cast<u32*>(h + o_segtab) + 4 * s2;
      row[0] = 0; row[1] = plan.sections[s2].group_base; row[2] = plan.sections[s2].n_nodes; row[3] = leaves;
      reinterpret_cast<u32*>(h + o_segmagic)[s2] = leaves > 1 ? (u32)((1ull << 32) / leaves + 1) : 0u;    // x / leaves = umulhi(x, this) for x < 65 536
    }
    BWTC_HIP_TRY(hipMemcpyAsync(base, h, tables_end, hipMemcpyHostToDevice, st));
  }
  WtTables t;
  t.first_run = ptr32(o_first);
  t.nsec = nsec;
  t.symtab = reinterpret_cast<const uint2*>(base + o_symtab);
  t.symcode = reinterpret_cast<const u64*>(base + o_symcode);
  t.lendense = reinterpret_cast<const uint2*>(base + o_lendense);
  t.over_first = ptr32(o_overfirst);
  t.over = reinterpret_cast<const uint4*>(base + o_over);
  t.pool = ptr32(o_pool);
  const u8* d_gtype = base + o_gtype;
  const u32* d_idgroup = dense_ids ? ptr32(o_idgroup) : nullptr;

  if (std::getenv("BWTC_HIP_DEBUG"))
    std::fprintf(stderr, "streams: %u sections, %u groups, step sort %s\n", nsec, n_groups,
                 segmented ? "segmented (two passes over the sections' own ids)" : dense_ids ? "over dense ids" : "over block-wide group numbers");
  if (std::getenv("BWTC_HIP_DEBUG"))
    std::fprintf(stderr, "streams: tables of %llu bytes staged and queued %.2f ms after entry\n", (unsigned long long)tables_end,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count());
  // expand
  u32* d_cnt = ptr32(o_cnt);
  hipLaunchKernelGGL(k_wt_count, dim3(ceil_div((u64)n_runs + 1, kWtTPB * kWtCountE)), dim3(kWtTPB), 0, st,
                     d_run_start, d_run_sym, n_runs, t, d_cnt);
  exclusive_scan_u32(d_cnt, (u64)n_runs + 1, ptr32(o_cnt_partial), st);
  if (segmented) {
    hipLaunchKernelGGL(k_wt_segments, dim3(1), dim3(64), 0, st, t.first_run, nsec, (const u32*)d_cnt, (u32)radix_tile<u32>(), ptr32(o_pstart),
                       ptr32(o_tilefirst), reinterpret_cast<uint4*>(base + o_segtab), ptr32(o_totals));
    BWTC_HIP_TRY(hipMemcpyAsync(e.h_small, ptr32(o_totals), 8, hipMemcpyDeviceToHost, st));
  } else {
    BWTC_HIP_TRY(hipMemcpyAsync(e.h_small, d_cnt + n_runs, 4, hipMemcpyDeviceToHost, st));
  }
  BWTC_HIP_TRY(e.wait());
  const u32 n_steps = e.h_small[0];
  if (n_steps == 0 || (u64)n_steps > plan.max_elements) return -3;
  const u32 n = segmented ? e.h_small[1] : n_steps;        // the length of the row of steps, padding included
  if ((u64)n > cap || n < n_steps || (segmented && n % radix_tile<u32>())) return -3;
  hipLaunchKernelGGL(k_wt_expand, dim3(ceil_div(n_runs, kWtExpandRuns)), dim3(kWtTPB), 0, st, d_run_start,
                     d_run_sym, n_runs, t, d_cnt, ptr32(o_k0), base + o_code, segmented ? reinterpret_cast<const uint4*>(base + o_segtab) : (const uint4*)nullptr);
  // sort by (group, leaf rank), keys only; the bit and the gap flag ride along in the word
  u32* ks = nullptr; u32* vs = nullptr;
  if (segmented) {
    hipLaunchKernelGGL(k_wt_pad_fill, dim3(nsec), dim3(kWtTPB), 0, st, t.first_run, (const u32*)d_cnt, (const u32*)ptr32(o_pstart), ptr32(o_k0),
                       base + o_code);
    radix_sort_keys_segmented(ptr32(o_k0), ptr32(o_k1), n, (int)kStepLeafShift, ptr32(o_tilefirst), nsec,
                              ptr32(o_table), ptr32(o_partial), st, &ks, base + o_code, base + o_compact);
  } else
  radix_sort_pairs<u32, u32>(ptr32(o_k0), ptr32(o_k1), (u32*)nullptr, (u32*)nullptr, n, key_bits,
                        ptr32(o_table), ptr32(o_partial), st, &ks, &vs, nullptr, (int)kStepLeafShift, false, true,
                        0, base + o_code, base + o_compact, true);     // planes: both byte arrays are free until the select kernel
  // select + compact + pack
  u32* d_gstart = ptr32(o_gstart);
  BWTC_HIP_TRY(hipMemsetAsync(d_gstart, 0xFF, (n_groups + 1) * 4ull, st));
  const u32 tiles = ceil_div(n, kWtTile);
  u8* d_code = base + o_code;
  u32* d_tile = ptr32(o_tile);
  const u32* d_tf = ptr32(o_tilefirst);
  const uint4* d_segtab = reinterpret_cast<const uint4*>(base + o_segtab);
  const u32* d_segmagic = ptr32(o_segmagic);
  if (segmented) hipLaunchKernelGGL((k_wt_select<false, true>), dim3(tiles), dim3(kWtTPB), 0, st, ks, n, d_gtype, d_idgroup, d_code, d_tile, d_gstart, d_tf, nsec, d_segtab, d_segmagic);
  else if (dense_ids) hipLaunchKernelGGL((k_wt_select<true, false>), dim3(tiles), dim3(kWtTPB), 0, st, ks, n, d_gtype, d_idgroup, d_code, d_tile, d_gstart, d_tf, nsec, d_segtab, d_segmagic);
  else hipLaunchKernelGGL((k_wt_select<false, false>), dim3(tiles), dim3(kWtTPB), 0, st, ks, n, d_gtype, d_idgroup, d_code, d_tile, d_gstart, d_tf, nsec, d_segtab, d_segmagic);
  BWTC_HIP_TRY(hipMemsetAsync(d_tile + tiles, 0, 4, st));
  exclusive_scan_u32(d_tile, (u64)tiles + 1, ptr32(o_tile_partial), st);
  hipLaunchKernelGGL(k_wt_group_pos, dim3(ceil_div(n_groups, kWtTPB / kWave)), dim3(kWtTPB), 0, st, d_gstart,
                     n_groups, n, d_code, d_tile, ptr32(o_gpos));
  BWTC_HIP_TRY(hipMemsetAsync(base + o_packed, 0, ((u64)ceil_div(n, 16) + 1) * 4ull, st));
  hipLaunchKernelGGL(k_wt_compact_pack, dim3(tiles), dim3(kWtTPB), 0, st, d_code, n, d_tile, ptr32(o_packed));
  BWTC_HIP_TRY(hipMemcpyAsync(e.h_small, d_tile + tiles, 4, hipMemcpyDeviceToHost, st));
  e.h_small[1] = 0;
  if (e.scan_chain.err) BWTC_HIP_TRY(hipMemcpyAsync(e.h_small + 1, e.scan_chain.err, 4, hipMemcpyDeviceToHost, st));   // a timed-out scan
  BWTC_HIP_TRY(hipMemcpyAsync(e.h_wt + h_gpos, ptr32(o_gpos), n_groups * 4ull, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(e.wait());
  coded_pos->assign((size_t)n_groups + 1, 0);
  std::memcpy(coded_pos->data(), e.h_wt + h_gpos, n_groups * 4ull);
  const u32 n_coded = e.h_small[0];
  if (n_coded > n) return -3;
  if (e.h_small[1]) { std::fprintf(stderr, "bwtc_hip: a chained scan timed out\n"); return -3; }
  const u32 words = ceil_div(n_coded, 16);
  if (!codes->reserve(words * 4ull + 16)) return -2;
  if (words) {
    if (async_copy) {
      // the packed streams go to the host on a stream of their own, under whatever the compute
      // stream does next (the next block's transform); e.codes_wait() before anybody reads them,
      // and before the next block's stream kernels reuse the packed buffer
      BWTC_HIP_TRY(e.ensure_d2h_stream());
      BWTC_HIP_TRY(hipEventRecord(e.ev_packed, st));
      BWTC_HIP_TRY(hipStreamWaitEvent(e.d2h_stream, e.ev_packed, 0));
      BWTC_HIP_TRY(hipMemcpyAsync(codes->data(), base + o_packed, words * 4ull, hipMemcpyDeviceToHost, e.d2h_stream));
      BWTC_HIP_TRY(hipEventRecord(e.ev_codes, e.d2h_stream));
      e.codes_in_flight = true;
    } else {
      BWTC_HIP_TRY(hipMemcpyAsync(codes->data(), base + o_packed, words * 4ull, hipMemcpyDeviceToHost, st));
    }
  }
  if (!async_copy) BWTC_HIP_TRY(e.wait());
  BWTC_HIP_TRY(hipGetLastError());
  (*coded_pos)[n_groups] = n_coded;
  for (u32 g = n_groups; g-- > 0;)
    if ((*coded_pos)[g] == 0xFFFFFFFFu) (*coded_pos)[g] = (*coded_pos)[g + 1];
  e.wt_elements = n_steps;
  e.wt_coded = n_coded;
  if (d_packed) *d_packed = ptr32(o_packed);
  return 0;
}

}  // namespace bwtc_hip
