// Device half of the 'H' coder (HuffmanEncoder::encodeData, HuffmanCoders.cpp:119-257):
// everything that is a pass over the transformed block.
//
//   runs +    run heads of every section (utils::calculateRunFrequenciesAndStoreRuns,
//   stats     Utils.cpp:150-170; a run never crosses a section start) -> run_start[], run_sym[],
//             and in the same sweep, per section: runs per symbol (the Huffman weights), total
//             gamma bits ('H'), run-length distribution ('B')
//   pack      for each of the two bit streams of a section (Huffman codes of the run symbols,
//             HuffmanCoders.cpp:200-226; gamma codes of the run lengths, :229-251): prefix sum
//             of the code lengths -> absolute bit position -> MSB-first bits OR-ed into the
//             output
// The small-table work in between (code lengths, canonical codes, shape, headers) is host
// code in entropy_host.cpp.  The run scanner and the per-section run statistics are shared
// with the 'B' coder (wavelet_section_stats_device below; the rest of 'B' is in
// wavelet_tree.hip / wavelet_encoder.hip).
#include "bwt_engine.hpp"
#include "entropy_host.hpp"
#include "wavelet_host.hpp"
#include "scan.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <chrono>
#include <vector>

namespace bwtc_hip {

constexpr int kRunTPB = 256;
constexpr int kRunE = 16;
constexpr int kRunTile = kRunTPB * kRunE;      // bytes per workgroup

// head bits of the 16 bytes starting at p0 (bit e = byte p0+e starts a run)
// a thread's sixteen bytes and the byte before them, loaded ahead of their use
struct RunBytes { uint4 v; u32 prev; bool fast; };
__device__ __forceinline__ RunBytes run_bytes_fetch(const u8* __restrict__ bwt, u32 size, u32 p0, bool aligned) {
  RunBytes rb;
  rb.v = make_uint4(0, 0, 0, 0); rb.prev = 0;
  rb.fast = aligned && p0 < size && p0 + kRunE <= size;
  if (rb.fast) { rb.v = *reinterpret_cast<const uint4*>(bwt + p0); rb.prev = p0 ? bwt[p0 - 1] : 0u; }
  return rb;
}

// pre: the bytes if they were fetched ahead (and could be: whole aligned rows).  plain: the caller
// knows that no section starts inside the tile, plain_first is then the first section after it.
__device__ __forceinline__ u32 run_head_bits(const u8* __restrict__ bwt, u32 size, u32 p0,
                                             const u32* __restrict__ s_sec, u32 nsec,
                                             u32* first_sec, u8 (&b)[kRunE], bool aligned,
                                             const RunBytes* pre = nullptr, bool plain = false, u32 plain_first = 0) {
  u32 heads = 0;
  *first_sec = nsec;
  if (p0 >= size) return 0;
  u8 prev;
  if (pre && pre->fast) {
    prev = (u8)pre->prev;
    const u32 w[4] = {pre->v.x, pre->v.y, pre->v.z, pre->v.w};
#pragma unroll
    for (int e = 0; e < kRunE; ++e) b[e] = (u8)(w[e >> 2] >> (8 * (e & 3)));
  } else {
    prev = p0 ? bwt[p0 - 1] : 0;
    if (aligned && p0 + kRunE <= size) {
      const uint4 v = *reinterpret_cast<const uint4*>(bwt + p0);
      const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < kRunE; ++e) b[e] = (u8)(w[e >> 2] >> (8 * (e & 3)));
    } else {
#pragma unroll
      for (int e = 0; e < kRunE; ++e) b[e] = (p0 + e < size) ? bwt[p0 + e] : 0;
    }
  }
#pragma unroll
  for (int e = 0; e < kRunE; ++e) {
    const u32 p = p0 + e;
    if (p < size && (p == 0 || b[e] != prev)) heads |= 1u << e;
    prev = b[e];
  }
  if (plain) { *first_sec = plain_first; return heads; }
  // section starts inside [p0, p0 + 16): binary search for the first one >= p0
  u32 lo = 0, hi = nsec;
  while (lo < hi) { u32 mid = (lo + hi) >> 1; if (s_sec[mid] < p0) lo = mid + 1; else hi = mid; }
  *first_sec = lo;
  for (u32 j = lo; j < nsec && s_sec[j] < p0 + kRunE && s_sec[j] < size; ++j)
    heads |= 1u << (s_sec[j] - p0);
  return heads;
}

__global__ __launch_bounds__(kRunTPB) void k_runs_count(const u8* __restrict__ bwt, u32 size,
                                                        const u32* __restrict__ sec_start,
                                                        u32 nsec, u32* __restrict__ tile_cnt, int aligned) {
  __shared__ u32 s_sec[256];
  __shared__ u32 scr[kRunTPB / kWave + 1];
  if (threadIdx.x < nsec) s_sec[threadIdx.x] = sec_start[threadIdx.x];
  __syncthreads();
  const u32 p0 = blockIdx.x * kRunTile + threadIdx.x * kRunE;
  u8 b[kRunE];
  u32 fs;
  const u32 heads = run_head_bits(bwt, size, p0, s_sec, nsec, &fs, b, aligned != 0);
  u32 total;
  block_scan_excl_add<kRunTPB>(__popc(heads), scr, &total);
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

// section of run r: last s with first_run[s] <= r
__device__ __forceinline__ u32 section_of(const u32* s_first, u32 nsec, u32 r) {
  u32 lo = 0, hi = nsec;           // invariant: first_run[lo] <= r < first_run[hi]
  while (hi - lo > 1) { u32 mid = (lo + hi) >> 1; if (s_first[mid] <= r) lo = mid; else hi = mid; }
  return lo;
}

__device__ __forceinline__ u32 gamma_bits(u32 len) { return 2u * (31u - (u32)__clz(len)) + 1u; }

// ---------------------------------------------------------------------------------------
// Fused front-end pass: run heads -> run_start[] / run_sym[] AND, in the same sweep over the
// transformed block, everything the coders' table builders need from the runs -- runs per symbol
// per section (the Huffman weights of both coders), the gamma bits per section ('H') and the
// run-length distribution per section ('B': dense table below kLenDense, overflow list above).
// A run's length is the distance to the next head: inside the thread's 16 bytes, else the first
// head of a later thread of the tile (LDS), else unknown here -- the tile's LAST run is "open"
// and is settled by k_open_runs (one run per tile) once run_start[] is complete.
// Replaces k_runs_emit + k_run_stats + k_run_length_dist (three sweeps) by one.
// ---------------------------------------------------------------------------------------
constexpr u32 kLenDense = 512;     // the table comes to the host with every block: 4096 entries per section were 1.3 MB of copy and a 0.1 ms loop
constexpr u32 kLenLds = 256;
constexpr u32 kLongDirect = 1u << 20;   // host: run lengths below this are counted in a table
constexpr u32 kNoRun = 0xFFFFFFFFu;

struct RunStatsOut {
  u32* run_freqs;                    // [section][256]
  unsigned long long* gbits;         // [section]   (GAMMA)
  u32* dense;                        // [section][kLenDense]   (DIST)
  u32* over_sec; u32* over_len; u32* over_count; u32 over_cap;
  u32* tile_open;                    // [tile] index of the tile's open run, kNoRun if it has none
};

template <bool GAMMA, bool DIST>
__device__ __forceinline__ void count_run(const RunStatsOut& o, u32 s, u32 sym, u32 len, bool in_s0,
                                          u32* hist, u32* s_len, u32 copy, u32& gsum, u32& n_len1, u32& n_len2) {
  if (in_s0) atomicAdd(&hist[sym * 8u + copy], 1u);
  else atomicAdd(&o.run_freqs[s * 256u + sym], 1u);
  if (GAMMA) {
    const u32 g = gamma_bits(len);
    if (in_s0) gsum += g;
    else atomicAdd(&o.gbits[s], (unsigned long long)g);
  }
  if (DIST) {
    // lengths 1 and 2 (four runs of five in text) are counted in registers: every lane adding to
    // the one LDS word of length 1 is a 64-way conflict per instruction
    if (in_s0 && len == 1u) ++n_len1;
    else if (in_s0 && len == 2u) ++n_len2;
    else if (in_s0 && len < kLenLds) atomicAdd(&s_len[len], 1u);
    else if (len < kLenDense) atomicAdd(&o.dense[(u64)s * kLenDense + len], 1u);
    else {
      const u32 slot = atomicAdd(o.over_count, 1u);
      if (slot < o.over_cap) { o.over_sec[slot] = s; o.over_len[slot] = len; }
    }
  }
}

template <bool GAMMA, bool DIST>
__global__ __launch_bounds__(kRunTPB) void k_runs_emit_stats(
    const u8* __restrict__ bwt, u32 size, const u32* __restrict__ sec_start, u32 nsec,
    const u32* __restrict__ tile_off, u32* __restrict__ run_start, u8* __restrict__ run_sym,
    u32* __restrict__ first_run, u32* __restrict__ n_runs, int aligned, RunStatsOut o, u32 ntiles,
    u32 tiles_per_wg) {
  __shared__ u32 s_sec[256];
  __shared__ u32 scr[kRunTPB / kWave + 1];
  __shared__ u32 s_fh[kRunTPB];               // position of the thread's first head, ~0 if it has none
  __shared__ u32 hist[256 * 8];               // runs per symbol of the section being accumulated, 8 copies
  __shared__ u32 s_len[kLenLds];
  __shared__ u32 s_g0;
  // the tile's runs, gathered here and written out by consecutive lanes (a thread storing its own
  // runs put some thirty partial-line requests on the memory system per instruction)
  __shared__ u32 s_rstart[kRunTile];
  __shared__ u8 s_rsym[kRunTile];
  if (threadIdx.x < nsec) s_sec[threadIdx.x] = sec_start[threadIdx.x];
  for (u32 i = threadIdx.x; i < 256 * 8; i += kRunTPB) hist[i] = 0;
  if (DIST) for (u32 i = threadIdx.x; i < kLenLds; i += kRunTPB) s_len[i] = 0;
  if (threadIdx.x == 0) s_g0 = 0;
  __syncthreads();
  // A workgroup walks a contiguous range of tiles and keeps the LDS statistics of the section its
  // tiles start in; they are flushed when that section changes and at the end (a flush per tile
  // is tens of thousands of global atomics on the same few hundred counters).
  auto flush = [&](u32 s) {                   // all threads; barriers around it are the caller's
    u32 c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { c += hist[threadIdx.x * 8u + k]; hist[threadIdx.x * 8u + k] = 0; }
    if (c) atomicAdd(&o.run_freqs[s * 256u + threadIdx.x], c);
    if (GAMMA && threadIdx.x == 0 && s_g0) { atomicAdd(&o.gbits[s], (unsigned long long)s_g0); s_g0 = 0; }
    if (DIST) for (u32 i = threadIdx.x; i < kLenLds; i += kRunTPB) { const u32 v = s_len[i]; if (v) { atomicAdd(&o.dense[(u64)s * kLenDense + i], v); s_len[i] = 0; } }
  };
  const u32 tile_begin = blockIdx.x * tiles_per_wg;
  const u32 tile_end = min(tile_begin + tiles_per_wg, ntiles);
  u32 acc_s = 0xFFFFFFFFu;                    // section the LDS statistics belong to
  const u32 copy = threadIdx.x & 7u;
  // The bytes of the next tile and its place in run_start[] are fetched before the current tile is
  // worked on, and the tile's section is found with uniform (scalar) loads: per tile the walk had
  // a memory latency, sixteen dependent LDS reads of two per-thread searches and two barriers in a row.
  RunBytes ahead = run_bytes_fetch(bwt, size, tile_begin * kRunTile + threadIdx.x * kRunE, aligned != 0);
  u32 off_ahead = tile_begin < tile_end ? tile_off[tile_begin] : 0u;
  for (u32 tile = tile_begin; tile < tile_end; ++tile) {
    const u32 t0 = tile * kRunTile;
    const u32 p0 = t0 + threadIdx.x * kRunE;
    const RunBytes mine = ahead;
    const u32 my_off = off_ahead;
    if (tile + 1 < tile_end) {
      ahead = run_bytes_fetch(bwt, size, p0 + kRunTile, aligned != 0);
      off_ahead = tile_off[tile + 1];
    }
    // section of the tile's first byte: its runs are counted in LDS
    u32 s0;
    { u32 lo = 0, hi = nsec; while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (sec_start[mid] <= t0) lo = mid; else hi = mid; } s0 = lo; }
    // no section starts inside the tile (nearly always): the threads need not look for one
    const bool plain = sec_start[s0] < t0 && (s0 + 1 >= nsec || sec_start[s0 + 1] >= t0 + (u32)kRunTile);
    __syncthreads();                          // the previous tile's LDS traffic is done
    if (s0 != acc_s) {
      if (acc_s != 0xFFFFFFFFu) { flush(acc_s); __syncthreads(); }
      acc_s = s0;
    }
    u8 b[kRunE];
    u32 sec;                                  // first section start >= p0
    const u32 heads = run_head_bits(bwt, size, p0, s_sec, nsec, &sec, b, aligned != 0, &mine, plain, s0 + 1u);
    s_fh[threadIdx.x] = heads ? p0 + (u32)__builtin_ctz(heads) : 0xFFFFFFFFu;
    u32 total;
    u32 r = block_scan_excl_add<kRunTPB>(__popc(heads), scr, &total) + my_off;
    // (block_scan's barriers also publish s_fh)
    const u32 s0_end = s0 + 1 < nsec ? s_sec[s0 + 1] : 0xFFFFFFFFu;
    u32 cur = sec ? sec - 1u : 0u;            // section of the byte before the next section start
    u32 gsum = 0, n_len1 = 0, n_len2 = 0;
    u32 open_run = kNoRun;
    if (heads) {
      // end of the thread's last run: the first head of a later thread, if the tile has one
      u32 next_after = 0xFFFFFFFFu;
      for (u32 t = threadIdx.x + 1; t < kRunTPB; ++t) { const u32 f = s_fh[t]; if (f != 0xFFFFFFFFu) { next_after = f; break; } }
      u32 rest = heads;
#pragma unroll
      for (int e = 0; e < kRunE; ++e) {
        const u32 p = p0 + e;                 // (heads has no bits at or beyond size)
        if ((heads >> e) & 1u) {
          s_rstart[r - my_off] = p;
          s_rsym[r - my_off] = b[e];
          if (sec < nsec && s_sec[sec] == p) { first_run[sec] = r; cur = sec; ++sec; }
          rest &= rest - 1u;                  // heads after this one
          const u32 nxt = rest ? p0 + (u32)__builtin_ctz(rest) : next_after;
          if (nxt != 0xFFFFFFFFu) count_run<GAMMA, DIST>(o, cur, b[e], nxt - p, p < s0_end && cur == s0, hist, s_len, copy, gsum, n_len1, n_len2);
          else open_run = r;
          ++r;
        }
        if (p == size - 1) { *n_runs = r; run_start[r] = size; first_run[nsec] = r; }
      }
    } else if (p0 < size && p0 + kRunE > size - 1u) {
      *n_runs = r; run_start[r] = size; first_run[nsec] = r;     // the block's last byte is here, its run started earlier
    }
    if (open_run != kNoRun) o.tile_open[tile] = open_run;
    __syncthreads();                          // the tile's runs are in LDS
    for (u32 i = threadIdx.x; i < total; i += kRunTPB) { run_start[my_off + i] = s_rstart[i]; run_sym[my_off + i] = s_rsym[i]; }
    if (GAMMA) {
      gsum = wave_scan_add(gsum, lane_id());
      if (lane_id() == kWave - 1 && gsum) atomicAdd(&s_g0, gsum);
    }
    if (DIST) {
      n_len1 = wave_scan_add(n_len1, lane_id());
      n_len2 = wave_scan_add(n_len2, lane_id());
      if (lane_id() == kWave - 1) { if (n_len1) atomicAdd(&s_len[1], n_len1); if (n_len2) atomicAdd(&s_len[2], n_len2); }
    }
  }
  __syncthreads();
  if (acc_s != 0xFFFFFFFFu) flush(acc_s);
}

// the open run of every tile (its end lies in a later tile), after run_start[] is complete
template <bool GAMMA, bool DIST>
__global__ __launch_bounds__(256) void k_open_runs(const u32* __restrict__ run_start,
                                                   const u8* __restrict__ run_sym,
                                                   const u32* __restrict__ first_run, u32 nsec,
                                                   u32 ntiles, RunStatsOut o) {
  __shared__ u32 s_first[257];                  // the search below: eight dependent reads, from LDS instead of L2
  for (u32 i = threadIdx.x; i <= nsec; i += 256u) s_first[i] = first_run[i];
  __syncthreads();
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= ntiles) return;
  const u32 r = o.tile_open[t];
  if (r == kNoRun) return;
  const u32 len = run_start[r + 1] - run_start[r];
  u32 lo = 0, hi = nsec;
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_first[mid] <= r) lo = mid; else hi = mid; }
  const u32 s = lo, sym = run_sym[r];
  atomicAdd(&o.run_freqs[s * 256u + sym], 1u);
  if (GAMMA) atomicAdd(&o.gbits[s], (unsigned long long)gamma_bits(len));
  if (DIST) {
    if (len < kLenDense) atomicAdd(&o.dense[(u64)s * kLenDense + len], 1u);
    else {
      const u32 slot = atomicAdd(o.over_count, 1u);
      if (slot < o.over_cap) { o.over_sec[slot] = s; o.over_len[slot] = len; }
    }
  }
}

// ---- bit packing ---------------------------------------------------------------------
// STREAM 0: Huffman code of the run symbol (clen/code tables per section)
// STREAM 1: gamma code of the run length (2*floor(log2 len)+1 bits, value = len)
template <int STREAM>
__device__ __forceinline__ u32 item_bits(u32 r, u32 s, const u32* __restrict__ run_start,
                                         const u8* __restrict__ run_sym,
                                         const u8* __restrict__ clen_tab) {
  if (STREAM == 0) return clen_tab[s * 256u + run_sym[r]];
  return gamma_bits(run_start[r + 1] - run_start[r]);
}

constexpr int kPackTPB = 256;
constexpr int kPackE = 8;
constexpr int kPackTile = kPackTPB * kPackE;

template <int STREAM>
__global__ __launch_bounds__(kPackTPB) void k_pack_count(const u32* __restrict__ run_start,
                                                         const u8* __restrict__ run_sym,
                                                         u32 n_runs,
                                                         const u32* __restrict__ first_run,
                                                         u32 nsec, const u8* __restrict__ clen_tab,
                                                         u32* __restrict__ tile_bits) {
  __shared__ u32 s_first[257];
  __shared__ u32 scr[kPackTPB / kWave + 1];
  for (u32 i = threadIdx.x; i <= nsec; i += kPackTPB) s_first[i] = first_run[i];
  __syncthreads();
  const u32 r0 = blockIdx.x * kPackTile + threadIdx.x * kPackE;
  u32 sum = 0;
  if (r0 < n_runs) {
    u32 s = section_of(s_first, nsec, r0);
    for (int e = 0; e < kPackE; ++e) {
      const u32 r = r0 + e;
      if (r >= n_runs) break;
      while (r >= s_first[s + 1]) ++s;
      sum += item_bits<STREAM>(r, s, run_start, run_sym, clen_tab);
    }
  }
  u32 total;
  block_scan_excl_add<kPackTPB>(sum, scr, &total);
  if (threadIdx.x == 0) tile_bits[blockIdx.x] = total;
}

// one workgroup: exclusive u64 prefix of the per-tile bit counts
__global__ __launch_bounds__(1024) void k_scan_tiles_u64(const u32* __restrict__ tile_bits,
                                                         unsigned long long* __restrict__ tile_base,
                                                         u32 ntiles) {
  __shared__ unsigned long long s_part[1024];
  const u32 per = (ntiles + 1023u) / 1024u;
  const u32 b = threadIdx.x * per;
  const u32 e = min(b + per, ntiles);
  unsigned long long s = 0;
  for (u32 i = b; i < e; ++i) s += tile_bits[i];
  s_part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int i = 0; i < 1024; ++i) { unsigned long long v = s_part[i]; s_part[i] = run; run += v; }
  }
  __syncthreads();
  unsigned long long off = s_part[threadIdx.x];
  for (u32 i = b; i < e; ++i) { tile_base[i] = off; off += tile_bits[i]; }
}

// per section: adj[s] = base_bit[s] - P[first_run[s]], P = global exclusive bit prefix.
// One wave per section sums the code lengths of the runs between the start of the tile that
// holds first_run[s] and first_run[s] itself.
template <int STREAM>
__global__ __launch_bounds__(kWave) void k_section_adjust(
    const u32* __restrict__ run_start, const u8* __restrict__ run_sym,
    const u32* __restrict__ first_run, u32 nsec, const u8* __restrict__ clen_tab,
    const unsigned long long* __restrict__ tile_base,
    const unsigned long long* __restrict__ base_bit, unsigned long long* __restrict__ adj) {
  const u32 s = blockIdx.x;
  if (s >= nsec) return;
  const u32 lane = threadIdx.x;
  const u32 r0 = first_run[s];
  const u32 t = r0 / kPackTile;
  u32 sum = 0;
  for (u32 r = t * kPackTile + lane; r < r0; r += kWave) {
    u32 sec = s;                              // runs before r0 belong to an earlier section
    while (r < first_run[sec]) --sec;
    sum += item_bits<STREAM>(r, sec, run_start, run_sym, clen_tab);
  }
  sum = wave_scan_add(sum, lane);
  if (lane == kWave - 1) adj[s] = base_bit[s] - (tile_base[t] + sum);
}

// ORs `nbits` bits of `value` into an MSB-first bit stream at bit position `pos` of the word
// array `words` (bit p is bit 7 - p%8 of byte p/8): work on big-endian 32-bit words and
// byte-swap each contribution into the little-endian store.
__device__ __forceinline__ void or_bits(u32* words, unsigned long long pos,
                                        unsigned long long value, u32 nbits) {
  while (nbits) {
    const u32 o = (u32)(pos & 31ull);
    const u32 take = min(nbits, 32u - o);
    const u32 chunk = (u32)((value >> (nbits - take)) & ((take == 32u) ? 0xFFFFFFFFull : ((1ull << take) - 1ull)));
    const u32 be = chunk << (32u - o - take);
    if (be) atomicOr(&words[pos >> 5], __builtin_bswap32(be));
    pos += take;
    nbits -= take;
  }
}

constexpr u32 kPackWinWords = 4096;   // LDS staging window: 16 KiB of output per tile

// The bits of one tile of runs cover a contiguous stretch of the output (plus, when the tile
// crosses a section start, the header bytes in between).  They are assembled in an LDS window
// with ds_or and flushed with one global OR per non-zero word; only tiles whose stretch does
// not fit the window fall back to one global OR per code.
template <int STREAM>
__global__ __launch_bounds__(kPackTPB) void k_pack_emit(
    const u32* __restrict__ run_start, const u8* __restrict__ run_sym, u32 n_runs,
    const u32* __restrict__ first_run, u32 nsec, const u8* __restrict__ clen_tab,
    const u32* __restrict__ code_tab, const unsigned long long* __restrict__ tile_base,
    const unsigned long long* __restrict__ adj, u32* __restrict__ out32) {
  __shared__ u32 s_first[257];
  __shared__ u32 scr[kPackTPB / kWave + 1];
  __shared__ u32 s_win[kPackWinWords];
  __shared__ unsigned long long s_lohi[2];
  for (u32 i = threadIdx.x; i <= nsec; i += kPackTPB) s_first[i] = first_run[i];
  for (u32 i = threadIdx.x; i < kPackWinWords; i += kPackTPB) s_win[i] = 0;
  __syncthreads();
  const u32 tile_first = blockIdx.x * kPackTile;
  const u32 tile_last = min(tile_first + (u32)kPackTile, n_runs) - 1u;   // last run of the tile
  const u32 r0 = tile_first + threadIdx.x * kPackE;
  u32 nb[kPackE];
  u32 sum = 0, s_begin = 0;
  if (r0 < n_runs) {
    u32 s = section_of(s_first, nsec, r0);
    s_begin = s;
#pragma unroll
    for (int e = 0; e < kPackE; ++e) {
      const u32 r = r0 + e;
      nb[e] = 0;
      if (r < n_runs) {
        while (r >= s_first[s + 1]) ++s;
        nb[e] = item_bits<STREAM>(r, s, run_start, run_sym, clen_tab);
      }
      sum += nb[e];
    }
  }
  u32 total;
  const u32 off = block_scan_excl_add<kPackTPB>(sum, scr, &total);
  unsigned long long P = tile_base[blockIdx.x] + off;
  // first and last bit position of the tile (positions grow with the run index)
  if (r0 < n_runs) {
    if (threadIdx.x == 0) s_lohi[0] = adj[s_begin] + P;
    if (tile_last >= r0 && tile_last < r0 + kPackE) {
      u32 s = s_begin;
      unsigned long long q = P;
      for (u32 r = r0; r < tile_last; ++r) q += nb[r - r0];
      while (tile_last >= s_first[s + 1]) ++s;
      s_lohi[1] = adj[s] + q + nb[tile_last - r0];
    }
  }
  __syncthreads();
  const unsigned long long wbase = s_lohi[0] >> 5;
  const unsigned long long nwords = ((s_lohi[1] + 31ull) >> 5) - wbase;
  const bool staged = nwords <= (unsigned long long)kPackWinWords;
  if (r0 < n_runs) {
    u32 s = s_begin;
#pragma unroll
    for (int e = 0; e < kPackE; ++e) {
      const u32 r = r0 + e;
      if (r >= n_runs) break;
      while (r >= s_first[s + 1]) ++s;
      unsigned long long value;
      if (STREAM == 0) value = code_tab[s * 256u + run_sym[r]];
      else value = run_start[r + 1] - run_start[r];
      const unsigned long long pos = adj[s] + P;
      if (staged) or_bits(s_win, pos - (wbase << 5), value, nb[e]);
      else or_bits(out32, pos, value, nb[e]);
      P += nb[e];
    }
  }
  if (!staged) return;
  __syncthreads();
  for (u32 w = threadIdx.x; w < (u32)nwords; w += kPackTPB) {
    const u32 v = s_win[w];
    if (v) atomicOr(&out32[wbase + w], v);
  }
}

// scatter of the host-built header pieces: piece i = bytes[src_off[i] .. +len[i]) -> out + dst_off[i]
__global__ __launch_bounds__(256) void k_place_pieces(const u8* __restrict__ bytes,
                                                      const unsigned long long* __restrict__ desc,
                                                      u8* __restrict__ out) {
  const unsigned long long dst = desc[3 * blockIdx.x], src = desc[3 * blockIdx.x + 1],
                           len = desc[3 * blockIdx.x + 2];
  for (unsigned long long i = threadIdx.x; i < len; i += 256) out[dst + i] = bytes[src + i];
}

// ---------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------
u64 huffman_compress_bound(u64 size) {
  // worst case per run: 47-bit code + 63-bit gamma is never reached; sections cost < 1 KiB
  // of headers each.  2x the block + headers is a safe and simple bound.
  return size * 2 + 256 * 1024 + 4096;
}

int huffman_encode_device(BwtEngine& e, const u8* d_bwt, u32 size, const u32* lf, u32 n_lf,
                          const u32* freqs, u8* d_out, u64 out_cap, u64* out_bytes) {
  BwtEngine::ScanScope scan_scope(e);
  hipStream_t st = e.stream;
  if (!freqs || !lf || !out_bytes || n_lf == 0 || n_lf > 256) return -1;
  if ((u64)size > e.cap) return -1;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  if (reinterpret_cast<uintptr_t>(d_out) & 3u) return -1;

  // ---- host: block header + sections (HuffmanEncoder::writeBlockHeader, :271-313)
  std::vector<uint8_t> head(6, 0);
  bwtc::writeBWTBlockHeader(lf, n_lf, head);
  const std::vector<uint32_t> sections = bwtc::deduceSections(freqs);
  const u32 nsec = (u32)sections.size();
  head.push_back((uint8_t)(nsec == 256 ? 0 : nsec));
  for (u32 s = 0; s < nsec; ++s) bwtc::utils::packInteger(sections[s], head);
  if (size == 0 || nsec == 0) {
    const u64 len = head.size() - 6;
    for (int i = 0; i < 6; ++i) head[i] = (uint8_t)(len >> (8 * (5 - i)));
    if (head.size() > out_cap) return -1;
    BWTC_HIP_TRY(hipMemcpyAsync(d_out, head.data(), head.size(), hipMemcpyHostToDevice, st));
    BWTC_HIP_TRY(e.wait());
    *out_bytes = head.size();
    return 0;
  }

  // ---- workspace carving (the BWT arrays are dead once the block has been emitted)
  u32* d_run_start = static_cast<u32*>(e.d_R1);                 // size + 1 words
  u8* d_run_sym = static_cast<u8*>(e.d_R2);                     // size bytes
  u32* d_tile = e.d_V0;                                         // per-tile counters / bases
  unsigned long long* d_tile_base = reinterpret_cast<unsigned long long*>(e.d_G0);
  u32* d_sec = reinterpret_cast<u32*>(e.d_ent);                 // small tables (2 MiB region)
  u32* d_sec_start = d_sec;                                     // 256
  u32* d_first_run = d_sec + 512;                               // 257
  u32* d_nruns = d_sec + 1024;                                  // 1
  u32* d_run_freqs = d_sec + 2048;                              // 256*256
  unsigned long long* d_gbits = reinterpret_cast<unsigned long long*>(d_sec + 2048 + 65536);  // 256
  unsigned long long* d_base_bit = d_gbits + 256;               // 256
  unsigned long long* d_adj = d_base_bit + 256;                 // 256
  u8* d_clen = reinterpret_cast<u8*>(d_adj + 256);              // 65536
  u32* d_code = reinterpret_cast<u32*>(d_clen + 65536);         // 65536 words
  unsigned long long* d_desc = reinterpret_cast<unsigned long long*>(d_code + 65536);  // 3*260
  u8* d_pieces = reinterpret_cast<u8*>(d_desc + 3 * 260);       // header bytes
  const u64 pieces_cap = (2u << 20) - (u64)(d_pieces - e.d_ent);

  std::vector<u32> sec_start(nsec);
  { u64 acc = 0; for (u32 s = 0; s < nsec; ++s) { sec_start[s] = (u32)acc; acc += sections[s]; } }
  BWTC_HIP_TRY(hipMemcpyAsync(d_sec_start, sec_start.data(), nsec * 4, hipMemcpyHostToDevice, st));
  BWTC_HIP_TRY(hipMemsetAsync(d_run_freqs, 0, (65536 + 512) * 4, st));   // run_freqs + gbits

  // ---- runs
  const u32 rtiles = ceil_div(size, kRunTile);
  const int aligned = (reinterpret_cast<uintptr_t>(d_bwt) & 15u) == 0;
  hipLaunchKernelGGL(k_runs_count, dim3(rtiles), dim3(kRunTPB), 0, st, d_bwt, size, d_sec_start,
                     nsec, d_tile, aligned);
  exclusive_scan_u32(d_tile, rtiles, e.d_partial, st);
  // run arrays and statistics in one sweep (k_runs_emit_stats), the tiles' open runs after it
  RunStatsOut so;
  so.run_freqs = d_run_freqs; so.gbits = d_gbits; so.dense = nullptr;
  so.over_sec = so.over_len = so.over_count = nullptr; so.over_cap = 0;
  so.tile_open = e.d_V1;
  BWTC_HIP_TRY(hipMemsetAsync(so.tile_open, 0xFF, (size_t)rtiles * 4, st));
  const u32 tiles_per_wg = ceil_div(rtiles, 2048);
  hipLaunchKernelGGL((k_runs_emit_stats<true, false>), dim3(ceil_div(rtiles, tiles_per_wg)), dim3(kRunTPB), 0, st, d_bwt, size, d_sec_start,
                     nsec, d_tile, d_run_start, d_run_sym, d_first_run, d_nruns, aligned, so, rtiles, tiles_per_wg);
  hipLaunchKernelGGL((k_open_runs<true, false>), dim3(ceil_div(rtiles, 256)), dim3(256), 0, st, d_run_start,
                     d_run_sym, d_first_run, nsec, rtiles, so);
  u32 n_runs = 0;
  BWTC_HIP_TRY(hipMemcpyAsync(e.h_small, d_nruns, 4, hipMemcpyDeviceToHost, st));
  std::vector<u32> h_run_freqs((size_t)nsec * 256);
  std::vector<unsigned long long> h_gbits(nsec);
  std::vector<u32> h_first(nsec + 1);
  BWTC_HIP_TRY(hipMemcpyAsync(h_run_freqs.data(), d_run_freqs, (size_t)nsec * 1024, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(hipMemcpyAsync(h_gbits.data(), d_gbits, (size_t)nsec * 8, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(hipMemcpyAsync(h_first.data(), d_first_run, (size_t)(nsec + 1) * 4, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(e.wait());
  n_runs = e.h_small[0];
  if (n_runs == 0 || n_runs > size) return -3;

  // ---- host: per-section tables and layout (encodeData, :133-198)
  std::vector<uint8_t> clen_tab((size_t)nsec * 256);
  std::vector<u32> code_tab((size_t)nsec * 256);
  std::vector<unsigned long long> base_code(nsec), base_gamma(nsec);
  std::vector<uint8_t> pieces;                  // all header bytes, concatenated
  std::vector<unsigned long long> desc;         // (dst, src, len) per piece
  desc.push_back(0); desc.push_back(0); desc.push_back(head.size());
  pieces.insert(pieces.end(), head.begin(), head.end());
  u64 pos = head.size();
  for (u32 s = 0; s < nsec; ++s) {
    uint64_t f64[256];
    for (int c = 0; c < 256; ++c) f64[c] = h_run_freqs[(size_t)s * 256 + c];
    uint8_t* cl = &clen_tab[(size_t)s * 256];
    bwtc::utils::calculateHuffmanLengths(f64, cl);
    bwtc::utils::computeHuffmanCodes(cl, &code_tab[(size_t)s * 256]);
    std::vector<uint8_t> sh;
    bwtc::utils::packInteger(h_first[s + 1] - h_first[s], sh);      // number of runs
    bwtc::serializeShape(cl, sh);
    desc.push_back(pos); desc.push_back(pieces.size()); desc.push_back(sh.size());
    pieces.insert(pieces.end(), sh.begin(), sh.end());
    pos += sh.size();
    u64 code_bits = 0;
    for (int c = 0; c < 256; ++c) code_bits += f64[c] * cl[c];
    base_code[s] = pos * 8;
    pos += (code_bits + 7) / 8;
    base_gamma[s] = pos * 8;
    pos += (h_gbits[s] + 7) / 8;
  }
  const u64 total = pos;
  if (total > out_cap || pieces.size() > pieces_cap) return -1;
  { const u64 len = total - 6; for (int i = 0; i < 6; ++i) pieces[i] = (uint8_t)(len >> (8 * (5 - i))); }

  BWTC_HIP_TRY(hipMemsetAsync(d_out, 0, (total + 3) / 4 * 4 <= out_cap ? (total + 3) / 4 * 4 : total, st));
  BWTC_HIP_TRY(hipMemcpyAsync(d_clen, clen_tab.data(), clen_tab.size(), hipMemcpyHostToDevice, st));
  BWTC_HIP_TRY(hipMemcpyAsync(d_code, code_tab.data(), code_tab.size() * 4, hipMemcpyHostToDevice, st));
  BWTC_HIP_TRY(hipMemcpyAsync(d_desc, desc.data(), desc.size() * 8, hipMemcpyHostToDevice, st));
  BWTC_HIP_TRY(hipMemcpyAsync(d_pieces, pieces.data(), pieces.size(), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_place_pieces, dim3((u32)(desc.size() / 3)), dim3(256), 0, st, d_pieces, d_desc, d_out);

  // ---- the two bit streams
  const u32 ptiles = ceil_div(n_runs, kPackTile);
  u32* out32 = reinterpret_cast<u32*>(d_out);
  BWTC_HIP_TRY(hipMemcpyAsync(d_base_bit, base_code.data(), nsec * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_pack_count<0>, dim3(ptiles), dim3(kPackTPB), 0, st, d_run_start, d_run_sym,
                     n_runs, d_first_run, nsec, d_clen, d_tile);
  hipLaunchKernelGGL(k_scan_tiles_u64, dim3(1), dim3(1024), 0, st, d_tile, d_tile_base, ptiles);
  hipLaunchKernelGGL(k_section_adjust<0>, dim3(nsec), dim3(kWave), 0, st, d_run_start, d_run_sym,
                     d_first_run, nsec, d_clen, d_tile_base, d_base_bit, d_adj);
  hipLaunchKernelGGL(k_pack_emit<0>, dim3(ptiles), dim3(kPackTPB), 0, st, d_run_start, d_run_sym,
                     n_runs, d_first_run, nsec, d_clen, d_code, d_tile_base, d_adj, out32);
  // gamma stream (stream order keeps d_base_bit/d_adj/d_tile reuse safe)
  BWTC_HIP_TRY(hipMemcpyAsync(d_base_bit, base_gamma.data(), nsec * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_pack_count<1>, dim3(ptiles), dim3(kPackTPB), 0, st, d_run_start, d_run_sym,
                     n_runs, d_first_run, nsec, d_clen, d_tile);
  hipLaunchKernelGGL(k_scan_tiles_u64, dim3(1), dim3(1024), 0, st, d_tile, d_tile_base, ptiles);
  hipLaunchKernelGGL(k_section_adjust<1>, dim3(nsec), dim3(kWave), 0, st, d_run_start, d_run_sym,
                     d_first_run, nsec, d_clen, d_tile_base, d_base_bit, d_adj);
  hipLaunchKernelGGL(k_pack_emit<1>, dim3(ptiles), dim3(kPackTPB), 0, st, d_run_start, d_run_sym,
                     n_runs, d_first_run, nsec, d_clen, d_code, d_tile_base, d_adj, out32);
  BWTC_HIP_TRY(e.wait());
  BWTC_HIP_TRY(hipGetLastError());
  *out_bytes = total;
  return 0;
}


// ---------------------------------------------------------------------------------------
// Run scanner of the wavelet coders: utils::calculateRunsAndCharacters (Utils.cpp:128-147),
// the first thing the WaveletTree constructor does with a section (WaveletTree.hpp:294-308).
// Per section: runs per symbol, total runs, and the run-length distribution
// (std::map<runLength, count> in the reference).  Lengths below kLenDense are counted in a
// dense per-section table (LDS-aggregated for the tile's first section), longer ones are
// appended to an overflow list; the host folds both into sorted (length, count) pairs.
// ---------------------------------------------------------------------------------------
int wavelet_section_stats_device(BwtEngine& e, const u8* d_bwt, u32 size, const u32* freqs,
                                 WaveletSectionStats* out, u32* d_run_start_arg, u8* d_run_sym_arg,
                                 const std::function<void()>* while_waiting) {
  BwtEngine::ScanScope scan_scope(e);
  hipStream_t st = e.stream;
  const auto t_in = std::chrono::steady_clock::now();
  if (!freqs || !out) return -1;
  if ((u64)size > e.cap) return -1;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  out->sections = bwtc::deduceSections(freqs);
  const u32 nsec = (u32)out->sections.size();
  out->run_freqs.assign((size_t)nsec * 256, 0);
  out->total_runs.assign(nsec, 0);
  out->dist.assign(nsec, std::vector<std::pair<u32, u32> >());
  if (size == 0 || nsec == 0) { if (while_waiting) (*while_waiting)(); return 0; }

  u32* d_run_start = d_run_start_arg ? d_run_start_arg : static_cast<u32*>(e.d_R1);
  u8* d_run_sym = d_run_sym_arg ? d_run_sym_arg : static_cast<u8*>(e.d_R2);
  u32* d_tile = e.d_V0;
  u32* d_sec = reinterpret_cast<u32*>(e.d_ent);
  u32* d_sec_start = d_sec;
  u32* d_first_run = d_sec + 512;
  u32* d_nruns = d_sec + 1024;
  u32* d_over_count = d_sec + 1025;
  u32* d_run_freqs = d_sec + 2048;
  unsigned long long* d_gbits = reinterpret_cast<unsigned long long*>(d_sec + 2048 + 65536);
  u32* d_dense = e.d_G0;                      // nsec * kLenDense words
  u32* d_over_sec = e.d_G1;
  u32* d_over_len = e.d_GRP;
  const u32 over_cap = (u32)std::min<u64>(e.cap, 1u << 24);
  if ((u64)nsec * kLenDense > e.cap) return -2;    // tiny contexts cannot hold the dense table

  std::vector<u32> sec_start(nsec);
  { u64 acc = 0; for (u32 s = 0; s < nsec; ++s) { sec_start[s] = (u32)acc; acc += out->sections[s]; } }
  BWTC_HIP_TRY(hipMemcpyAsync(d_sec_start, sec_start.data(), nsec * 4, hipMemcpyHostToDevice, st));
  BWTC_HIP_TRY(hipMemsetAsync(d_run_freqs, 0, (65536 + 512) * 4, st));
  BWTC_HIP_TRY(hipMemsetAsync(d_over_count, 0, 4, st));
  BWTC_HIP_TRY(hipMemsetAsync(d_dense, 0, (u64)nsec * kLenDense * 4, st));
  const u32 rtiles = ceil_div(size, kRunTile);
  const int aligned = (reinterpret_cast<uintptr_t>(d_bwt) & 15u) == 0;
  hipLaunchKernelGGL(k_runs_count, dim3(rtiles), dim3(kRunTPB), 0, st, d_bwt, size, d_sec_start,
                     nsec, d_tile, aligned);
  exclusive_scan_u32(d_tile, rtiles, e.d_partial, st);
  RunStatsOut so;
  so.run_freqs = d_run_freqs; so.gbits = d_gbits; so.dense = d_dense;
  so.over_sec = d_over_sec; so.over_len = d_over_len; so.over_count = d_over_count; so.over_cap = over_cap;
  so.tile_open = e.d_V1;
  BWTC_HIP_TRY(hipMemsetAsync(so.tile_open, 0xFF, (size_t)rtiles * 4, st));
  const u32 tiles_per_wg = ceil_div(rtiles, 2048);
  hipLaunchKernelGGL((k_runs_emit_stats<false, true>), dim3(ceil_div(rtiles, tiles_per_wg)), dim3(kRunTPB), 0, st, d_bwt, size, d_sec_start,
                     nsec, d_tile, d_run_start, d_run_sym, d_first_run, d_nruns, aligned, so, rtiles, tiles_per_wg);
  hipLaunchKernelGGL((k_open_runs<false, true>), dim3(ceil_div(rtiles, 256)), dim3(256), 0, st, d_run_start,
                     d_run_sym, d_first_run, nsec, rtiles, so);
  // everything the host needs, into page-locked memory with two copies and one wait (copies into
  // pageable vectors are staged by the runtime one after the other, the GPU idle in between)
  const u64 sec_words = 2048 + 65536;                 // section starts, first runs, counts, runs per symbol
  const u64 dense_words = (u64)nsec * kLenDense;
  // ... and the head of the overflow list with them: how long it is is only known afterwards, and a
  // second round trip for a few thousand long runs cost half a millisecond
  const u32 over_head = std::min<u32>(over_cap, 1u << 16);
  int rcs = e.reserve_stats((sec_words + dense_words + 2ull * over_head) * 4);
  if (rcs) return rcs;
  u32* h_sec = reinterpret_cast<u32*>(e.h_stats);
  u32* h_dense = h_sec + sec_words;
  u32* h_osec = h_dense + dense_words;
  u32* h_olen = h_osec + over_head;
  BWTC_HIP_TRY(hipMemcpyAsync(h_sec, d_sec, sec_words * 4, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(hipMemcpyAsync(h_dense, d_dense, dense_words * 4, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(hipMemcpyAsync(h_osec, d_over_sec, (size_t)over_head * 4, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(hipMemcpyAsync(h_olen, d_over_len, (size_t)over_head * 4, hipMemcpyDeviceToHost, st));
  const auto tq = std::chrono::steady_clock::now();
  if (while_waiting) (*while_waiting)();              // the GPU has a millisecond of scanning to do
  const auto tw0 = std::chrono::steady_clock::now();
  BWTC_HIP_TRY(e.wait());
  const auto tw1 = std::chrono::steady_clock::now();
  const u32 n_over = h_sec[1025];
  std::memcpy(out->run_freqs.data(), h_sec + 2048, (size_t)nsec * 1024);
  std::vector<u32> first(h_sec + 512, h_sec + 512 + nsec + 1);
  const u32* dense = h_dense;
  const u32 n_runs = h_sec[1024];
  if (n_runs == 0 || n_runs > size) return -3;
  if (n_over > over_cap) return -2;
  std::vector<u32> osec(n_over), olen(n_over);
  if (n_over && n_over <= over_head) {
    std::memcpy(osec.data(), h_osec, (size_t)n_over * 4);
    std::memcpy(olen.data(), h_olen, (size_t)n_over * 4);
  } else if (n_over) {
    BWTC_HIP_TRY(hipMemcpyAsync(osec.data(), d_over_sec, (size_t)n_over * 4, hipMemcpyDeviceToHost, st));
    BWTC_HIP_TRY(hipMemcpyAsync(olen.data(), d_over_len, (size_t)n_over * 4, hipMemcpyDeviceToHost, st));
    BWTC_HIP_TRY(e.wait());
  }
  BWTC_HIP_TRY(hipGetLastError());
  out->first_run = first;
  if (n_over && e.long_count.empty()) e.long_count.assign(kLongDirect, 0);
  std::vector<std::vector<u32> > long_runs(nsec);
  for (u32 i = 0; i < n_over; ++i) long_runs[osec[i]].push_back(olen[i]);
  for (u32 s = 0; s < nsec; ++s) {
    out->total_runs[s] = first[s + 1] - first[s];
    std::vector<std::pair<u32, u32> >& d = out->dist[s];
    for (u32 l = 1; l < kLenDense; ++l) {
      const u32 c = dense[(size_t)s * kLenDense + l];
      if (c) d.push_back(std::make_pair(l, c));
    }
    // long runs: counted through a direct table where they fit it, so that only the DISTINCT
    // lengths are sorted (a text block has some ten thousand runs of 512 bytes and more, few
    // hundred different lengths; sorting the runs themselves took half a millisecond per block)
    std::vector<u32>& lr = long_runs[s];
    const size_t first_long = d.size();
    std::vector<u32> rest;
    for (size_t i = 0; i < lr.size(); ++i) {
      if (lr[i] < kLongDirect) { if (e.long_count[lr[i]]++ == 0) d.push_back(std::make_pair(lr[i], 0u)); }
      else rest.push_back(lr[i]);
    }
    for (size_t i = first_long; i < d.size(); ++i) { d[i].second = e.long_count[d[i].first]; e.long_count[d[i].first] = 0; }
    std::sort(rest.begin(), rest.end());
    for (size_t i = 0; i < rest.size();) {
      size_t j = i;
      while (j < rest.size() && rest[j] == rest[i]) ++j;
      d.push_back(std::make_pair(rest[i], (u32)(j - i)));
      i = j;
    }
    std::sort(d.begin() + first_long, d.end());
  }
  if (std::getenv("BWTC_HIP_DEBUG"))
    std::fprintf(stderr, "run statistics: queued in %.2f ms, hook %.2f ms, waited %.2f ms, %u long runs, tables read in %.2f ms\n",
                 std::chrono::duration<double, std::milli>(tq - t_in).count(), std::chrono::duration<double, std::milli>(tw0 - tq).count(),
                 std::chrono::duration<double, std::milli>(tw1 - tw0).count(), n_over,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw1).count());
  return 0;
}


}  // namespace bwtc_hip
