// Device-wide exclusive prefix sum over u32 (in place).  Two forms: three launches and no
// spin-waits (tile reduce -> one-workgroup scan of the tile sums -> tile scan + offset), and ONE
// launch of chained tiles with decoupled look-back (below), used when the calling thread's context
// has set up a ScanChain.
#pragma once
#include "common.hpp"

namespace bwtc_hip {

constexpr int kScanTPB = 256;
constexpr int kScanE = 16;
constexpr int kScanTile = kScanTPB * kScanE;  // 4096 items per workgroup

static __global__ __launch_bounds__(kScanTPB) void k_scan_reduce(const u32* __restrict__ in,
                                                          u32* __restrict__ partial, u64 n) {
  __shared__ u32 scratch[kScanTPB / kWave + 1];
  const u64 base = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x * kScanE;
  u32 s = 0;
  if (base + kScanE <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
    for (int i = 0; i < kScanE / 4; ++i) { uint4 v = p[i]; s += v.x + v.y + v.z + v.w; }
  } else {
    for (int i = 0; i < kScanE; ++i) if (base + i < n) s += in[base + i];
  }
  u32 total;
  block_scan_excl_add<kScanTPB>(s, scratch, &total);
  if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// Exclusive scan of `count` values by ONE workgroup of 1024 threads (count up to a few
// million is fine).  Optionally stores the grand total.
static __global__ __launch_bounds__(1024) void k_scan_single(u32* __restrict__ data, u32 count,
                                                      u32* __restrict__ total_out) {
  __shared__ u32 scratch[1024 / kWave + 1];
  const u32 per = (count + 1023) / 1024;
  const u32 b = threadIdx.x * per;
  const u32 e = min(b + per, count);
  u32 s = 0;
  for (u32 i = b; i < e; ++i) s += data[i];
  u32 total;
  u32 off = block_scan_excl_add<1024>(s, scratch, &total);
  for (u32 i = b; i < e; ++i) { u32 v = data[i]; data[i] = off; off += v; }
  if (total_out && threadIdx.x == 0) *total_out = total;
}

// RAW: `partial` holds the tiles' totals as k_scan_reduce left them, and every workgroup adds up the
// ones before its own (at most kScanRawTiles words from the L2) instead of a one-workgroup scan between
// the two launches: the radix sorter's offset tables are 2048 tiles, scanned 33 times per block, and
// that launch was 6 us of one workgroup with the GPU idle around it.
constexpr u32 kScanRawTiles = 4096;
template <bool RAW>
static __global__ __launch_bounds__(kScanTPB) void k_scan_apply(u32* __restrict__ data,
                                                         const u32* __restrict__ partial, u64 n) {
  __shared__ u32 scratch[kScanTPB / kWave + 1];
  u32 before = 0;
  if (RAW) {
    u32 acc = 0;
    for (u32 i = threadIdx.x; i < blockIdx.x; i += kScanTPB) acc += partial[i];
    block_scan_excl_add<kScanTPB>(acc, scratch, &before);          // every thread gets the total
  } else {
    before = partial[blockIdx.x];
  }
  const u64 base = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x * kScanE;
  u32 v[kScanE];
  const bool full = base + kScanE <= n;
  if (full) {
    const uint4* p = reinterpret_cast<const uint4*>(data + base);
#pragma unroll
    for (int i = 0; i < kScanE / 4; ++i) {
      uint4 q = p[i];
      v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < kScanE; ++i) v[i] = (base + i < n) ? data[base + i] : 0u;
  }
  u32 s = 0;
#pragma unroll
  for (int i = 0; i < kScanE; ++i) s += v[i];
  u32 total;
  u32 off = block_scan_excl_add<kScanTPB>(s, scratch, &total) + before;
#pragma unroll
  for (int i = 0; i < kScanE; ++i) { u32 t = v[i]; v[i] = off; off += t; }
  if (full) {
    uint4* p = reinterpret_cast<uint4*>(data + base);
#pragma unroll
    for (int i = 0; i < kScanE / 4; ++i)
      p[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < kScanE; ++i) if (base + i < n) data[base + i] = v[i];
  }
}

// ---- one launch: chained tiles with decoupled look-back -------------------------------------------
// A suffix sort makes some hundred scans per block (one per radix pass, over the tile histograms),
// each small: three launches cost more in launch boundaries and in reading the table twice than
// the scan itself.  Here a tile's workgroup takes a ticket (so every tile before it has started:
// waiting for them cannot deadlock), publishes its sum, looks back over the tiles before it --
// a wave at a time, 64 status words per look -- until it meets a tile whose inclusive prefix is
// known, and publishes its own.  A status word carries its value: {epoch, flag, sum} in 64 bits,
// written and read with relaxed agent-scope atomics, nothing else to order.  The epoch (one per
// scan) makes stale words of earlier scans read as "not there yet", so nothing is cleared between
// scans, and the ticket counter runs on (the host knows where each scan's tickets start).  Spins
// are bounded: a tile that waits absurdly long sets the error word and gives up (the caller reads
// the word back with its results).
struct ScanChain {
  unsigned long long* status = nullptr;   // one word per tile
  unsigned long long* ticket = nullptr;   // running ticket counter
  u32* err = nullptr;                     // device word, set to 1 on a timed-out wait
  u32 cap_tiles = 0;
  unsigned long long issued = 0;          // host: tickets handed to earlier scans
  u32 epoch = 0;                          // host: this many scans so far
};
// the chain of the context the calling thread is working for (null: three-launch scans)
inline ScanChain*& current_scan_chain() { static thread_local ScanChain* c = nullptr; return c; }

constexpr unsigned long long kScanFlagSum = 1ull << 32, kScanFlagPrefix = 2ull << 32;

static __global__ __launch_bounds__(kScanTPB) void k_scan_chained(u32* __restrict__ data, u64 n, unsigned long long* __restrict__ status,
                                                           unsigned long long* __restrict__ ticket, unsigned long long ticket_base,
                                                           u32 epoch, u32* __restrict__ err) {
  __shared__ u32 scratch[kScanTPB / kWave + 1];
  __shared__ u32 s_tile, s_prefix;
  if (threadIdx.x == 0) s_tile = (u32)(atomicAdd(ticket, 1ull) - ticket_base);
  __syncthreads();
  const u32 tile = s_tile;
  const u64 base = (u64)tile * kScanTile + (u64)threadIdx.x * kScanE;
  u32 v[kScanE];
  const bool full = base + kScanE <= n;
  if (full) {
    const uint4* p = reinterpret_cast<const uint4*>(data + base);
#pragma unroll
    for (int i = 0; i < kScanE / 4; ++i) {
      uint4 q = p[i];
      v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < kScanE; ++i) v[i] = (base + i < n) ? data[base + i] : 0u;
  }
  u32 s = 0;
#pragma unroll
  for (int i = 0; i < kScanE; ++i) s += v[i];
  u32 total;
  u32 off = block_scan_excl_add<kScanTPB>(s, scratch, &total);
  const unsigned long long tag = (unsigned long long)epoch << 34;
  if (threadIdx.x < kWave) {                          // the first wave publishes and looks back
    const u32 lane = threadIdx.x;
    u32 prefix = 0;
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(&status[0], tag | kScanFlagPrefix | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&status[tile], tag | kScanFlagSum | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int hi = (int)tile - 1;                         // nearest tile not yet taken in
      u32 spins = 0;
      bool done = false;
      while (!done) {
        const int j = hi - (int)lane;
        unsigned long long w = 0;
        if (j >= 0) w = __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool there = j >= 0 && (w >> 34) == epoch && (w & (kScanFlagSum | kScanFlagPrefix)) != 0;
        const bool incl = there && (w & kScanFlagPrefix) != 0;
        const u64 miss = __ballot(j >= 0 && !there);
        const u64 inc = __ballot(incl);
        // lanes before the first missing one are usable; among them, up to the first inclusive one
        const u32 usable = miss ? (u32)__builtin_ctzll(miss) : (u32)kWave;
        const u32 first_incl = inc ? (u32)__builtin_ctzll(inc) : (u32)kWave;
        u32 take = 0;
        if (first_incl < usable) { take = first_incl + 1; done = true; }
        else { take = usable; if (hi - (int)usable < 0 && miss == 0) done = true; }   // ran past tile 0 (cannot: tile 0 is inclusive)
        u32 part = (lane < take && j >= 0) ? (u32)w : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, kWave);
        prefix += part;
        hi -= (int)take;
        if (!done && take == 0 && ++spins > (1u << 22)) { if (lane == 0) atomicExch(err, 1u); done = true; }
        if (!done && take == 0) __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) __hip_atomic_store(&status[tile], tag | kScanFlagPrefix | (unsigned long long)(u32)(prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_prefix = prefix;
  }
  __syncthreads();
  off += s_prefix;
#pragma unroll
  for (int i = 0; i < kScanE; ++i) { u32 t = v[i]; v[i] = off; off += t; }
  if (full) {
    uint4* p = reinterpret_cast<uint4*>(data + base);
#pragma unroll
    for (int i = 0; i < kScanE / 4; ++i)
      p[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < kScanE; ++i) if (base + i < n) data[base + i] = v[i];
  }
}

// partial must hold ceil(n / kScanTile) words.  n up to 2^32-1 items whose total fits u32.
static inline void exclusive_scan_u32(u32* data, u64 n, u32* partial, hipStream_t st) {
  if (n == 0) return;
  const u32 tiles = ceil_div(n, kScanTile);
  if (tiles == 1) {
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, data, (u32)n, (u32*)nullptr);
    return;
  }
  ScanChain* ch = current_scan_chain();
  if (ch && ch->status && tiles <= ch->cap_tiles) {
    ch->epoch = (ch->epoch + 1u) & ((1u << 30) - 1u);
    if (ch->epoch == 0) ch->epoch = 1;                // status words start zeroed: epoch 0 is never used
    hipLaunchKernelGGL(k_scan_chained, dim3(tiles), dim3(kScanTPB), 0, st, data, n, ch->status, ch->ticket, ch->issued, ch->epoch, ch->err);
    ch->issued += tiles;
    return;
  }
  hipLaunchKernelGGL(k_scan_reduce, dim3(tiles), dim3(kScanTPB), 0, st, data, partial, n);
  if (tiles <= kScanRawTiles) {
    hipLaunchKernelGGL(k_scan_apply<true>, dim3(tiles), dim3(kScanTPB), 0, st, data, partial, n);
    return;
  }
  hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, partial, tiles, (u32*)nullptr);
  hipLaunchKernelGGL(k_scan_apply<false>, dim3(tiles), dim3(kScanTPB), 0, st, data, partial, n);
}

}  // namespace bwtc_hip
