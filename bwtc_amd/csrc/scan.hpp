// Device-wide exclusive prefix sum over u32 (in place), three launches and no spin-waits:
// tile reduce -> one-workgroup scan of the tile sums -> tile scan + offset.
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

static __global__ __launch_bounds__(kScanTPB) void k_scan_apply(u32* __restrict__ data,
                                                         const u32* __restrict__ partial, u64 n) {
  __shared__ u32 scratch[kScanTPB / kWave + 1];
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
  u32 off = block_scan_excl_add<kScanTPB>(s, scratch, &total) + partial[blockIdx.x];
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
  hipLaunchKernelGGL(k_scan_reduce, dim3(tiles), dim3(kScanTPB), 0, st, data, partial, n);
  hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, partial, tiles, (u32*)nullptr);
  hipLaunchKernelGGL(k_scan_apply, dim3(tiles), dim3(kScanTPB), 0, st, data, partial, n);
}

}  // namespace bwtc_hip
