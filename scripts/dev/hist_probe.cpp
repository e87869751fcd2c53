// scratch: what bounds k_radix_hist_plane (one byte per item in, a 256-bin histogram per tile of 4096 out)?
// build on the GPU box: hipcc -O3 --offload-arch=gfx950 -Ibwtc_amd/csrc -Iinclude scripts/dev/hist_probe.cpp -o /tmp/hist_probe
#include "radix_sort.hpp"
#include <cstdio>
#include <cstdint>
#include <vector>
using namespace bwtc_hip;

// B: the kernel as it is, but the table gets one word per workgroup (the floor without its stores)
__global__ __launch_bounds__(kRadixTPB) void k_hist_nostore(const u8* __restrict__ plane, u32* __restrict__ table, u64 n, u32 ntiles) {
  constexpr int HALF = kRadixTPB / 2, B = 16;
  constexpr u32 C = 16;
  __shared__ __attribute__((aligned(16))) u32 hist[2][kRadixBins * C];
  {
    uint4* z = reinterpret_cast<uint4*>(&hist[0][0]);
    for (u32 i = threadIdx.x; i < 2u * kRadixBins * C / 4u; i += kRadixTPB) z[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  const u32 half = threadIdx.x / HALF, t = threadIdx.x % HALF;
  const u32 tile = blockIdx.x * 2u + half;
  const u64 base = (u64)tile * (kRadixTPB * 8) + (u64)t * B;
  const u32 copy = t & (C - 1u);
  u32* h = hist[half];
  const uint4 q = *reinterpret_cast<const uint4*>(plane + base);
  const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int b = 0; b < 16; ++b) atomicAdd(&h[((w[b >> 2] >> (8 * (b & 3))) & 255u) * C + copy], 1u);
  __syncthreads();
  u32 acc = 0;
  for (u32 bin = t; bin < (u32)kRadixBins; bin += HALF) {
    const uint4* r4 = reinterpret_cast<const uint4*>(&hist[half][bin * C]);
    u32 c = 0;
#pragma unroll
    for (u32 r = 0; r < C / 4u; ++r) { const uint4 v = r4[r]; c += v.x + v.y + v.z + v.w; }
    acc ^= c;
  }
  if (acc == 0xFFFFFFFFu) table[tile] = acc;
}

// C: TP consecutive tiles per workgroup, two at a time; all loads up front; a bin's copies are summed and cleared
// by the same thread; the counts wait in LDS and leave as whole 64-byte runs per bin
template <int TP>
__global__ __launch_bounds__(kRadixTPB) void k_hist_multi(const u8* __restrict__ plane, u32* __restrict__ table, u64 n, u32 ntiles) {
  constexpr int HALF = kRadixTPB / 2, B = 16, IT = TP / 2;
  constexpr u32 C = 16;
  __shared__ __attribute__((aligned(16))) u32 hist[2][kRadixBins * C];
  __shared__ __attribute__((aligned(16))) u32 res[kRadixBins][TP];
  const u32 half = threadIdx.x / HALF, t = threadIdx.x % HALF;
  const u32 tile0 = blockIdx.x * TP;
  uint4 q[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const u32 tile = tile0 + 2u * it + half;
    const u64 base = (u64)tile * (kRadixTPB * 8) + (u64)t * B;
    q[it] = tile < ntiles ? *reinterpret_cast<const uint4*>(plane + base) : make_uint4(0, 0, 0, 0);
  }
  u32* h = hist[half];
  uint4* mine = reinterpret_cast<uint4*>(&h[t * C]);
#pragma unroll
  for (u32 r = 0; r < C / 4u; ++r) mine[r] = make_uint4(0, 0, 0, 0);
  const u32 copy = t & (C - 1u);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    __syncthreads();
    const u32 w[4] = {q[it].x, q[it].y, q[it].z, q[it].w};
#pragma unroll
    for (int b = 0; b < 16; ++b) atomicAdd(&h[((w[b >> 2] >> (8 * (b & 3))) & 255u) * C + copy], 1u);
    __syncthreads();
    u32 c = 0;
#pragma unroll
    for (u32 r = 0; r < C / 4u; ++r) { const uint4 v = mine[r]; c += v.x + v.y + v.z + v.w; mine[r] = make_uint4(0, 0, 0, 0); }
    res[t][2 * it + half] = c;
  }
  __syncthreads();
  // 256 bins x TP tiles: thread -> (bin, quarter of the tiles)
  constexpr int PER = TP / 2;                      // words per thread
  const u32 bin = threadIdx.x / 2u, part = threadIdx.x % 2u;
  const uint4* src = reinterpret_cast<const uint4*>(&res[bin][part * PER]);
  uint4* dst = reinterpret_cast<uint4*>(&table[(u64)bin * ntiles + tile0 + part * PER]);
#pragma unroll
  for (int i = 0; i < PER / 4; ++i) dst[i] = src[i];
}

// D: the kernel as it is, tile pairs dealt out so that the 16 workgroups that fill one 128-byte line of a bin's row
// run on one XCD (workgroup b runs on XCD b % 8)
__global__ __launch_bounds__(kRadixTPB) void k_hist_xcd(const u8* __restrict__ plane, u32* __restrict__ table, u64 n, u32 ntiles) {
  constexpr int HALF = kRadixTPB / 2, B = 16;
  constexpr u32 C = 16;
  __shared__ __attribute__((aligned(16))) u32 hist[2][kRadixBins * C];
  {
    uint4* z = reinterpret_cast<uint4*>(&hist[0][0]);
    for (u32 i = threadIdx.x; i < 2u * kRadixBins * C / 4u; i += kRadixTPB) z[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  const u32 half = threadIdx.x / HALF, t = threadIdx.x % HALF;
  const u32 b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
  const u32 pair = (slot >> 4) * 128u + xcd * 16u + (slot & 15u);
  const u32 tile = pair * 2u + half;
  const u64 base = (u64)tile * (kRadixTPB * 8) + (u64)t * B;
  const u32 copy = t & (C - 1u);
  u32* h = hist[half];
  const uint4 q = *reinterpret_cast<const uint4*>(plane + base);
  const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int bb = 0; bb < 16; ++bb) atomicAdd(&h[((w[bb >> 2] >> (8 * (bb & 3))) & 255u) * C + copy], 1u);
  __syncthreads();
  for (u32 bin = t; bin < (u32)kRadixBins; bin += HALF) {
    const uint4* r4 = reinterpret_cast<const uint4*>(&hist[half][bin * C]);
    u32 c = 0;
#pragma unroll
    for (u32 r = 0; r < C / 4u; ++r) { const uint4 v = r4[r]; c += v.x + v.y + v.z + v.w; }
    table[(u64)bin * ntiles + tile] = c;
  }
}

// D2: D + the load issued before the clear + a layout in which a bin's copies are summed without bank conflicts:
// word of (bin, copy) = ((copy / 4) * 256 + (bin ^ (copy / 4))) * 4 + copy % 4
template <u32 C>
__global__ __launch_bounds__(kRadixTPB) void k_hist_xcd2(const u8* __restrict__ plane, u32* __restrict__ table, u64 n, u32 ntiles) {
  constexpr int HALF = kRadixTPB / 2, B = 16;
  constexpr u32 Q = C / 4;
  __shared__ __attribute__((aligned(16))) u32 hist[2][kRadixBins * C];
  const u32 half = threadIdx.x / HALF, t = threadIdx.x % HALF;
  const u32 b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
  const u32 pair = (slot >> 4) * 128u + xcd * 16u + (slot & 15u);
  const u32 tile = pair * 2u + half;
  const u64 base = (u64)tile * (kRadixTPB * 8) + (u64)t * B;
  const uint4 q = *reinterpret_cast<const uint4*>(plane + base);
  {
    uint4* z = reinterpret_cast<uint4*>(&hist[0][0]);
    for (u32 i = threadIdx.x; i < 2u * kRadixBins * C / 4u; i += kRadixTPB) z[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  const u32 copy = t & (C - 1u), cq = copy >> 2;
  u32* h = hist[half] + cq * 1024u + (copy & 3u);
  const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int bb = 0; bb < 16; ++bb) atomicAdd(&h[(((w[bb >> 2] >> (8 * (bb & 3))) & 255u) ^ cq) * 4u], 1u);
  __syncthreads();
  {
    const u32 bin = t;
    u32 c = 0;
#pragma unroll
    for (u32 r = 0; r < Q; ++r) { const uint4 v = *reinterpret_cast<const uint4*>(&hist[half][(r * 256u + (bin ^ r)) * 4u]); c += v.x + v.y + v.z + v.w; }
    table[(u64)bin * ntiles + tile] = c;
  }
}

int main() {
  const u64 n = 1ull << 28;
  const u32 ntiles = (u32)(n / 4096);
  u8* plane; u32 *table, *table2;
  hipMalloc(&plane, n); hipMalloc(&table, (u64)ntiles * 256 * 4); hipMalloc(&table2, (u64)ntiles * 256 * 4);
  std::vector<u8> h(n);
  uint64_t x = 88172645463325252ull;
  for (u64 i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (u8)(x >> 32); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 10; ++r) launch(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.1f us\n", name, ms * 100.0);
  };
  auto check = [&](const char* name) {
    std::vector<u32> a((u64)ntiles * 256), b((u64)ntiles * 256);
    hipMemcpy(a.data(), table, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), table2, b.size() * 4, hipMemcpyDeviceToHost);
    u64 bad = 0; for (u64 i = 0; i < a.size(); ++i) bad += a[i] != b[i];
    printf("  %s: %llu words differ\n", name, (unsigned long long)bad);
  };
  for (int kind = 0; kind < 2; ++kind) {
    if (kind == 1) { for (u64 i = 0; i < n; ++i) h[i] = (u8)((i >> 12) & 3); printf("-- skewed plane (four values, long runs)\n"); }
    else printf("-- random plane\n");
    hipMemcpy(plane, h.data(), n, hipMemcpyHostToDevice);
    hipMemset(table, 0, (u64)ntiles * 1024); hipMemset(table2, 0, (u64)ntiles * 1024);
    run("A as built", [&] { hipLaunchKernelGGL((k_radix_hist_plane<u64, 8>), dim3((ntiles + 1) / 2), dim3(kRadixTPB), 0, 0, (const u8*)plane, table, n, ntiles, SegArgs()); });
    run("B no stores", [&] { hipLaunchKernelGGL(k_hist_nostore, dim3(ntiles / 2), dim3(kRadixTPB), 0, 0, (const u8*)plane, table2, n, ntiles); });
    run("C 16 tiles per workgroup", [&] { hipLaunchKernelGGL((k_hist_multi<16>), dim3(ntiles / 16), dim3(kRadixTPB), 0, 0, (const u8*)plane, table2, n, ntiles); });
    check("C16");
    run("C 8 tiles per workgroup", [&] { hipLaunchKernelGGL((k_hist_multi<8>), dim3(ntiles / 8), dim3(kRadixTPB), 0, 0, (const u8*)plane, table2, n, ntiles); });
    check("C8");
    run("C 32 tiles per workgroup", [&] { hipLaunchKernelGGL((k_hist_multi<32>), dim3(ntiles / 32), dim3(kRadixTPB), 0, 0, (const u8*)plane, table2, n, ntiles); });
    check("C32");
    run("D xcd-aware pairs", [&] { hipLaunchKernelGGL(k_hist_xcd, dim3(ntiles / 2), dim3(kRadixTPB), 0, 0, (const u8*)plane, table2, n, ntiles); });
    check("D");
    run("D2 xcd + swizzle, 16 copies", [&] { hipLaunchKernelGGL((k_hist_xcd2<16>), dim3(ntiles / 2), dim3(kRadixTPB), 0, 0, (const u8*)plane, table2, n, ntiles); });
    check("D2/16");
    run("D2 xcd + swizzle, 8 copies", [&] { hipLaunchKernelGGL((k_hist_xcd2<8>), dim3(ntiles / 2), dim3(kRadixTPB), 0, 0, (const u8*)plane, table2, n, ntiles); });
    check("D2/8");
  }
  return 0;
}
