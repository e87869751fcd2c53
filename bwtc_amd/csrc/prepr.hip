// The pair-replacing pre-stage (`--prepr p...`) on the GPU: pair statistics and the replaced text.
// Replaces the two sweeps of preprocessors/PairReplacer.cpp -- analyseData (:53-63 with PairReplacer.hpp:47-67)
// and writeReplacedVersion (:330-400); the choice of pairs and the grammar stay host code (prepr_host.cpp).
//
//   k_pair_stats   65 536 pair counters + 256 byte counters.  A workgroup keeps a quarter of the pair counters
//                  (16 384 words = 64 KiB of LDS, the quarter = the top two bits of the pair's first byte) and walks
//                  many tiles before it flushes the non-zero ones with global atomics; grid.y = 4 quarters, so the
//                  text is read four times (1 GB for a 256 MiB block -- 0.3 ms of HBM) instead of 65 536 counters
//                  of 4 bytes not fitting one workgroup.  The reference counts the pair that ends at position i
//                  always for even i and, for odd i, only when it differs from the pair ending at i - 1
//                  (analyseData0): three equal bytes in a row, with a virtual zero byte before the text.
//   k_pr_heads / k_pr_head_scan   (only when a pair of EQUAL bytes is replaced)  where the run of equal bytes
//                  that reaches into a tile starts: along such a run the replaced pairs are taken from its start.
//   k_pr_emit<COUNT>   what every position turns into -- nothing (second byte of a replaced pair), the pair's
//                  variable, its own two-byte escape, or itself -- counted per tile, scanned, written.
#include "bwt_engine.hpp"
#include "prepr_host.hpp"

namespace bwtc_hip {

constexpr int kPrTPB = 256;
constexpr u32 kPrPer = 16;                       // positions per thread
constexpr u32 kPrTile = kPrTPB * kPrPer;         // 4096 positions per tile

// the thread's bytes d[j0 - 2 .. j0 + 16] (zero outside the text) -> c[0 .. 18]
__device__ __forceinline__ void pr_load(const u8* __restrict__ d, u64 n, u64 j0, u32 (&c)[19]) {
  if (j0 >= 2 && j0 + 17 <= n && ((reinterpret_cast<uintptr_t>(d) + j0) & 15u) == 0) {
    const uint4 v = *reinterpret_cast<const uint4*>(d + j0);
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) c[2 + k] = (w[k >> 2] >> (8 * (k & 3))) & 255u;
    c[0] = d[j0 - 2]; c[1] = d[j0 - 1]; c[18] = d[j0 + 16];
  } else {
#pragma unroll
    for (int k = 0; k < 19; ++k) {
      const u64 j = j0 + (u64)k;                  // position + 2
      c[k] = (j >= 2 && j - 2 < n) ? d[j - 2] : 0u;
    }
  }
}

__global__ __launch_bounds__(kPrTPB) void k_pair_stats(const u8* __restrict__ d, u64 n, u32* __restrict__ pair_cnt,
                                                       u32* __restrict__ byte_cnt) {
  __shared__ u32 h[16384];
  __shared__ u32 hb[256];
  const u32 quarter = blockIdx.y;
  for (u32 k = threadIdx.x; k < 16384u; k += kPrTPB) h[k] = 0;
  hb[threadIdx.x] = 0;
  __syncthreads();
  const u64 ntiles = (n + kPrTile - 1) / kPrTile;
  for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const u64 j0 = tile * kPrTile + (u64)threadIdx.x * kPrPer;
    if (j0 >= n) continue;
    u32 c[19];
    pr_load(d, n, j0, c);
#pragma unroll
    for (u32 k = 0; k < kPrPer; ++k) {
      const u64 i = j0 + k;
      if (i >= n) break;
      const u32 cur = c[2 + k], prev = c[1 + k], before = c[k];
      if (quarter == 0) atomicAdd(&hb[cur], 1u);
      if (i == 0 || (prev >> 6) != quarter) continue;
      if ((i & 1) && before == prev && prev == cur) continue;          // the same pair as the one ending at i - 1
      atomicAdd(&h[((prev & 63u) << 8) | cur], 1u);
    }
  }
  __syncthreads();
  for (u32 k = threadIdx.x; k < 16384u; k += kPrTPB) if (h[k]) atomicAdd(&pair_cnt[quarter * 16384u + k], h[k]);
  if (quarter == 0 && hb[threadIdx.x]) atomicAdd(&byte_cnt[threadIdx.x], hb[threadIdx.x]);
}

// last run head (position + 1; a head: position 0, or a byte that differs from the one before) of every tile
__global__ __launch_bounds__(kPrTPB) void k_pr_heads(const u8* __restrict__ d, u64 n, u32* __restrict__ tile_head) {
  __shared__ u32 scr[kPrTPB / kWave + 1];
  const u64 j0 = (u64)blockIdx.x * kPrTile + (u64)threadIdx.x * kPrPer;
  u32 last = 0;
  if (j0 < n) {
    u32 c[19];
    pr_load(d, n, j0, c);
#pragma unroll
    for (u32 k = 0; k < kPrPer; ++k) {
      const u64 j = j0 + k;
      if (j < n && (j == 0 || c[1 + k] != c[2 + k])) last = (u32)j + 1u;
    }
  }
  u32 total;
  (void)block_scan_incl_max<kPrTPB>(last, scr, &total);
  if (threadIdx.x == 0) tile_head[blockIdx.x] = total;
}
// exclusive running maximum over the tiles, in place (one workgroup; at most a few hundred thousand tiles)
__global__ __launch_bounds__(1024) void k_pr_head_scan(u32* __restrict__ tile_head, u32 ntiles) {
  __shared__ u32 scr[1024 / kWave + 1];
  __shared__ u32 s_incl[1024];
  __shared__ u32 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (u32 base = 0; base < ntiles; base += 1024u) {
    const u32 i = base + threadIdx.x;
    const u32 v = i < ntiles ? tile_head[i] : 0u;
    u32 total;
    s_incl[threadIdx.x] = block_scan_incl_max<1024>(v, scr, &total);
    __syncthreads();
    const u32 excl = max(carry, threadIdx.x ? s_incl[threadIdx.x - 1] : 0u);
    if (i < ntiles) tile_head[i] = excl;
    __syncthreads();
    if (threadIdx.x == 0) carry = max(carry, total);
    __syncthreads();
  }
}

// What the thread's sixteen positions turn into.  len[k] in {0, 1, 2}, bytes b0 / b1.
// SELF: some replaced pair has two equal bytes; run_before = where the run that reaches position j0 - 1 starts.
template <bool SELF>
__device__ __forceinline__ void pr_kinds(const u32 (&c)[19], u64 j0, u64 n, const uint16_t* __restrict__ table, u32 common,
                                         u64 run_before, u32 (&len)[kPrPer], u32 (&b0)[kPrPer], u32 (&b1)[kPrPer]) {
  const u32 nothing = common << 8 | common;
  // the pair starting at position j0 - 1 (was it replaced? then j0 is its second byte)
  u64 run = run_before;
  bool prev_fired = false;
  if (j0 >= 1) {
    const u32 t = table[c[1] << 8 | c[2]];
    prev_fired = j0 < n && t != nothing && (t & 255u) == common;
    if (SELF && prev_fired && c[1] == c[2]) prev_fired = (((j0 - 1) - run) & 1ull) == 0;
  }
#pragma unroll
  for (u32 k = 0; k < kPrPer; ++k) {
    const u64 j = j0 + k;
    len[k] = 0; b0[k] = 0; b1[k] = 0;
    if (j >= n) continue;
    const u32 cur = c[2 + k], next = c[3 + k];                // next is 0 behind the text, as the reference looks it up
    if (SELF && (j == 0 || c[1 + k] != cur)) run = j;
    const u32 t = table[cur << 8 | next];
    bool fires = j + 1 < n && t != nothing && (t & 255u) == common;
    if (SELF && fires && cur == next) fires = ((j - run) & 1ull) == 0;
    if (prev_fired) { prev_fired = false; continue; }        // the second byte of a replaced pair
    if (fires) { len[k] = 1; b0[k] = t >> 8; prev_fired = true; }
    else if (j + 1 < n ? t == nothing : (t & 255u) == common) { len[k] = 1; b0[k] = cur; }
    else { len[k] = 2; b0[k] = t >> 8; b1[k] = t & 255u; }
  }
}

template <bool SELF, bool COUNT>
__global__ __launch_bounds__(kPrTPB) void k_pr_emit(const u8* __restrict__ d, u64 n, const uint16_t* __restrict__ table, u32 common,
                                                    const u32* __restrict__ tile_head, u32* __restrict__ tile_count,
                                                    const u32* __restrict__ tile_offset, u8* __restrict__ dst) {
  __shared__ u32 scr[kPrTPB / kWave + 1];
  __shared__ u32 s_incl[kPrTPB];
  const u64 j0 = (u64)blockIdx.x * kPrTile + (u64)threadIdx.x * kPrPer;
  u32 c[19];
  pr_load(d, n, j0 < n ? j0 : 0, c);
  u64 run_before = 0;
  if (SELF) {
    // where the run that reaches j0 - 1 starts: the last head before j0 -- in earlier tiles (tile_head, exclusive
    // maximum) or among the earlier threads' positions of this tile
    u32 mine = 0;
    if (j0 < n) {
#pragma unroll
      for (u32 k = 0; k < kPrPer; ++k) { const u64 j = j0 + k; if (j < n && (j == 0 || c[1 + k] != c[2 + k])) mine = (u32)j + 1u; }
    }
    u32 total;
    const u32 incl = block_scan_incl_max<kPrTPB>(mine, scr, &total);
    s_incl[threadIdx.x] = incl;
    __syncthreads();
    const u32 head = max(tile_head[blockIdx.x], threadIdx.x ? s_incl[threadIdx.x - 1] : 0u);
    run_before = head ? (u64)head - 1ull : 0ull;
    __syncthreads();
  }
  u32 len[kPrPer], b0[kPrPer], b1[kPrPer];
  u32 mine = 0;
  if (j0 < n) {
    pr_kinds<SELF>(c, j0, n, table, common, run_before, len, b0, b1);
#pragma unroll
    for (u32 k = 0; k < kPrPer; ++k) mine += len[k];
  } else {
#pragma unroll
    for (u32 k = 0; k < kPrPer; ++k) len[k] = 0;
  }
  u32 total;
  u32 at = block_scan_excl_add<kPrTPB>(mine, scr, &total);
  if (COUNT) { if (threadIdx.x == 0) tile_count[blockIdx.x] = total; return; }
  at += tile_offset[blockIdx.x];
#pragma unroll
  for (u32 k = 0; k < kPrPer; ++k) {
    if (len[k] >= 1) dst[at++] = (u8)b0[k];
    if (len[k] == 2) dst[at++] = (u8)b1[k];
  }
}

// ---- one PairReplacer over a device-resident text ---------------------------------------------------
// d_src (n bytes) -> d_dst (room for 2 n): statistics on the device, the choice on the host (grammar updated),
// the replaced text on the device.  *n_out = its length, *replaced = pairs replaced.
int pair_replace_device(BwtEngine& e, bwtc::prepr::Grammar& grammar, const u8* d_src, u64 n, u8* d_dst, u64* n_out, u32* replaced) {
  hipStream_t st = e.stream;
  if (n < 3 || n >= (1ull << 31)) return -1;                       // PairReplacer::analyseData asserts length > 2
  BWTC_HIP_TRY(hipSetDevice(e.device));
  const u32 ntiles = (u32)((n + kPrTile - 1) / kPrTile);
  // workspace: counters, replacement table, per-tile words -- in the transform's arena (free between blocks)
  u32* d_pairs = e.d_V0;                                           // 65536 + 256 words
  u32* d_bytes = d_pairs + 65536;
  uint16_t* d_table = reinterpret_cast<uint16_t*>(e.d_V1);         // 65536 half words
  u32* d_head = e.d_G0;                                            // ntiles words each
  u32* d_count = e.d_G1;
  if ((u64)ntiles + 1 > e.cap || e.cap < (1u << 17)) return -1;
  BWTC_HIP_TRY(hipMemsetAsync(d_pairs, 0, (65536 + 256) * 4, st));
  hipLaunchKernelGGL(k_pair_stats, dim3(std::min<u32>(ntiles, 512u), 4), dim3(kPrTPB), 0, st, d_src, n, d_pairs, d_bytes);
  std::vector<u32> counts(65536 + 256);
  BWTC_HIP_TRY(hipMemcpyAsync(counts.data(), d_pairs, counts.size() * 4, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(e.wait());
  std::vector<uint64_t> pf(65536), bf(256);
  for (size_t i = 0; i < 65536; ++i) pf[i] = counts[i];
  for (size_t i = 0; i < 256; ++i) bf[i] = counts[65536 + i];
  bwtc::prepr::Replacements r;
  bwtc::prepr::decideReplacements(&grammar, bf.data(), pf.data(), &r);
  *replaced = r.count;
  if (r.count == 0) {                                              // nothing to do: the text as it is (same bytes as the reference writes)
    BWTC_HIP_TRY(hipMemcpyAsync(d_dst, d_src, n, hipMemcpyDeviceToDevice, st));
    BWTC_HIP_TRY(e.wait());
    *n_out = n;
    return 0;
  }
  BWTC_HIP_TRY(hipMemcpyAsync(d_table, r.table.data(), 65536 * 2, hipMemcpyHostToDevice, st));
  if (r.self_pairs) {
    hipLaunchKernelGGL(k_pr_heads, dim3(ntiles), dim3(kPrTPB), 0, st, d_src, n, d_head);
    hipLaunchKernelGGL(k_pr_head_scan, dim3(1), dim3(1024), 0, st, d_head, ntiles);
    hipLaunchKernelGGL((k_pr_emit<true, true>), dim3(ntiles), dim3(kPrTPB), 0, st, d_src, n, (const uint16_t*)d_table, (u32)r.common,
                       (const u32*)d_head, d_count, (const u32*)nullptr, (u8*)nullptr);
  } else {
    hipLaunchKernelGGL((k_pr_emit<false, true>), dim3(ntiles), dim3(kPrTPB), 0, st, d_src, n, (const uint16_t*)d_table, (u32)r.common,
                       (const u32*)nullptr, d_count, (const u32*)nullptr, (u8*)nullptr);
  }
  // the tiles' offsets: their counts (at most 8192 each) scanned on the host -- a few hundred KB down and up
  std::vector<u32> tc(ntiles);
  BWTC_HIP_TRY(hipMemcpyAsync(tc.data(), d_count, (size_t)ntiles * 4, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(e.wait());
  u64 total = 0;
  for (u32 t = 0; t < ntiles; ++t) { const u32 v = tc[t]; tc[t] = (u32)total; total += v; }
  if (total >= (1ull << 32)) return -1;
  BWTC_HIP_TRY(hipMemcpyAsync(d_count, tc.data(), (size_t)ntiles * 4, hipMemcpyHostToDevice, st));
  if (r.self_pairs)
    hipLaunchKernelGGL((k_pr_emit<true, false>), dim3(ntiles), dim3(kPrTPB), 0, st, d_src, n, (const uint16_t*)d_table, (u32)r.common,
                       (const u32*)d_head, (u32*)nullptr, (const u32*)d_count, d_dst);
  else
    hipLaunchKernelGGL((k_pr_emit<false, false>), dim3(ntiles), dim3(kPrTPB), 0, st, d_src, n, (const uint16_t*)d_table, (u32)r.common,
                       (const u32*)nullptr, (u32*)nullptr, (const u32*)d_count, d_dst);
  BWTC_HIP_TRY(e.wait());
  BWTC_HIP_TRY(hipGetLastError());
  *n_out = total;
  return 0;
}

}  // namespace bwtc_hip
