// GPU inverse Burrows-Wheeler transform.  Replaces InverseBWTransform::doTransform(BWTBlock&)
// (bwtransforms/InverseBWT.cpp:47-51) and the MTL-SA walk it drives
// (bwtransforms/MtlSaInverseBWT.cpp:246-362).
//
// Row model (same as the forward path): N = size + 1 rows, L[i] = block[i] for i < size,
// L[size] = block[eob] (InverseBWT.cpp:49), row eob = LF[0] is the end-of-block row.
// LF(i) = 1 + #{j != eob : L[j] < L[i]} + #{j < i, j != eob : L[j] == L[i]},  LF(eob) = 0
// (test/LFpowersTest.cpp:85-105); the original block is  out[i] = L[LF^i(0)], i < size.
//
// The reference walks this chain serially (from up to 256 starting points).  LF is one
// N-cycle, so the walk is LIST RANKING, done here in three data-parallel steps:
//   1. LF for all rows = the destination computation of ONE stable 8-bit radix pass
//      (tile histograms -> scan -> ballot ranking); stored packed with L as u64 words.
//   2. every 64th row is a splitter: one lane per splitter walks to the next splitter
//      (next splitter, steps); pointer jumping over the 64x shorter splitter list gives each
//      splitter its distance from row 0.
//   3. the splitters walk their sublists again and write the characters to their final
//      positions.
// 2N dependent 8-byte gathers in total, spread over N/64 lanes: latency is hidden by
// parallelism instead of by the reference's 256-way interleaving.
#include "bwt_engine.hpp"
#include "scan.hpp"

namespace bwtc_hip {

constexpr int kInvTPB = 512;
constexpr int kInvE = 8;
constexpr int kInvTile = kInvTPB * kInvE;     // rows per workgroup
constexpr int kInvWaves = kInvTPB / kWave;
constexpr u32 kSplit = 64;                    // one splitter per kSplit rows

__device__ __forceinline__ u32 inv_L(const u8* __restrict__ bwt, u32 size, u32 eob, u32 row) {
  return row < size ? bwt[row] : bwt[eob];
}

// tile histogram of L (the end-of-block row does not count) -> table[c * ntiles + tile]
__global__ __launch_bounds__(kInvTPB) void k_inv_hist(const u8* __restrict__ bwt, u32 size,
                                                      u32 eob, u32* __restrict__ table,
                                                      u32 ntiles) {
  __shared__ u32 hist[256 * 16];
  for (u32 i = threadIdx.x; i < 256 * 16; i += kInvTPB) hist[i] = 0;
  __syncthreads();
  const u32 n = size + 1u;
  const u32 base = blockIdx.x * kInvTile + threadIdx.x;
  const u32 copy = threadIdx.x & 15u;
#pragma unroll
  for (int e = 0; e < kInvE; ++e) {
    const u32 row = base + e * kInvTPB;
    if (row < n && row != eob) atomicAdd(&hist[inv_L(bwt, size, eob, row) * 16u + copy], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    u32 c = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) c += hist[threadIdx.x * 16u + r];
    table[(u64)threadIdx.x * ntiles + blockIdx.x] = c;
  }
}

// LF of every row of the tile, packed with the row's character: lfl[row] = LF | L << 32
__global__ __launch_bounds__(kInvTPB) void k_inv_lf(const u8* __restrict__ bwt, u32 size, u32 eob,
                                                    const u32* __restrict__ table, u32 ntiles,
                                                    u64* __restrict__ lfl) {
  __shared__ u32 s_cnt[kInvWaves][256];
  for (u32 i = threadIdx.x; i < kInvWaves * 256; i += kInvTPB) (&s_cnt[0][0])[i] = 0;
  __syncthreads();
  const u32 n = size + 1u;
  const u32 wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const u32 wrow = blockIdx.x * kInvTile + wave * (kWave * kInvE) + lane;
  u32 c[kInvE], r[kInvE];
  volatile lds_u32* my_cnt = lds_volatile(&s_cnt[wave][0]);
#pragma unroll
  for (int e = 0; e < kInvE; ++e) {
    const u32 row = wrow + e * kWave;
    const bool ok = row < n && row != eob;
    c[e] = row < n ? inv_L(bwt, size, eob, row) : 0u;
    u32 below, peers;
    wave_digit_rank<8>(c[e], ok, &below, &peers);
    u32 prev = 0;
    if (ok) prev = my_cnt[c[e]];
    __builtin_amdgcn_wave_barrier();
    r[e] = prev + below;
    if (ok && below + 1u == peers) my_cnt[c[e]] = prev + peers;
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  if (threadIdx.x < 256) {                     // per character: exclusive prefix over waves
    u32 run = table[(u64)threadIdx.x * ntiles + blockIdx.x] + 1u;   // +1: row 0 is the EOB's target
    for (int w = 0; w < kInvWaves; ++w) { const u32 t = s_cnt[w][threadIdx.x]; s_cnt[w][threadIdx.x] = run; run += t; }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < kInvE; ++e) {
    const u32 row = wrow + e * kWave;
    if (row < n) {
      const u32 lf = (row == eob) ? 0u : s_cnt[wave][c[e]] + r[e];
      lfl[row] = (u64)lf | ((u64)c[e] << 32);
    }
  }
}

// splitter s = row s * kSplit.  Walk to the next splitter: nxt[s] = its index, len[s] = steps.
// LF as computed by k_inv_lf is a permutation of the rows whatever the input, so the walk from
// a splitter always ends (at the latest back at itself).  max_len = longest sublist: the bound
// k_inv_check_lf uses for its own walk.
__global__ __launch_bounds__(256) void k_inv_walk1(const u64* __restrict__ lfl, u32 n_split,
                                                   u32* __restrict__ nxt, u32* __restrict__ len,
                                                   u32* __restrict__ max_len) {
  const u32 s = blockIdx.x * 256u + threadIdx.x;
  u32 steps = 0;
  if (s < n_split) {
    u32 row = (u32)lfl[(u64)s * kSplit];
    steps = 1;
    while (row % kSplit != 0u) { row = (u32)lfl[row]; ++steps; }
    nxt[s] = row / kSplit;
    len[s] = steps;
  }
  const u32 lane = lane_id();
  const u32 m = wave_scan_max(steps, lane);
  if (lane == kWave - 1 && m) atomicMax(max_len, m);
}

// Pointer jumping over the splitter list.  The list is the cycle 0 -> nxt[0] -> ... -> 0; the
// node whose successor is 0 becomes the tail (successor = n_split).  After ceil(log2 n_split)
// rounds dist[s] = number of rows from splitter s to the end of the cycle.
__global__ __launch_bounds__(256) void k_inv_jump_init(u32* __restrict__ nxt, u32 n_split) {
  const u32 s = blockIdx.x * 256u + threadIdx.x;
  if (s < n_split && nxt[s] == 0u) nxt[s] = n_split;
}
__global__ __launch_bounds__(256) void k_inv_jump(const u32* __restrict__ nxt_in,
                                                  const u32* __restrict__ dist_in,
                                                  u32* __restrict__ nxt_out,
                                                  u32* __restrict__ dist_out, u32 n_split) {
  const u32 s = blockIdx.x * 256u + threadIdx.x;
  if (s >= n_split) return;
  const u32 t = nxt_in[s];
  u32 d = dist_in[s];
  u32 nn = t;
  if (t < n_split) { d += dist_in[t]; nn = nxt_in[t]; }
  nxt_out[s] = nn;
  dist_out[s] = d;
}

// second walk: splitter s starts at output position n - dist[s] and emits its sublist
__global__ __launch_bounds__(256) void k_inv_walk2(const u64* __restrict__ lfl,
                                                   const u32* __restrict__ dist,
                                                   const u32* __restrict__ len, u32 n_split, u32 n,
                                                   u32 size, u8* __restrict__ out) {
  const u32 s = blockIdx.x * 256u + threadIdx.x;
  if (s >= n_split) return;
  u32 pos = n - dist[s];
  const u32 steps = len[s];
  u64 v = lfl[(u64)s * kSplit];
  for (u32 j = 0; j < steps; ++j) {
    if (pos < size) out[pos] = (u8)(v >> 32);
    ++pos;
    v = lfl[(u32)v];
  }
}

// checks LF powers against the ranking: row lf[k] must sit k * (n / n_lf) steps before the end
// of the cycle's walk from the EOB, i.e. at distance (n - k*x) from row 0 ... done on the
// splitter level only when the row is a splitter; otherwise by a short walk.
// A healthy block's LF is ONE cycle through all n rows: splitter 0 is then n rows from the end
// of the walk (bad |= 2 otherwise), and any row reaches a splitter within max_len steps (a row
// that does not lies on a cycle of its own: bad |= 1, and the walk stops there instead of
// circling for ever).
__global__ void k_inv_check_lf(const u64* __restrict__ lfl, const u32* __restrict__ dist,
                               const u32* __restrict__ lf, u32 n_lf, u32 n,
                               const u32* __restrict__ max_len, u32* __restrict__ bad) {
  const u32 k = threadIdx.x;
  if (k == 0) { if (dist[0] != n) atomicOr(bad, 2u); return; }
  if (k >= n_lf) return;
  const u32 x = n / n_lf;
  // LF^j(eob) is the row of suffix n - j, which is row number (j) on the walk from row 0
  // counted as p_{j-1}: p_0 = row 0 = LF(eob).  So row lf[k] must be p_{k*x - 1}.
  u32 row = lf[k];
  u32 steps = 0;
  const u32 limit = *max_len;
  while (row % kSplit != 0u && steps <= limit) { row = (u32)lfl[row]; ++steps; }
  if (row % kSplit != 0u) { atomicOr(bad, 1u); return; }
  const u32 at = n - dist[row / kSplit];          // index of that splitter on the walk
  const u32 idx = at >= steps ? at - steps : at + n - steps;
  if (idx != k * x - 1u) atomicOr(bad, 1u);
}

int inverse_bwt_device(BwtEngine& e, const u8* d_bwt, u8* d_out, u32 size, const u32* lf, u32 n_lf) {
  BwtEngine::ScanScope scan_scope(e);
  hipStream_t st = e.stream;
  if (!lf || n_lf == 0 || n_lf > 256) return -1;
  if (size == 0) return 0;
  if ((u64)size + 1 > e.cap) return -1;
  const u32 n = size + 1u;
  const u32 eob = lf[0];
  if (eob > size) return -1;
  for (u32 k = 1; k < n_lf; ++k) if (lf[k] > size) return -4;   // not a row of this block
  BWTC_HIP_TRY(hipSetDevice(e.device));
  BWTC_HIP_TRY(hipEventRecord(e.ev_begin, st));

  u64* lfl = static_cast<u64*>(e.d_R1);
  const u32 ntiles = ceil_div(n, kInvTile);
  hipLaunchKernelGGL(k_inv_hist, dim3(ntiles), dim3(kInvTPB), 0, st, d_bwt, size, eob, e.d_table, ntiles);
  exclusive_scan_u32(e.d_table, (u64)ntiles * 256, e.d_partial, st);
  hipLaunchKernelGGL(k_inv_lf, dim3(ntiles), dim3(kInvTPB), 0, st, d_bwt, size, eob, e.d_table, ntiles, lfl);

  const u32 n_split = (n + kSplit - 1u) / kSplit;
  u32 *nxtA = e.d_V0, *nxtB = e.d_V1, *distA = e.d_G0, *distB = e.d_G1, *len = e.d_GRP;
  const u32 sblocks = ceil_div(n_split, 256);
  BWTC_HIP_TRY(hipMemsetAsync(e.d_small, 0, 1024 * 4, st));
  u32* d_maxlen = e.d_small + 601;
  hipLaunchKernelGGL(k_inv_walk1, dim3(sblocks), dim3(256), 0, st, lfl, n_split, nxtA, len, d_maxlen);
  BWTC_HIP_TRY(hipMemcpyAsync(distA, len, (u64)n_split * 4, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_inv_jump_init, dim3(sblocks), dim3(256), 0, st, nxtA, n_split);
  for (u32 span = 1; span < n_split; span *= 2) {
    hipLaunchKernelGGL(k_inv_jump, dim3(sblocks), dim3(256), 0, st, nxtA, distA, nxtB, distB, n_split);
    { u32* t = nxtA; nxtA = nxtB; nxtB = t; }
    { u32* t = distA; distA = distB; distB = t; }
  }
  for (u32 k = 0; k < n_lf; ++k) e.h_small[256 + k] = lf[k];   // pinned: the async copy reads it later
  BWTC_HIP_TRY(hipMemcpyAsync(e.d_small + 256, e.h_small + 256, n_lf * 4, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_inv_check_lf, dim3(1), dim3(256), 0, st, lfl, distA, e.d_small + 256, n_lf, n,
                     d_maxlen, e.d_small + 600);
  hipLaunchKernelGGL(k_inv_walk2, dim3(sblocks), dim3(256), 0, st, lfl, distA, len, n_split, n, size, e.d_out);
  BWTC_HIP_TRY(hipMemcpyAsync(d_out, e.d_out, size, hipMemcpyDeviceToDevice, st));
  BWTC_HIP_TRY(hipMemcpyAsync(e.h_small + 600, e.d_small + 600, 4, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(hipEventRecord(e.ev_end, st));
  BWTC_HIP_TRY(e.wait());
  BWTC_HIP_TRY(hipGetLastError());
  (void)hipEventElapsedTime(&e.stats.ms_total, e.ev_begin, e.ev_end);
  e.stats.n = n;
  if (e.h_small[600]) return -4;                    // an LF power is not on the LF walk
  return 0;
}

}  // namespace bwtc_hip
