// Shared device/host helpers for the gfx950 BWT kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

namespace bwtc_hip {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

constexpr int kWave = 64;

#define BWTC_HIP_TRY(expr)                                                            \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "bwtc_hip: %s failed: %s (%s:%d)\n", #expr,                \
                   hipGetErrorString(e_), __FILE__, __LINE__);                        \
      return e_ == hipErrorOutOfMemory ? -2 : -3;                                     \
    }                                                                                 \
  } while (0)

// Calls whose failure cannot be returned on the spot (inside void helpers, between launches)
// leave their first error here; every transform / coder entry point checks it at its read-back,
// next to hipGetLastError().  Per thread: a context is driven by one thread at a time.
inline hipError_t& sticky_error() { static thread_local hipError_t e = hipSuccess; return e; }
inline void note_error(hipError_t rc) { if (rc != hipSuccess && sticky_error() == hipSuccess) sticky_error() = rc; }
inline hipError_t take_sticky_error() { const hipError_t e = sticky_error(); sticky_error() = hipSuccess; return e; }

static inline u32 ceil_div(u64 a, u64 b) { return (u32)((a + b - 1) / b); }
static inline int bit_width_u64(u64 v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

__device__ __forceinline__ u32 lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Inclusive prefix sum across the 64 lanes of a wave.
__device__ __forceinline__ u32 wave_scan_add(u32 v, u32 lane) {
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    u32 t = __shfl_up(v, o, kWave);
    if (lane >= (u32)o) v += t;
  }
  return v;
}

// Inclusive prefix max across the 64 lanes of a wave.
__device__ __forceinline__ u32 wave_scan_max(u32 v, u32 lane) {
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    u32 t = __shfl_up(v, o, kWave);
    if (lane >= (u32)o) v = max(v, t);
  }
  return v;
}

// Lanes of the wave whose (valid) digit equals this lane's digit.  BITS ballots.
template <int BITS>
__device__ __forceinline__ u64 match_any(u32 d, bool valid) {
  u64 m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < BITS; ++b) {
    bool bit = (d >> b) & 1u;
    u64 bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return m;
}

// For every valid lane: how many LOWER lanes of the wave hold the same (valid) digit, and how many
// lanes hold it in all -- match_any() + two popcounts, written for the instruction count: the
// ranking loops of the radix passes are bound by VALU issue (rocprofv3: SQ_ACTIVE_INST_VALU 0.99
// of the busy cycles for 32-bit keys), not by memory.  Per digit bit: the bit as 0 / -1 (one
// v_bfe_i32), its ballot, and per 32-lane half one xor that leaves the lanes whose bit differs;
// those are OR-ed up as "not a peer".  The rank is then v_mbcnt of the peer mask (no lane mask, no
// 64-bit shift) and the highest peer is the lane with rank + 1 == count.
template <int BITS>
__device__ __forceinline__ void wave_digit_rank(u32 d, bool valid, u32* rank, u32* count) {
  const u64 vb = __ballot(valid);
  u32 nlo = ~(u32)vb, nhi = ~(u32)(vb >> 32);
#pragma unroll
  for (int b = 0; b < BITS; ++b) {
    u32 x;                                // bit b of d as 0 / -1; spelled out, or the compiler makes a shift pair of it
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(x) : "v"(d), "n"(b));
    const u64 bal = __ballot(x != 0u);
    nlo = __builtin_amdgcn_bitop3_b32(nlo, (u32)bal, x, 0xF6);           // nlo | (bal ^ x), one v_bitop3_b32
    nhi = __builtin_amdgcn_bitop3_b32(nhi, (u32)(bal >> 32), x, 0xF6);
  }
  const u32 mlo = ~nlo, mhi = ~nhi;
  *rank = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
  *count = (u32)__popc(mlo) + (u32)__popc(mhi);
}

// LDS words that a wave reads and rewrites in lockstep (the per-wave digit counters): volatile, and
// typed as LDS so that the accesses are ds_read / ds_write (a volatile generic pointer compiles to
// system-scope FLAT loads and stores, which go through the vector memory path).
typedef __attribute__((address_space(3))) u32 lds_u32;
__device__ __forceinline__ volatile lds_u32* lds_volatile(u32* p) { return (volatile lds_u32*)p; }

// Block-wide exclusive sum for TPB threads (TPB multiple of 64, <= 1024).
// `scratch` needs TPB/64 + 1 words of LDS.  Returns the exclusive prefix of v; *total gets
// the block total.  Contains two barriers; every thread of the block must call it.
template <int TPB>
__device__ __forceinline__ u32 block_scan_excl_add(u32 v, u32* scratch, u32* total) {
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x / kWave;
  constexpr int NW = TPB / kWave;
  u32 incl = wave_scan_add(v, lane);
  if (lane == kWave - 1) scratch[wave] = incl;
  __syncthreads();
  u32 wave_off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    u32 t = scratch[w];
    if ((u32)w < wave) wave_off += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return wave_off + incl - v;
}

// Block-wide inclusive max, same contract as above (identity 0).
template <int TPB>
__device__ __forceinline__ u32 block_scan_incl_max(u32 v, u32* scratch, u32* total) {
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x / kWave;
  constexpr int NW = TPB / kWave;
  u32 incl = wave_scan_max(v, lane);
  if (lane == kWave - 1) scratch[wave] = incl;
  __syncthreads();
  u32 wave_off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    u32 t = scratch[w];
    if ((u32)w < wave) wave_off = max(wave_off, t);
    tot = max(tot, t);
  }
  __syncthreads();
  *total = tot;
  return max(wave_off, incl);
}

}  // namespace bwtc_hip
