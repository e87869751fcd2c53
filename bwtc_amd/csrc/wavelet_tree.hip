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

__device__ __forceinline__ uint2 wt_length_entry(const WtTables& t, u32 s, u32 len) {
  if (len < kWtLenDense) return t.lendense[(u64)s * kWtLenDense + len];
  u32 lo = t.over_first[s], hi = t.over_first[s + 1];
  while (lo < hi) { u32 mid = (lo + hi) >> 1; if (t.over[mid].x < len) lo = mid + 1; else hi = mid; }
  const uint4 e = t.over[lo];
  return make_uint2(e.y, e.z);
}

// steps per run -> cnt[r]; cnt[n_runs] = 0 so that the exclusive scan leaves the total there
__global__ __launch_bounds__(kWtTPB) void k_wt_count(const u32* __restrict__ run_start,
                                                     const u8* __restrict__ run_sym, u32 n_runs,
                                                     WtTables t, u32* __restrict__ cnt) {
  __shared__ u32 s_first[257];
  for (u32 i = threadIdx.x; i <= t.nsec; i += kWtTPB) s_first[i] = t.first_run[i];
  __syncthreads();
  const u32 r = blockIdx.x * kWtTPB + threadIdx.x;
  if (r > n_runs) return;
  if (r == n_runs) { cnt[r] = 0; return; }
  const u32 s = wt_section_of(s_first, t.nsec, r);
  const uint2 sy = t.symtab[s * 256u + run_sym[r]];
  u32 c = sy.y & 255u;
  if (sy.y >> 16) c += wt_length_entry(t, s, run_start[r + 1] - run_start[r]).y;
  cnt[r] = c;
}

// One thread per run; the step words of a workgroup's 256 runs are consecutive in the output
// (off[] is the prefix of the step counts), so they are assembled in LDS and written by
// consecutive lanes -- a thread storing its own four or five words would touch a cache line
// per lane and instruction.  Workgroups whose runs expand to more than the LDS window (very long
// codes) write directly.
constexpr u32 kWtExpandWin = 6144;    // words of LDS staging: 24 steps per run on average

__global__ __launch_bounds__(kWtTPB) void k_wt_expand(const u32* __restrict__ run_start,
                                                      const u8* __restrict__ run_sym, u32 n_runs,
                                                      WtTables t, const u32* __restrict__ off,
                                                      u32* __restrict__ key, u8* __restrict__ plane) {
  __shared__ u32 s_first[257];
  __shared__ u32 s_out[kWtExpandWin];
  for (u32 i = threadIdx.x; i <= t.nsec; i += kWtTPB) s_first[i] = t.first_run[i];
  __syncthreads();
  const u32 r0 = blockIdx.x * kWtTPB;
  const u32 r = r0 + threadIdx.x;
  const u32 r_end = min(r0 + (u32)kWtTPB, n_runs);
  const u32 o0 = off[r0], o1 = off[r_end];               // off[n_runs] = total
  const bool staged = o1 - o0 <= kWtExpandWin;
  // plane: the sort's first digit of every step word (radix_sort.hpp, digit planes)
  auto put = [&](u32 o, u32 v) { if (staged) s_out[o - o0] = v; else { key[o] = v; plane[o] = (u8)(v >> kStepLeafShift); } };
  if (r < n_runs) {
    const u32 s = wt_section_of(s_first, t.nsec, r);
    const u32 c = run_sym[r];
    const uint2 sy = t.symtab[s * 256u + c];
    // steps at depth > (common prefix with the previous run's code) carry the gap flag
    u32 common = 0;
    if (r > s_first[s]) common = (u32)__clzll(t.symcode[s * 256u + c] ^ t.symcode[s * 256u + run_sym[r - 1]]);
    u32 o = off[r];
    const u32 ns = sy.y & 255u;
    for (u32 i = 0; i < ns; ++i, ++o) put(o, t.pool[sy.x + i] | ((i > common ? 1u : 0u) << kStepGapShift));
    if (sy.y >> 16) {
      const uint2 le = wt_length_entry(t, s, run_start[r + 1] - run_start[r]);
      const u32 leaf = ((sy.y >> 8) & 255u) << kStepLeafShift;
      for (u32 i = 0; i < le.y; ++i, ++o) put(o, t.pool[le.x + i] | leaf);
    }
  }
  if (!staged) return;
  __syncthreads();
  for (u32 i = threadIdx.x; i < o1 - o0; i += kWtTPB) {
    const u32 v = s_out[i];
    key[o0 + i] = v;
    plane[o0 + i] = (u8)(v >> kStepLeafShift);
  }
}

constexpr u8 kWtSkip = 0xFF;

// sorted order: what encodeTreeBF does with the bit -> code[j] = bit | gap << 1, or kWtSkip;
// coded elements per tile -> tile_count; first element of every group -> group_start
__global__ __launch_bounds__(kWtTPB) void k_wt_select(const u32* __restrict__ key, u32 n,
                                                      const u8* __restrict__ group_type,
                                                      u8* __restrict__ code,
                                                      u32* __restrict__ tile_count,
                                                      u32* __restrict__ group_start) {
  __shared__ u32 scratch[kWtTPB / kWave + 1];
  const u32 base = blockIdx.x * kWtTile;
  u32 coded = 0;
  for (int e = 0; e < kWtE; ++e) {
    const u32 j = base + e * kWtTPB + threadIdx.x;
    if (j >= n) break;
    const u32 k = key[j];
    const u32 g = (k >> kStepGroupShift) & kStepGroupMask;
    const u32 kp = j ? key[j - 1] : 0u;
    const bool first = j == 0 || ((kp >> kStepGroupShift) & kStepGroupMask) != g;
    if (first) group_start[g] = j;
    const u32 bit = k & 1u;
    const u8 type = group_type[g];
    u8 c;
    if (type == bwtc::wavelet::kRoot || type == bwtc::wavelet::kInteger) {
      c = (u8)bit;
    } else {
      const u32 gap = (k >> kStepGapShift) & 1u;
      const u32 prev = first ? (bit ^ 1u) : (kp & 1u);
      bool keep = true;
      if (type == bwtc::wavelet::kBothLeaves) keep = gap != 0;
      else if (type == bwtc::wavelet::kLeftLeaf) keep = (prev | gap) != 0;
      c = keep ? (u8)(bit | (gap << 1)) : kWtSkip;
    }
    code[j] = c;
    coded += c != kWtSkip;
  }
  u32 total;
  block_scan_excl_add<kWtTPB>(coded, scratch, &total);
  if (threadIdx.x == 0) tile_count[blockIdx.x] = total;
}

// keeps the coded elements, in order, one byte each
__global__ __launch_bounds__(kWtTPB) void k_wt_compact(const u8* __restrict__ code, u32 n,
                                                       const u32* __restrict__ tile_off,
                                                       u8* __restrict__ out) {
  __shared__ u32 scratch[kWtTPB / kWave + 1];
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
  u32 o = tile_off[blockIdx.x] + block_scan_excl_add<kWtTPB>(mine, scratch, &total);
#pragma unroll
  for (int e = 0; e < kWtE; ++e) if (c[e] != kWtSkip) out[o++] = c[e];
}

// 16 coded elements -> one 32-bit word, 2 bits each, element i in bits 2i..2i+1
__global__ __launch_bounds__(kWtTPB) void k_wt_pack(const u8* __restrict__ in, u32 n,
                                                    u32* __restrict__ out) {
  const u32 w = blockIdx.x * kWtTPB + threadIdx.x;
  const u32 base = w * 16u;
  if (base >= n) return;
  u32 word = 0;
  if (base + 16u <= n) {
    const uint4 v = *reinterpret_cast<const uint4*>(in + base);
    const u32 q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 16; ++e) word |= ((q[e >> 2] >> (8 * (e & 3))) & 3u) << (2 * e);
  } else {
    for (u32 e = 0; base + e < n; ++e) word |= (u32)(in[base + e] & 3u) << (2 * e);
  }
  out[w] = word;
}

// coded elements before the first element of every group
__global__ __launch_bounds__(kWtTPB) void k_wt_group_pos(const u32* __restrict__ group_start,
                                                         u32 n_groups, u32 n,
                                                         const u8* __restrict__ code,
                                                         const u32* __restrict__ tile_off,
                                                         u32* __restrict__ coded_pos) {
  const u32 g = blockIdx.x * kWtTPB + threadIdx.x;
  if (g >= n_groups) return;
  const u32 j = group_start[g];
  if (j >= n) { coded_pos[g] = 0xFFFFFFFFu; return; }     // empty group, the host fills it in
  const u32 tile = j / kWtTile;
  u32 c = tile_off[tile];
  for (u32 i = tile * kWtTile; i < j; ++i) c += code[i] != kWtSkip;
  coded_pos[g] = c;
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

// Runs of the block are in d_run_start / d_run_sym as wavelet_section_stats_device left them.  On success coded_pos has plan.group_type.size() + 1 entries and `codes` holds the
// packed elements.
int wavelet_streams_device(BwtEngine& e, const u32* d_run_start, const u8* d_run_sym, u32 n_runs, const std::vector<u32>& first_run,
                           const bwtc::wavelet::StreamPlan& plan, std::vector<u32>* coded_pos,
                           PinnedBytes* codes, bool async_copy) {
  hipStream_t st = e.stream;
  const u32 nsec = (u32)plan.sections.size();
  const u32 n_groups = (u32)plan.group_type.size();
  if (nsec == 0 || nsec > 256 || first_run.size() != nsec + 1 || n_runs == 0) return -1;
  BWTC_HIP_TRY(e.codes_wait());                       // the previous block's packed streams have left the device
  if (plan.max_elements + kWtTile >= (1ull << 32)) return -5;       // step indices are 32 bit
  BWTC_HIP_TRY(hipSetDevice(e.device));
  const u64 cap = plan.max_elements;
  const int key_bits = (int)kStepGroupShift + bit_width_u64(n_groups ? n_groups - 1 : 0);

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
  const u64 tables_end = at;
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
  int rc = e.reserve_wavelet(at, tables_end);
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

  // expand
  u32* d_cnt = ptr32(o_cnt);
  hipLaunchKernelGGL(k_wt_count, dim3(ceil_div((u64)n_runs + 1, kWtTPB)), dim3(kWtTPB), 0, st,
                     d_run_start, d_run_sym, n_runs, t, d_cnt);
  exclusive_scan_u32(d_cnt, (u64)n_runs + 1, ptr32(o_cnt_partial), st);
  BWTC_HIP_TRY(hipMemcpyAsync(e.h_small, d_cnt + n_runs, 4, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(e.wait());
  const u32 n = e.h_small[0];
  if (n == 0 || (u64)n > cap) return -3;
  hipLaunchKernelGGL(k_wt_expand, dim3(ceil_div(n_runs, kWtTPB)), dim3(kWtTPB), 0, st, d_run_start,
                     d_run_sym, n_runs, t, d_cnt, ptr32(o_k0), base + o_code);
  // sort by (group, leaf rank), keys only; the bit and the gap flag ride along in the word
  u32* ks = nullptr; u32* vs = nullptr;
  radix_sort_pairs<u32>(ptr32(o_k0), ptr32(o_k1), nullptr, nullptr, n, key_bits,
                        ptr32(o_table), ptr32(o_partial), st, &ks, &vs, nullptr, (int)kStepLeafShift, false, true,
                        0, base + o_code, base + o_compact, true);     // planes: both byte arrays are free until the select kernel
  // select + compact + pack
  u32* d_gstart = ptr32(o_gstart);
  BWTC_HIP_TRY(hipMemsetAsync(d_gstart, 0xFF, (n_groups + 1) * 4ull, st));
  const u32 tiles = ceil_div(n, kWtTile);
  u8* d_code = base + o_code;
  u32* d_tile = ptr32(o_tile);
  hipLaunchKernelGGL(k_wt_select, dim3(tiles), dim3(kWtTPB), 0, st, ks, n, d_gtype, d_code, d_tile, d_gstart);
  BWTC_HIP_TRY(hipMemsetAsync(d_tile + tiles, 0, 4, st));
  exclusive_scan_u32(d_tile, (u64)tiles + 1, ptr32(o_tile_partial), st);
  hipLaunchKernelGGL(k_wt_group_pos, dim3(ceil_div(n_groups, kWtTPB)), dim3(kWtTPB), 0, st, d_gstart,
                     n_groups, n, d_code, d_tile, ptr32(o_gpos));
  u8* d_compact = base + o_compact;
  hipLaunchKernelGGL(k_wt_compact, dim3(tiles), dim3(kWtTPB), 0, st, d_code, n, d_tile, d_compact);
  BWTC_HIP_TRY(hipMemcpyAsync(e.h_small, d_tile + tiles, 4, hipMemcpyDeviceToHost, st));
  coded_pos->assign((size_t)n_groups + 1, 0);
  BWTC_HIP_TRY(hipMemcpyAsync(coded_pos->data(), ptr32(o_gpos), n_groups * 4ull, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(e.wait());
  const u32 n_coded = e.h_small[0];
  if (n_coded > n) return -3;
  const u32 words = ceil_div(n_coded, 16);
  if (!codes->reserve(words * 4ull + 16)) return -2;
  if (words) {
    hipLaunchKernelGGL(k_wt_pack, dim3(ceil_div(words, kWtTPB)), dim3(kWtTPB), 0, st, d_compact, n_coded,
                       ptr32(o_packed));
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
  e.wt_elements = n;
  e.wt_coded = n_coded;
  return 0;
}

}  // namespace bwtc_hip
