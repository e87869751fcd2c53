// The adaptive models of the 'B' coder on the GPU: the lane functions of wavelet_gpu_models.hpp as
// kernels, one lane per chunk (element passes) or per slot-chunk (slot-space passes).  Input: the
// block's packed streams as wavelet_tree.hip left them in HBM; output: one 16-bit word per coded
// element (bit, probability of the coded bit), copied to the host for the range coders.
// A block of 526 M coded elements is 260 K chunks: every pass has the whole chip's lanes busy
// with short serial walks; the passes are bound by instruction issue and LDS latency, not by HBM
// (they read 0.13 GB of streams a few times and write 1 GB once).
#include "bwt_engine.hpp"
#include "scan.hpp"
#include "wavelet_gpu_models.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace bwtc_hip {

namespace gm = bwtc::wavelet::gm;

constexpr int kGmTPB = 256;
constexpr int kGmStateTPB = 1024;
constexpr int kGmSmallEnds = 960;                   // words of the engine's small page-locked mirror used here
constexpr int kGmStateE = 8;

__global__ __launch_bounds__(kGmTPB) void k_gm_map(const u32* __restrict__ packed, const gm::Chunk* __restrict__ chunks,
                                                   const gm::Task* __restrict__ tasks, const u32* __restrict__ order, u32 nc,
                                                   u64* __restrict__ cmap) {
  const u32 gid = blockIdx.x * kGmTPB + threadIdx.x;
  if (gid >= nc) return;
  const u32 c = order[gid];                          // lanes in type order: a wave walks one kind of loop
  const gm::Chunk ch = chunks[c];
  cmap[c] = gm::laneMap(packed, ch.begin, ch.end, tasks[ch.task_first & 0x7FFFFFFFu].type);
}

// Composition scan of the chunks' maps in coding order -> every chunk's start states, in three
// launches: per tile of 8192 chunks the threads' exclusive prefixes inside the tile and the tile's
// map (many workgroups), a serial walk over the tiles (a map application each), and the walk of
// every thread over its eight chunks from its true start state.
__global__ __launch_bounds__(kGmStateTPB) void k_gm_state_tile(const gm::Chunk* __restrict__ chunks, const u64* __restrict__ cmap,
                                                               u32 nc, u64* __restrict__ excl, u64* __restrict__ tagg) {
  __shared__ u64 s_agg[kGmStateTPB];
  const u32 tid = threadIdx.x;
  const u32 c0 = blockIdx.x * (kGmStateTPB * kGmStateE) + tid * kGmStateE;
  u64 agg = gm::kMapIdentity;
#pragma unroll
  for (int i = 0; i < kGmStateE; ++i) {
    if (c0 + i < nc) {
      if (chunks[c0 + i].task_first >> 31) agg = gm::mapConstGapsInts(agg, 2, 1);
      agg = gm::mapCompose(agg, cmap[c0 + i]);
    }
  }
  s_agg[tid] = agg;
  __syncthreads();
  for (u32 off = 1; off < kGmStateTPB; off <<= 1) {
    const u64 v = tid >= off ? gm::mapCompose(s_agg[tid - off], agg) : agg;
    __syncthreads();
    s_agg[tid] = agg = v;
    __syncthreads();
  }
  excl[blockIdx.x * kGmStateTPB + tid] = tid ? s_agg[tid - 1] : gm::kMapIdentity;
  if (tid == kGmStateTPB - 1) tagg[blockIdx.x] = agg;
}
__global__ void k_gm_state_top(const u64* __restrict__ tagg, u32 ntiles, u32 state_in, u32* __restrict__ tstate,
                               u32* __restrict__ state_out) {
  if (threadIdx.x || blockIdx.x) return;
  u32 st = gm::packState(state_in & 7u, 2, 1);
  for (u32 t = 0; t < ntiles; ++t) { tstate[t] = st; st = gm::mapApply(tagg[t], st); }
  *state_out = st & 7u;
}
// ends[i] = the carried state after the block when it starts in state i: known before the state is
// (a stream farmed over contexts hands the state on without waiting for this block's passes)
__global__ void k_gm_state_ends(const u64* __restrict__ tagg, u32 ntiles, u32* __restrict__ ends) {
  if (blockIdx.x || threadIdx.x >= 8) return;
  u32 st = gm::packState(threadIdx.x, 2, 1);
  for (u32 t = 0; t < ntiles; ++t) st = gm::mapApply(tagg[t], st);
  ends[threadIdx.x] = st & 7u;
}
__global__ __launch_bounds__(kGmStateTPB) void k_gm_state_apply(const gm::Chunk* __restrict__ chunks, const u64* __restrict__ cmap,
                                                                u32 nc, const u64* __restrict__ excl, const u32* __restrict__ tstate,
                                                                u32* __restrict__ cstate) {
  const u32 tid = threadIdx.x;
  const u32 c0 = blockIdx.x * (kGmStateTPB * kGmStateE) + tid * kGmStateE;
  if (c0 >= nc) return;
  u32 st = gm::mapApply(excl[blockIdx.x * kGmStateTPB + tid], tstate[blockIdx.x]);
#pragma unroll
  for (int i = 0; i < kGmStateE; ++i) {
    if (c0 + i < nc) {
      if (chunks[c0 + i].task_first >> 31) st = gm::packState(st & 7u, 2, 1);
      cstate[c0 + i] = st;
      st = gm::mapApply(cmap[c0 + i], st);
    }
  }
}

__global__ __launch_bounds__(kGmTPB) void k_gm_count(const u32* __restrict__ packed, const gm::Chunk* __restrict__ chunks,
                                                     const gm::Task* __restrict__ tasks, const u32* __restrict__ cstate,
                                                     const u32* __restrict__ order, u32 nc, u32* __restrict__ cnt) {
  __shared__ u32 tab[gm::kSlotStride][kGmTPB];
  const u32 gid = blockIdx.x * kGmTPB + threadIdx.x;
  if (gid >= nc) return;
  const u32 c = order[gid];
  const gm::Chunk ch = chunks[c];
  gm::laneCount(packed, ch.begin, ch.end, tasks[ch.task_first & 0x7FFFFFFFu].type, cstate[c], &tab[0][threadIdx.x], kGmTPB);
  for (u32 k = 0; k < gm::kSlots; ++k) cnt[(u64)k * nc + c] = tab[k][threadIdx.x];
}

// sb[k * nt + t] = where stream (slot k, task t) starts in slot space; sb[15 nt] = total
__global__ __launch_bounds__(kGmTPB) void k_gm_streams(const u32* __restrict__ base, const gm::Task* __restrict__ tasks,
                                                       u32 nc, u32 nt, u32* __restrict__ sb) {
  const u32 i = blockIdx.x * kGmTPB + threadIdx.x;
  if (i > gm::kSlots * nt) return;
  if (i == gm::kSlots * nt) { sb[i] = base[(u64)gm::kSlots * nc]; return; }
  const u32 k = i / nt, t = i - k * nt;
  sb[i] = base[(u64)k * nc + tasks[t].first_chunk];
}

__global__ __launch_bounds__(kGmTPB) void k_gm_partition(const u32* __restrict__ packed, const gm::Chunk* __restrict__ chunks,
                                                         const gm::Task* __restrict__ tasks, const u32* __restrict__ cstate,
                                                         const u32* __restrict__ base, const u32* __restrict__ order, u32 nc,
                                                         u32* __restrict__ sbits) {
  __shared__ u32 pos[gm::kSlotStride][kGmTPB];
  __shared__ u32 acc[gm::kSlotStride][kGmTPB];
  const u32 gid = blockIdx.x * kGmTPB + threadIdx.x;
  if (gid >= nc) return;
  const u32 c = order[gid];
  const gm::Chunk ch = chunks[c];
  for (u32 k = 0; k < gm::kSlots; ++k) { pos[k][threadIdx.x] = base[(u64)k * nc + c]; acc[k][threadIdx.x] = 0; }
  gm::lanePartition(packed, ch.begin, ch.end, tasks[ch.task_first & 0x7FFFFFFFu].type, cstate[c], &pos[0][threadIdx.x],
                        &acc[0][threadIdx.x], kGmTPB, sbits);
}

// The same pass with whole lines leaving the workgroup (BWTC_HIP_GM_PARTITION=lines; measured, not the default): its
// chunks are CONSECUTIVE (no type order here), so in every slot its bits fill one stretch of slot space.  The
// stretches are assembled in LDS -- a lane takes its next position with one returning ds_add and sets a one-bit with
// a ds_or -- and stored word by word, consecutive lanes consecutive words; only a stretch's first and last word are
// shared with the neighbouring workgroups and OR-ed in.  k_gm_partition above lets every lane store its own words as
// they fill: a line of slot space is completed by eight lanes at different times while 130 000 lanes hold fifteen
// open lines each, so lines leave the L2 part-filled and come back (8 x the bytes).  But the pass is
// bound by instruction issue, not by those bytes: this form needs 32 KiB of LDS per 128 lanes (two waves per SIMD
// instead of five) and takes 0.99 ms per 256 MiB text block against 0.63; non-temporal word stores in the first
// form: 0.82 ms.
constexpr int kGmPartTPB = 128;
constexpr u32 kGmPartWords = kGmPartTPB * gm::kChunk / 32 + 32;     // every slot's stretch may start and end inside a word

template <u32 TYPE>
__device__ __forceinline__ void gm_partition_walk(const u32* __restrict__ packed, u32 begin, u32 end, u32 state,
                                                  u32* pos, u32* region) {
  gm::Machines m = {state & 7u, (state >> 3) & 3u, (state >> 5) & 3u};
  gm::forElements(packed, begin, end, [&](u32, u32 v) {
    const u32 slot = gm::stepMachinesT<TYPE>(v, m);
    const u32 p = __hip_atomic_fetch_add(&pos[slot * kGmPartTPB], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (v & 1u) (void)__hip_atomic_fetch_or(&region[p >> 5], 1u << (p & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  });
}

__global__ __launch_bounds__(kGmPartTPB) void k_gm_partition_lines(const u32* __restrict__ packed, const gm::Chunk* __restrict__ chunks,
                                                                   const gm::Task* __restrict__ tasks, const u32* __restrict__ cstate,
                                                                   const u32* __restrict__ base, u32 nc, u32* __restrict__ sbits) {
  __shared__ u32 region[kGmPartWords];
  __shared__ u32 pos[gm::kSlots][kGmPartTPB];
  __shared__ u32 s_g0[gm::kSlots], s_nw[gm::kSlots], s_off[gm::kSlots + 1];
  const u32 tid = threadIdx.x;
  const u32 c0 = blockIdx.x * kGmPartTPB, c1 = min(c0 + (u32)kGmPartTPB, nc);
  const u32 c = c0 + tid;
  for (u32 i = tid; i < kGmPartWords; i += kGmPartTPB) region[i] = 0;
  if (tid < gm::kSlots) {
    const u32 g0 = base[(u64)tid * nc + c0], g1 = base[(u64)tid * nc + c1];     // base[k nc + nc] = base[(k + 1) nc]: the array is one prefix
    s_g0[tid] = g0 >> 5;
    s_nw[tid] = g1 > g0 ? ((g1 - 1u) >> 5) - (g0 >> 5) + 1u : 0u;
  }
  __syncthreads();
  if (tid == 0) {
    u32 at = 0;
    for (u32 k = 0; k < gm::kSlots; ++k) { s_off[k] = at; at += s_nw[k]; }
    s_off[gm::kSlots] = at;
  }
  __syncthreads();
  if (s_off[gm::kSlots] > kGmPartWords) return;          // cannot happen (see kGmPartWords); the block then fails its element count
  if (c < nc) {
    const gm::Chunk ch = chunks[c];
    for (u32 k = 0; k < gm::kSlots; ++k) pos[k][tid] = (s_off[k] << 5) + (base[(u64)k * nc + c] - (s_g0[k] << 5));
    const u32 type = tasks[ch.task_first & 0x7FFFFFFFu].type;
    const u32 st = cstate[c];
    using namespace gm;
    BWTC_GM_BY_TYPE(type, gm_partition_walk<TYPE>(packed, ch.begin, ch.end, st, &pos[0][tid], region));
  }
  __syncthreads();
  for (u32 k = 0; k < gm::kSlots; ++k) {
    const u32 nw = s_nw[k], off = s_off[k], g0 = s_g0[k];
    for (u32 w = tid; w < nw; w += kGmPartTPB) {
      const u32 v = region[off + w];
      if (w == 0 || w + 1u == nw) { if (v) atomicOr(&sbits[g0 + w], v); }
      else sbits[g0 + w] = v;
    }
  }
}

__global__ __launch_bounds__(kGmTPB) void k_gm_bracket(const u32* __restrict__ sbits, const u32* __restrict__ sb, u32 ns, u32 nt,
                                                       u32 total, u32 nsc, gm::SlotMap* __restrict__ smap,
                                                       gm::SlotMap* __restrict__ snaps, u32* __restrict__ err) {
  const u32 j = blockIdx.x * kGmTPB + threadIdx.x;
  if (j >= nsc) return;
  u32 e = 0;
  smap[j] = gm::laneBracket(sbits, sb, ns, nt, total, j, snaps, &e);
  if (e) atomicOr(err, e);
}

// level (a): one wave per group, lane = candidate of the group's first bracket
__global__ __launch_bounds__(64) void k_gm_chain_group(const gm::SlotMap* __restrict__ smap, u32 nsc, u32 gsize,
                                                       unsigned short* __restrict__ gmap, unsigned short* __restrict__ gL) {
  const u32 g = blockIdx.x, cand = threadIdx.x;
  if (cand >= 32) return;
  gmap[g * 32u + cand] = (unsigned short)gm::laneChainGroup(smap, nsc, gsize, g, cand);
  if (cand == 0) gL[g] = (unsigned short)(smap[g * gsize].lo_x0 & 0xFFFFu);
}
// level (b): the groups' maps into LDS, one thread walks them
__global__ __launch_bounds__(1024) void k_gm_chain_top(const unsigned short* __restrict__ gmap, const unsigned short* __restrict__ gL,
                                                       u32 ng, unsigned short* __restrict__ tg, u32* __restrict__ err) {
  __shared__ unsigned short s_map[gm::kChainGroups * 32];
  __shared__ unsigned short s_L[gm::kChainGroups], s_t[gm::kChainGroups];
  for (u32 i = threadIdx.x; i < ng * 32u; i += 1024) s_map[i] = gmap[i];
  for (u32 i = threadIdx.x; i < ng; i += 1024) s_L[i] = gL[i];
  __syncthreads();
  if (threadIdx.x == 0) { u32 e = 0; gm::laneChainTop(s_map, s_L, ng, s_t, &e); if (e) atomicOr(err, e); }
  __syncthreads();
  for (u32 i = threadIdx.x; i < ng; i += 1024) tg[i] = s_t[i];
}
// level (c): lane per group
__global__ __launch_bounds__(64) void k_gm_chain_fill(const gm::SlotMap* __restrict__ smap, u32 nsc, u32 gsize, u32 ng,
                                                      const unsigned short* __restrict__ tg, unsigned short* __restrict__ sstart,
                                                      u32* __restrict__ err) {
  const u32 g = blockIdx.x * 64 + threadIdx.x;
  if (g >= ng) return;
  u32 e = 0;
  gm::laneChainFill(smap, nsc, gsize, g, tg, sstart, &e);
  if (e) atomicOr(err, e);
}

__global__ __launch_bounds__(kGmTPB) void k_gm_emit(const u32* __restrict__ packed, const gm::Chunk* __restrict__ chunks,
                                                    const gm::Task* __restrict__ tasks, const u32* __restrict__ cstate,
                                                    const u32* __restrict__ base, const u32* __restrict__ sb,
                                                    const u32* __restrict__ sbits, const gm::SlotMap* __restrict__ snaps,
                                                    const gm::SlotMap* __restrict__ smap, const unsigned short* __restrict__ sstart,
                                                    const u32* __restrict__ order, u32 nc, u32 nt, unsigned short* __restrict__ out,
                                                    u32* __restrict__ err) {
  __shared__ u32 q[gm::kSlotStride][kGmTPB];
  __shared__ u32 stage[32][kGmTPB];                  // a lane's line of output (64 elements) on its way out
  const u32 gid = blockIdx.x * kGmTPB + threadIdx.x;
  if (gid >= nc) return;
  const u32 c = order[gid];
  const gm::Chunk ch = chunks[c];
  const u32 t = ch.task_first & 0x7FFFFFFFu;
  u32 e = 0;
  gm::laneEmit(packed, ch.begin, ch.end, tasks[t].type, cstate[c], t, c, nc, nt, base, sb, sbits, snaps, smap, sstart,
               &q[0][threadIdx.x], kGmTPB, out, &e, &stage[0][threadIdx.x]);
  if (e) atomicOr(err, e);
}

static inline u64 gm_align(u64 v) { return (v + 255) / 256 * 256; }

int BwtEngine::reserve_models(u64 device_bytes, u64 host_bytes, u64 w_bytes) {
  if (device_bytes > gm_bytes) {
    if (d_gm) (void)hipFree(d_gm);
    d_gm = nullptr; gm_bytes = 0;
    const u64 want = gm_align(device_bytes + device_bytes / 8);
    BWTC_HIP_TRY(hipMalloc(&d_gm, want));
    gm_bytes = want;
  }
  if (host_bytes > h_gm_bytes) {
    if (h_gm) (void)hipHostFree(h_gm);
    h_gm = nullptr; h_gm_bytes = 0;
    const u64 want = gm_align(host_bytes + host_bytes / 8);
    BWTC_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_gm), want, hipHostMallocDefault));
    h_gm_bytes = want;
  }
  if (w_bytes > gm_w_bytes) {
    if (d_gm_w) (void)hipFree(d_gm_w);
    d_gm_w = nullptr; gm_w_bytes = 0;
    const u64 want = gm_align(w_bytes + w_bytes / 8);
    BWTC_HIP_TRY(hipMalloc(&d_gm_w, want));
    gm_w_bytes = want;
  }
  return 0;
}

// The passes in two parts, so that a stream farmed over several contexts can do everything that does
// not depend on the blocks before this one early (wavelet_models_prepare: tables, chunk maps, the
// tiles' state maps) and the rest once the carried state is known (wavelet_models_run).
// d_packed: the block's packed streams (n_coded elements) in HBM.  Nothing is waited for, except:
// early_state != null -> the state after the block is read back right after the state scan (a wait
// of some tens of microseconds) and returned, before the long passes are queued.
int wavelet_models_prepare(BwtEngine& e, const u32* d_packed, u32 n_coded, const bwtc::wavelet::StreamPlan& plan,
                           const std::vector<u32>& coded_pos, GmPass* g, bool read_ends, hipStream_t side) {
  BwtEngine::ScanScope scan_scope(e);
  hipStream_t st = side ? side : e.stream;             // side: the passes run beside the next block's transform (wavelet_models_device)
  g->side = side;
  std::vector<gm::Task> tasks;
  std::vector<gm::Chunk> chunks;
  gm::buildTasks(plan, coded_pos.data(), &tasks, &chunks);
  const u32 nt = (u32)tasks.size(), nc = (u32)chunks.size();
  if (nt == 0 || nc == 0 || n_coded == 0) return -1;
  g->d_packed = d_packed; g->n_coded = n_coded; g->nt = nt; g->nc = nc;
  g->ns = gm::kSlots * nt;
  g->nsc = ceil_div(n_coded, gm::kSlotChunk);
  g->n_base = (u64)gm::kSlots * nc + 1;
  g->ntiles = ceil_div(nc, kGmStateTPB * kGmStateE);
  g->gsize = gm::chainGroupSize(g->nsc);
  g->ng = ceil_div(g->nsc, g->gsize);

  u64 at = 0;
  auto take = [&](u64 bytes) { const u64 o = at; at = gm_align(at + bytes); return o; };
  const u64 o_tasks = take((u64)nt * sizeof(gm::Task));
  const u64 o_chunks = take((u64)nc * sizeof(gm::Chunk));
  const u64 o_order = take((u64)nc * 4);
  const u64 tables_end = at;
  const u64 o_excl = take((u64)g->ntiles * kGmStateTPB * 8), o_tagg = take((u64)g->ntiles * 8), o_tstate = take((u64)g->ntiles * 4);
  const u64 o_cmap = take((u64)nc * 8);
  const u64 o_cstate = take((u64)nc * 4);
  const u64 o_base = take(g->n_base * 4);
  const u64 o_partial = take(((u64)ceil_div(g->n_base, kScanTile) + 1) * 4);
  const u64 o_sb = take(((u64)g->ns + 1) * 4);
  const u64 o_sbits = take(((u64)n_coded / 32 + 16) * 4);
  const u64 o_smap = take((u64)g->nsc * sizeof(gm::SlotMap));
  const u64 o_sstart = take(((u64)g->nsc + 1) * 2);
  const u64 o_gmap = take((u64)g->ng * 64), o_gL = take((u64)g->ng * 2), o_tg = take((u64)g->ng * 2);
  const u64 o_snaps = take(((u64)n_coded / gm::kSample + 4) * sizeof(gm::SlotMap));
  const u64 o_tail = take(16 + 32);                   // + the eight possible states after the block
  // h_gm is one staging buffer for every block (and reserve_models may replace it): the previous block's upload
  // of its tables must have left it.  Until now that followed from waits elsewhere (the transform's read-backs).
  if (e.gm_upload_pending) {
    BWTC_HIP_TRY(hipEventSynchronize(e.ev_gm_upload));
    e.gm_upload_pending = false;
  }
  int rc = e.reserve_models(at, tables_end, ((u64)n_coded + 8) * 2);
  if (rc) return rc;
  u8* base = static_cast<u8*>(e.d_gm);
  std::memcpy(e.h_gm + o_tasks, tasks.data(), (size_t)nt * sizeof(gm::Task));
  std::memcpy(e.h_gm + o_chunks, chunks.data(), (size_t)nc * sizeof(gm::Chunk));
  {
    // lanes in type order (a wave then walks ONE kind of loop; a task's chunks stay together)
    u32* order = reinterpret_cast<u32*>(e.h_gm + o_order);
    u32 first[5] = {0, 0, 0, 0, 0};
    for (u32 c = 0; c < nc; ++c) ++first[tasks[chunks[c].task_first & 0x7FFFFFFFu].type + 1];
    for (int t = 1; t < 5; ++t) first[t] += first[t - 1];
    for (u32 c = 0; c < nc; ++c) order[first[tasks[chunks[c].task_first & 0x7FFFFFFFu].type]++] = c;
  }
  BWTC_HIP_TRY(hipMemcpyAsync(base, e.h_gm, tables_end, hipMemcpyHostToDevice, st));
  if (!e.ev_gm_upload) BWTC_HIP_TRY(hipEventCreateWithFlags(&e.ev_gm_upload, hipEventDisableTiming));
  BWTC_HIP_TRY(hipEventRecord(e.ev_gm_upload, st));
  e.gm_upload_pending = true;
  g->d_tasks = base + o_tasks; g->d_chunks = base + o_chunks; g->d_order = base + o_order;
  g->d_excl = base + o_excl; g->d_tagg = base + o_tagg; g->d_tstate = base + o_tstate;
  g->d_cmap = base + o_cmap; g->d_cstate = base + o_cstate; g->d_base = base + o_base; g->d_partial = base + o_partial;
  g->d_sb = base + o_sb; g->d_sbits = base + o_sbits; g->d_smap = base + o_smap; g->d_sstart = base + o_sstart;
  g->d_gmap = base + o_gmap; g->d_gL = base + o_gL; g->d_tg = base + o_tg; g->d_snaps = base + o_snaps; g->d_tail = base + o_tail;

  BWTC_HIP_TRY(hipMemsetAsync(g->d_tail, 0, 16, st));
  BWTC_HIP_TRY(hipMemsetAsync(g->d_sbits, 0, ((u64)n_coded / 32 + 16) * 4, st));
  BWTC_HIP_TRY(hipMemsetAsync(static_cast<u32*>(g->d_base) + g->n_base - 1, 0, 4, st));
  hipLaunchKernelGGL(k_gm_map, dim3(ceil_div(nc, kGmTPB)), dim3(kGmTPB), 0, st, d_packed, (const gm::Chunk*)g->d_chunks,
                     (const gm::Task*)g->d_tasks, (const u32*)g->d_order, nc, (u64*)g->d_cmap);
  hipLaunchKernelGGL(k_gm_state_tile, dim3(g->ntiles), dim3(kGmStateTPB), 0, st, (const gm::Chunk*)g->d_chunks, (const u64*)g->d_cmap,
                     nc, (u64*)g->d_excl, (u64*)g->d_tagg);
  if (read_ends) {
    // (the caller hands the state from context to context: it waits here, once, for its OWN device half
    // instead of making the stream's other contexts wait at _queue)
    u32* d_ends = static_cast<u32*>(g->d_tail) + 4;
    hipLaunchKernelGGL(k_gm_state_ends, dim3(1), dim3(64), 0, st, (const u64*)g->d_tagg, g->ntiles, d_ends);
    BWTC_HIP_TRY(hipMemcpyAsync(e.h_small + kGmSmallEnds, d_ends, 32, hipMemcpyDeviceToHost, st));
    BWTC_HIP_TRY(e.wait());
    for (int i = 0; i < 8; ++i) g->ends[i] = e.h_small[kGmSmallEnds + i];
    g->ends_ready = true;
  }
  g->ready = true;
  return 0;
}

int wavelet_models_run(BwtEngine& e, const GmPass& g, u32 state_in, uint16_t* h_w, u32* h_tail, u32* early_state) {
  BwtEngine::ScanScope scan_scope(e);
  hipStream_t st = g.side ? static_cast<hipStream_t>(g.side) : e.stream;
  if (!g.ready) return -1;
  const u32 nc = g.nc, nt = g.nt, ns = g.ns, nsc = g.nsc, n_coded = g.n_coded;
  const gm::Task* d_tasks = static_cast<const gm::Task*>(g.d_tasks);
  const gm::Chunk* d_chunks = static_cast<const gm::Chunk*>(g.d_chunks);
  const u32* d_order = static_cast<const u32*>(g.d_order);
  u32* d_cstate = static_cast<u32*>(g.d_cstate);
  u32* d_base = static_cast<u32*>(g.d_base);
  u32* d_sb = static_cast<u32*>(g.d_sb);
  u32* d_sbits = static_cast<u32*>(g.d_sbits);
  gm::SlotMap* d_smap = static_cast<gm::SlotMap*>(g.d_smap);
  unsigned short* d_sstart = static_cast<unsigned short*>(g.d_sstart);
  gm::SlotMap* d_snaps = static_cast<gm::SlotMap*>(g.d_snaps);
  u32* d_tail = static_cast<u32*>(g.d_tail);                        // [0] state after the block, [1] error flags, [2] elements counted, [3] scan error
  unsigned short* d_w = static_cast<unsigned short*>(e.d_gm_w);
  const dim3 gc(ceil_div(nc, kGmTPB)), gs(ceil_div(nsc, kGmTPB)), tpb(kGmTPB);
  hipLaunchKernelGGL(k_gm_state_top, dim3(1), dim3(64), 0, st, (const u64*)g.d_tagg, g.ntiles, state_in, (u32*)g.d_tstate, d_tail);
  if (early_state) {
    if (g.ends_ready) {
      *early_state = g.ends[state_in & 7u];
    } else {
      BWTC_HIP_TRY(hipMemcpyAsync(e.h_small + kGmSmallEnds, d_tail, 4, hipMemcpyDeviceToHost, st));
      BWTC_HIP_TRY(e.wait());
      *early_state = e.h_small[kGmSmallEnds];
    }
  }
  hipLaunchKernelGGL(k_gm_state_apply, dim3(g.ntiles), dim3(kGmStateTPB), 0, st, d_chunks, (const u64*)g.d_cmap, nc, (const u64*)g.d_excl,
                     (const u32*)g.d_tstate, d_cstate);
  hipLaunchKernelGGL(k_gm_count, gc, tpb, 0, st, g.d_packed, d_chunks, d_tasks, d_cstate, d_order, nc, d_base);
  exclusive_scan_u32(d_base, g.n_base, static_cast<u32*>(g.d_partial), st);
  hipLaunchKernelGGL(k_gm_streams, dim3(ceil_div((u64)ns + 1, kGmTPB)), tpb, 0, st, d_base, d_tasks, nc, nt, d_sb);
  if (e.gm_partition_lines)
    hipLaunchKernelGGL(k_gm_partition_lines, dim3(ceil_div(nc, kGmPartTPB)), dim3(kGmPartTPB), 0, st, g.d_packed, d_chunks, d_tasks, d_cstate,
                       d_base, nc, d_sbits);
  else
    hipLaunchKernelGGL(k_gm_partition, gc, tpb, 0, st, g.d_packed, d_chunks, d_tasks, d_cstate, d_base, d_order, nc, d_sbits);
  hipLaunchKernelGGL(k_gm_bracket, gs, tpb, 0, st, d_sbits, d_sb, ns, nt, n_coded, nsc, d_smap, d_snaps, d_tail + 1);
  hipLaunchKernelGGL(k_gm_chain_group, dim3(g.ng), dim3(64), 0, st, d_smap, nsc, g.gsize, (unsigned short*)g.d_gmap, (unsigned short*)g.d_gL);
  hipLaunchKernelGGL(k_gm_chain_top, dim3(1), dim3(1024), 0, st, (const unsigned short*)g.d_gmap, (const unsigned short*)g.d_gL, g.ng,
                     (unsigned short*)g.d_tg, d_tail + 1);
  hipLaunchKernelGGL(k_gm_chain_fill, dim3(ceil_div(g.ng, 64)), dim3(64), 0, st, d_smap, nsc, g.gsize, g.ng, (const unsigned short*)g.d_tg,
                     d_sstart, d_tail + 1);
  // d_gm_w is one buffer for every block: the copy of the previous block's w-elements (d2h stream, ev_codes) must have
  // read it before this block's emit pass writes it (so far the callers' codes_wait saw to that)
  if (e.codes_in_flight && e.ev_codes) BWTC_HIP_TRY(hipStreamWaitEvent(st, e.ev_codes, 0));
  hipLaunchKernelGGL(k_gm_emit, gc, tpb, 0, st, g.d_packed, d_chunks, d_tasks, d_cstate, d_base, d_sb, d_sbits, d_snaps, d_smap,
                     d_sstart, d_order, nc, nt, d_w, d_tail + 1);
  // total of the scan = every element counted once (else the tables do not describe the streams)
  BWTC_HIP_TRY(hipMemcpyAsync(d_tail + 2, d_base + g.n_base - 1, 4, hipMemcpyDeviceToDevice, st));
  if (e.scan_chain.err) BWTC_HIP_TRY(hipMemcpyAsync(d_tail + 3, e.scan_chain.err, 4, hipMemcpyDeviceToDevice, st));   // a timed-out scan
  BWTC_HIP_TRY(e.ensure_d2h_stream());
  BWTC_HIP_TRY(hipEventRecord(e.ev_models, st));
  BWTC_HIP_TRY(hipStreamWaitEvent(e.d2h_stream, e.ev_models, 0));
  BWTC_HIP_TRY(hipMemcpyAsync(h_w, d_w, (u64)n_coded * 2, hipMemcpyDeviceToHost, e.d2h_stream));
  BWTC_HIP_TRY(hipMemcpyAsync(h_tail, d_tail, 16, hipMemcpyDeviceToHost, e.d2h_stream));
  BWTC_HIP_TRY(hipEventRecord(e.ev_codes, e.d2h_stream));
  e.codes_in_flight = true;
  if (std::getenv("BWTC_HIP_DEBUG"))
    std::fprintf(stderr, "models on the device: %u coded elements, %u tasks, %u chunks, %u slot-chunks, state in %u\n",
                 n_coded, nt, nc, nsc, state_in);
  return 0;
}

int wavelet_models_device(BwtEngine& e, const u32* d_packed, u32 n_coded, const bwtc::wavelet::StreamPlan& plan,
                          const std::vector<u32>& coded_pos, u32 state_in, uint16_t* h_w, u32* h_tail) {
  // One context's stream of blocks (the _begin flow): nothing here waits for the device, so the passes -- 2.9 ms of
  // kernels per 256 MiB text block that share nothing with the suffix sorter but HBM -- go to a stream of their own and
  // run beside the NEXT block's transform.  Ordered by events: they start when this block's packed streams are there
  // (the main stream's position now); the main stream takes the workspace of the packed streams back only when the emit
  // pass has read them (ev_gm_done, waited for by the next block's stream kernels: wavelet_streams_device).
  hipStream_t side = nullptr;
  if (e.models_side_stream) {
    if (!e.gm_stream) {
      BWTC_HIP_TRY(hipStreamCreateWithFlags(&e.gm_stream, hipStreamNonBlocking));
      BWTC_HIP_TRY(hipEventCreateWithFlags(&e.ev_packed_ready, hipEventDisableTiming));
      BWTC_HIP_TRY(hipEventCreateWithFlags(&e.ev_gm_done, hipEventDisableTiming));
    }
    BWTC_HIP_TRY(hipEventRecord(e.ev_packed_ready, e.stream));
    BWTC_HIP_TRY(hipStreamWaitEvent(e.gm_stream, e.ev_packed_ready, 0));
    side = e.gm_stream;
  }
  GmPass g;
  int rc = wavelet_models_prepare(e, d_packed, n_coded, plan, coded_pos, &g, false, side);
  if (rc == 0) rc = wavelet_models_run(e, g, state_in, h_w, h_tail, nullptr);
  if (side) {
    // (also after an error: whatever was queued there must be through before the workspace changes hands)
    const hipError_t h = hipEventRecord(e.ev_gm_done, side);
    e.gm_done_pending = h == hipSuccess;
    if (h != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(side); }
  }
  return rc;
}

}  // namespace bwtc_hip
