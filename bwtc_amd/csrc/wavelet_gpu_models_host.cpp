// Host side of wavelet_gpu_models.hpp: the task / chunk tables of a block, and the passes run lane
// by lane on the host (what wavelet_gpu_models.hip launches as kernels), for the CPU tests.
#include "wavelet_gpu_models.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "wavelet_host.hpp"

namespace bwtc {
namespace wavelet {
namespace gm {

void buildTasks(const ::bwtc::wavelet::StreamPlan& plan, const u32* coded_pos, std::vector<Task>* tasks,
                std::vector<Chunk>* chunks) {
  tasks->clear();
  chunks->clear();
  // model tasks in coding order: per section the symbol tree's nodes, then the integer levels (one
  // model reset per level, WaveletTree.hpp:792-797) -- the order of StreamCoder's constructor
  for (size_t s = 0; s < plan.sections.size(); ++s) {
    const ::bwtc::wavelet::StreamPlan::Section& sec = plan.sections[s];
    if (sec.level_first.size() < 2) continue;
    const u32* pos = coded_pos + sec.group_base;
    for (u32 g = 0; g < sec.n_nodes; ++g) {
      if (pos[g + 1] <= pos[g]) continue;
      const u32 gt = plan.group_type[sec.group_base + g];
      const u32 type = gt == kRoot ? kTRoot : gt == kBothLeaves ? kTGaps : kTInner;
      tasks->push_back(Task{pos[g], pos[g + 1], type, 0});
    }
    for (size_t l = 0; l + 1 < sec.level_first.size(); ++l) {
      const u32 b = pos[sec.level_first[l]], e = pos[sec.level_first[l + 1]];
      if (e > b) tasks->push_back(Task{b, e, kTInts, 0});
    }
  }
  for (size_t t = 0; t < tasks->size(); ++t) {
    Task& task = (*tasks)[t];
    task.first_chunk = static_cast<u32>(chunks->size());
    for (u32 b = task.begin; b < task.end;) {
      const u32 cell_end = (b / kChunk + 1u) * kChunk;
      const u32 e = std::min(task.end, cell_end);
      chunks->push_back(Chunk{b, e, static_cast<u32>(t) | (b == task.begin ? 1u << 31 : 0u)});
      b = e;
    }
  }
}

bool modelsOnHostLanes(const u32* packed, u32 total, const std::vector<Task>& tasks, const std::vector<Chunk>& chunks,
                       u32* state, unsigned short* out) {
  const u32 nc = static_cast<u32>(chunks.size()), nt = static_cast<u32>(tasks.size());
  if (nc == 0 || nt == 0) return true;
  u32 err = 0;
  // pass 1 + the composition scan (the kernel composes per thread, scans the aggregates and
  // applies; here: aggregates of eight chunks, checked against the plain walk)
  std::vector<u64> cmap(nc);
  for (u32 c = 0; c < nc; ++c) cmap[c] = laneMap(packed, chunks[c].begin, chunks[c].end, tasks[chunks[c].task_first & 0x7FFFFFFFu].type);
  std::vector<u32> cstate(nc);
  u32 st = packState(*state & 7u, 2, 1);
  for (u32 c0 = 0; c0 < nc; c0 += 8) {
    u64 agg = kMapIdentity;
    u32 walk = st;
    for (u32 c = c0; c < std::min(nc, c0 + 8); ++c) {
      if (chunks[c].task_first >> 31) { agg = mapConstGapsInts(agg, 2, 1); walk = packState(walk & 7u, 2, 1); }
      cstate[c] = walk;
      agg = mapCompose(agg, cmap[c]);
      walk = mapApply(cmap[c], walk);
    }
    if (mapApply(agg, st) != walk) return false;
    st = walk;
  }
  *state = st & 7u;
  // pass 2 + exclusive scan: slot space is slot-major, then chunk
  std::vector<u32> base(static_cast<size_t>(kSlots) * nc + 1, 0);
  {
    u32 tab[kSlotStride];
    for (u32 c = 0; c < nc; ++c) {
      laneCount(packed, chunks[c].begin, chunks[c].end, tasks[chunks[c].task_first & 0x7FFFFFFFu].type, cstate[c], tab, 1);
      for (u32 k = 0; k < kSlots; ++k) base[static_cast<size_t>(k) * nc + c] = tab[k];
    }
    u32 run = 0;
    for (size_t i = 0; i < base.size(); ++i) { const u32 v = base[i]; base[i] = run; run += v; }
    if (run != total) return false;
  }
  // streams
  const u32 ns = kSlots * nt;
  std::vector<u32> sb(ns + 1);
  for (u32 k = 0; k < kSlots; ++k)
    for (u32 t = 0; t < nt; ++t) sb[k * nt + t] = base[static_cast<size_t>(k) * nc + tasks[t].first_chunk];
  sb[ns] = total;
  // pass 3
  std::vector<u32> sbits(total / 32 + 16, 0);
  {
    u32 pos[kSlotStride], acc[kSlotStride];
    for (u32 c = 0; c < nc; ++c) {
      for (u32 k = 0; k < kSlots; ++k) { pos[k] = base[static_cast<size_t>(k) * nc + c]; acc[k] = 0; }
      lanePartition(packed, chunks[c].begin, chunks[c].end, tasks[chunks[c].task_first & 0x7FFFFFFFu].type, cstate[c], pos, acc, 1, sbits.data());
    }
  }
  // passes 4-6
  const u32 nsc = (total + kSlotChunk - 1) / kSlotChunk;
  std::vector<SlotMap> smap(nsc), snaps(total / kSample + 4);
  for (u32 j = 0; j < nsc; ++j) smap[j] = laneBracket(sbits.data(), sb.data(), ns, nt, total, j, snaps.data(), &err);
  if (err) return false;
  if (std::getenv("BWTC_HIP_DEBUG")) {
    u32 open_end = 0, far = 0;
    for (u32 j = 0; j < nsc; ++j) { open_end += smap[j].mask != 0; far += (j * kSlotChunk - sb[streamAt(sb.data(), ns, j * kSlotChunk)]) > kWarm; }
    std::fprintf(stderr, "model lanes: %u elements, %u tasks, %u chunks, %u slot-chunks (%u start from a bracket, %u end with an open bracket)\n",
                 total, nt, nc, nsc, far, open_end);
  }
  std::vector<unsigned short> sstart(nsc + 1, 0);
  {
    const u32 gsize = chainGroupSize(nsc), ng = (nsc + gsize - 1) / gsize;
    std::vector<unsigned short> gmap(static_cast<size_t>(ng) * 32), gL(ng), tg(ng);
    for (u32 g = 0; g < ng; ++g) {
      gL[g] = static_cast<unsigned short>(smap[g * gsize].lo_x0 & 0xFFFFu);
      for (u32 cand = 0; cand < 32; ++cand) gmap[g * 32 + cand] = static_cast<unsigned short>(laneChainGroup(smap.data(), nsc, gsize, g, cand));
    }
    laneChainTop(gmap.data(), gL.data(), ng, tg.data(), &err);
    for (u32 g = 0; g < ng; ++g) laneChainFill(smap.data(), nsc, gsize, g, tg.data(), sstart.data(), &err);
  }
  if (err) return false;
  // pass 7
  u32 q[kSlotStride], stage[32];
  for (u32 c = 0; c < nc; ++c) {
    const u32 t = chunks[c].task_first & 0x7FFFFFFFu;
    laneEmit(packed, chunks[c].begin, chunks[c].end, tasks[t].type, cstate[c], t, c, nc, nt, base.data(), sb.data(), sbits.data(),
             snaps.data(), smap.data(), sstart.data(), q, 1, out, &err, stage);
  }
  return err == 0;
}

}  // namespace gm
}  // namespace wavelet
}  // namespace bwtc
