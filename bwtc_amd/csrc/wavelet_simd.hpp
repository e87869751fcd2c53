// Stage 1 of the stream coder (the adaptive models of coder 'B'), sixteen groups at a time.
//
// One group's predictor recurrence is a serial chain, but groups are independent (fresh
// predictors at every group start, wavelet_host.hpp), so a thread steps sixteen groups in
// lockstep, one per AVX-512 lane: per step every lane reads its 2-bit element, selects its
// predictor (15 per lane, kept as 15 vectors), emits the probability, moves the predictor
// and its three small state machines.  Lanes that finish are refilled from a source that may
// span several blocks, so long and short groups share vectors; when too few lanes are left
// the remaining groups are finished with the scalar loop.  Heads and tails that do not fill a
// 16-element word of the packed streams are scalar as well.  Results are identical to
// modelGroup() by construction (same integer operations per element).
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace bwtc {
namespace wavelet {

struct ModelGroupDesc {
  const uint8_t* codes;      // packed streams of the group's block
  uint16_t* prob;            // probabilities of the group's block
  uint64_t begin, end;       // elements [begin, end)
  uint8_t type;              // GroupType
  uint32_t mainState;        // main model's state at the group's start
  void* cookie;              // handed back through done()
};

class GroupSource {
 public:
  virtual ~GroupSource() {}
  // prefer: 0 any, 1 integer-level groups, 2 symbol-tree groups; the preference is a hint (a
  // vector whose lanes are all of one kind skips the other kinds' predictors), any group may come
  virtual bool next(ModelGroupDesc* g, int prefer) = 0;     // false: none available now
  virtual void done(void* cookie) = 0;          // the group is fully modelled
  // An engine with nothing left asks before it returns; false = more groups have arrived
  virtual bool retire(int prefer) { (void)prefer; return true; }
};

// AVX-512 (F, BW, VL, DQ) present on this CPU and not switched off with BWTC_HIP_SIMD=0
bool simdModelsAvailable();

// Models groups from `src` until it is empty and every lane has drained.
void runModelLanes(GroupSource& src, int prefer = 0);

// ---- fused engine: models AND range coder, sixteen SECTIONS at a time -------------------------
// A lane walks one section's chain from its first group to its last: per element it selects the
// predictor, takes its probability, codes the bit with it (exactly BitEncoder's arithmetic,
// wavelet_rc.hpp) and moves the predictor -- so the 2-byte probability of every coded element
// never goes through memory, and a block under way holds only its packed streams.  At a group
// border the lane gets fresh predictors and the main model's state at that point (which follows
// from the streams).  The price is the chain's pace: a lane advances one element per vector step
// (about 16 ns), so a section of 280 M elements stays under way for 4.5 s; the engine is for
// pipelines some eighty blocks deep, where there are always sixteen chains to step.
struct FusedGroup {
  uint64_t begin, end;       // elements of the block's packed streams
  uint8_t type;              // GroupType
  uint8_t mainState;         // main model's state at the group's start
};
struct FusedSectionDesc {
  const uint8_t* codes;
  const FusedGroup* groups;  // non-empty groups of the section in coding order; consecutive ranges
  size_t n_groups;
  std::vector<uint8_t>* out; // holds the section's prefix; receives the coded bytes and the flush
  void* cookie;
};
class SectionSource {
 public:
  virtual ~SectionSource() {}
  virtual bool next(FusedSectionDesc* s) = 0;   // false: none available now
  virtual void done(void* cookie) = 0;          // the section is coded and flushed
  virtual bool retire() { return true; }
};
void runFusedLanes(SectionSource& src, std::atomic<uint64_t>* busy_ns = nullptr);

}  // namespace wavelet
}  // namespace bwtc
