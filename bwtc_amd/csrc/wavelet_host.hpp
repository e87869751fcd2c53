// Host half of the wavelet coder 'B' (WaveletEncoder, WaveletCoders.cpp:44-219;
// WaveletTree<vector<bool>>, WaveletTree.hpp; BitEncoder, BitCoders.cpp:59-113; the 'B'
// models of probmodels/ProbabilityModel.cpp:38-75).  The GPU delivers the runs of every
// section (huffman_frontend.hip); what remains is bit-serial: building the Huffman-shaped
// tree of (symbol, run-length-code) bit vectors and range-coding them breadth-first with
// adaptive models.  Sections are independent streams (WaveletCoders.cpp:62-68) except for ONE
// carried value: FSM8::resetModel() does not reset the model's current state
// (probmodels/FSM.hpp:196-205), so the state a section starts its root with is the state the
// previous section ended in.  That dependency is resolved with a cheap bits-only pass that
// yields every section's state transition, after which all sections are range-coded in
// parallel on host threads.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

#include "wavelet_rc.hpp"
#include "wavelet_simd.hpp"

namespace bwtc {
namespace wavelet {

// One section as the GPU front-end hands it over.
struct SectionRuns {
  const uint8_t* symbols;      // run symbols, in order
  const uint32_t* starts;      // run start offsets; starts[n_runs] = end of the section's last run
  uint64_t n_runs;
  const uint32_t* run_freqs;   // [256] runs per symbol
  const std::pair<uint32_t, uint32_t>* dist;   // (run length, count) ascending by length
  size_t n_dist;
};

// Encoded payload of one section: packed bitsInRoot, shape bytes, range-coded bytes.
struct SectionOutput {
  std::vector<uint8_t> bytes;
};

// Encodes all sections of one block.  fsm8_state is the probability model's carried state
// (4 for a fresh encoder) and is updated to the state after the last section.
// `model` is the coder letter = the main probability model (giveProbabilityModel,
// probmodels/ProbabilityModel.cpp:47-76): 'B' FSM8 (the default), 'b' FSM<6,
// EvenIntervalPredictor<4>>, 'u' EvenIntervalPredictor<4>; only 'B' carries a state.
void encodeSections(const std::vector<SectionRuns>& sections, unsigned threads,
                    uint32_t* fsm8_state, std::vector<SectionOutput>* out, char model = 'B');
inline bool isWaveletModel(char c) { return c == 'B' || c == 'b' || c == 'u'; }

// ---- device-built streams -------------------------------------------------------------------
// With the tree shapes known (they follow from the run statistics alone) every coded bit has a
// fixed place: the bits of one tree node are the bits its runs contribute, in run order, and
// the nodes are coded in the breadth-first order of encodeTreeBF (WaveletTree.hpp:637-809),
// the integer subtrees level by level after the symbol tree.  planStreams() numbers those
// places ("groups") and emits lookup tables from which the GPU (wavelet_tree.hip) expands every
// run into (group, bit, gap flag) steps, sorts them into coding order and drops the bits
// encodeTreeBF skips.  codeStreams() then only runs the adaptive models and the range
// coder over the finished streams.
constexpr uint32_t kStepBitShift = 0;      // step word: bit 0 = the bit,
constexpr uint32_t kStepGapShift = 1;      //   bit 1 = gap flag (symbol-tree steps); bits 0-1 are below the sort key
constexpr uint32_t kStepLeafShift = 2;     //   bits 2..9 = rank of the symbol leaf (integer steps),
constexpr uint32_t kStepGroupShift = 10;   //   bits 10..31 = group number (block-wide); or bits 2..31 = dense id (StreamPlan::id_group)
constexpr uint32_t kMaxDenseIds = 1u << 22;
constexpr uint32_t kMaxGroups = 1u << 22;
constexpr uint32_t kLenDense = 512;        // run lengths below this are looked up in a dense table (uploaded with every block: 4 KiB per section)

enum GroupType { kRoot = 0, kBothLeaves = 1, kLeftLeaf = 2, kInner = 3, kInteger = 4 };

struct StreamPlan {
  struct Section {
    std::vector<uint8_t> prefix;           // packed bitsInRoot + tree shape
    uint32_t group_base;                   // first group of the section
    uint32_t n_nodes;                      // symbol-tree nodes that are coded (groups 0..n_nodes-1)
    uint32_t n_leaves;                     // symbol leaves (the leaf ranks of the integer steps are below this)
    std::vector<uint32_t> level_first;     // integer levels: first group of each level, then the end
  };
  std::vector<Section> sections;
  std::vector<uint8_t> group_type;         // per block-wide group
  // device tables
  std::vector<uint32_t> symtab;            // [section][256] x {pool offset, steps | leaf rank << 8 | live << 16}
  std::vector<uint64_t> symcode;           // [section][256] the symbol's code, first bit in bit 63 (gap flags:
                                           //   common prefix of consecutive runs' codes, see wavelet_tree.hip)
  std::vector<uint32_t> lendense;          // [section][kLenDense] x {pool offset, steps}
  std::vector<uint32_t> over_first;        // [section + 1] ranges into `over`
  std::vector<uint32_t> over;              // x {length, pool offset, steps, 0}, ascending by length
  std::vector<uint32_t> pool;              // step words
  uint64_t max_elements;                   // upper bound of the number of steps of the block
  // Dense sort keys: instead of (group, leaf rank) as two bit fields -- 8 bits of leaf rank under every
  // group number, nearly all of them unused -- a step word carries  id << 2 | gap << 1 | bit  with
  // id = id_base(group) + leaf rank, where a symbol-tree node takes one id and an integer node as
  // many as its section has symbol leaves.  A 256 MiB text block has some 40 000 ids: the step sort
  // needs two 8-bit passes instead of three.  id_group[id] = the id's group; empty: the two-field layout.
  std::vector<uint32_t> id_group;
};

// false: the shapes need something the device path does not do (too many groups, a symbol code
// longer than 64 bits, a length code that is a prefix of another); the caller then uses
// encodeSections().
bool planStreams(const std::vector<SectionRuns>& sections, StreamPlan* plan);

// The coder over finished streams.  coded_pos[g] = index of group g's first coded element in
// `codes` (2 bits per element, bit 0 = the bit, bit 1 = the gap flag), coded_pos[groups] =
// total.  Two kinds of independent tasks, both ordered largest first:
//   model(k)        the probability of every element of one group (adaptive models; a group
//                   starts with fresh predictors, the main model's state at its start follows
//                   from the few bits before it)
//   codeSection(k)  the range coder over one section -- the one serial chain of the format --
//                   once all of the section's groups are modelled
// plan, coded_pos and codes must outlive the object.
class StreamCoder {
 public:
  StreamCoder(const StreamPlan& plan, const uint32_t* coded_pos, const uint8_t* codes, uint32_t fsm8_state,
              char model = 'B');
  // For a block whose models ran elsewhere (wavelet_gpu_models.hpp): no model tasks, no packed
  // streams; only the sections' chains over  w = bit << 15 | probability of the coded bit.
  struct Modelled {};
  StreamCoder(const StreamPlan& plan, const uint32_t* coded_pos, uint32_t end_state, Modelled);
  uint32_t endState() const { return end_state_; }        // the carried state after the block
  uint64_t elements() const;                              // size of the probability buffer
  size_t modelTasks() const { return tasks_.size(); }
  void model(size_t k, uint16_t* prob) const;
  void modelPair(size_t k1, size_t k2, uint16_t* prob) const;   // two groups stepped alternately by one thread
  // the same task as a descriptor for the 16-lane engine (wavelet_simd.hpp; coder 'B' only)
  void describe(size_t k, uint16_t* prob, ModelGroupDesc* d) const;
  char modelLetter() const { return model_; }
  uint64_t taskElements(size_t k) const { return tasks_[k].end - tasks_[k].begin; }
  bool taskIsInteger(size_t k) const { return tasks_[k].type == kInteger; }
  size_t sectionTasks() const { return sections_.size(); }
  uint64_t largestSectionElements() const;                // coded elements of the longest range-coder chain
  uint64_t sectionElements(size_t k) const;               // coded elements of section task k (tasks are sorted largest first)
  void codeSection(size_t k, const uint16_t* prob, std::vector<SectionOutput>* out) const;   // out[section]
  // the same for sections taken from a shared cursor, two chains at a time in one thread;
  // returns how many sections this call finished
  size_t codeSectionsPaired(std::atomic<size_t>* cursor, const uint16_t* prob, std::vector<SectionOutput>* out) const;
  // sections from the cursor, one chain at a time (a chain alone runs at its full speed: the
  // block's longest one sets its latency), over w-elements (runChainW)
  size_t codeSectionsW(std::atomic<size_t>* cursor, const uint16_t* w, std::vector<SectionOutput>* out) const;
  void codeSectionW(size_t k, const uint16_t* w, std::vector<SectionOutput>* out) const;     // section task k alone
  void describeChainW(size_t k, const uint16_t* w, std::vector<SectionOutput>* out, ChainDesc* d) const;
  // section task k as a chain for the 16-lane coder engine (wavelet_rc.hpp): puts the section's
  // prefix into out[section] and describes the elements to code
  void describeChain(size_t k, const uint16_t* prob, std::vector<SectionOutput>* out, ChainDesc* d) const;
  // section task k for the fused model + coder engine (wavelet_simd.hpp): its non-empty model groups
  // in coding order with the main model's state at each group's start (-> *groups, which must
  // outlive the coding), the section's prefix into out[section]
  void describeSectionFused(size_t k, std::vector<SectionOutput>* out, std::vector<FusedGroup>* groups,
                            FusedSectionDesc* d) const;

 private:
  struct Task { uint32_t section, group; uint64_t begin, end; uint8_t type; };
  uint32_t stateBefore(size_t s, uint32_t group) const;
  void startSection(size_t k, std::vector<SectionOutput>* out, void* chain) const;
  const StreamPlan& plan_;
  const uint32_t* pos_;
  const uint8_t* codes_;
  uint32_t end_state_;
  char model_;
  std::vector<uint32_t> start_;
  std::vector<Task> tasks_;
  std::vector<uint32_t> sections_;
};

// All of it on `threads` threads; `scratch` (optional) keeps the probability buffer between calls.
void codeStreams(const StreamPlan& plan, const uint32_t* coded_pos, const uint8_t* codes,
                 unsigned threads, uint32_t* fsm8_state, std::vector<SectionOutput>* out,
                 std::vector<uint16_t>* scratch = nullptr, char model = 'B');

// What wavelet_tree.hip computes, stated with plain host loops (expand with gap flags, stable
// sort, select, pack); it also derives the gap flags the reference's way and fails when the two
// disagree.  Only the host-only test hook bwtc_hip_host_wavelet_streams calls it, so that
// planStreams/codeStreams can be checked where there is no GPU; false = the plan's tables do
// not cover a run of the input.
bool expandStreamsOnHost(const StreamPlan& plan, const std::vector<SectionRuns>& sections,
                         std::vector<uint32_t>* coded_pos, std::vector<uint8_t>* codes);

}  // namespace wavelet
}  // namespace bwtc
