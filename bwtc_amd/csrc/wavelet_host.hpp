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
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

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
void encodeSections(const std::vector<SectionRuns>& sections, unsigned threads,
                    uint32_t* fsm8_state, std::vector<SectionOutput>* out);

}  // namespace wavelet
}  // namespace bwtc
