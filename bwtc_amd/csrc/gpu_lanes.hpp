// Range-coder chains on the GPU (round 5): the sections of a 'B' block that are NOT its longest -- the ones the host
// codes on 16-lane AVX-512 engines (wavelet_rc.hpp, runCoderLanes) -- as one GPU lane each, reading the block's
// w-elements where the model passes left them, in HBM.  BitEncoder (BitCoders.cpp:59-113) is one serial chain per
// section by the format (WaveletCoders.cpp:62-68): a lane is slow (tens of nanoseconds per element), but a 256 MiB
// text block has 29 such sections, a few hundred lanes cover every block under way, and the host keeps only the
// chain that sets a block's latency.  Same integer arithmetic per element as runChainW: same bytes.
#pragma once
#include <atomic>
#include <cstdint>
#include <functional>

#include "wavelet_rc.hpp"

namespace bwtc_hip {

struct GpuLaneStats {
  std::atomic<uint64_t> chains{0}, elements{0}, slices{0}, bytes{0}, host_chains{0};
  std::atomic<uint64_t> slice_ns{0}, slice_steps{0};   // time of the slices' kernels; the most elements one lane coded, summed over the slices
};

// Codes chains from `src` (chains of w-elements) until it is empty and every lane has drained, as
// bwtc::wavelet::runCoderLanes does on the host.  dev_w(cookie) = the device address of the chain's block's w-elements
// (element 0), or null: that chain is coded here on the calling thread instead.  Returns 0, or a negative code after
// which every chain it had taken has still been coded (on the calling thread) and handed back.
int runCoderLanesGpu(bwtc::wavelet::ChainSource& src, int device, const std::function<const uint16_t*(void*)>& dev_w,
                     GpuLaneStats* stats);

}  // namespace bwtc_hip
