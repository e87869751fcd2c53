// See gpu_lanes.hpp.
#include "gpu_lanes.hpp"

#include <hip/hip_runtime.h>
#include <time.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace bwtc_hip {

namespace {

struct GpuLane {                 // one chain in flight; the kernel advances i, lo, size and says how many bytes it wrote
  const uint16_t* w;             // the block's w-elements in HBM (element 0)
  uint64_t i, e;                 // next element, one past the last
  uint32_t lo, size;             // BitEncoder's interval: (low, high - low - 1), as CoderChain keeps it
  uint8_t* out;                  // this lane's output chunk
  uint32_t used, pad;
};

// One element (wavelet_rc.cpp, runChainW): w = bit << 15 | probability of the coded bit (12 bits); at most four bytes leave.
__device__ __forceinline__ void rc_step(uint32_t x, uint32_t& lo, uint32_t& size, uint8_t* o, uint32_t& used) {
  const uint32_t bit = x >> 15, m = x & 0x7FFFu;
  const uint32_t ns = (uint32_t)(((int64_t)((uint64_t)size * m) + ((int64_t)bit - 2049)) >> 12);
  lo += (bit - 1u) & (size - ns);
  size = ns;
  while (((lo ^ (lo + size + 1u)) & 0xFF000000u) == 0u) { o[used++] = (uint8_t)(lo >> 24); lo <<= 8; size = (size << 8) + 510u; }
}

// A slice: every busy lane codes up to max_elems elements of its chain, or until its chunk is nearly full.  One wave
// per workgroup (the waves spread over the CUs; a lane's speed is its dependency chain, not the CU's throughput).
// Elements are read eight at a time (16 bytes), 32 elements ahead: a lane walks its own stretch of the array, and a
// 64-byte line from HBM is two microseconds away.
__global__ __launch_bounds__(64) void k_rc_slice(GpuLane* __restrict__ lanes, int n, uint32_t max_elems, uint32_t out_cap) {
  const int l = blockIdx.x * 64 + threadIdx.x;
  if (l >= n) return;
  GpuLane L = lanes[l];
  if (L.i >= L.e || L.w == nullptr) return;
  uint64_t i = L.i;
  const uint64_t stop = L.e - i > (uint64_t)max_elems ? i + max_elems : L.e;
  uint32_t lo = L.lo, size = L.size, used = 0;
  const uint16_t* __restrict__ w = L.w;
  uint8_t* __restrict__ o = L.out;
  while (i < stop && used + 4u <= out_cap) {
    if ((i & 7u) == 0u && i + 8u <= stop) {
      if (used + 32u > out_cap) break;                       // not room for a group: the next slice goes on from here
      const uint4* q = reinterpret_cast<const uint4*>(w + i);
      uint4 r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3];     // (past the chain's end: other sections' elements, or the buffer's slack)
      while (i + 8u <= stop && used + 32u <= out_cap) {
        const uint32_t v[4] = {r0.x, r0.y, r0.z, r0.w};
        r0 = r1; r1 = r2; r2 = r3; r3 = q[4];
        ++q;
#pragma unroll
        for (int k = 0; k < 4; ++k) { rc_step(v[k] & 0xFFFFu, lo, size, o, used); rc_step(v[k] >> 16, lo, size, o, used); }
        i += 8u;
      }
    } else {
      rc_step(w[i], lo, size, o, used);                      // up to the next multiple of eight, and the chain's last few
      ++i;
    }
  }
  lanes[l].i = i; lanes[l].lo = lo; lanes[l].size = size; lanes[l].used = used;
}

constexpr int kGpuLanes = 256;
constexpr uint32_t kChunk = 1u << 19;            // bytes a lane may write per slice
constexpr uint32_t kSliceElems = 1u << 20;
constexpr uint64_t kHostOnly = 1u << 16;         // shorter chains are not worth a lane

struct Engine {                                   // per process and device; one engine runs at a time
  std::mutex mu;
  int device = -1;
  hipStream_t st = nullptr;
  hipEvent_t ev = nullptr;
  GpuLane* d_lanes = nullptr;
  GpuLane* h_lanes = nullptr;                     // page-locked
  uint8_t* d_out = nullptr;
  uint8_t* h_out = nullptr;                       // page-locked
  bool ok = false;
  bool init(int dev) {
    if (ok && device == dev) return true;
    if (hipSetDevice(dev) != hipSuccess) return false;
    device = dev;
    int lo_prio = 0, hi_prio = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo_prio) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return false;
    if (hipMalloc(reinterpret_cast<void**>(&d_lanes), sizeof(GpuLane) * kGpuLanes) != hipSuccess) return false;
    if (hipHostMalloc(reinterpret_cast<void**>(&h_lanes), sizeof(GpuLane) * kGpuLanes, hipHostMallocDefault) != hipSuccess) return false;
    if (hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)kChunk * kGpuLanes) != hipSuccess) return false;
    if (hipHostMalloc(reinterpret_cast<void**>(&h_out), (size_t)kChunk * kGpuLanes, hipHostMallocDefault) != hipSuccess) return false;
    ok = true;
    return true;
  }
  // waits for everything queued on st, sleeping (a spinning wait would cost the CPU this engine exists to save)
  bool wait() {
    if (hipEventRecord(ev, st) != hipSuccess) return false;
    for (;;) {
      const hipError_t q = hipEventQuery(ev);
      if (q == hipSuccess) return true;
      if (q != hipErrorNotReady) return false;
      timespec ts; ts.tv_sec = 0; ts.tv_nsec = 500000;
      nanosleep(&ts, nullptr);
    }
  }
};

Engine& engine() { static Engine e; return e; }

}  // namespace

int runCoderLanesGpu(bwtc::wavelet::ChainSource& src, int device, const std::function<const uint16_t*(void*)>& dev_w,
                     GpuLaneStats* stats) {
  using bwtc::wavelet::ChainDesc;
  using bwtc::wavelet::CoderChain;
  Engine& E = engine();
  std::lock_guard<std::mutex> only_one(E.mu);
  bool gpu = hipSetDevice(device) == hipSuccess && E.init(device);
  if (!gpu) { (void)hipGetLastError(); std::fprintf(stderr, "bwtc_hip: the GPU lane engine could not start; its chains are coded on the host\n"); }
  ChainDesc desc[kGpuLanes];
  CoderChain chain[kGpuLanes];                     // lo / size / i / used live here between slices
  bool busy[kGpuLanes];
  for (int l = 0; l < kGpuLanes; ++l) busy[l] = false;
  auto on_host = [&](ChainDesc& d) {               // the whole chain on this thread (short chains, chains without device elements, errors)
    CoderChain c;
    c.start(d.begin, d.end, d.out);
    bwtc::wavelet::runChainW(c, d.w, c.e);
    c.finish();
    src.done(d.cookie);
    if (stats) ++stats->host_chains;
  };
  auto rest_on_host = [&](int l) {                 // what is left of lane l's chain
    CoderChain& c = chain[l];
    bwtc::wavelet::runChainW(c, desc[l].w, c.e);
    c.finish();
    src.done(desc[l].cookie);
    busy[l] = false;
    if (stats) ++stats->host_chains;
  };
  bool dry = false;
  int rc = 0;
  for (;;) {
    int nbusy = 0;
    for (int l = 0; l < kGpuLanes; ++l) {
      while (!busy[l] && !dry) {
        if (!src.next(&desc[l])) { dry = true; break; }
        const uint16_t* dw = (gpu && desc[l].w) ? dev_w(desc[l].cookie) : nullptr;
        if (!dw || desc[l].end - desc[l].begin < kHostOnly) { on_host(desc[l]); continue; }
        CoderChain& c = chain[l];
        c.start(desc[l].begin, desc[l].end, desc[l].out);
        GpuLane& L = E.h_lanes[l];
        L.w = dw; L.i = c.i; L.e = c.e; L.lo = c.lo; L.size = c.size; L.out = E.d_out + (size_t)l * kChunk; L.used = 0; L.pad = 0;
        busy[l] = true;
        if (stats) ++stats->chains;
      }
      if (!busy[l]) { E.h_lanes[l].w = nullptr; E.h_lanes[l].i = E.h_lanes[l].e = 0; E.h_lanes[l].used = 0; }
      nbusy += busy[l] ? 1 : 0;
    }
    if (nbusy == 0) {
      if (dry && src.retire()) return rc;
      dry = false;
      continue;
    }
    // one slice
    bool good = gpu;
    if (good) {
      good = hipMemcpyAsync(E.d_lanes, E.h_lanes, sizeof(GpuLane) * kGpuLanes, hipMemcpyHostToDevice, E.st) == hipSuccess;
      const auto t_slice = std::chrono::steady_clock::now();
      if (good) {
        hipLaunchKernelGGL(k_rc_slice, dim3(kGpuLanes / 64), dim3(64), 0, E.st, E.d_lanes, kGpuLanes, kSliceElems, kChunk);
        good = hipMemcpyAsync(E.h_lanes, E.d_lanes, sizeof(GpuLane) * kGpuLanes, hipMemcpyDeviceToHost, E.st) == hipSuccess && E.wait();
        if (good && stats) {
          stats->slice_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_slice).count();
          uint64_t most = 0;
          for (int l = 0; l < kGpuLanes; ++l) if (busy[l]) most = std::max<uint64_t>(most, E.h_lanes[l].i - chain[l].i);
          stats->slice_steps += most;
        }
      }
      if (good) {
        for (int l = 0; l < kGpuLanes && good; ++l)
          if (busy[l] && E.h_lanes[l].used)
            good = hipMemcpyAsync(E.h_out + (size_t)l * kChunk, E.d_out + (size_t)l * kChunk, E.h_lanes[l].used, hipMemcpyDeviceToHost, E.st) == hipSuccess;
        good = good && E.wait();
      }
    }
    if (!good) {
      // the device side failed: nothing of this slice is used (the lanes' host copies still hold the state before it);
      // every chain in flight, and every later one, is finished on this thread
      (void)hipGetLastError();
      if (gpu) std::fprintf(stderr, "bwtc_hip: the GPU lane engine lost its device; its chains are finished on the host\n");
      gpu = false;
      rc = -3;
      for (int l = 0; l < kGpuLanes; ++l) if (busy[l]) rest_on_host(l);
      dry = false;
      continue;
    }
    if (stats) ++stats->slices;
    for (int l = 0; l < kGpuLanes; ++l) {
      if (!busy[l]) continue;
      CoderChain& c = chain[l];
      const GpuLane& L = E.h_lanes[l];
      if (L.used) {
        c.out->resize(c.used + L.used);
        std::memcpy(c.out->data() + c.used, E.h_out + (size_t)l * kChunk, L.used);
        c.used += L.used;
        if (stats) stats->bytes += L.used;
      }
      if (stats) stats->elements += L.i - c.i;
      c.i = L.i; c.lo = L.lo; c.size = L.size;
      E.h_lanes[l].used = 0;
      if (c.i >= c.e) {
        c.finish();
        src.done(desc[l].cookie);
        busy[l] = false;
      }
    }
    dry = false;                                   // ask again: other blocks may have arrived
  }
}

}  // namespace bwtc_hip
