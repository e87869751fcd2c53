// BwtEngine: device-resident Burrows-Wheeler transform of one block on one MI355X.
//
// Replaces the work of Divsufsorter / SAISBWTransform behind
// BWTransform::doTransform (bwtransforms/BWTransform.hpp:53-58, BWTransform.cpp:52-64):
// suffix sort of T = reverse(block) + '\0' under "proper prefix sorts first", BWT
// emission with the reference's end-of-block and LFpowers conventions
// (bwtransforms/divsufsort.c:328-404,480-522) and the 256-bin histogram.
//
// Algorithm (integer / index work, HBM-bound, no MFMA):
//   1. load+reverse+histogram, 4-byte big-endian keys            (streaming)
//   2. LSD radix sort of (key32, suffix) pairs                   (radix_sort.hpp)
//   3. ranking: group heads -> rank[], singleton suffixes are final, the rest is
//      compacted into the active list (suffix, global slot, dense group id)
//   4. prefix doubling with discarding: per round gather rank[s+h], radix sort the active
//      list by (group, rank[s+h]), re-rank, compact; stop when nothing is active
//   5. BWT gather T[SA-1], end-of-block patch, LF powers from the final rank[] (= ISA)
#pragma once
#include "common.hpp"
#include "bwtc_hip.h"
#include "radix_sort.hpp"
#include "scan.hpp"
#include "wavelet_host.hpp"
#include "wavelet_pipeline.hpp"
#include <sys/mman.h>
#include <condition_variable>
#include <future>
#include <map>
#include <mutex>
#include <thread>
#include <memory>
#include <utility>
#include <cstdlib>
#include <functional>
#include <vector>

namespace bwtc_hip {

struct RrEmit;
struct RrLong;
struct FinList;
struct FinShallow;
struct FinOutcome { u32 hard = 0, hard_depth = 0xFFFFFFFFu, left = 0, local = 0; };
// A finisher list lives in up to kFinRegions regions of its arrays (bwt_engine.hip, k_finish)
constexpr u32 kFinRegions = 16;
struct FinRegions { u32 nreg; u32 wfirst[kFinRegions + 1]; u32 ebase[kFinRegions]; u32 ecount[kFinRegions]; };   // a finisher run: entries of groups too large (and their smallest depth), entries its last pass left tied

// ---- device side of the 'B' block pipeline (host side: wavelet_pipeline.hpp) ----------------
// Page-locked host bytes: the packed streams are copied from the device straight into the
// block's own buffer.  Recycled between blocks (locking pages is slow), contents lost on growth.
// Where the system will not lock that many pages (96 blocks under way hold 12 GB per context) the
// bytes are ordinary memory: the copy is then staged by the runtime, slower but the same bytes.
struct PinnedGauge {                                // bytes of host staging held by this process, now and at most
  std::atomic<uint64_t> now{0}, peak{0};
  void add(uint64_t n) {
    const uint64_t v = now.fetch_add(n) + n;
    uint64_t p = peak.load();
    while (p < v && !peak.compare_exchange_weak(p, v)) {}
  }
  void sub(uint64_t n) { now.fetch_sub(n); }
  static PinnedGauge& get() { static PinnedGauge g; return g; }
};

// Page-locked host memory, the fast way: an anonymous mapping of transparent huge pages, touched by the
// calling thread (on its own NUMA node, outside the driver), then registered -- 50 ms per 1.2 GB where
// hipHostMalloc takes 200 ms, most of it under the driver's lock, and 130 ms more to free
// (scripts/dev/pin_bench.cpp; the copies run at the same 57 GB/s).  Only for buffers of 32 MiB and more
// (the w-elements and staging buffers of full-size blocks): a mapping of its own, never memory carved
// from the allocator's heap, so nothing else ever lives in, next to or after a registered range.
// *kind: 2 registered mapping, 1 hipHostMalloc, 0 nothing could be locked.
constexpr size_t kLockedHugePage = (size_t)2 << 20;
constexpr size_t kLockedMapFrom = (size_t)32 << 20;
inline size_t lockedMapBytes(size_t n) { return (n + kLockedHugePage - 1) / kLockedHugePage * kLockedHugePage; }
inline void* lockedHostAlloc(size_t n, int* kind) {
  void* q = nullptr;
  if (n >= kLockedMapFrom && !std::getenv("BWTC_HIP_NO_HUGE_PAGES")) {
    const size_t bytes = lockedMapBytes(n);
    // over-map by one huge page and trim to a 2 MiB boundary
    void* raw = mmap(nullptr, bytes + kLockedHugePage, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (raw != MAP_FAILED) {
      const uintptr_t r = reinterpret_cast<uintptr_t>(raw);
      const uintptr_t al = (r + kLockedHugePage - 1) / kLockedHugePage * kLockedHugePage;
      if (al > r) (void)munmap(raw, al - r);
      const size_t tail = (r + bytes + kLockedHugePage) - (al + bytes);
      if (tail) (void)munmap(reinterpret_cast<void*>(al + bytes), tail);
      q = reinterpret_cast<void*>(al);
      (void)madvise(q, bytes, MADV_HUGEPAGE);
      for (size_t i = 0; i < bytes; i += 4096) static_cast<volatile char*>(q)[i] = 0;
      // Huge pages only where the first touch got them: from here on the range is left alone (no later
      // collapse by khugepaged under a registered range -- the device's mapping of a page that the kernel
      // is replacing is what round 3's write fault looked like).
      (void)madvise(q, bytes, MADV_NOHUGEPAGE);
      if (hipHostRegister(q, bytes, hipHostRegisterDefault) == hipSuccess) {
        if (std::getenv("BWTC_HIP_DEBUG")) std::fprintf(stderr, "pinned: registered [%p, %p) (%zu MiB, mapping of its own)\n", q, static_cast<char*>(q) + bytes, bytes >> 20);
        *kind = 2;
        return q;
      }
      (void)hipGetLastError();
      (void)munmap(q, bytes);
      q = nullptr;
    }
  }
  if (hipHostMalloc(&q, n ? n : 1, hipHostMallocDefault) == hipSuccess) { *kind = 1; return q; }
  (void)hipGetLastError();                         // not an error of the caller's stream
  *kind = 0;
  return nullptr;
}
// `n`: the size the buffer was asked for (kind 2 needs it to unmap)
inline void lockedHostFree(void* p, int kind, size_t n) {
  if (!p) return;
  if (kind == 2) {
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return; }   // still known to the device: the mapping stays
    if (std::getenv("BWTC_HIP_DEBUG")) std::fprintf(stderr, "pinned: unregistered and unmapped [%p, %p)\n", p, static_cast<char*>(p) + lockedMapBytes(n));
    (void)munmap(p, lockedMapBytes(n));
  } else if (kind == 1) (void)hipHostFree(p);
  else std::free(p);
}

class PinnedBytes {
 public:
  PinnedBytes() : p_(nullptr), n_(0), locked_(0) {}
  ~PinnedBytes() { release(); }
  PinnedBytes(const PinnedBytes&) = delete;
  PinnedBytes& operator=(const PinnedBytes&) = delete;
  void swap(PinnedBytes& o) { std::swap(p_, o.p_); std::swap(n_, o.n_); std::swap(locked_, o.locked_); }
  bool reserve(size_t n) {
    if (n <= n_) return true;
    release();
    void* q = lockedHostAlloc(n + n / 8, &locked_);
    if (!q) {
      if (std::getenv("BWTC_HIP_DEBUG")) std::fprintf(stderr, "PinnedBytes: %zu bytes could not be page-locked, using ordinary memory\n", n + n / 8);
      q = std::malloc(n + n / 8);
      if (!q) return false;
    }
    p_ = static_cast<uint8_t*>(q);
    n_ = n + n / 8;
    PinnedGauge::get().add(n_);
    return true;
  }
  uint8_t* data() { return p_; }
  size_t size() const { return n_; }
 private:
  void release() {
    if (p_) { lockedHostFree(p_, locked_, n_); PinnedGauge::get().sub(n_); }
    p_ = nullptr; n_ = 0;
  }
  uint8_t* p_;
  size_t n_;
  int locked_;                                   // see lockedHostAlloc
};

// Page-locked buffers of one size class, recycled between blocks.  Locking a gigabyte of pages
// takes 60 ms: once blocks overlap, a helper thread allocates a few buffers AHEAD of the thread that
// feeds the GPU (never more than `limit` in all), so that a stream's first blocks do not each stop
// for it.
class PinnedPool {
 public:
  ~PinnedPool() { stop(); }
  // a recycled (or pre-allocated) buffer, if there is one
  bool take(PinnedBytes* into) {
    std::lock_guard<std::mutex> g(mu_);
    if (free_.empty()) return false;
    into->swap(*free_.back());
    free_.pop_back();
    cv_.notify_all();
    return true;
  }
  void give(PinnedBytes* from, size_t keep_at_most) {
    std::lock_guard<std::mutex> g(mu_);
    if (from->size() && free_.size() < keep_at_most) {
      free_.push_back(std::unique_ptr<PinnedBytes>(new PinnedBytes()));
      free_.back()->swap(*from);
    }
  }
  // the caller allocated one itself (counts against the limit of the helper)
  void noteAllocated() { std::lock_guard<std::mutex> g(mu_); ++made_; }
  // from now on keep up to four spare buffers of `bytes` ready, `limit` buffers in all
  void allocateAhead(int device, size_t bytes, unsigned limit) {
    std::lock_guard<std::mutex> g(mu_);
    if (running_ || bytes == 0) return;
    running_ = true; quit_ = false;
    helper_ = std::thread([this, device, bytes, limit] {
      (void)hipSetDevice(device);
      std::unique_lock<std::mutex> l(mu_);
      while (!quit_ && made_ < limit) {
        if (free_.size() >= 4) { cv_.wait(l); continue; }
        l.unlock();
        std::unique_ptr<PinnedBytes> b(new PinnedBytes());
        const bool ok = b->reserve(bytes);
        l.lock();
        if (!ok) break;
        ++made_;
        free_.push_back(std::move(b));
      }
    });
  }
  void stop() {
    { std::lock_guard<std::mutex> g(mu_); quit_ = true; cv_.notify_all(); }
    if (helper_.joinable()) helper_.join();
    std::lock_guard<std::mutex> g(mu_);
    running_ = false;
  }
  void clear() { stop(); std::lock_guard<std::mutex> g(mu_); free_.clear(); made_ = 0; }
  // the stream needs fewer buffers than were made: the spare ones above `keep` go back to the system
  void trim(size_t keep) {
    std::lock_guard<std::mutex> g(mu_);
    while (free_.size() > keep) { free_.pop_back(); if (made_) --made_; }
  }
 private:
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<std::unique_ptr<PinnedBytes> > free_;
  std::thread helper_;
  unsigned made_ = 0;
  bool running_ = false, quit_ = false;
};

// The model passes of one block between their two parts (wavelet_gpu_models.hip): where its tables
// and intermediates live in the context's workspace.
struct GmPass {
  bool ready = false;
  const u32* d_packed = nullptr;
  u32 n_coded = 0, nt = 0, nc = 0, ns = 0, nsc = 0, ntiles = 0, gsize = 0, ng = 0;
  u64 n_base = 0;
  void *d_tasks = nullptr, *d_chunks = nullptr, *d_order = nullptr, *d_excl = nullptr, *d_tagg = nullptr, *d_tstate = nullptr,
       *d_cmap = nullptr, *d_cstate = nullptr, *d_base = nullptr, *d_partial = nullptr, *d_sb = nullptr, *d_sbits = nullptr,
       *d_smap = nullptr, *d_sstart = nullptr, *d_gmap = nullptr, *d_gL = nullptr, *d_tg = nullptr, *d_snaps = nullptr,
       *d_tail = nullptr;
  u32 ends[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // the carried state after the block, by the state it starts in
  bool ends_ready = false;
  void* side = nullptr;                        // the stream the passes run on when it is not the context's (wavelet_models_device)
};

// A block of the pipeline whose streams came from this device: owns the page-locked bytes.
struct WaveletSectionStats;
struct DeviceWaveletJob : WaveletJob {
  ~DeviceWaveletJob() { if (plan_future.valid()) plan_future.wait(); }   // the planner reads this object
  PinnedBytes codes_owner;
  PinnedBytes w_owner;                 // device-modelled blocks: the w-elements (2 bytes per coded element)
  u32* h_tail = nullptr;               //   {state after the block, error flags, elements counted} (in w_owner, behind the elements)
  u32 gm_state_in = 0;                 //   the carried state the device passes were given
  GmPass gm;                           // farmed streams: the passes' first part is done, the rest waits for the state (_queue)
  // between wavelet_encode_prepare and wavelet_encode_queue
  bool streams_ready = false;          // plan + coded_pos + codes are in place for HostPipeline::queue
  bool host_route = false;             // coded by encodeSections at queue time instead
  bool queued = false;
  bool joins_from_callback = false;    // farmed stream: the block joins the host pipeline from a callback on the d2h stream
  std::atomic<bool> callback_seen{false};
  bool half = false;                   // scanned and (being) planned; streams not made yet
  bool copying = false;                // streams made, their copy to the host may still be in flight
  int run_buf = 0;                     // which of the engine's run-array buffers holds its runs
  u32 block_size = 0;
  std::future<bool> plan_future;       // planStreams on a helper thread
  std::vector<bwtc::wavelet::SectionRuns> host_secs;
  std::vector<u8> host_run_sym;
  std::vector<u32> host_run_start;
  std::shared_ptr<WaveletSectionStats> host_stats;
  unsigned host_threads = 0;
};

struct KeyPlan;
struct BwtEngine {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;   // uploads that overlap the kernels (bwtc_hip_memcpy_to_device_async)
  hipEvent_t ev_copy = nullptr;
  hipStream_t d2h_stream = nullptr;    // packed wavelet streams to the host, under the next block's transform
  hipEvent_t ev_packed = nullptr, ev_codes = nullptr, ev_models = nullptr;
  bool codes_in_flight = false;
  hipError_t ensure_d2h_stream();
  hipError_t codes_wait();             // the copy issued by wavelet_streams_device(async_copy) has landed
  u32 max_block = 0;     // largest block size in bytes
  u64 cap = 0;           // suffix capacity = max_block + 1

  // device workspace
  void* arena = nullptr;
  u64 arena_bytes = 0;
  u8* d_T = nullptr;       // cap + 32
  u8* d_out = nullptr;     // cap + 32
  u8* d_in = nullptr;      // staging for host blocks: cap + 32
  u32* d_SA = nullptr;     // cap
  u32* d_rank = nullptr;   // cap
  void* d_R1 = nullptr;    // 8*cap : K32a|K32b, later K64a
  void* d_R2 = nullptr;    // 8*cap : K64b
  u32* d_V0 = nullptr;     // cap
  u32* d_V1 = nullptr;     // cap
  u32* d_G0 = nullptr;     // cap
  u32* d_G1 = nullptr;     // cap
  u32* d_GRP = nullptr;    // cap
  u8* d_C0 = nullptr;      // cap : carried characters of the active list (ping-pong)
  u8* d_C1 = nullptr;
  u8* d_P0 = nullptr;      // cap : digit planes of the radix passes (radix_sort.hpp)
  u8* d_P1 = nullptr;
  u32* d_W0 = nullptr;     // cap : second key words of the long-key initial sort (ping-pong)
  u32* d_W1 = nullptr;
  u32* d_table = nullptr;  // radix tables
  u32* d_partial = nullptr;
  u32* d_aggA = nullptr;   // rerank tile aggregates
  u32* d_aggB = nullptr;
  u32* d_aggC = nullptr;
  u32* d_agg_part = nullptr;   // totals per 4096 tile aggregates (k_rerank_scan_tiles)
  u32* d_sweep = nullptr;  // chained radix sort: histograms, bases, tickets (SweepWs::small_words)
  bool use_sweep = false;  // BWTC_HIP_SORT=sweep selects the chained single-read passes (measured
                           // slower on MI355X than the XCD-contiguous three-kernel passes)
  u8* d_comp = nullptr;    // compressed-block staging, huffman_compress_bound(max_block)
  u8* d_ent = nullptr;     // 2 MiB of small tables for the entropy front-end
  u32* d_small = nullptr;  // [0..255] freqs, [256..511] lf, [512] pidx, [520..521] counts
  u32* h_small = nullptr;  // pinned mirror of d_small
  u8* h_stage = nullptr;   // pinned staging, cap + 32
  void* d_wt = nullptr;    // wavelet stream workspace, sized by the block's step count (wavelet_tree.hip)
  u64 wt_bytes = 0;
  u8* h_wt = nullptr;      // pinned: tables up, packed streams down
  u64 h_wt_bytes = 0;
  u8* h_stats = nullptr;   // pinned: the run scanner's statistics come down in two copies and one wait
  u64 h_stats_bytes = 0;
  int reserve_stats(u64 bytes);
  // the adaptive models on the device (wavelet_gpu_models.hip): tables and slot space, the
  // w-elements of the block whose copy to the host is under way, pinned staging for the tables
  void* d_gm = nullptr;  u64 gm_bytes = 0;
  void* d_gm_w = nullptr; u64 gm_w_bytes = 0;
  u8* h_gm = nullptr;    u64 h_gm_bytes = 0;
  hipStream_t gm_stream = nullptr;     // the model passes of the _begin flow: beside the next block's transform
  hipEvent_t ev_packed_ready = nullptr, ev_gm_done = nullptr;
  bool gm_done_pending = false;        // the main stream must wait for ev_gm_done before the packed streams' workspace is written again
  bool models_side_stream = true;      // BWTC_HIP_MODELS_STREAM=0: the passes stay on the context's stream (round 4)
  hipEvent_t ev_gm_upload = nullptr;   // the tables' upload from h_gm has finished (the next block may fill h_gm)
  bool gm_upload_pending = false;
  int reserve_models(u64 device_bytes, u64 host_bytes, u64 w_bytes);
  ScanChain scan_chain;                // single-launch scans (scan.hpp): BWTC_HIP_SCAN=chained switches them on (slower here)
  // while alive, the calling thread's scans go through this context's chain
  struct ScanScope {
    ScanChain* prev;
    explicit ScanScope(BwtEngine& e) : prev(current_scan_chain()) { current_scan_chain() = e.scan_chain.status ? &e.scan_chain : nullptr; }
    ~ScanScope() { current_scan_chain() = prev; }
  };
  bool device_models = true;           // BWTC_HIP_MODELS=host: the models stay on the worker threads (all routes of round 2)
  std::vector<u32> long_count;   // host scratch of the run statistics: counts of long run lengths, all zero between calls
  // blocks of the 'B' coder between _begin and _end (wavelet_pipeline.hpp)
  // run arrays of the 'B' coder's scanner, two buffers: block i is scanned while block i-1's runs still
  // wait for their stream kernels (wavelet_encoder.hip)
  u32* d_run_start[2] = {nullptr, nullptr};
  u8* d_run_sym[2] = {nullptr, nullptr};
  int next_run_buf = 0;
  int reserve_run_arrays();
  std::shared_ptr<DeviceWaveletJob> half_job;   // scanned + planned, its streams still to be made
  std::shared_ptr<DeviceWaveletJob> copy_job;   // streams made, copy to the host in flight, not queued yet (_begin flow)
  std::weak_ptr<DeviceWaveletJob> gm_pending;   // _prepare / _queue flow: the job whose model passes wait in the workspace for their state
  bool deferred_queue = false;        // _begin is in use: a begun block joins the stream one or two calls later
  bool async_streams_copy = false;    // set by _begin around its _prepare
  HostPipeline* pipeline = nullptr;   // worker threads, lane engines, coder tasks ('B'; made by the first block)
  std::vector<int> worker_cpus;       // bwtc_hip_set_worker_cpus: where those threads may run (empty: anywhere)
  std::map<u64, std::shared_ptr<DeviceWaveletJob> > jobs;
  u64 next_ticket = 1;
  unsigned max_inflight = 16;          // BWTC_HIP_WAVELET_DEPTH
  u64 huge_group_elements = 32u << 20; // BWTC_HIP_HUGE_MI: groups this large are modelled by scalar tasks, not lanes
  PinnedPool codes_pool, w_pool;                                       // recycled: no fresh pages per block
  std::vector<std::unique_ptr<RawBuffer<uint16_t> > > prob_free;
  u64 wt_elements = 0, wt_coded = 0;   // steps / coded elements of the last block
  bool gm_partition_lines = false;     // BWTC_HIP_GM_PARTITION=lines: the model passes' partition assembled in LDS and stored as whole lines
  bool wt_segmented = true;            // BWTC_HIP_SEG_STEPS=0: the 'B' coder's step sort over block-wide group numbers (three passes)
  bool wavelet_on_host = false;        // BWTC_HIP_WAVELET=host: build the trees on the host instead

  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_wait = nullptr;        // blocking-sync event (hipEventBlockingSync)
  int wait_mode = 0;                   // 0 auto: block while the 'B' worker pool exists, spin otherwise; 1 spin; 2 block (BWTC_HIP_SYNC)
  // Waits for the context's stream.  hipStreamSynchronize spins, which costs the calling thread
  // (and a runtime helper thread) a CPU or two for the whole device time of a block; harmless
  // when nothing else runs, but the 'B' coder's worker threads need those CPUs, so the wait then
  // sleeps on an interrupt instead.
  hipError_t wait();
  static constexpr int kWaitSlots = 64;
  float wait_us[kWaitSlots] = {};      // how long each of a block's waits took for the previous blocks (sleepy_wait)
  int wait_seq = 0;
  hipError_t sleepy_wait(hipEvent_t ev, int slot);
  static constexpr int kMaxSortEvents = 160;
  hipEvent_t ev_sort[kMaxSortEvents];
  int n_sort_events = 0;

  bwtc_hip_stats stats;
  int stream_start_error = 0;   // bwtc_hip_wavelet_start / _reset could not let the old stream's blocks join: the next _begin / _prepare says so
  u32 wavelet_state = 4;   // FSM8 state carried from block to block by one WaveletEncoder
  char wavelet_model = 'B'; // coder letter = main probability model ('B', 'b' or 'u')
  ScatterProbe probe;

  static u64 workspace_bytes(u32 max_block);
  int init(int dev, u32 max_block_size);
  void release();
  int reserve_wavelet(u64 device_bytes, u64 host_bytes);

  // Sorts the suffixes of d_T[0..n-1]; on return d_SA holds the suffix array and d_rank
  // its inverse.  d_T must be followed by >= 8 zero bytes.
  // hist = byte histogram of d_T[0..n-1] (drives the width of the initial sort key).
  // em != nullptr: instead of storing the suffix array the ranking kernels write the
  // transform's bytes, out[slot] = T[SA[slot]-1] for slot < out_n, as suffixes become final
  // (blocks for which can_carry() holds only; see k_rerank_apply).
  struct EmitTarget { u8* out; u32 out_n; u32 n_lf; };
  int suffix_sort(u32 n, const u32* hist, bool lone_sentinel, const EmitTarget* em = nullptr);
  // radix sort front door: picks the chained single-read passes or the classic ones
  template <typename K>
  void sort_pairs(K* k0, K* k1, u32* v0, u32* v1, u64 n, int nbits, K** ks, u32** vs,
                  bool probe_it, int bit_lo = 0, u64 n_holes = 0, bool plane_ready = false, bool values_descend = false);
  // one ranking step of the suffix sorter (bwt_engine.hip)
  struct RankBuffers { void* rec_keys; void* rec_free; u32* v_keys; u32* v_free; u32* aglob; u32* aglob_next; };
  struct RankResult { u32 m, groups; u64* ks; u32* vs; void* rec_other; u32* v_other; bool finish; u32 text_chars; bool carry; };
  template <typename K, bool INIT>
  int rank_step(const K* ks, const u32* vs, u32 m, u32 n, u32 short_len, K kmask, RankBuffers& rb,
                struct RrEmit re, bool emit, u64 h_next, RankResult* res, u32 split = 0,
                const struct RrLong* lg = nullptr, bool text = false, bool carry_in = true, bool raw_out = false);
  bool dense_route = true;   // BWTC_HIP_DENSE=0: always the list-order route (random rank[s+h] gather)
  bool digit_planes = true;  // BWTC_HIP_PLANES=0: every histogram pass reads the keys
  int window_bits = 16;      // BWTC_HIP_WINDOW_BITS: rank[] is updated / read in windows of n >> window_bits suffixes
  bool split_index = true;   // BWTC_HIP_SPLIT_INDEX=0: the initial sort always carries 32-bit suffix numbers
  bool gram_keys = true;     // BWTC_HIP_GRAMS=0: initial keys are always base-sigma numbers (no dense gram codes)
  int gram_count_override = 0;
  struct GramPlan { int g = 0, G = 0, b = 0; u32 top = 0; };
  int plan_grams(const KeyPlan& plan, u32 n, const u8* d_lut, GramPlan* gp);
  bool no_emit = false;      // BWTC_HIP_NO_EMIT: suffix array + gather even for blocks that could carry
  bool long_keys = true;     // BWTC_HIP_LONG=0: never the long-key initial sort
  int long_items_per_thread = 8;   // BWTC_HIP_LONG_E=6: tiles of 3072 items (three workgroups per CU instead of two)
  u32 gram_min_n = 1u << 22; // blocks below this keep the plain base-sigma key (BWTC_HIP_GRAM_MIN_N: tests)
  bool code_failed = false;  // this block's code table could not be built: it is sorted again with gram keys
  bool code_keys = true;     // BWTC_HIP_KEYS=grams: the long key is made of dense gram codes (round 4) instead of the order-1 prefix code
  int code_bits = 72;        // BWTC_HIP_CODE_BITS=N (40..72): bits of a code key (the low 32 in the second word)
  u32* d_pairs = nullptr;    // code keys: sampled character-pair counts (kPairReplicas x 65536 words), and the code table (257 x 256 words)
  u32* d_codes = nullptr;
  u32* d_parkS = nullptr;    // finisher route: the hard list (suffix; head slot << 32 | slot) while the shallow list takes its rounds
  u64* d_parkHP = nullptr;
  u32 parked = 0;            // this block: entries waiting there
  u32 park_holes = 0;        //   of which empty (all ones: they sort to the list's end)
  u32* d_hardS = nullptr;    // finisher route: the groups too large for a window (suffix; head slot << 32 | slot; character | depth << 8), until their code round
  u64* d_hardHP = nullptr;
  unsigned short* d_hardC = nullptr;
  bool ranks_live = false;   // this block: rank[] has been completed (every ranking step keeps it exact from then on)
  u32 fin_floor = 12;        // BWTC_HIP_FIN_FLOOR: hard groups that share fewer characters take their rounds first, on their own
  bool finisher = true;      // BWTC_HIP_FINISHER=0: after the long-key sort straight into the doubling rounds
  int fin_window = 1024;     // BWTC_HIP_FIN_WINDOW=2048: entries a finisher workgroup sorts
  int fin_max_group = 256;   // BWTC_HIP_FIN_GROUP=256|512|1024: the largest group the finisher settles (larger ones take the text rounds)
  int fin_words = 2;         // BWTC_HIP_FIN_WORDS=2|3|4: 8-character words a finisher pass compares
  bool fin_shape_fixed = false;  // either of the two was given: every block takes that shape
  bool fin_wide_short = true;    // BWTC_HIP_FIN_WIDE=0: a short list (the long keys left at most n / 24 suffixes tied) keeps the default shape;
                                 // else it takes windows of 2048 entries and groups of up to 1024 members, and no hard list is left for the rounds
  int fin_window_blk = 1024;     // this block's shape (finisher_passes and local_pass must agree: the local list's groups are as large as the passes left them)
  int fin_group_blk = 256;
  bool long_direct = true;   // BWTC_HIP_LONG_DIRECT=0: every pass of the long-key sort takes its histogram from a digit plane (else: the second word's passes read the words)
  int fin_rounds = 3;        // BWTC_HIP_FIN_ROUNDS=1..4 (with two words): rounds of fin_words words a finisher pass makes (the members a round leaves tied are compared again inside the workgroup)
  int fin_max_passes = 8;    // BWTC_HIP_FIN_PASSES: finisher passes at most before what is still tied takes the doubling rounds
  int text_rounds = 6;       // BWTC_HIP_TEXT_ROUNDS: rounds that compare the text itself before rank[] is completed for doubling
  bool text_rounds_fixed = false;
  bool fin_active = false;   // this block takes the finisher route: finished suffixes also go to d_SA
  bool bridged = false;      // this block: the finisher handed its rest to the doubling rounds (rank[] is complete)
  bool lf_noted = false;     // this block: the LF powers were noted as suffixes became final (rank[] is not complete)
  void complete_ranks(u32 n, const u32* list_sfx, const u32* list_slot, u32 m, void* pairs_region, void* tmp_region);
  // finisher passes over list a (m entries, one region; b: spare list arrays): groups too large for a window go to
  // d_hardS / d_hardHP / d_hardC, shallow ones to shal, what the last pass leaves tied is appended to d_parkS / d_parkHP (`parked`)
  int finisher_passes(u32 n, u32 m, struct FinList a, struct FinList b, struct RrEmit& re, struct FinShallow shal, FinOutcome* fo, bool keep_local = false, int max_passes = 1 << 30);
  // the local list (deep repeats in small groups, doubled group by group beside the global list: bwt_engine.hip, local_pass)
  bool local_rounds = true;  // BWTC_HIP_LOCAL_ROUNDS=0: what the finisher's passes leave joins the waiting list, as in round 4
  u32* d_LP0 = nullptr; u32* d_LH0 = nullptr; u32* d_LH1 = nullptr; unsigned short* d_LC1 = nullptr;
  u32* d_US = nullptr; u32* d_UR = nullptr;          // k_finish<RANK>'s notes: suffix (or none), its new rank, by list position
  u32 local_m = 0, local_depth = 0;
  int local_home = 0, passes_done = 0;
  bool local_pending = false;
  FinRegions local_rg, local_upd;                    // where the local list lives now; where the last pass's notes lie
  struct FinList local_list(int home) const;
  int local_updates();
  int local_pass(u32 n, u64 h_global, struct RrEmit& re);
  // a raw list (suffix; head slot << 32 | slot) in (S, HP) -> a sorted list the rounds understand, in res / rb
  int dress_list(u32 n, u32 total, u32* S, u64* HP, RankBuffers& rb, RankResult* res, u32 holes = 0);
  int long_grams_override = 0;   // BWTC_HIP_LONG_G2=N: N grams in the second key word
  void scatter_rank_pairs(u32* pairs, u32* tmp, u32 m, u32 n);
  int load_text(const u8* d_src, u32 ncopy, u32 n, bool reverse, u32* hist_T);
  // Device-resident block transform: d_src (size bytes) -> d_dst (size bytes).
  // raw=false: block semantics (reverse, sentinel, EOB patch).  raw=true: d_src already is
  // T (size = length), position pidx keeps its input byte.
  int transform(const u8* d_src, u8* d_dst, u32 size, bool raw, u32* lf, u32 n_lf, u32* freqs);
};

u64 huffman_compress_bound(u64 size);
// 'H' coder over an already transformed block that is resident on the device: writes the
// complete BWT-block record (48-bit length, BWTBlock header, sections, payloads) to d_out.
int huffman_encode_device(BwtEngine& e, const u8* d_bwt, u32 size, const u32* lf, u32 n_lf,
                          const u32* freqs, u8* d_out, u64 out_cap, u64* out_bytes);

// What the WaveletTree constructor first computes for every section of a transformed block
// (utils::calculateRunsAndCharacters, Utils.cpp:128-147; WaveletTree.hpp:294-308).
struct WaveletSectionStats {
  std::vector<u32> sections;                                  // section lengths
  std::vector<u32> run_freqs;                                 // [section][256] runs per symbol
  std::vector<u64> total_runs;                                // runs per section
  std::vector<std::vector<std::pair<u32, u32> > > dist;       // per section: (run length, count), ascending
  std::vector<u32> first_run;                                 // first run of every section, + total
};
// d_run_start / d_run_sym: where the runs go (null: the transform's workspace, e.d_R1 / e.d_R2)
// while_waiting: host work to do after the scanner's kernels and copies are queued, before the wait for them
int wavelet_section_stats_device(BwtEngine& e, const u8* d_bwt, u32 size, const u32* freqs,
                                 WaveletSectionStats* out, u32* d_run_start = nullptr, u8* d_run_sym = nullptr,
                                 const std::function<void()>* while_waiting = nullptr);

// Steps of all runs of the block sorted into coding order, skipped bits dropped, the packed
// streams copied into `codes` (wavelet_tree.hip); plan comes from bwtc::wavelet::planStreams.
int wavelet_streams_device(BwtEngine& e, const u32* d_run_start, const u8* d_run_sym, u32 n_runs, const std::vector<u32>& first_run,
                           const bwtc::wavelet::StreamPlan& plan, std::vector<u32>* coded_pos,
                           PinnedBytes* codes, bool async_copy = false, const u32** d_packed = nullptr);
// The adaptive models of a block whose packed streams are in HBM (wavelet_gpu_models.hip): queues
// the passes and the copy of the w-elements to h_w, of {state after, error flags, count} to h_tail.
int wavelet_models_device(BwtEngine& e, const u32* d_packed, u32 n_coded, const bwtc::wavelet::StreamPlan& plan,
                          const std::vector<u32>& coded_pos, u32 state_in, uint16_t* h_w, u32* h_tail);
// the same in two parts: what does not depend on the carried state (read_ends: and, after a wait for the
// device, the state after the block for each of the eight states it can start in), and the rest
// (early_state != null: the state after the block is returned before the long passes are queued -- from
// that table, else read back with a short wait)
int wavelet_models_prepare(BwtEngine& e, const u32* d_packed, u32 n_coded, const bwtc::wavelet::StreamPlan& plan,
                           const std::vector<u32>& coded_pos, GmPass* g, bool read_ends = false, hipStream_t side = nullptr);
int wavelet_models_run(BwtEngine& e, const GmPass& g, u32 state_in, uint16_t* h_w, u32* h_tail, u32* early_state);

// WaveletEncoder: writeBlockHeader + encodeData + finishBlock (WaveletCoders.cpp:173-219,
// 112-157, 159-163) for a device-resident transformed block, in two halves so that blocks
// overlap (wavelet_pipeline.hpp): _begin does the device work (run scanner, stream kernels),
// advances the encoder's carried model state and queues the block's models and range coders
// on the engine's worker threads (created with `threads` threads by the first call); _end waits
// for that block and reports the record's size (the bytes are in `out` given to _begin).
int wavelet_encode_begin(BwtEngine& e, const u8* d_bwt, u32 size, const u32* lf, u32 n_lf,
                         const u32* freqs, unsigned threads, u8* out, u64 out_cap, u64* ticket);
int wavelet_encode_end(BwtEngine& e, u64 ticket, u64* out_bytes);
// _begin in two halves, for a stream whose blocks are spread over several contexts: _prepare is
// everything that does not depend on earlier blocks, _queue hands the block to the workers with
// the carried model state after the previous block and reports the state after this one.
int wavelet_encode_prepare(BwtEngine& e, const u8* d_bwt, u32 size, const u32* lf, u32 n_lf,
                           const u32* freqs, unsigned threads, u8* out, u64 out_cap, u64* ticket,
                           const std::function<void()>* while_scanning = nullptr);
int wavelet_encode_queue(BwtEngine& e, u64 ticket, u32 state_in, u32* state_out);
// Blocks begun with _begin join their stream one or two calls later; this makes every such block
// join NOW, in ticket order, with the stream's current state and model -- what must happen before
// the context starts another stream (bwtc_hip_wavelet_start / _reset).
int wavelet_join_begun(BwtEngine& e);
void wavelet_pipeline_release(BwtEngine& e);

// Inverse transform of a device-resident transformed block (d_bwt and d_out may alias).
// Returns -4 when an LF power does not lie on the LF walk (corrupt header or data).
int inverse_bwt_device(BwtEngine& e, const u8* d_bwt, u8* d_out, u32 size, const u32* lf, u32 n_lf);

}  // namespace bwtc_hip
