// Host side of the block pipeline of the 'B' coder.  The device work of a block (BWT, run
// scanner, stream kernels) takes a tenth of the time its range coder needs on the host, and the
// range coder of a block is a handful of serial chains, so blocks are overlapped: the calling
// thread does the device work of block i+1 while a fixed pool of worker threads models and
// range-codes blocks i, i-1, ...  The one value that links consecutive blocks (the main
// model's state, probmodels/FSM.hpp:196-205) follows from the last bits of a block's streams
// and is known before its coding starts, so the bytes are those of a strictly sequential
// encoder.
#pragma once
#include <malloc.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "wavelet_host.hpp"

namespace bwtc_hip {

// host time spent in the two stages (summed over threads), for BWTC_HIP_DEBUG
struct StageClock {
  std::atomic<uint64_t> model_ns{0}, coder_ns{0}, blocks{0}, finished{0};
  std::atomic<uint64_t> scalar_ns{0};               // the part of coder_ns spent in the scalar tasks of device-modelled blocks
  std::atomic<uint64_t> latency_ns{0};             // begun -> record finished, summed over the finished blocks
  std::atomic<int> unfinished{0};      // blocks begun and not yet finished by the workers
};

// Fixed set of threads over one queue ordered by (block, submission order): whatever the
// oldest block still needs runs first, so the block the caller will collect next is never
// overtaken by the work of newer ones.  Rank 0 is for the cross-block model engines.
class WorkerPool {
 public:
  explicit WorkerPool(unsigned threads) : seq_(0), stop_(false) {
    for (unsigned i = 0; i < (threads ? threads : 1u); ++i) workers_.push_back(std::thread([this] { loop(); }));
  }
  // the workers may run on exactly these CPUs from now on (empty: anywhere the process may)
  void setCpus(const std::vector<int>& cpus) {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (cpus.empty()) { if (sched_getaffinity(0, sizeof set, &set) != 0) return; }
    else for (size_t i = 0; i < cpus.size(); ++i) if (cpus[i] >= 0 && cpus[i] < CPU_SETSIZE) CPU_SET(cpus[i], &set);
    for (size_t i = 0; i < workers_.size(); ++i) (void)pthread_setaffinity_np(workers_[i].native_handle(), sizeof set, &set);
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
    cv_.notify_all();
    for (size_t i = 0; i < workers_.size(); ++i) workers_[i].join();
  }
  unsigned size() const { return static_cast<unsigned>(workers_.size()); }
  void submit(uint64_t rank, std::function<void()> f) {
    { std::lock_guard<std::mutex> g(mu_); q_.insert(std::make_pair(std::make_pair(rank, seq_++), std::move(f))); }
    cv_.notify_one();
  }
  // the tasks keep their order: first of `fs` runs first
  void submit(uint64_t rank, std::vector<std::function<void()> >& fs) {
    { std::lock_guard<std::mutex> g(mu_); for (size_t i = 0; i < fs.size(); ++i) q_.insert(std::make_pair(std::make_pair(rank, seq_++), std::move(fs[i]))); }
    cv_.notify_all();
  }

 private:
  void loop() {
    // a notch below the calling thread: it feeds the GPU and must not wait for a core
    (void)setpriority(PRIO_PROCESS, static_cast<id_t>(syscall(SYS_gettid)), 5);
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [this] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;                      // stop_ and drained
        f = std::move(q_.begin()->second);
        q_.erase(q_.begin());
      }
      f();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::map<std::pair<uint64_t, uint64_t>, std::function<void()> > q_;
  uint64_t seq_;
  std::vector<std::thread> workers_;
  bool stop_;
};

// Large scratch that is written before it is read: not zero-filled, and the pages are first
// touched by whoever writes them (the worker threads, in parallel, for the probabilities).
template <typename T>
class RawBuffer {
 public:
  RawBuffer() : p_(nullptr), n_(0) {}
  ~RawBuffer() { std::free(p_); }
  RawBuffer(const RawBuffer&) = delete;
  RawBuffer& operator=(const RawBuffer&) = delete;
  void swap(RawBuffer& o) { std::swap(p_, o.p_); std::swap(n_, o.n_); }
  // contents are lost when it grows
  bool reserve(size_t n) {
    if (n <= n_) return true;
    std::free(p_);
    n_ = 0;
    p_ = static_cast<T*>(std::malloc((n + n / 8) * sizeof(T)));
    if (!p_) return false;
    n_ = n + n / 8;
    return true;
  }
  T* data() { return p_; }
  size_t size() const { return n_; }
 private:
  T* p_;
  size_t n_;
};

struct WaveletJob;

// The groups of all blocks under way, oldest block first and largest group first within a
// block, for the 16-lane model engines (wavelet_simd.hpp).  Lanes are refilled across block
// borders: one block's few huge groups (the root of its largest section holds a tenth of its
// elements) share vectors with the next blocks' groups instead of running in a nearly empty
// vector.  Engines are few (the models are a small part of the host work; the range coders
// need the other threads) and come and go with the work: begin() starts one when there are
// fewer than the limit, an engine that finds nothing left retires -- both decided under the
// source's lock.
class BlockGroupSource : public bwtc::wavelet::GroupSource {
 public:
  BlockGroupSource() { engines_[0] = engines_[1] = engines_[2] = 0; }
  // queues a block's groups from `first_group` on (the list is largest first; the caller keeps
  // the few huge ones for the scalar loop, which finishes a single chain four times sooner than
  // a lane does); returns the preference (1 integer-level groups, 2 symbol-tree groups) of the
  // engine the caller should start, 0 for none
  int add(const std::shared_ptr<WaveletJob>& job, size_t first_group, unsigned max_engines);
  bool next(bwtc::wavelet::ModelGroupDesc* d, int prefer);
  void done(void* cookie);
  bool retire(int prefer) {
    std::lock_guard<std::mutex> g(mu_);
    if (!queue_.empty()) return false;
    --engines_[prefer];
    return true;
  }
  // set once: what to do when a block's last group is modelled
  std::function<void(const std::shared_ptr<WaveletJob>&)> on_block_modelled;
 private:
  // a block's groups by kind, each list largest first; [0] integer levels, [1] symbol-tree nodes
  struct Entry { std::shared_ptr<WaveletJob> job; std::vector<uint32_t> list[2]; size_t at[2]; };
  std::mutex mu_;
  std::deque<Entry> queue_;
  unsigned engines_[3];
};

// The sections of all modelled blocks, oldest block first and largest section first within a
// block, for the 16-lane coder engines (wavelet_rc.hpp): a block's one or two long chains share
// vectors with the chains of the blocks behind it instead of running alone.
class BlockChainSource : public bwtc::wavelet::ChainSource {
 public:
  BlockChainSource() : engines_(0) {}
  // queues the section tasks [first, last) of a modelled block (largest first); true = the caller
  // should start another engine
  bool add(const std::shared_ptr<WaveletJob>& job, size_t first, size_t last, unsigned max_engines);
  bool next(bwtc::wavelet::ChainDesc* d);
  void done(void* cookie);
  bool retire() {
    std::lock_guard<std::mutex> g(mu_);
    if (!queue_.empty()) return false;
    --engines_;
    return true;
  }
  std::function<void(WaveletJob&)> on_block_coded;     // set once: the block's last section is flushed
 private:
  struct Entry { std::shared_ptr<WaveletJob> job; size_t at, end; };
  std::mutex mu_;
  std::deque<Entry> queue_;
  unsigned engines_;
};

// The sections of all queued blocks for the fused model + coder engines (wavelet_simd.hpp).
class BlockSectionSource : public bwtc::wavelet::SectionSource {
 public:
  BlockSectionSource() : engines_(0) {}
  bool add(const std::shared_ptr<WaveletJob>& job, unsigned max_engines);   // true: start another engine
  bool next(bwtc::wavelet::FusedSectionDesc* d);
  void done(void* cookie);
  bool retire() {
    std::lock_guard<std::mutex> g(mu_);
    if (!queue_.empty()) return false;
    --engines_;
    return true;
  }
  std::function<void(WaveletJob&)> on_block_coded;
 private:
  struct Entry { std::shared_ptr<WaveletJob> job; size_t at; };
  std::mutex mu_;
  std::deque<Entry> queue_;
  unsigned engines_;
};

// One block between bwtc_hip_wavelet_encode_device_begin and ..._end.
struct WaveletJob : std::enable_shared_from_this<WaveletJob> {
  virtual ~WaveletJob() {}
  std::vector<uint8_t> record;                       // header + section sizes, then the sections, finished in place
  bwtc::wavelet::StreamPlan plan;
  std::vector<uint32_t> coded_pos;
  const uint8_t* codes = nullptr;                    // packed streams (2 bits per coded element); the owner of the
                                                     //   bytes is whoever made the job (see DeviceWaveletJob)
  RawBuffer<uint16_t> prob;                          // probability of every coded element
  std::unique_ptr<bwtc::wavelet::StreamCoder> coder;
  std::vector<bwtc::wavelet::SectionOutput> outs;
  std::vector<std::vector<bwtc::wavelet::FusedGroup> > fused_groups;   // fused engines: the sections' group lists
  bool fused = false;                                // coded by the fused engines (no `prob`); set before queue()
  const uint16_t* w = nullptr;                       // modelled on the device: bit << 15 | probability of the coded bit,
                                                     //   per coded element (wavelet_gpu_models.hpp); no model tasks then
  uint32_t w_end_state = 4;                          //   and the carried state after the block
  bool failed = false;                 // the block could not be coded as its stream needs (out of memory at the join,
                                       // a host half that ends in another state than the one handed on): _end reports it
  std::mutex mu;
  std::condition_variable cv;
  size_t models_left = 0, sections_left = 0;
  std::atomic<size_t> section_cursor{0};            // sections handed to the paired range-coder tasks
  bool done = false;
  uint64_t rank = 0;                                 // the block's place in the worker pool's order (its ticket)
  uint8_t* user_out = nullptr;
  uint64_t user_cap = 0;
  std::chrono::steady_clock::time_point t_queued, t_modelled, t_finished;   // BWTC_HIP_DEBUG timeline
  std::chrono::steady_clock::time_point t_begun;    // when the block's device half started (set by whoever made the job)
};

}  // namespace bwtc_hip

namespace bwtc_hip {

// Everything that happens to a block on the host once its streams exist: models (lane engines
// fed across blocks, scalar tasks for the huge groups), range coders (two chains per task), the
// record.  Host code only -- the device side (wavelet_encoder.hip) and the CPU tests
// (tests/cpp/host_pipeline_test.cpp) drive the same object.
class HostPipeline {
 public:
  // depth = how many blocks the caller keeps under way at most.  It picks the range-coder route:
  // the 16-lane engines code sixteen sections at a time for a third of the scalar loop's host
  // time, but a lane advances ITS chain four to five times slower than the scalar loop, so a
  // block whose longest section holds half of its elements (the text blocks) stays under way for
  // three seconds instead of one -- that only pays when the pipeline is deep enough to hide it
  // (>= kLaneDepth blocks, 1.2 GB of host memory each).  Shallower pipelines keep the scalar
  // chains, two per task (codeSectionsPaired).  BWTC_HIP_CODER_LANES=1 / 0 overrides.
  HostPipeline(unsigned threads, uint64_t huge_group_elements, unsigned depth = 16);
  // (Measured again with the GPU side at 45 ms per block: the lane route then needs more than 64
  // blocks under way to beat the scalar chains, and from 56 blocks on the fused engines beat both
  // -- so by default the lanes are only used when asked for.)
  static constexpr unsigned kLaneDepth = 1u << 30;
  ~HostPipeline();                                   // joins the workers; every queued block must be finished
  // job: record (header part), plan, coded_pos, codes, prob (room for the coded elements), rank and
  // user_out/user_cap set.  Builds the block's coder from the carried model state and queues its
  // work; returns the state to carry into the next block.
  // expect_end (optional): the state the caller has already handed to the next block; a block whose
  // models end elsewhere is marked failed before any of its tasks runs
  uint32_t queue(const std::shared_ptr<WaveletJob>& job, uint32_t fsm8_state, char model, const uint32_t* expect_end = nullptr);
  static void wait(WaveletJob& job);
  static bool waitFor(WaveletJob& job, unsigned milliseconds);   // true: done
  // a block without coded elements (or coded elsewhere): closes the record right away
  static void finishNow(WaveletJob& job);
  unsigned threads() const { return pool_.size(); }
  void setWorkerCpus(const std::vector<int>& cpus) { pool_.setCpus(cpus); }
  // Lets a worker thread drop the last reference to a collected block: its record, section
  // outputs and tables are a few hundred MB of heap, and returning them to the system costs the
  // caller's thread -- the one that feeds the GPU -- several milliseconds per block.
  void dispose(std::shared_ptr<WaveletJob>&& job) {
    std::shared_ptr<WaveletJob> j(std::move(job));
    // The record and the outputs change hands here and now (a few swaps), and what is left of the block
    // is freed by the next worker that looks for work: as a task of the lowest priority -- as this was
    // at first -- it waited while the workers were busy, that is for as long as the stream ran at full
    // rate, and every collected block stayed allocated (110 MB per block, 60 GB after 700 blocks).
    keepSpares(*j);
    pool_.submit(0, [this, j]() mutable {
      j.reset();
      // what the blocks' smaller tables leave behind in the allocator's arenas (freed by other threads
      // than the ones that allocated them) goes back to the system now and then: a long stream at full
      // rate otherwise sat on 30-70 GB of freed memory beside the 36 GB it uses
      if ((++disposed_ & 31u) == 0) (void)malloc_trim(0);
    });
  }
  // The record and the sections' outputs of a collected block -- 100 MB of touched pages for a 256 MiB
  // text block -- go to the next block that needs them instead of back to the allocator (which gave
  // every block fresh pages to fault in, and whose per-thread arenas grew the process to 60-95 GB
  // over a long stream where 36 GB are in use).
  void keepSpares(WaveletJob& job) {
    std::lock_guard<std::mutex> g(spare_mu_);
    if (spare_outs_.size() < kSpares && !job.outs.empty()) { spare_outs_.emplace_back(); spare_outs_.back().swap(job.outs); }
    if (spare_records_.size() < kSpares && job.record.capacity() > 4096) { spare_records_.emplace_back(); spare_records_.back().swap(job.record); }
  }
  void takeSpareOutputs(std::vector<bwtc::wavelet::SectionOutput>* outs) {
    std::lock_guard<std::mutex> g(spare_mu_);
    if (!spare_outs_.empty()) { outs->swap(spare_outs_.back()); spare_outs_.pop_back(); }
  }
  void takeSpareRecord(std::vector<uint8_t>* record) {
    std::lock_guard<std::mutex> g(spare_mu_);
    if (!spare_records_.empty()) { record->swap(spare_records_.back()); spare_records_.pop_back(); }
  }
  // Route of the next block of a 'B' stream: true = the fused engines (the job then needs no
  // probability array).  Even a pipeline built for them sends a block the two-stage way (parallel
  // models, scalar chains -- a third of the latency) while fewer than kFusedBacklog blocks are
  // under way: a lone block, or the first blocks of a stream, would crawl through nearly empty
  // vectors.  The caller sets WaveletJob::fused from this before queue().
  bool fusedNow(char model) const { return max_fused_engines_ && model == 'B' && clock.unfinished.load() >= (int)kFusedBacklog; }
  // From which announced depth on the fused engines pay.  Measured at ONE point -- 256 MiB text blocks (280 M coded
  // elements in the longest section = 4.5 s in a lane), 16 threads, a block begun every 45 ms: 48 deep = the scalar
  // route's rate, 56 the break-even, 64 deep 1.5x, 96 deep 2.3x -- i.e. at 0.56 of the blocks that are begun while a
  // lane holds the longest section.  Away from that point the same ratio is used: fusedDepth(seconds between blocks).
  static constexpr unsigned kFusedDepth = 56;
  static unsigned fusedDepth(double block_interval_s) {
    if (!(block_interval_s > 0.0)) return kFusedDepth;
    const double d = 0.5625 * 4.48 / block_interval_s;
    return d < 24.0 ? 24u : d > 256.0 ? 256u : static_cast<unsigned>(d + 0.5);
  }
  static constexpr unsigned kFusedBacklog = 6;
  // Blocks a caller has to keep under way for the measured rate: a block's mean time from begun to record
  // finished over the mean time between two begins, a quarter more, and two (0 until four blocks have finished).
  unsigned depthNeeded() const {
    const uint64_t n = clock.finished.load(), iv = begin_interval_ns_.load();
    if (n < 4 || iv == 0) return 0;
    const double latency = static_cast<double>(clock.latency_ns.load()) / static_cast<double>(n);
    return static_cast<unsigned>(1.25 * latency / static_cast<double>(iv)) + 3u;
  }
  double beginIntervalSeconds() const { return begin_interval_ns_.load() * 1e-9; }
  void noteBegin() {                                 // called once per block, by the thread that feeds the GPU
    const uint64_t now = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(
        std::chrono::steady_clock::now().time_since_epoch()).count());
    const uint64_t last = last_begin_ns_.exchange(now);
    if (last && now > last) {
      const uint64_t d = now - last, old = begin_interval_ns_.load();
      begin_interval_ns_.store(old ? (old * 7 + d) / 8 : d);
    }
  }
  StageClock clock;
 private:
  std::atomic<uint64_t> last_begin_ns_{0}, begin_interval_ns_{0};
  enum { kSpares = 40 };
  std::atomic<uint32_t> disposed_{0};
  std::mutex spare_mu_;
  std::vector<std::vector<bwtc::wavelet::SectionOutput> > spare_outs_;
  std::vector<std::vector<uint8_t> > spare_records_;
  // n sections' outputs, empty, with whatever room a spare set brings along
  void freshOutputs(WaveletJob& job, size_t n) {
    if (job.outs.empty()) takeSpareOutputs(&job.outs);
    job.outs.resize(n);
    for (size_t s = 0; s < n; ++s) job.outs[s].bytes.clear();
  }
  void submitSections(const std::shared_ptr<WaveletJob>& job);
  void finish(WaveletJob& job);
  BlockGroupSource groups_;                          // before the pool: the workers are joined first
  BlockChainSource chains_;                          // sections for the 16-lane coder engines
  BlockChainSource long_chains_;                     // very long sections: scalar engines, two chains at a time
  BlockChainSource w_chains_;                        // device-modelled blocks: their shorter sections, for the 16-lane engines
  BlockChainSource w_long_chains_;                   //   their longest sections, for scalar engines that step two chains alternately
  unsigned max_w_pair_engines_;                      //   at most this many of those (0: every long section has a scalar task of its own)
  unsigned w_lanes_;                                 //   chains an engine holds (16: one vector, 32: two stepped alternately)
  unsigned max_w_engines_;                           //   at most this many of them (0: no lanes, every chain scalar)
  uint64_t w_long_chain_;                            //   sections this long keep a scalar task of their own
  BlockSectionSource fused_sections_;                // fused model + coder engines
  WorkerPool pool_;
  uint64_t huge_;
  unsigned max_engines_;
  unsigned max_coder_engines_;                       // 16-lane range-coder engines (0: scalar pairs per block)
  unsigned max_pair_engines_;                        // scalar engines for the very long chains
  uint64_t long_chain_;                              // elements from which a chain counts as very long
  unsigned max_fused_engines_;                       // > 0: 'B' blocks go to the fused engines, no probability arrays
  size_t coder_tasks_;
};

inline int BlockGroupSource::add(const std::shared_ptr<WaveletJob>& job, size_t first_group, unsigned max_engines) {
  Entry e;
  e.job = job;
  e.at[0] = e.at[1] = 0;
  for (size_t k = first_group; k < job->coder->modelTasks(); ++k)
    e.list[job->coder->taskIsInteger(k) ? 0 : 1].push_back(static_cast<uint32_t>(k));
  std::lock_guard<std::mutex> g(mu_);
  queue_.push_back(std::move(e));
  if (engines_[1] + engines_[2] >= max_engines) return 0;
  // engines of one kind keep their lanes to one kind of group while there is enough of it: a
  // vector of integer-level lanes needs 3 of the 15 predictor selects
  const int prefer = engines_[1] * 2 < engines_[2] || (engines_[2] > 0 && engines_[1] == 0 && max_engines > 1) ? 1 : 2;
  ++engines_[prefer];
  return prefer;
}

inline bool BlockGroupSource::next(bwtc::wavelet::ModelGroupDesc* d, int prefer) {
  std::shared_ptr<WaveletJob> job;
  size_t k = 0;
  {
    std::lock_guard<std::mutex> g(mu_);
    while (!queue_.empty() && queue_.front().at[0] >= queue_.front().list[0].size() &&
           queue_.front().at[1] >= queue_.front().list[1].size())
      queue_.pop_front();
    if (queue_.empty()) return false;
    Entry* from = nullptr;
    int kind = prefer == 1 ? 0 : 1;
    for (size_t i = 0; i < queue_.size() && prefer; ++i)       // the preferred kind, oldest block first
      if (queue_[i].at[kind] < queue_[i].list[kind].size()) { from = &queue_[i]; break; }
    if (!from) {                                               // none left anywhere: whatever the oldest block has
      from = &queue_.front();
      kind = from->at[0] < from->list[0].size() ? 0 : 1;
      if (prefer == 0 && from->at[1] < from->list[1].size() &&
          (kind == 1 || from->list[1][from->at[1]] < from->list[0][from->at[0]])) kind = 1;   // largest first
    }
    job = from->job;
    k = from->list[kind][from->at[kind]++];
  }
  job->coder->describe(k, job->prob.data(), d);
  d->cookie = job.get();
  return true;
}

inline bool BlockChainSource::add(const std::shared_ptr<WaveletJob>& job, size_t first, size_t last, unsigned max_engines) {
  if (first >= last) return false;
  Entry e;
  e.job = job;
  e.at = first;
  e.end = last;
  std::lock_guard<std::mutex> g(mu_);
  queue_.push_back(std::move(e));
  if (engines_ >= max_engines) return false;
  ++engines_;
  return true;
}

inline bool BlockChainSource::next(bwtc::wavelet::ChainDesc* d) {
  std::shared_ptr<WaveletJob> job;
  size_t k = 0;
  {
    std::lock_guard<std::mutex> g(mu_);
    while (!queue_.empty() && queue_.front().at >= queue_.front().end) queue_.pop_front();
    if (queue_.empty()) return false;
    job = queue_.front().job;
    k = queue_.front().at++;
  }
  if (job->w) job->coder->describeChainW(k, job->w, &job->outs, d);
  else job->coder->describeChain(k, job->prob.data(), &job->outs, d);
  d->cookie = job.get();
  return true;
}

inline void BlockChainSource::done(void* cookie) {
  WaveletJob* job = static_cast<WaveletJob*>(cookie);
  const std::shared_ptr<WaveletJob> keep = job->shared_from_this();   // whoever waits for the block may drop it the moment it is done
  bool last;
  { std::lock_guard<std::mutex> g(job->mu); last = --job->sections_left == 0; }
  if (last) on_block_coded(*job);
}

inline bool BlockSectionSource::add(const std::shared_ptr<WaveletJob>& job, unsigned max_engines) {
  Entry e;
  e.job = job;
  e.at = 0;
  std::lock_guard<std::mutex> g(mu_);
  queue_.push_back(std::move(e));
  if (engines_ >= max_engines) return false;
  ++engines_;
  return true;
}

inline bool BlockSectionSource::next(bwtc::wavelet::FusedSectionDesc* d) {
  std::shared_ptr<WaveletJob> job;
  size_t k = 0;
  {
    std::lock_guard<std::mutex> g(mu_);
    while (!queue_.empty() && queue_.front().at >= queue_.front().job->coder->sectionTasks()) queue_.pop_front();
    if (queue_.empty()) return false;
    job = queue_.front().job;
    k = queue_.front().at++;
  }
  job->coder->describeSectionFused(k, &job->outs, &job->fused_groups[k], d);
  d->cookie = job.get();
  return true;
}

inline void BlockSectionSource::done(void* cookie) {
  WaveletJob* job = static_cast<WaveletJob*>(cookie);
  const std::shared_ptr<WaveletJob> keep = job->shared_from_this();
  bool last;
  { std::lock_guard<std::mutex> g(job->mu); last = --job->sections_left == 0; }
  if (last) on_block_coded(*job);
}

inline void BlockGroupSource::done(void* cookie) {
  WaveletJob* job = static_cast<WaveletJob*>(cookie);
  bool last;
  { std::lock_guard<std::mutex> g(job->mu); last = --job->models_left == 0; }
  if (last) on_block_modelled(job->shared_from_this());
}

}  // namespace bwtc_hip
