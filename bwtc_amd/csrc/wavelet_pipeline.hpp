// Host side of the block pipeline of the 'B' coder.  The device work of a block (BWT, run
// scanner, stream kernels) takes a tenth of the time its range coder needs on the host, and the
// range coder of a block is a handful of serial chains, so blocks are overlapped: the calling
// thread does the device work of block i+1 while a fixed pool of worker threads models and
// range-codes blocks i, i-1, ...  The one value that links consecutive blocks (the main
// model's state, probmodels/FSM.hpp:196-205) follows from the last bits of a block's streams
// and is known before its coding starts, so the bytes are those of a strictly sequential
// encoder.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "wavelet_host.hpp"

namespace bwtc_hip {

// Fixed set of threads, two-ended queue: new blocks' work goes to the back, the follow-up work
// of blocks already under way to the front, so the oldest block finishes first.
class WorkerPool {
 public:
  explicit WorkerPool(unsigned threads) : stop_(false) {
    for (unsigned i = 0; i < (threads ? threads : 1u); ++i) workers_.push_back(std::thread([this] { loop(); }));
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
    cv_.notify_all();
    for (size_t i = 0; i < workers_.size(); ++i) workers_[i].join();
  }
  unsigned size() const { return static_cast<unsigned>(workers_.size()); }
  void submit(std::function<void()> f, bool front = false) {
    { std::lock_guard<std::mutex> g(mu_); if (front) q_.push_front(std::move(f)); else q_.push_back(std::move(f)); }
    cv_.notify_one();
  }
  // the tasks keep their order: first of `fs` runs first
  void submitFront(std::vector<std::function<void()> >& fs) {
    { std::lock_guard<std::mutex> g(mu_); for (size_t i = fs.size(); i-- > 0;) q_.push_front(std::move(fs[i])); }
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
        f = std::move(q_.front());
        q_.pop_front();
      }
      f();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()> > q_;
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

// One block between bwtc_hip_wavelet_encode_device_begin and ..._end.
struct WaveletJob {
  std::vector<uint8_t> record;                       // header + section sizes, then the sections, finished in place
  bwtc::wavelet::StreamPlan plan;
  std::vector<uint32_t> coded_pos;
  RawBuffer<uint8_t> codes;                          // packed streams, copied out of the engine's pinned buffer
  RawBuffer<uint16_t> prob;                          // probability of every coded element
  std::unique_ptr<bwtc::wavelet::StreamCoder> coder;
  std::vector<bwtc::wavelet::SectionOutput> outs;
  std::mutex mu;
  std::condition_variable cv;
  size_t models_left = 0, sections_left = 0;
  bool done = false;
  uint8_t* user_out = nullptr;
  uint64_t user_cap = 0;
};

}  // namespace bwtc_hip
