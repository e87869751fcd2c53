// Block farm over several GPU contexts for ONE compressed stream (north star:
// "blocks partition embarrassingly across the 8 GPUs of one node, one BWTBlock per GPU, no
// collectives").  The reference's loop, Compressor.cpp:100-108 over the slices that
// PrecompressorBlock::sliceIntoBlocks made (PrecompressorBlock.cpp:123-134), turned into:
//
//   reader (caller's thread)   fills page-locked buffers with consecutive blocks; block k goes
//                              to worker k mod N
//   worker d (one thread,      uploads (copy stream, the next block's upload runs under this
//   one context per device)    block's kernels), transforms, runs the device half of the coder;
//                              'B': takes its turn to join the stream -- the main model's carried
//                              state after block k-1 is handed on through
//                              bwtc_hip_wavelet_encode_queue (FSM8 keeps its state over
//                              resetModel, probmodels/FSM.hpp:196-205; it is known as soon as a
//                              block's streams exist) -- and collects the record when the
//                              context's worker threads have coded it
//   writer (caller's thread)   takes the records in block order
//
// Neighbouring slices are staged (copied to page-locked memory and on to the device), never
// transformed in place: slice i's sentinel slot is slice i+1's first byte.
// The bytes are those of the sequential encoder whatever the number of devices; the same device
// may be listed twice (two contexts on one GPU), which is how the GPU test checks that.
#pragma once
#include <pthread.h>
#include <sched.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "bwtc_hip.hpp"

namespace bwtc {

class BlockFarm {
 public:
  enum { kStaging = 4 };                 // page-locked input buffers per worker
  // coder: 'H' or a wavelet letter.  depth = blocks one context keeps between the device half and
  // the collected record ('B'); host threads are split evenly between the contexts.
  BlockFarm(const std::vector<int>& devices, uint32 maxBlock, char coder, uint32 startingPoints, unsigned depth = 12)
      : m_maxBlock(maxBlock), m_coder(coder), m_sp(startingPoints), m_depth(depth ? depth : 1), m_submitted(0),
        m_inputDone(false), m_nextQueue(0), m_state(4), m_nextWrite(0), m_failed(false) {
    if (devices.empty()) { std::fprintf(stderr, "bwtc-hip: block farm without devices\n"); std::exit(1); }
    // (what the process may keep busy -- the cgroup's quota counts, not only the hardware threads: a
    // farm that started 64 workers per context inside a 16-CPU quota was throttled as a whole)
    const unsigned cpus = std::max(1u, (unsigned)bwtc_hip_host_usable_cpus());
    m_threadsPerContext = std::max(1u, std::min(64u, cpus / (unsigned)devices.size()));
    for (size_t d = 0; d < devices.size(); ++d) {
      std::unique_ptr<Worker> w(new Worker());
      w->device = devices[d];
      hipFatal(bwtc_hip_create(devices[d], maxBlock, &w->ctx), "bwtc_hip_create");
      if (coder != 'H') hipFatal(bwtc_hip_wavelet_start(w->ctx, coder), "bwtc_hip_wavelet_start");
      for (int b = 0; b < kStaging; ++b) {
        w->h_in[b] = static_cast<byte*>(bwtc_hip_host_alloc(w->ctx, (uint64)maxBlock + 64));
        if (!w->h_in[b]) hipFatal(-2, "farm staging buffers");
        w->h_free[b] = true;
      }
      for (int b = 0; b < 2; ++b) {
        w->d_in[b] = bwtc_hip_malloc(w->ctx, (uint64)maxBlock + 64);
        if (!w->d_in[b]) hipFatal(-2, "farm device buffers");
      }
      if (coder == 'H') {
        w->d_comp = bwtc_hip_malloc(w->ctx, bwtc_hip_compress_bound(maxBlock));
        if (!w->d_comp) hipFatal(-2, "farm record buffer");
      }
      m_workers.push_back(std::move(w));
    }
    // Every context's worker threads on the NUMA node of its GPU, on CPUs of their own: the contexts
    // of one node split that node's CPUs (bwtc_hip_host_cpu_slice); BWTC_HIP_FARM_AFFINITY=0 leaves
    // the threads where the system puts them.
    const char* aff = std::getenv("BWTC_HIP_FARM_AFFINITY");
    if (!(aff && aff[0] == '0')) {
      std::vector<int> node(m_workers.size());
      for (size_t d = 0; d < m_workers.size(); ++d) node[d] = bwtc_hip_numa_node(m_workers[d]->ctx);
      for (size_t d = 0; d < m_workers.size(); ++d) {
        uint32_t idx = 0, cnt = 0;
        for (size_t o = 0; o < m_workers.size(); ++o) if (node[o] == node[d]) { if (o < d) ++idx; ++cnt; }
        std::vector<uint32_t> cpus(4096);
        const int n = bwtc_hip_host_cpu_slice(node[d], idx, cnt, cpus.data(), (uint32_t)cpus.size());
        if (n > 0) {
          m_workers[d]->cpus.assign(cpus.begin(), cpus.begin() + n);
          hipFatal(bwtc_hip_set_worker_cpus(m_workers[d]->ctx, cpus.data(), (uint32_t)n), "bwtc_hip_set_worker_cpus");
          m_threadsPerContext = std::max(1u, std::min(m_threadsPerContext, (unsigned)n));
        }
      }
    }
    for (size_t d = 0; d < m_workers.size(); ++d) m_workers[d]->thread = std::thread([this, d] { run(*m_workers[d]); });
  }

  ~BlockFarm() {
    finishInput();
    for (size_t d = 0; d < m_workers.size(); ++d) if (m_workers[d]->thread.joinable()) m_workers[d]->thread.join();
    for (size_t d = 0; d < m_workers.size(); ++d) {
      Worker& w = *m_workers[d];
      for (int b = 0; b < 2; ++b) bwtc_hip_free(w.ctx, w.d_in[b]);
      for (int b = 0; b < kStaging; ++b) bwtc_hip_host_free(w.ctx, w.h_in[b]);
      if (w.d_comp) bwtc_hip_free(w.ctx, w.d_comp);
      // the context first: destroying it lets blocks still under way finish, and they write their records into the
      // pending buffers (a worker that gave up early leaves some behind)
      bwtc_hip_destroy(w.ctx);
      for (size_t i = 0; i < w.outFree.size(); ++i) std::free(w.outFree[i]);
      for (size_t i = 0; i < w.pending.size(); ++i) std::free(w.pending[i].out);
    }
  }

  size_t devices() const { return m_workers.size(); }

  // ---- reader side ---------------------------------------------------------------------------
  // Page-locked buffer (maxBlock bytes) for the next block; waits while all of the target worker's
  // staging buffers are still in use.  (Four of them: with two, reading a block -- 60 ms for 256 MiB
  // from the page cache -- and the worker's 75 ms per block took turns instead of overlapping.)
  byte* nextInput() {
    Worker& w = *m_workers[m_submitted % m_workers.size()];
    std::unique_lock<std::mutex> g(m_mu);
    int free_one = -1;
    m_cv.wait(g, [&] {
      for (int b = 0; b < kStaging && free_one < 0; ++b) if (w.h_free[b]) free_one = b;
      return free_one >= 0 || m_failed;
    });
    m_fill = free_one < 0 ? 0 : free_one;
    return w.h_in[m_fill];
  }
  // the buffer from nextInput() holds `size` bytes: block number = order of submission
  void submit(uint32 size) {
    Worker& w = *m_workers[m_submitted % m_workers.size()];
    {
      std::lock_guard<std::mutex> g(m_mu);
      w.h_free[m_fill] = false;
      Job j; j.index = m_submitted; j.size = size; j.buf = m_fill;
      w.jobs.push_back(j);
      ++m_submitted;
    }
    m_cv.notify_all();
  }
  void finishInput() {
    { std::lock_guard<std::mutex> g(m_mu); m_inputDone = true; }
    m_cv.notify_all();
  }

  // ---- writer side ---------------------------------------------------------------------------
  // Next record in block order.  wait = false: only if it is ready.  Returns false when nothing
  // (more) can be delivered: every submitted block is out (after finishInput) or wait was false.
  bool nextRecord(std::vector<byte>* record, uint32* blockSize, bool wait) {
    std::unique_lock<std::mutex> g(m_mu);
    for (;;) {
      std::map<size_t, Done>::iterator it = m_done.find(m_nextWrite);
      if (it != m_done.end()) {
        record->swap(it->second.record);
        *blockSize = it->second.size;
        m_done.erase(it);
        ++m_nextWrite;
        return true;
      }
      if (m_failed) { std::fprintf(stderr, "bwtc-hip: a farm worker failed\n"); std::exit(1); }
      if (!wait || (m_inputDone && m_nextWrite >= m_submitted)) return false;
      m_cv.wait(g);
    }
  }
  size_t written() const { return m_nextWrite; }

 private:
  struct Job { size_t index; uint32 size; int buf; int dev; };   // buf: host staging buffer, dev: device buffer
  struct Done { std::vector<byte> record; uint32 size; };
  // out: where the context's workers leave the record -- compress_bound(maxBlock) bytes, NOT
  // zero-filled (only the record's bytes are ever touched) and recycled through Worker::outFree
  struct Pending { size_t index; uint32 size; uint64_t ticket; byte* out; };
  struct Worker {
    int device = 0;
    bwtc_hip_ctx* ctx = nullptr;
    byte* h_in[kStaging] = {nullptr, nullptr, nullptr, nullptr};
    void* d_in[2] = {nullptr, nullptr};
    bool h_free[kStaging] = {true, true, true, true};
    void* d_comp = nullptr;
    std::deque<Job> jobs;                 // assigned, not yet taken (guarded by the farm's mutex)
    std::deque<Pending> pending;          // 'B': queued on the context, record not collected yet (worker's own)
    std::vector<byte*> outFree;           // 'B': record buffers not in use (at most depth + 1 exist per worker)
    std::vector<uint32_t> cpus;           // where this context's worker threads (and its feeding thread) run
    size_t taken = 0;                     // jobs taken so far: job number t uses device buffer t & 1
    std::thread thread;
  };

  void fail(int rc, const char* what) {
    std::fprintf(stderr, "bwtc-hip: %s failed with code %d\n", what, rc);
    { std::lock_guard<std::mutex> g(m_mu); m_failed = true; }
    m_cv.notify_all();
  }
  void deliver(size_t index, uint32 size, std::vector<byte>& record) {
    {
      std::lock_guard<std::mutex> g(m_mu);
      Done& d = m_done[index];
      d.record.swap(record);
      d.size = size;
    }
    m_cv.notify_all();
  }
  bool collectOldest(Worker& w) {
    Pending p = w.pending.front();
    w.pending.pop_front();
    uint64_t n = 0;
    const int rc = bwtc_hip_wavelet_encode_end(w.ctx, p.ticket, &n);
    if (rc) { std::free(p.out); fail(rc, "bwtc_hip_wavelet_encode_end"); return false; }
    std::vector<byte> record(p.out, p.out + (size_t)n);              // the record's bytes only
    w.outFree.push_back(p.out);
    deliver(p.index, p.size, record);
    return true;
  }

  // one worker = one thread = one context
  void run(Worker& w) {
    if (!w.cpus.empty()) {                    // the thread that feeds the GPU sits with its workers
      cpu_set_t set;
      CPU_ZERO(&set);
      for (size_t i = 0; i < w.cpus.size(); ++i) if (w.cpus[i] < CPU_SETSIZE) CPU_SET(w.cpus[i], &set);
      (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }
    bool have = false, uploaded = false;      // `cur` taken / its upload issued
    Job cur;
    for (;;) {
      if (!have) {
        std::unique_lock<std::mutex> g(m_mu);
        // while waiting for work, finished records are worth collecting ('B'): do not sleep on them
        m_cv.wait(g, [&] { return !w.jobs.empty() || m_inputDone || m_failed; });
        if (m_failed) return;
        if (w.jobs.empty()) break;                                 // input done and nothing left for this worker
        cur = w.jobs.front();
        w.jobs.pop_front();
        cur.dev = (int)(w.taken++ & 1);
        have = true;
        uploaded = false;
      }
      int rc;
      const bool debug = std::getenv("BWTC_HIP_DEBUG") != nullptr;
      const auto tA = std::chrono::steady_clock::now();
      auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count(); };
      if (!uploaded) {
        rc = bwtc_hip_memcpy_to_device_async(w.ctx, w.d_in[cur.dev], w.h_in[cur.buf], cur.size);
        if (rc) return fail(rc, "bwtc_hip_memcpy_to_device_async");
      }
      rc = bwtc_hip_copy_wait(w.ctx);
      if (rc) return fail(rc, "bwtc_hip_copy_wait");
      // this block's staging buffer is free again; the next block of this worker, if the reader
      // has already delivered it, goes up while this one is transformed
      Job next;
      bool haveNext = false;
      {
        std::lock_guard<std::mutex> g(m_mu);
        w.h_free[cur.buf] = true;
        if (!w.jobs.empty()) { next = w.jobs.front(); w.jobs.pop_front(); next.dev = (int)(w.taken++ & 1); haveNext = true; }
      }
      m_cv.notify_all();
      if (haveNext) {
        // the other device buffer: its last user (the block before this one) is done with it
        rc = bwtc_hip_memcpy_to_device_async(w.ctx, w.d_in[next.dev], w.h_in[next.buf], next.size);
        if (rc) return fail(rc, "bwtc_hip_memcpy_to_device_async");
      }
      uint8_t* d_blk = static_cast<uint8_t*>(w.d_in[cur.dev]);
      uint32 lf[256], freqs[256];
      std::memset(freqs, 0, sizeof freqs);
      const uint32 n_lf = bwtc_hip_n_lf(cur.size, m_sp);
      const auto tB = std::chrono::steady_clock::now();
      rc = bwtc_hip_bwt_block_device(w.ctx, d_blk, d_blk, cur.size, lf, n_lf, freqs);
      if (rc) return fail(rc, "bwtc_hip_bwt_block_device");
      const auto tC = std::chrono::steady_clock::now();
      if (m_coder == 'H') {
        uint64_t n = 0;
        rc = bwtc_hip_huffman_encode_device(w.ctx, d_blk, cur.size, lf, n_lf, freqs, static_cast<uint8_t*>(w.d_comp),
                                            bwtc_hip_compress_bound(m_maxBlock), &n);
        if (rc) return fail(rc, "bwtc_hip_huffman_encode_device");
        std::vector<byte> out((size_t)n);                              // sized once the record's length is known
        rc = bwtc_hip_memcpy_to_host(w.ctx, out.data(), w.d_comp, n);
        if (rc) return fail(rc, "bwtc_hip_memcpy_to_host");
        deliver(cur.index, cur.size, out);
      } else {
        while (w.pending.size() >= m_depth) if (!collectOldest(w)) return;
        const auto tD = std::chrono::steady_clock::now();
        Pending p;
        p.index = cur.index; p.size = cur.size;
        const uint64_t cap = bwtc_hip_compress_bound(m_maxBlock);
        if (w.outFree.empty()) {
          p.out = static_cast<byte*>(std::malloc((size_t)cap));
          if (!p.out) return fail(-2, "farm record buffer");
        } else { p.out = w.outFree.back(); w.outFree.pop_back(); }
        rc = bwtc_hip_wavelet_encode_device_prepare(w.ctx, d_blk, cur.size, lf, n_lf, freqs, m_threadsPerContext,
                                                    p.out, cap, &p.ticket);
        if (rc) { std::free(p.out); return fail(rc, "bwtc_hip_wavelet_encode_device_prepare"); }
        const auto tE = std::chrono::steady_clock::now();
        std::chrono::steady_clock::time_point tF;
        {
          // the block's place in the stream: after block index - 1, whichever context holds that one
          std::unique_lock<std::mutex> g(m_mu);
          m_cv.wait(g, [&] { return m_nextQueue == cur.index || m_failed; });
          if (m_failed) return;
          uint32_t after = 0;
          tF = std::chrono::steady_clock::now();
          rc = bwtc_hip_wavelet_encode_queue(w.ctx, p.ticket, m_state, &after);
          if (rc == 0) { m_state = after; ++m_nextQueue; }
        }
        m_cv.notify_all();
        if (rc) { std::free(p.out); return fail(rc, "bwtc_hip_wavelet_encode_queue"); }
        if (debug)
          std::fprintf(stderr, "farm: block %zu: upload %.1f ms, transform %.1f, collect %.1f, prepare %.1f, turn %.1f, queue %.1f\n",
                       cur.index, ms(tA, tB), ms(tB, tC), ms(tC, tD), ms(tD, tE), ms(tE, tF), ms(tF, std::chrono::steady_clock::now()));
        w.pending.push_back(p);
      }
      if (haveNext) { cur = next; have = true; uploaded = true; } else have = false;
    }
    while (!w.pending.empty()) if (!collectOldest(w)) return;
  }

  uint32 m_maxBlock;
  char m_coder;
  uint32 m_sp;
  unsigned m_depth, m_threadsPerContext;
  std::vector<std::unique_ptr<Worker> > m_workers;
  std::mutex m_mu;
  std::condition_variable m_cv;
  size_t m_submitted;
  int m_fill = 0;
  bool m_inputDone;
  size_t m_nextQueue;                     // 'B': the block whose turn it is to join the stream
  uint32_t m_state;                       //      the carried model state after block m_nextQueue - 1
  std::map<size_t, Done> m_done;
  size_t m_nextWrite;
  bool m_failed;
};

// Compressor.cpp:65-118 over a BlockFarm: a reader thread fills the staging buffers in block order,
// the caller's thread writes the records in block order (one thread doing both spent 60 ms reading and
// 15 ms writing per 256 MiB block, one after the other: more than a GPU needs for the block).
inline size_t Compressor::compressFarmed(const std::vector<int>& devices, uint32 startingPoints, unsigned depth) {
  size_t compressedSize = writeGlobalHeader();
  const size_t bs = bwtBlockSize();
  BlockFarm farm(devices, (uint32)bs, m_options.entropyCoder, startingPoints, depth ? depth : 12);
  std::thread reader([&] {
    for (;;) {
      byte* buf = farm.nextInput();
      const size_t got = m_in->readBlock(buf, bs);
      if (got == 0) break;
      farm.submit((uint32)got);
    }
    farm.finishInput();
  });
  std::vector<byte> record;
  uint32 blockSize = 0;
  while (farm.nextRecord(&record, &blockSize, true)) {
    compressedSize += writeBlockPrefix(blockSize);
    m_out->writeBlock(record.data(), record.data() + record.size());
    compressedSize += record.size();
  }
  reader.join();
  m_out->writeByte(0); ++compressedSize;                          // writeEmptyHeader, Compressor.cpp:115
  m_out->flush();
  return compressedSize;
}

}  // namespace bwtc
