// WaveletEncoder for a device-resident transformed block (WaveletCoders.cpp:173-219 header,
// :112-157 encodeData, :159-163 finishBlock), split into a device half and a host half so that
// consecutive blocks overlap (wavelet_pipeline.hpp):
//   _begin  block header, run scanner + statistics (huffman_frontend.hip), plan
//           (wavelet_host.cpp planStreams), stream kernels (wavelet_tree.hip), hand-over of the
//           carried model state, models + range coders queued on the worker pool
//   _end    waits for the block's last section and reports the record
#include "bwt_engine.hpp"
#include "entropy_host.hpp"
#include "wavelet_host.hpp"
#include "wavelet_pipeline.hpp"
#include <algorithm>
#include <chrono>
#include <future>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <utility>
#include <vector>

namespace bwtc_hip {

// The device half of a block comes in two parts with a piece of host work between them:
//   scan     run scanner + section statistics (one sweep over the transformed block); the runs go
//            to one of the engine's two run-array buffers, not into the transform's workspace
//   plan     (host, ~3 ms) code sets, tree shapes, coding order of every section -- started on a
//            helper thread as soon as the statistics are on the host
//   streams  expand / sort / select / pack kernels, copy of the packed streams to the host
// With _begin called block after block, the GPU would idle through every plan.  So _begin(i) does
// scan(i), starts plan(i), and only then runs streams(i-1) -- whose plan was made while the GPU
// transformed block i -- and queues block i-1 on the host pipeline.  Block i's second part happens
// in the next _begin, or when somebody needs it (_queue / _end of that block).  The records are
// the same; a block just joins the host pipeline one call later.

// second part of a block whose first part is done: streams (or the host route), buffers
static int wavelet_finish_device_half(BwtEngine& e, const std::shared_ptr<DeviceWaveletJob>& jobp, bool async_copy = false,
                                      bool farm_prepare = false) {
  DeviceWaveletJob& job = *jobp;
  if (!job.half) return 0;
  job.half = false;
  if (!job.plan_future.valid()) return 0;                 // empty block: nothing to plan or code
  const bool debug = std::getenv("BWTC_HIP_DEBUG") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  const bool planned = job.plan_future.get();
  const auto t1 = std::chrono::steady_clock::now();
  const WaveletSectionStats& st = *job.host_stats;
  const u32 nsec = (u32)st.sections.size();
  const u32 n_runs = st.first_run[nsec];
  const bool on_device = !e.wavelet_on_host && planned && job.plan.max_elements + (1u << 16) < (1ull << 32);
  if (!on_device && !e.wavelet_on_host)
    std::fprintf(stderr, "bwtc_hip: block of %u bytes is outside the stream kernels' range (%llu steps, %zu groups); "
                 "its wavelet trees are built by the host route instead (same bytes, much slower)\n", job.block_size,
                 (unsigned long long)job.plan.max_elements, job.plan.group_type.size());
  const u32* d_run_start = e.d_run_start[job.run_buf];
  const u8* d_run_sym = e.d_run_sym[job.run_buf];
  if (on_device) {
    const bool codes_recycled = e.codes_pool.take(&job.codes_owner);
    // The adaptive models on the device (wavelet_gpu_models.hip) need the carried state BEFORE this
    // block: known when every block begun before it has joined the stream -- the _begin flow, where
    // the block before the previous one joined while the scanner ran.  (_prepare / _queue callers
    // learn the state at _queue time: their models stay on the worker threads.)
    const bool on_gpu = e.device_models && e.wavelet_model == 'B' && e.deferred_queue;
    // A stream farmed over contexts (_prepare / _queue) learns the state at _queue time: the passes'
    // first part (tables, chunk maps, the tiles' state maps) is done now, the rest then.
    const bool on_gpu_later = e.device_models && e.wavelet_model == 'B' && !e.deferred_queue && farm_prepare;
    if (on_gpu)
      for (std::map<u64, std::shared_ptr<DeviceWaveletJob> >::iterator o = e.jobs.begin(); o != e.jobs.end() && o->first < job.rank; ++o)
        if (!o->second->queued) { const int rq = wavelet_encode_queue(e, o->first, e.wavelet_state, &e.wavelet_state); if (rq) return rq; }
    const u32* d_packed = nullptr;
    int rc = wavelet_streams_device(e, d_run_start, d_run_sym, n_runs, st.first_run, job.plan, &job.coded_pos, &job.codes_owner, async_copy, &d_packed);
    if (rc) return rc;
    if (!codes_recycled) e.codes_pool.noteAllocated();
    job.copying = async_copy;
    job.codes = job.codes_owner.data();
    const auto t2 = std::chrono::steady_clock::now();
    if (!e.pipeline) {
      e.pipeline = new HostPipeline(job.host_threads, e.huge_group_elements, e.max_inflight);
      if (!e.worker_cpus.empty()) e.pipeline->setWorkerCpus(e.worker_cpus);
    }
    if (on_gpu_later && e.wt_coded) {
      // (another prepared block's passes may still wait in the workspace: its turn is lost, its models
      // go to the worker threads)
      if (std::shared_ptr<DeviceWaveletJob> other = e.gm_pending.lock()) if (!other->queued) other->gm.ready = false;
      const bool w_recycled = e.w_pool.take(&job.w_owner);
      const u64 w_bytes = ((u64)e.wt_coded * 2 + 63) / 64 * 64;
      const size_t had = job.w_owner.size();
      if (!job.w_owner.reserve(w_bytes + 64)) return -2;
      if (!w_recycled || job.w_owner.size() != had) e.w_pool.noteAllocated();
      if (e.max_inflight > 2) {                       // as below: the next blocks' buffers are made off this thread
        e.w_pool.allocateAhead(e.device, (size_t)(w_bytes + w_bytes / 64 + 64), e.max_inflight + 1);
        e.codes_pool.allocateAhead(e.device, job.codes_owner.size(), e.max_inflight + 1);
      }
      job.h_tail = reinterpret_cast<u32*>(job.w_owner.data() + w_bytes);
      job.h_tail[0] = job.h_tail[1] = job.h_tail[2] = job.h_tail[3] = 0xFFFFFFFFu;
      rc = wavelet_models_prepare(e, d_packed, (u32)e.wt_coded, job.plan, job.coded_pos, &job.gm, true);
      if (rc) return rc;
      e.gm_pending = jobp;
    }
    if (on_gpu && e.wt_coded) {
      const bool w_recycled = e.w_pool.take(&job.w_owner);
      const u64 w_bytes = ((u64)e.wt_coded * 2 + 63) / 64 * 64;
      const size_t had = job.w_owner.size();
      if (!job.w_owner.reserve(w_bytes + 64)) return -2;
      if (!w_recycled || job.w_owner.size() != had) e.w_pool.noteAllocated();
      if (e.max_inflight > 2) {                       // a pipelined context: the next blocks' buffers are made ahead, off this thread
        e.w_pool.allocateAhead(e.device, (size_t)(w_bytes + w_bytes / 64 + 64), e.max_inflight + 1);
        e.codes_pool.allocateAhead(e.device, job.codes_owner.size(), e.max_inflight + 1);
      }
      job.h_tail = reinterpret_cast<u32*>(job.w_owner.data() + w_bytes);
      job.h_tail[0] = job.h_tail[1] = job.h_tail[2] = job.h_tail[3] = 0xFFFFFFFFu;
      job.gm_state_in = e.wavelet_state;
      rc = wavelet_models_device(e, d_packed, (u32)e.wt_coded, job.plan, job.coded_pos, e.wavelet_state,
                                 reinterpret_cast<uint16_t*>(job.w_owner.data()), job.h_tail);
      if (rc) return rc;
      job.w = reinterpret_cast<const uint16_t*>(job.w_owner.data());
      job.copying = true;                             // the w-elements are on their way (d2h stream) whatever async_copy says
      if (debug)
        std::fprintf(stderr, "wavelet: models of %llu coded elements queued on the device %.2f ms after the streams\n",
                     (unsigned long long)e.wt_coded, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
    }
    job.fused = !job.w && !job.gm.ready && e.pipeline->fusedNow(e.wavelet_model);
    if (!job.fused && !job.w && !job.gm.ready) {
      if (!e.prob_free.empty()) { job.prob.swap(*e.prob_free.back()); e.prob_free.pop_back(); }
      if (!job.prob.reserve(e.wt_coded + 8)) return -2;
    }
    if (debug) {
      uint64_t cnt[8] = {0}, el[8] = {0};
      const uint32_t* pos = job.coded_pos.data();
      for (size_t g = 0; g + 1 < job.coded_pos.size(); ++g) {
        const uint64_t n = pos[g + 1] - pos[g];
        const int b = n >= (32u << 20) ? 7 : n >= (16u << 20) ? 6 : n >= (8u << 20) ? 5 : n >= (4u << 20) ? 4 : n >= (1u << 20) ? 3 : n >= (1u << 16) ? 2 : n >= 256 ? 1 : 0;
        ++cnt[b]; el[b] += n;
      }
      std::fprintf(stderr, "wavelet groups by coded size (<256, <64Ki, <1Mi, <4Mi, <8Mi, <16Mi, <32Mi, more): ");
      for (int b = 0; b < 8; ++b) std::fprintf(stderr, "%llu/%.1fM ", (unsigned long long)cnt[b], el[b] * 1e-6);
      std::fprintf(stderr, "\n");
      std::fprintf(stderr, "wavelet: %u runs, %llu steps, %llu coded, %zu groups; waited %.1f ms for the plan, "
                   "device streams %.1f ms, %u threads\n", n_runs,
                   (unsigned long long)e.wt_elements, (unsigned long long)e.wt_coded, job.plan.group_type.size(),
                   std::chrono::duration<double, std::milli>(t1 - t0).count(),
                   std::chrono::duration<double, std::milli>(t2 - t1).count(), e.pipeline->threads());
    }
    job.streams_ready = true;
  } else {
    // the runs themselves: symbols and start offsets, as the scanner left them
    job.host_run_sym.resize(n_runs);
    job.host_run_start.resize((size_t)n_runs + 1);
    BWTC_HIP_TRY(hipMemcpyAsync(job.host_run_sym.data(), d_run_sym, n_runs, hipMemcpyDeviceToHost, e.stream));
    BWTC_HIP_TRY(hipMemcpyAsync(job.host_run_start.data(), d_run_start, ((size_t)n_runs + 1) * 4, hipMemcpyDeviceToHost, e.stream));
    BWTC_HIP_TRY(e.wait());
    for (u32 s = 0; s < nsec; ++s) {
      job.host_secs[s].symbols = job.host_run_sym.data() + st.first_run[s];
      job.host_secs[s].starts = job.host_run_start.data() + st.first_run[s];
    }
    job.host_route = true;
  }
  return 0;
}

// the block that still waits for its second part, if any; async_copy: its streams' copy to the
// host is left in flight and the block becomes e.copy_job
static int wavelet_finish_pending_half(BwtEngine& e, bool async_copy) {
  if (!e.half_job) return 0;
  std::shared_ptr<DeviceWaveletJob> jobp;
  jobp.swap(e.half_job);
  const int rc = wavelet_finish_device_half(e, jobp, async_copy);
  if (rc == 0 && async_copy && jobp->copying) e.copy_job = jobp;
  return rc;
}

int wavelet_encode_begin(BwtEngine& e, const u8* d_bwt, u32 size, const u32* lf, u32 n_lf,
                         const u32* freqs, unsigned threads, u8* out, u64 out_cap, u64* ticket) {
  // Block i-2: its packed streams were copied to the host under the transform of block i; it joins
  // the host pipeline now.  Block i-1: its plan was made while the GPU transformed block i; its
  // streams are made below, their copy is left in flight.  (Blocks join the stream in order.)
  e.deferred_queue = true;                            // begun blocks join the stream later, or when they are needed
  int rc;
  // block i-2 joins the host pipeline while the GPU scans block i (its descriptors take the
  // calling thread a few hundred microseconds to build)
  int rc_queue = 0;
  const bool debug = std::getenv("BWTC_HIP_DEBUG") != nullptr;
  const std::function<void()> join_earlier = [&e, &rc_queue, debug] {
    if (!e.copy_job) return;
    const auto t0 = std::chrono::steady_clock::now();
    std::shared_ptr<DeviceWaveletJob> ready;
    ready.swap(e.copy_job);
    if (!ready->queued) rc_queue = wavelet_encode_queue(e, ready->rank, e.wavelet_state, &e.wavelet_state);
    if (debug) std::fprintf(stderr, "begin: block joined the host pipeline in %.2f ms (under the run scanner)\n",
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  std::shared_ptr<DeviceWaveletJob> prev = e.half_job;
  e.async_streams_copy = true;
  rc = wavelet_encode_prepare(e, d_bwt, size, lf, n_lf, freqs, threads, out, out_cap, ticket, &join_earlier);   // scan(i), then streams(i-1)
  e.async_streams_copy = false;
  if (rc) return rc;
  if (rc_queue) return rc_queue;
  if (prev && !prev->queued && !prev->copying) {      // no copy in flight (host route, empty block): it can join at once
    rc = wavelet_encode_queue(e, prev->rank, e.wavelet_state, &e.wavelet_state);
    if (rc) return rc;
  }
  return 0;
}

// Second half of _begin: the block joins its stream.  state_in = the main model's carried state
// after the previous block of the stream (whichever context coded that one), *state_out = the
// state after this block.
static int wavelet_encode_queue_unguarded(BwtEngine& e, u64 ticket, u32 state_in, u32* state_out);

// A block whose entry into the stream failed half way is marked failed and done: _end then reports the
// error instead of waiting for a record that nobody is making.
int wavelet_encode_queue(BwtEngine& e, u64 ticket, u32 state_in, u32* state_out) {
  const int rc = wavelet_encode_queue_unguarded(e, ticket, state_in, state_out);
  if (rc) {
    std::map<u64, std::shared_ptr<DeviceWaveletJob> >::iterator it = e.jobs.find(ticket);
    if (it != e.jobs.end() && it->second->queued) {
      DeviceWaveletJob& job = *it->second;
      std::lock_guard<std::mutex> g(job.mu);
      if (!job.done) { job.failed = true; job.done = true; job.cv.notify_all(); }
    }
  }
  return rc;
}

static int wavelet_encode_queue_unguarded(BwtEngine& e, u64 ticket, u32 state_in, u32* state_out) {
  std::map<u64, std::shared_ptr<DeviceWaveletJob> >::iterator it = e.jobs.find(ticket);
  if (it == e.jobs.end() || !state_out) return -1;
  std::shared_ptr<DeviceWaveletJob> jobp = it->second;
  DeviceWaveletJob& job = *jobp;
  if (job.queued) return -1;
  if (job.half) {                                    // its streams are not made yet: now
    if (e.half_job == jobp) e.half_job.reset();
    const int rc = wavelet_finish_device_half(e, jobp);
    if (rc) return rc;
  }
  if (job.copying) {                                 // its streams are on their way to the host
    if (e.copy_job == jobp) e.copy_job.reset();
    job.copying = false;
    if (e.codes_wait() != hipSuccess) return -3;
  }
  job.queued = true;
  if (job.gm.ready && job.streams_ready && e.wavelet_model == 'B') {
    // farmed stream, models on the device: the rest of the passes with the state that has just
    // arrived; the state after the block is looked up (_prepare left it for every state the block can
    // start in: nothing here waits for the device), the block joins the host pipeline from a host
    // callback when its w-elements have landed
    u32 after = state_in;
    int rc = wavelet_models_run(e, job.gm, state_in, reinterpret_cast<uint16_t*>(job.w_owner.data()), job.h_tail, &after);
    job.gm.ready = false;
    if (rc) return rc;
    job.w = reinterpret_cast<const uint16_t*>(job.w_owner.data());
    job.gm_state_in = state_in;
    job.w_end_state = after;
    *state_out = after;
    struct Join { std::shared_ptr<DeviceWaveletJob> job; HostPipeline* pipe; u32 state_in; };
    Join* j = new Join{jobp, e.pipeline, state_in};
    const hipError_t hrc = hipLaunchHostFunc(e.d2h_stream, [](void* p) {
      std::unique_ptr<Join> j(static_cast<Join*>(p));
      DeviceWaveletJob& job = *j->job;
      job.callback_seen.store(true);
      const u32 n_coded = job.coded_pos.empty() ? 0u : job.coded_pos.back();
      const bool ok = job.h_tail[1] == 0 && job.h_tail[2] == n_coded && job.h_tail[3] == 0 && job.h_tail[0] == job.w_end_state &&
                      !std::getenv("BWTC_HIP_TEST_MODELS_FALLBACK");
      if (!ok) {                                      // flagged: the worker threads model it from the packed streams
        job.w = nullptr;
        if (!job.prob.reserve(static_cast<size_t>(n_coded) + 8)) { job.failed = true; HostPipeline::finishNow(job); return; }
      }
      // (the state handed on came from the device's state maps: a host half that ends elsewhere means
      // the stream after this block is wrong -- reported at _end, never passed over)
      const u32 handed_on = job.w_end_state;
      (void)j->pipe->queue(j->job, j->state_in, 'B', &handed_on);
    }, j);
    if (hrc != hipSuccess) {
      // the copies of the w-elements and of the tail words are queued: nothing may hand the block's buffers on
      // (the caller marks it failed, _end gives w_owner back to the pool) while they can still be written
      delete j;
      (void)hipStreamSynchronize(e.d2h_stream);
      (void)hipGetLastError();
      return -3;
    }
    job.joins_from_callback = true;
    return 0;
  }
  if (job.streams_ready && !job.w && !job.fused && job.prob.size() < static_cast<size_t>(job.coded_pos.empty() ? 0 : job.coded_pos.back()) + 8) {
    // a block whose device passes lost their turn (see wavelet_finish_device_half): host models after all
    if (!job.prob.reserve(static_cast<size_t>(job.coded_pos.empty() ? 0 : job.coded_pos.back()) + 8)) return -2;
  }
  if (job.host_route) {
    // shapes the stream kernels do not take: the library's own tree builder, here and now
    u32 st = state_in;
    bwtc::wavelet::encodeSections(job.host_secs, job.host_threads, &st, &job.outs, e.wavelet_model);
    *state_out = st;
    job.host_run_sym.clear(); job.host_run_sym.shrink_to_fit();
    job.host_run_start.clear(); job.host_run_start.shrink_to_fit();
    HostPipeline::finishNow(job);
    return 0;
  }
  if (!job.streams_ready) { *state_out = state_in; HostPipeline::finishNow(job); return 0; }
  if (job.w) {
    // modelled on the device: the copy has landed (above); the passes report themselves
    const u32 n_coded = job.coded_pos.empty() ? 0u : job.coded_pos.back();
    if (std::getenv("BWTC_HIP_TEST_MODELS_FALLBACK")) job.h_tail[1] |= 0x80u;    // tests: pretend the passes flagged the block
    const bool ok = job.h_tail[1] == 0 && job.h_tail[2] == n_coded && job.h_tail[3] == 0 && job.h_tail[0] < 8 && job.gm_state_in == state_in &&
                    e.wavelet_model == 'B';
    if (ok) {
      job.w_end_state = job.h_tail[0];
    } else {
      // never seen; kept so that a surprise costs time, not bytes: the block's models run on the
      // worker threads from its packed streams (which came down as well)
      std::fprintf(stderr, "bwtc_hip: the device models of a block were not used (flags %u, counted %u of %u, state %u/%u); "
                   "its models run on the host instead\n", job.h_tail[1], job.h_tail[2], n_coded, job.gm_state_in, state_in);
      job.w = nullptr;
      if (!job.prob.reserve(static_cast<size_t>(n_coded) + 8)) return -2;
    }
  }
  if (job.fused && e.wavelet_model != 'B') {
    // prepared for the fused engines (no probability array) while the stream's model was 'B', queued
    // after a switch to 'b' / 'u': those go the two-stage way and need the array
    job.fused = false;
    if (!job.prob.reserve(static_cast<size_t>(job.coded_pos.empty() ? 0 : job.coded_pos.back()) + 8)) return -2;
  }
  *state_out = e.pipeline->queue(jobp, state_in, e.wavelet_model);
  return 0;
}

// First half of _begin: everything that does not depend on the blocks before this one.  (The
// streams of THIS block are made later, see above; what is made here are the streams of the block
// prepared before it.)
int wavelet_encode_prepare(BwtEngine& e, const u8* d_bwt, u32 size, const u32* lf, u32 n_lf,
                           const u32* freqs, unsigned threads, u8* out, u64 out_cap, u64* ticket,
                           const std::function<void()>* while_scanning) {
  if (!freqs || !lf || !ticket || n_lf == 0 || n_lf > 256) return -1;
  if (e.stream_start_error) { const int rc = e.stream_start_error; e.stream_start_error = 0; return rc; }   // bwtc_hip_wavelet_reset has no result of its own
  // bounded number of blocks under way; only ..._end frees a place, so waiting here could never help
  if (e.jobs.size() >= e.max_inflight) return -6;
  std::shared_ptr<DeviceWaveletJob> jobp(new DeviceWaveletJob());
  DeviceWaveletJob& job = *jobp;
  job.rank = e.next_ticket;
  job.t_begun = std::chrono::steady_clock::now();
  job.user_out = out;
  job.user_cap = out_cap;
  job.block_size = size;
  job.host_threads = threads;
  // header: WaveletEncoder::writeBlockHeader, WaveletCoders.cpp:173-219
  std::vector<uint8_t>& rec = job.record;
  if (e.pipeline) e.pipeline->takeSpareRecord(&rec);   // a collected block's record: its pages are there already
  rec.assign(6, 0);
  bwtc::writeBWTBlockHeader(lf, n_lf, rec);
  job.host_stats.reset(new WaveletSectionStats());
  WaveletSectionStats& st = *job.host_stats;
  int rc = e.reserve_run_arrays();
  if (rc) return rc;
  job.run_buf = e.next_run_buf;
  e.next_run_buf ^= 1;
  const bool debug_t = std::getenv("BWTC_HIP_DEBUG") != nullptr;
  const auto tp0 = std::chrono::steady_clock::now();
  rc = wavelet_section_stats_device(e, d_bwt, size, freqs, &st, e.d_run_start[job.run_buf], e.d_run_sym[job.run_buf], while_scanning);
  if (rc) return rc;
  const auto tp1 = std::chrono::steady_clock::now();
  const u32 nsec = (u32)st.sections.size();
  rec.push_back((uint8_t)(nsec == 256 ? 0 : nsec));
  for (u32 s = 0; s < nsec; ++s) bwtc::utils::packInteger(st.sections[s], rec);
  job.half = true;
  if (size && nsec) {
    job.host_secs.assign(nsec, bwtc::wavelet::SectionRuns());
    for (u32 s = 0; s < nsec; ++s) {
      bwtc::wavelet::SectionRuns& r = job.host_secs[s];
      r.symbols = nullptr;
      r.starts = nullptr;
      r.n_runs = st.first_run[s + 1] - st.first_run[s];
      r.run_freqs = &st.run_freqs[(size_t)s * 256];
      r.dist = st.dist[s].data();
      r.n_dist = st.dist[s].size();
    }
    // Tree bit vectors and traversal on the GPU (wavelet_tree.hip), models + range coder on the
    // host.  Shapes the device path does not take (planStreams) and BWTC_HIP_WAVELET=host go
    // through the host's own tree builder instead; both are this library's code and give the
    // same bytes.  The plan is made on a helper thread while the GPU does something else.
    DeviceWaveletJob* jp = jobp.get();
    const bool want_plan = !e.wavelet_on_host;
    job.plan_future = std::async(std::launch::async, [jp, want_plan] { return want_plan && bwtc::wavelet::planStreams(jp->host_secs, &jp->plan); });
  }
  // the block prepared before this one: its streams, now that its plan has had the time of a transform
  const auto tp2 = std::chrono::steady_clock::now();
  rc = wavelet_finish_pending_half(e, e.async_streams_copy);
  if (rc) return rc;
  if (debug_t)
    std::fprintf(stderr, "prepare: run scanner + statistics %.2f ms, planner started in %.2f ms, streams of the previous block %.2f ms\n",
                 std::chrono::duration<double, std::milli>(tp1 - tp0).count(), std::chrono::duration<double, std::milli>(tp2 - tp1).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp2).count());
  e.half_job = jobp;
  *ticket = e.next_ticket++;
  e.jobs[*ticket] = jobp;
  if (!e.deferred_queue && e.device_models && e.wavelet_model == 'B' && !e.wavelet_on_host) {
    // _prepare / _queue flow with the models on the device: the streams of THIS block are made now
    // (its plan is waited for: a few milliseconds of GPU idle, once per block and context), so that
    // _queue -- which callers serialise over all contexts of a stream -- only has the state-dependent
    // passes to launch
    e.half_job.reset();
    rc = wavelet_finish_device_half(e, jobp, true, true);
    if (rc) return rc;
  }
  return 0;
}

int wavelet_encode_end(BwtEngine& e, u64 ticket, u64* out_bytes) {
  std::map<u64, std::shared_ptr<DeviceWaveletJob> >::iterator it = e.jobs.find(ticket);
  if (it == e.jobs.end() || !out_bytes) return -1;
  std::shared_ptr<DeviceWaveletJob> jobp = it->second;
  DeviceWaveletJob& job = *jobp;
  if (!job.queued) {
    // _begin defers a block's entry into the stream to the next _begin; nobody else may have
    // blocks of this context waiting (the farm's _prepare / _queue callers queue explicitly)
    if (!e.deferred_queue) return -1;                 // prepared but never given its place in a stream
    // blocks join the stream in order: the ones begun before this one first
    for (std::map<u64, std::shared_ptr<DeviceWaveletJob> >::iterator o = e.jobs.begin(); o != e.jobs.end() && o->first <= ticket; ++o) {
      if (o->second->queued) continue;
      const int rc = wavelet_encode_queue(e, o->first, e.wavelet_state, &e.wavelet_state);
      if (rc) return rc;
    }
  }
  if (job.joins_from_callback) {
    // The block joins the host pipeline from a callback on the d2h stream.  A stream (or context) in error never
    // runs it: look at the stream while waiting, and turn that into an error instead of waiting for ever.
    while (!HostPipeline::waitFor(job, 200)) {
      const hipError_t q = e.d2h_stream ? hipStreamQuery(e.d2h_stream) : hipSuccess;
      if (q != hipSuccess && q != hipErrorNotReady) {
        (void)hipGetLastError();
        std::fprintf(stderr, "bwtc_hip: the copy stream of a block under way reports %s; the block is given up\n", hipGetErrorString(q));
        e.jobs.erase(it);
        return -3;                                    // its buffers are NOT recycled: a copy may still own them
      }
      if (q == hipSuccess && HostPipeline::waitFor(job, 2000)) break;   // the stream has drained: the callback has run (or never will)
      if (q == hipSuccess && !job.callback_seen.load()) {
        std::fprintf(stderr, "bwtc_hip: a block's join callback did not run although its stream has drained\n");
        e.jobs.erase(it);
        return -3;
      }
    }
  } else HostPipeline::wait(job);
  if (std::getenv("BWTC_HIP_DEBUG") && job.coder)
    std::fprintf(stderr, "wavelet block %llu: queued -> modelled %.0f ms, -> finished %.0f ms, -> collected %.0f ms\n",
                 (unsigned long long)ticket,
                 std::chrono::duration<double, std::milli>(job.t_modelled - job.t_queued).count(),
                 std::chrono::duration<double, std::milli>(job.t_finished - job.t_modelled).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - job.t_finished).count());
  e.jobs.erase(it);
  // keep the two big buffers: fresh ones would be paged in again for every block
  e.codes_pool.give(&job.codes_owner, e.max_inflight + 2);
  e.w_pool.give(&job.w_owner, e.max_inflight + 2);
  if (job.prob.size() && e.prob_free.size() < e.max_inflight) {
    e.prob_free.push_back(std::unique_ptr<RawBuffer<uint16_t> >(new RawBuffer<uint16_t>()));
    e.prob_free.back()->swap(job.prob);
  }
  *out_bytes = job.record.size();
  const int rc = job.failed ? -2 : job.record.size() <= job.user_cap ? 0 : -1;
  if (e.pipeline) {                                   // freed by a worker, not by the thread that feeds the GPU
    std::shared_ptr<WaveletJob> last(std::move(jobp));
    e.pipeline->dispose(std::move(last));
  }
  return rc;
}

int wavelet_join_begun(BwtEngine& e) {
  if (!e.deferred_queue) return 0;                    // _prepare / _queue callers place their blocks themselves
  for (std::map<u64, std::shared_ptr<DeviceWaveletJob> >::iterator it = e.jobs.begin(); it != e.jobs.end(); ++it) {
    if (it->second->queued) continue;
    const int rc = wavelet_encode_queue(e, it->first, e.wavelet_state, &e.wavelet_state);
    if (rc) return rc;
  }
  return 0;
}

void wavelet_pipeline_release(BwtEngine& e) {
  // a block begun with _begin joins the stream one call later: the last one still has to
  (void)wavelet_join_begun(e);
  e.half_job.reset();
  e.copy_job.reset();
  for (std::map<u64, std::shared_ptr<DeviceWaveletJob> >::iterator it = e.jobs.begin(); it != e.jobs.end(); ++it)
    if (it->second->queued) HostPipeline::wait(*it->second);   // uncollected blocks are finished, not abandoned half way
  e.jobs.clear();
  if (e.pipeline && std::getenv("BWTC_HIP_DEBUG") && e.pipeline->clock.blocks)
    std::fprintf(stderr, "wavelet pipeline: %llu blocks; host time in models %.3f s, in range coders %.3f s (summed over threads)\n",
                 (unsigned long long)e.pipeline->clock.blocks.load(), e.pipeline->clock.model_ns.load() * 1e-9,
                 e.pipeline->clock.coder_ns.load() * 1e-9);
  if (e.pipeline && std::getenv("BWTC_HIP_DEBUG") && e.pipeline->clock.scalar_ns.load())
    std::fprintf(stderr, "wavelet pipeline: %.3f s of the range coders' time in the scalar tasks of the longest sections, the rest in the lane engines\n",
                 e.pipeline->clock.scalar_ns.load() * 1e-9);
  delete e.pipeline;                              // joins the workers (engines retire when nothing is queued)
  e.pipeline = nullptr;
  e.codes_pool.clear();
  e.w_pool.clear();
  e.prob_free.clear();
}

}  // namespace bwtc_hip
