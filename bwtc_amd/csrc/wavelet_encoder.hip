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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <utility>
#include <vector>

namespace bwtc_hip {

// Appends the finished sections to the record and closes it (finishBlock,
// WaveletCoders.cpp:159-163); the worker that finishes a block's last section runs this.
static void finish_wavelet_job(WaveletJob& job, StageClock* clock = nullptr) {
  if (clock) --clock->unfinished;
  std::vector<uint8_t>& rec = job.record;
  size_t total = rec.size();
  for (size_t s = 0; s < job.outs.size(); ++s) total += job.outs[s].bytes.size();
  rec.reserve(total);
  for (size_t s = 0; s < job.outs.size(); ++s) rec.insert(rec.end(), job.outs[s].bytes.begin(), job.outs[s].bytes.end());
  const u64 len = rec.size() - 6;
  for (int i = 0; i < 6; ++i) rec[i] = (uint8_t)(len >> (8 * (5 - i)));
  if (job.user_out && rec.size() <= job.user_cap) std::memcpy(job.user_out, rec.data(), rec.size());
  job.t_finished = std::chrono::steady_clock::now();
  { std::lock_guard<std::mutex> g(job.mu); job.done = true; }
  job.cv.notify_all();
}

// Every group of the block is modelled: its sections' range coders go to the front of the
// queue, ahead of newer blocks' work.
static void submit_sections(const std::shared_ptr<WaveletJob>& jobp, WorkerPool* pool, StageClock* clock) {
  jobp->t_modelled = std::chrono::steady_clock::now();
  // two tasks per block, each stepping two sections' chains at a time (StreamCoder::
  // codeSectionsPaired): the pair costs a quarter less host time than two chains run apart, and
  // with two tasks the block's longest chain is alone again (at its own full speed) as soon as
  // the other sections are used up
  std::vector<std::function<void()> > next;
  // (one task -- the longest chain paired all the way -- costs a tenth less host time again but
  // lengthens every block by 0.3 s; measured slower over 48 blocks, equal over 96)
  static const size_t kTasks = [] { const char* v = std::getenv("BWTC_HIP_CODER_TASKS"); return v && std::atoi(v) > 0 ? (size_t)std::atoi(v) : (size_t)2; }();
  const size_t want = kTasks;
  const size_t engines = std::min<size_t>(want, (jobp->coder->sectionTasks() + 1) / 2);
  for (size_t q = 0; q < std::max<size_t>(engines, 1); ++q) {
    next.push_back([jobp, clock] {
      WaveletJob& j = *jobp;
      const auto t0 = std::chrono::steady_clock::now();
      const size_t did = j.coder->codeSectionsPaired(&j.section_cursor, j.prob.data(), &j.outs);
      clock->coder_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      bool fin;
      { std::lock_guard<std::mutex> g(j.mu); j.sections_left -= did; fin = did > 0 && j.sections_left == 0; }
      if (fin) finish_wavelet_job(j, clock);
    });
  }
  pool->submit(jobp->rank, next);
}

int wavelet_encode_begin(BwtEngine& e, const u8* d_bwt, u32 size, const u32* lf, u32 n_lf,
                         const u32* freqs, unsigned threads, u8* out, u64 out_cap, u64* ticket) {
  if (!freqs || !lf || !ticket || n_lf == 0 || n_lf > 256) return -1;
  // bounded number of blocks under way: wait for the oldest one (its result stays collectable)
  while (e.jobs.size() >= e.max_inflight) {
    WaveletJob& oldest = *e.jobs.begin()->second;
    std::unique_lock<std::mutex> g(oldest.mu);
    if (oldest.done) return -6;                     // finished blocks must be collected with ..._end first
    oldest.cv.wait(g, [&] { return oldest.done; });
  }
  std::shared_ptr<WaveletJob> jobp(new WaveletJob());
  WaveletJob& job = *jobp;
  job.rank = e.next_ticket;
  job.user_out = out;
  job.user_cap = out_cap;
  // header: WaveletEncoder::writeBlockHeader, WaveletCoders.cpp:173-219
  std::vector<uint8_t>& rec = job.record;
  rec.assign(6, 0);
  bwtc::writeBWTBlockHeader(lf, n_lf, rec);
  WaveletSectionStats st;
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = wavelet_section_stats_device(e, d_bwt, size, freqs, &st);
  if (rc) return rc;
  const u32 nsec = (u32)st.sections.size();
  rec.push_back((uint8_t)(nsec == 256 ? 0 : nsec));
  for (u32 s = 0; s < nsec; ++s) bwtc::utils::packInteger(st.sections[s], rec);
  bool queued = false;
  if (size && nsec) {
    const u32 n_runs = st.first_run[nsec];
    std::vector<bwtc::wavelet::SectionRuns> secs(nsec);
    for (u32 s = 0; s < nsec; ++s) {
      bwtc::wavelet::SectionRuns& r = secs[s];
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
    // same bytes.
    const bool debug = std::getenv("BWTC_HIP_DEBUG") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const bool on_device = !e.wavelet_on_host && bwtc::wavelet::planStreams(secs, &job.plan) &&
                           job.plan.max_elements + (1u << 16) < (1ull << 32);
    if (!on_device && !e.wavelet_on_host)
      std::fprintf(stderr, "bwtc_hip: block of %u bytes is outside the stream kernels' range (%llu steps, %zu groups); "
                   "its wavelet trees are built by the host route instead (same bytes, much slower)\n", size,
                   (unsigned long long)job.plan.max_elements, job.plan.group_type.size());
    if (on_device) {
      const auto t1 = std::chrono::steady_clock::now();
      if (!e.codes_free.empty()) { job.codes.swap(*e.codes_free.back()); e.codes_free.pop_back(); }
      rc = wavelet_streams_device(e, n_runs, st.first_run, job.plan, &job.coded_pos, &job.codes);
      if (rc) return rc;
      const auto t2 = std::chrono::steady_clock::now();
      job.coder.reset(new bwtc::wavelet::StreamCoder(job.plan, job.coded_pos.data(), job.codes.data(), e.wavelet_state,
                                                      e.wavelet_model));
      if (!e.prob_free.empty()) { job.prob.swap(*e.prob_free.back()); e.prob_free.pop_back(); }
      if (!job.prob.reserve(job.coder->elements() + 8)) return -2;
      e.wavelet_state = job.coder->endState();        // nothing can fail from here on: the stream moves on
      job.outs.assign(nsec, bwtc::wavelet::SectionOutput());
      job.t_queued = std::chrono::steady_clock::now();
      job.models_left = job.coder->modelTasks();
      job.sections_left = job.coder->sectionTasks();
      if (!e.pool) e.pool = new WorkerPool(threads);
      const auto t3 = std::chrono::steady_clock::now();
      if (job.models_left == 0 || job.sections_left == 0) {
        finish_wavelet_job(job);
      } else {
        ++e.stage_clock.unfinished;
        WorkerPool* pool = e.pool;
        if (e.wavelet_model == 'B' && bwtc::wavelet::simdModelsAvailable()) {
          // sixteen groups per thread at a time, lanes refilled across blocks (wavelet_simd.hpp,
          // BlockGroupSource).  A few engines keep up with the GPU; the range coders need the rest.
          if (!e.model_groups) {
            e.model_groups = new BlockGroupSource();
            StageClock* clock = &e.stage_clock;
            e.model_groups->on_block_modelled = [pool, clock](const std::shared_ptr<WaveletJob>& j) { submit_sections(j, pool, clock); };
          }
          BlockGroupSource* groups = e.model_groups;
          static const unsigned kEngineShare = [] { const char* v = std::getenv("BWTC_HIP_ENGINE_SHARE"); return (unsigned)(v && std::atoi(v) > 0 ? std::atoi(v) : 4); }();
          const unsigned max_engines = std::max(1u, (pool->size() + kEngineShare - 1) / kEngineShare);
          // the block's few huge groups keep their own scalar tasks: a lane would hold the block's
          // range coders back for as long as the largest of them takes at a lane's pace
          if (debug) {
            uint64_t cnt[8] = {0}, el[8] = {0};
            for (size_t k = 0; k < job.coder->modelTasks(); ++k) {
              const uint64_t n = job.coder->taskElements(k);
              const int b = n >= (32u << 20) ? 7 : n >= (16u << 20) ? 6 : n >= (8u << 20) ? 5 : n >= (4u << 20) ? 4 : n >= (1u << 20) ? 3 : n >= (1u << 16) ? 2 : n >= 256 ? 1 : 0;
              ++cnt[b]; el[b] += n;
            }
            std::fprintf(stderr, "wavelet groups by size (<256, <64Ki, <1Mi, <4Mi, <8Mi, <16Mi, <32Mi, more): ");
            for (int b = 0; b < 8; ++b) std::fprintf(stderr, "%llu/%.1fM ", (unsigned long long)cnt[b], el[b] * 1e-6);
            std::fprintf(stderr, "\n");
          }
          const uint64_t kHuge = e.huge_group_elements;
          size_t huge = 0;
          while (huge < job.coder->modelTasks() && job.coder->taskElements(huge) >= kHuge) ++huge;
          {
            StageClock* clock = &e.stage_clock;
            std::vector<std::function<void()> > own;
            for (size_t k = 0; k < huge; k += 2) {
              const size_t k2 = k + 1 < huge ? k + 1 : k;              // two at a time: a chain alone leaves the core half idle
              own.push_back([jobp, pool, k, k2, clock] {
                WaveletJob& j = *jobp;
                const auto t0 = std::chrono::steady_clock::now();
                if (k2 != k) j.coder->modelPair(k, k2, j.prob.data()); else j.coder->model(k, j.prob.data());
                clock->model_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                bool last;
                { std::lock_guard<std::mutex> g(j.mu); j.models_left -= (k2 != k ? 2 : 1); last = j.models_left == 0; }
                if (last) submit_sections(jobp, pool, clock);
              });
            }
            pool->submit(job.rank, own);
          }
          const int prefer = huge < job.coder->modelTasks() ? groups->add(jobp, huge, max_engines) : 0;
          if (prefer) {
            StageClock* clock = &e.stage_clock;
            pool->submit(0, [groups, clock, prefer] {
              const auto t0 = std::chrono::steady_clock::now();
              bwtc::wavelet::runModelLanes(*groups, prefer);
              clock->model_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            });
          }
        } else {
          for (size_t k = 0; k < job.coder->modelTasks(); ++k) {
            StageClock* clock = &e.stage_clock;
            pool->submit(job.rank, [jobp, pool, k, clock] {
              WaveletJob& j = *jobp;
              const auto t0 = std::chrono::steady_clock::now();
              j.coder->model(k, j.prob.data());
              clock->model_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
              bool last;
              { std::lock_guard<std::mutex> g(j.mu); last = --j.models_left == 0; }
              if (last) submit_sections(jobp, pool, clock);
            });
          }
        }
      }
      if (debug) {
        const auto t4 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "wavelet: %u runs, %llu steps, %llu coded, %zu groups; run scanner %.1f ms, plan %.1f ms, "
                     "device streams %.1f ms, hand-over %.1f ms, queueing %.1f ms on %u threads\n", n_runs,
                     (unsigned long long)e.wt_elements, (unsigned long long)e.wt_coded, job.plan.group_type.size(),
                     std::chrono::duration<double, std::milli>(t0 - t_begin).count(),
                     std::chrono::duration<double, std::milli>(t1 - t0).count(),
                     std::chrono::duration<double, std::milli>(t2 - t1).count(),
                     std::chrono::duration<double, std::milli>(t3 - t2).count(),
                     std::chrono::duration<double, std::milli>(t4 - t3).count(), e.pool->size());
      }
      queued = true;
    } else {
      // the runs themselves: symbols and start offsets, left in the workspace by the scanner
      std::vector<u8> run_sym(n_runs);
      std::vector<u32> run_start((size_t)n_runs + 1);
      BWTC_HIP_TRY(hipMemcpyAsync(run_sym.data(), e.d_R2, n_runs, hipMemcpyDeviceToHost, e.stream));
      BWTC_HIP_TRY(hipMemcpyAsync(run_start.data(), e.d_R1, ((size_t)n_runs + 1) * 4, hipMemcpyDeviceToHost, e.stream));
      BWTC_HIP_TRY(e.wait());
      for (u32 s = 0; s < nsec; ++s) {
        secs[s].symbols = run_sym.data() + st.first_run[s];
        secs[s].starts = run_start.data() + st.first_run[s];
      }
      bwtc::wavelet::encodeSections(secs, threads, &e.wavelet_state, &job.outs, e.wavelet_model);
    }
  }
  if (!queued) finish_wavelet_job(job);
  ++e.stage_clock.blocks;
  *ticket = e.next_ticket++;
  e.jobs[*ticket] = jobp;
  return 0;
}

int wavelet_encode_end(BwtEngine& e, u64 ticket, u64* out_bytes) {
  std::map<u64, std::shared_ptr<WaveletJob> >::iterator it = e.jobs.find(ticket);
  if (it == e.jobs.end() || !out_bytes) return -1;
  std::shared_ptr<WaveletJob> jobp = it->second;
  WaveletJob& job = *jobp;
  { std::unique_lock<std::mutex> g(job.mu); job.cv.wait(g, [&] { return job.done; }); }
  if (std::getenv("BWTC_HIP_DEBUG") && job.coder)
    std::fprintf(stderr, "wavelet block %llu: queued -> modelled %.0f ms, -> finished %.0f ms, -> collected %.0f ms\n",
                 (unsigned long long)ticket,
                 std::chrono::duration<double, std::milli>(job.t_modelled - job.t_queued).count(),
                 std::chrono::duration<double, std::milli>(job.t_finished - job.t_modelled).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - job.t_finished).count());
  e.jobs.erase(it);
  // keep the two big buffers: fresh ones would be paged in again for every block
  if (job.codes.size() && e.codes_free.size() < e.max_inflight) {
    e.codes_free.push_back(std::unique_ptr<PinnedBytes>(new PinnedBytes()));
    e.codes_free.back()->swap(job.codes);
  }
  if (job.prob.size() && e.prob_free.size() < e.max_inflight) {
    e.prob_free.push_back(std::unique_ptr<RawBuffer<uint16_t> >(new RawBuffer<uint16_t>()));
    e.prob_free.back()->swap(job.prob);
  }
  *out_bytes = job.record.size();
  return job.record.size() <= job.user_cap ? 0 : -1;
}

void wavelet_pipeline_release(BwtEngine& e) {
  for (std::map<u64, std::shared_ptr<WaveletJob> >::iterator it = e.jobs.begin(); it != e.jobs.end(); ++it) {
    WaveletJob& job = *it->second;                // uncollected blocks are finished, not abandoned half way
    std::unique_lock<std::mutex> g(job.mu);
    job.cv.wait(g, [&] { return job.done; });
  }
  e.jobs.clear();
  if (std::getenv("BWTC_HIP_DEBUG") && e.stage_clock.blocks)
    std::fprintf(stderr, "wavelet pipeline: %llu blocks; host time in models %.3f s, in range coders %.3f s (summed over threads)\n",
                 (unsigned long long)e.stage_clock.blocks.load(), e.stage_clock.model_ns.load() * 1e-9, e.stage_clock.coder_ns.load() * 1e-9);
  delete e.pool;                                  // joins the workers (engines retire when nothing is queued)
  e.pool = nullptr;
  delete e.model_groups;
  e.model_groups = nullptr;
  e.codes_free.clear();
  e.prob_free.clear();
}

}  // namespace bwtc_hip
