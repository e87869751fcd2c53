// See wavelet_pipeline.hpp.
#include "wavelet_pipeline.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace bwtc_hip {

namespace {
unsigned envNumber(const char* name, unsigned fallback) {
  const char* v = std::getenv(name);
  return v && std::atoi(v) > 0 ? static_cast<unsigned>(std::atoi(v)) : fallback;
}
uint64_t since(const std::chrono::steady_clock::time_point& t0) {
  return static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
}
}  // namespace

HostPipeline::HostPipeline(unsigned threads, uint64_t huge_group_elements, unsigned depth)
    : groups_(), pool_(threads), huge_(huge_group_elements) {
  // a few lane engines keep up with the GPU; the range coders need the other threads
  const unsigned P = pool_.size();
  const char* force = std::getenv("BWTC_HIP_CODER_LANES");
  const bool lanes = bwtc::wavelet::simdCoderAvailable() && (force ? force[0] == '1' : depth >= kLaneDepth);
  // two tasks per block, each stepping two sections' chains at a time (StreamCoder::
  // codeSectionsPaired): the pair costs a quarter less host time than two chains run apart, and
  // with two tasks the block's longest chain is alone again (at its own full speed) as soon as
  // the other sections are used up.
  coder_tasks_ = envNumber("BWTC_HIP_CODER_TASKS", 2);
  // Engines are long-running tasks, so their numbers split the threads between the stages.  With
  // the lane coder a text block costs 0.5 core-seconds of models and 0.35 of range coding
  // (measured on the GPU box's EPYC 9575F): 7 + 6 of 16 threads, the rest for the scalar tasks
  // of the huge groups.  Without it the range coders are per-block tasks and the model engines
  // get a quarter of the threads, as measured best in round 1.
  max_engines_ = std::max(1u, envNumber("BWTC_HIP_MODEL_ENGINES", lanes ? (P * 7 + 15) / 16 : (P + 3) / 4));
  max_coder_engines_ = lanes ? envNumber("BWTC_HIP_CODER_ENGINES", std::max(1u, (P * 6 + 15) / 16)) : 0u;
  // Optional middle way (BWTC_HIP_LONG_CHAIN_MI=n, off by default): chains of n Mi elements and
  // more go to scalar engines that step two such chains, of different blocks, alternately, the
  // others to the lanes.  Measured no better than either pure route at any depth.
  long_chain_ = static_cast<uint64_t>(std::getenv("BWTC_HIP_LONG_CHAIN_MI") ? std::atoi(std::getenv("BWTC_HIP_LONG_CHAIN_MI")) : 0) << 20;
  if (std::getenv("BWTC_HIP_LONG_CHAIN_ELEMENTS")) long_chain_ = std::strtoull(std::getenv("BWTC_HIP_LONG_CHAIN_ELEMENTS"), nullptr, 10);   // tests: small blocks
  if (long_chain_ == 0) long_chain_ = ~static_cast<uint64_t>(0);
  max_pair_engines_ = lanes && long_chain_ != ~static_cast<uint64_t>(0) ? envNumber("BWTC_HIP_PAIR_ENGINES", std::max(1u, (P * 3 + 7) / 8)) : 0u;
  // Fused engines (models + coder in one lane loop, no probability arrays): every thread runs
  // one; for pipelines deep enough to keep sixteen chains per engine going through the 4.5 s a
  // text block's longest chain takes at a lane's pace.  BWTC_HIP_FUSED=1 / 0 overrides.
  const char* fforce = std::getenv("BWTC_HIP_FUSED");
  const bool fused = bwtc::wavelet::simdCoderAvailable() && bwtc::wavelet::simdModelsAvailable() &&
                     (fforce ? fforce[0] == '1' : depth >= kFusedDepth);
  max_fused_engines_ = fused ? envNumber("BWTC_HIP_FUSED_ENGINES", P) : 0u;
  groups_.on_block_modelled = [this](const std::shared_ptr<WaveletJob>& j) { submitSections(j); };
  chains_.on_block_coded = [this](WaveletJob& j) { finish(j); };
  fused_sections_.on_block_coded = [this](WaveletJob& j) { finish(j); };
  long_chains_.on_block_coded = [this](WaveletJob& j) { finish(j); };
  w_chains_.on_block_coded = [this](WaveletJob& j) { finish(j); };
  w_long_chains_.on_block_coded = [this](WaveletJob& j) { finish(j); };
  // Device-modelled blocks leave only the range coders to the host.  A lane of the 16-lane engine
  // advances ITS chain five times slower than the scalar loop (9 ns against 1.7 per element), so
  // the block's longest section(s) get a scalar task each (they set the block's latency; queue()
  // says which), the others share the lanes with the sections of the blocks behind them (a third of
  // the scalar loop's host time per element).
  // (engines: the lanes cost 0.25 core-seconds per text block, the scalar chain 0.47; measured with 16
  // threads and 51 ms of GPU per block: 2 engines 91 ms per block, 4 just short, 6 and 8 keep up.  Round 4,
  // 39 ms of GPU per block, the process at its CPU quota: 6 engines 51.5 ms per block, 4 engines 47.0, 3 engines
  // 58 -- an engine costs a whole thread however full its lanes are, and six were two-thirds full)
  max_w_engines_ = bwtc::wavelet::simdCoderAvailable() ? envNumber("BWTC_HIP_W_ENGINES", std::max(1u, P / 4)) : 0u;
  if (std::getenv("BWTC_HIP_W_ENGINES") && std::atoi(std::getenv("BWTC_HIP_W_ENGINES")) == 0) max_w_engines_ = 0;
  // The long sections of two blocks stepped alternately by one thread (runChainPairW): 1.25 ns per
  // element instead of 1.65-1.9 on the GPU box's EPYC on random elements (scripts/dev/rc_fma_bench.cpp).
  // (opt-in, BWTC_HIP_W_PAIR_ENGINES=n: in the pipeline, on real streams, the pairs saved nothing --
  // 0.79 core-seconds per text block either way -- and every block was under way for 1.3 s instead of 0.7)
  max_w_pair_engines_ = envNumber("BWTC_HIP_W_PAIR_ENGINES", 0);
  // BWTC_HIP_W_LANES=32: an engine steps two vectors of sixteen chains alternately.  Measured no faster per
  // core (0.87 core-seconds per text block with six engines against 0.82): the vector step is bound by
  // its instruction count (transposes, the byte events), not by its dependency chain.
  w_lanes_ = envNumber("BWTC_HIP_W_LANES", 16);
  w_long_chain_ = static_cast<uint64_t>(envNumber("BWTC_HIP_W_LONG_MI", 8)) << 20;
  if (std::getenv("BWTC_HIP_LONG_CHAIN_ELEMENTS")) w_long_chain_ = std::strtoull(std::getenv("BWTC_HIP_LONG_CHAIN_ELEMENTS"), nullptr, 10);   // tests
}

HostPipeline::~HostPipeline() {}

void HostPipeline::wait(WaveletJob& job) {
  std::unique_lock<std::mutex> g(job.mu);
  job.cv.wait(g, [&] { return job.done; });
}

bool HostPipeline::waitFor(WaveletJob& job, unsigned milliseconds) {
  std::unique_lock<std::mutex> g(job.mu);
  return job.cv.wait_for(g, std::chrono::milliseconds(milliseconds), [&] { return job.done; });
}

// Appends the finished sections to the record and closes it (finishBlock,
// WaveletCoders.cpp:159-163); the worker that finishes a block's last section runs this.
void HostPipeline::finishNow(WaveletJob& job) {
  std::vector<uint8_t>& rec = job.record;
  size_t total = rec.size();
  for (size_t s = 0; s < job.outs.size(); ++s) total += job.outs[s].bytes.size();
  rec.reserve(total);
  for (size_t s = 0; s < job.outs.size(); ++s) rec.insert(rec.end(), job.outs[s].bytes.begin(), job.outs[s].bytes.end());
  const uint64_t len = rec.size() - 6;
  for (int i = 0; i < 6; ++i) rec[i] = static_cast<uint8_t>(len >> (8 * (5 - i)));
  if (job.user_out && rec.size() <= job.user_cap) std::memcpy(job.user_out, rec.data(), rec.size());
  job.t_finished = std::chrono::steady_clock::now();
  { std::lock_guard<std::mutex> g(job.mu); job.done = true; job.cv.notify_all(); }
}

void HostPipeline::finish(WaveletJob& job) {
  --clock.unfinished;
  finishNow(job);
  if (job.t_begun.time_since_epoch().count())
    clock.latency_ns += static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(job.t_finished - job.t_begun).count());
  ++clock.finished;
}

// Every group of the block is modelled: its sections' range coders.
void HostPipeline::submitSections(const std::shared_ptr<WaveletJob>& jobp) {
  jobp->t_modelled = std::chrono::steady_clock::now();
  // (a lone block, or the first blocks of a stream, keep the scalar chains even in a deep
  // pipeline: a third of the latency, and there is nothing to share vectors with yet)
  if (max_coder_engines_ && clock.unfinished.load() >= (int)kFusedBacklog) {
    // section tasks are sorted largest first: [0, n_long) are the very long ones
    size_t n_long = 0;
    while (max_pair_engines_ && n_long < jobp->coder->sectionTasks() && jobp->coder->sectionElements(n_long) >= long_chain_) ++n_long;
    if (long_chains_.add(jobp, 0, n_long, max_pair_engines_)) {
      pool_.submit(0, [this] { bwtc::wavelet::runCoderLanes(long_chains_, static_cast<int>(envNumber("BWTC_HIP_LONG_LANES", 2)), &clock.coder_ns); });
    }
    if (chains_.add(jobp, n_long, jobp->coder->sectionTasks(), max_coder_engines_)) {
      pool_.submit(0, [this] { bwtc::wavelet::runCoderLanes(chains_, 16, &clock.coder_ns); });
    }
    return;
  }
  std::vector<std::function<void()> > next;
  // ... and more of them when no section dominates (256 equal sections of a block of random
  // bytes): as many as the block's elements are multiples of its longest chain
  const uint64_t longest = std::max<uint64_t>(1, jobp->coder->largestSectionElements());
  const size_t balanced = static_cast<size_t>(std::min<uint64_t>(pool_.size(), jobp->coder->elements() / longest));
  const size_t engines = std::max<size_t>(1, std::min<size_t>(std::max(coder_tasks_, balanced), (jobp->coder->sectionTasks() + 1) / 2));
  for (size_t q = 0; q < engines; ++q) {
    next.push_back([this, jobp] {
      WaveletJob& j = *jobp;
      const auto t0 = std::chrono::steady_clock::now();
      const size_t did = j.coder->codeSectionsPaired(&j.section_cursor, j.prob.data(), &j.outs);
      clock.coder_ns += since(t0);
      bool fin;
      { std::lock_guard<std::mutex> g(j.mu); j.sections_left -= did; fin = did > 0 && j.sections_left == 0; }
      if (fin) finish(j);
    });
  }
  pool_.submit(jobp->rank, next);
}

uint32_t HostPipeline::queue(const std::shared_ptr<WaveletJob>& jobp, uint32_t fsm8_state, char model, const uint32_t* expect_end) {
  WaveletJob& job = *jobp;
  noteBegin();
  if (job.w) {
    // The models ran on the device: what is left are the sections' range-coder chains, each on its
    // own (a chain alone advances at its full speed, and the block's longest one is its latency),
    // as many tasks as the block's elements are multiples of its longest chain, at least two.
    job.coder.reset(new bwtc::wavelet::StreamCoder(job.plan, job.coded_pos.data(), job.w_end_state, bwtc::wavelet::StreamCoder::Modelled()));
    freshOutputs(job, job.plan.sections.size());
    job.t_queued = job.t_modelled = std::chrono::steady_clock::now();
    job.models_left = 0;
    job.sections_left = job.coder->sectionTasks();
    ++clock.blocks;
    if (job.sections_left == 0) { finishNow(job); ++clock.finished; return job.w_end_state; }
    ++clock.unfinished;
    if (std::getenv("BWTC_HIP_DEBUG")) {
      std::fprintf(stderr, "w-route: %zu sections, elements (Mi):", job.coder->sectionTasks());
      for (size_t k = 0; k < job.coder->sectionTasks() && k < 12; ++k) std::fprintf(stderr, " %.1f", job.coder->sectionElements(k) / 1048576.0);
      std::fprintf(stderr, " ...; %u lane engines at most, scalar from %.0f Mi\n", max_w_engines_, w_long_chain_ / 1048576.0);
    }
    if (max_w_engines_) {
      // scalar tasks: sections that a lane (five times slower per chain) would hold for longer than
      // the block's longest section needs in the scalar loop -- those set the block's latency --
      // and never sections below the floor; section tasks are sorted largest first
      const uint64_t cut = std::max<uint64_t>(w_long_chain_, job.coder->largestSectionElements() / 5);
      size_t n_long = 0;
      while (n_long < job.coder->sectionTasks() && job.coder->sectionElements(n_long) >= cut) ++n_long;
      std::vector<std::function<void()> > own;
      if (max_w_pair_engines_ && n_long) {
        if (w_long_chains_.add(jobp, 0, n_long, max_w_pair_engines_))
          pool_.submit(0, [this] { bwtc::wavelet::runCoderLanes(w_long_chains_, 2, &clock.coder_ns); });
      } else
      for (size_t k = 0; k < n_long; ++k) {
        own.push_back([this, jobp, k] {
          WaveletJob& j = *jobp;
          const auto t0 = std::chrono::steady_clock::now();
          j.coder->codeSectionW(k, j.w, &j.outs);
          const uint64_t dt = since(t0);
          clock.coder_ns += dt;
          clock.scalar_ns += dt;
          bool fin;
          { std::lock_guard<std::mutex> g(j.mu); fin = --j.sections_left == 0; }
          if (fin) finish(j);
        });
      }
      if (!own.empty()) pool_.submit(job.rank, own);
      if (w_chains_.add(jobp, n_long, job.coder->sectionTasks(), max_w_engines_))
        pool_.submit(0, [this] { bwtc::wavelet::runCoderLanes(w_chains_, static_cast<int>(w_lanes_), &clock.coder_ns); });
      return job.w_end_state;
    }
    const uint64_t longest = std::max<uint64_t>(1, job.coder->largestSectionElements());
    const size_t balanced = static_cast<size_t>(std::min<uint64_t>(pool_.size(), (job.coder->elements() + longest - 1) / longest));
    const size_t engines = std::max<size_t>(1, std::min<size_t>(std::max<size_t>(2, balanced), job.coder->sectionTasks()));
    std::vector<std::function<void()> > next;
    for (size_t q = 0; q < engines; ++q) {
      next.push_back([this, jobp] {
        WaveletJob& j = *jobp;
        const auto t0 = std::chrono::steady_clock::now();
        const size_t did = j.coder->codeSectionsW(&j.section_cursor, j.w, &j.outs);
        clock.coder_ns += since(t0);
        bool fin;
        { std::lock_guard<std::mutex> g(j.mu); j.sections_left -= did; fin = did > 0 && j.sections_left == 0; }
        if (fin) finish(j);
      });
    }
    pool_.submit(job.rank, next);
    return job.w_end_state;
  }
  job.coder.reset(new bwtc::wavelet::StreamCoder(job.plan, job.coded_pos.data(), job.codes, fsm8_state, model));
  const uint32_t next_state = job.coder->endState();
  if (expect_end && *expect_end != next_state) job.failed = true;
  freshOutputs(job, job.plan.sections.size());
  job.t_queued = std::chrono::steady_clock::now();
  job.models_left = job.coder->modelTasks();
  job.sections_left = job.coder->sectionTasks();
  ++clock.blocks;
  if (job.models_left == 0 || job.sections_left == 0) { finishNow(job); ++clock.finished; return next_state; }
  ++clock.unfinished;
  if (model == 'B' && max_fused_engines_ && job.fused) {
    job.models_left = 0;
    job.fused_groups.assign(job.coder->sectionTasks(), std::vector<bwtc::wavelet::FusedGroup>());
    job.t_modelled = job.t_queued;
    if (fused_sections_.add(jobp, max_fused_engines_))
      pool_.submit(0, [this] { bwtc::wavelet::runFusedLanes(fused_sections_, &clock.coder_ns); });
    return next_state;
  }
  if (model == 'B' && bwtc::wavelet::simdModelsAvailable()) {
    // Sixteen groups per thread at a time, lanes refilled across blocks (wavelet_simd.hpp,
    // BlockGroupSource).  The block's few huge groups keep scalar tasks of their own, two groups
    // stepped alternately per task: a lane would hold the block's range coders back for as long
    // as the largest of them takes at a lane's pace.
    size_t huge = 0;
    while (huge < job.coder->modelTasks() && job.coder->taskElements(huge) >= huge_) ++huge;
    std::vector<std::function<void()> > own;
    for (size_t k = 0; k < huge; k += 2) {
      const size_t k2 = k + 1 < huge ? k + 1 : k;
      own.push_back([this, jobp, k, k2] {
        WaveletJob& j = *jobp;
        const auto t0 = std::chrono::steady_clock::now();
        if (k2 != k) j.coder->modelPair(k, k2, j.prob.data()); else j.coder->model(k, j.prob.data());
        clock.model_ns += since(t0);
        bool last;
        { std::lock_guard<std::mutex> g(j.mu); j.models_left -= (k2 != k ? 2 : 1); last = j.models_left == 0; }
        if (last) submitSections(jobp);
      });
    }
    pool_.submit(job.rank, own);
    const int prefer = huge < job.coder->modelTasks() ? groups_.add(jobp, huge, max_engines_) : 0;
    if (prefer) {
      pool_.submit(0, [this, prefer] {
        const auto t0 = std::chrono::steady_clock::now();
        bwtc::wavelet::runModelLanes(groups_, prefer);
        clock.model_ns += since(t0);
      });
    }
  } else {
    for (size_t k = 0; k < job.coder->modelTasks(); ++k) {
      pool_.submit(job.rank, [this, jobp, k] {
        WaveletJob& j = *jobp;
        const auto t0 = std::chrono::steady_clock::now();
        j.coder->model(k, j.prob.data());
        clock.model_ns += since(t0);
        bool last;
        { std::lock_guard<std::mutex> g(j.mu); last = --j.models_left == 0; }
        if (last) submitSections(jobp);
      });
    }
  }
  return next_state;
}

}  // namespace bwtc_hip
