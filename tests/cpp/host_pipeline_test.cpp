// CPU-only test of the host side of the 'B' block pipeline (bwtc_amd/csrc/wavelet_pipeline.hpp):
// several blocks under way at once through HostPipeline -- worker pool ordered by block age, lane
// engines fed across blocks (where the CPU has AVX-512), scalar tasks for huge groups (threshold
// lowered so that they occur), paired range-coder chains, the model state carried from block to
// block -- against the oracle's sequential encoder.  The device passes are replaced by their
// host restatement (expandStreamsOnHost); no device call is made.
// TEST INFRASTRUCTURE: links the oracle (liboracle.so) as transform + reference encoder, and the
// product library for its internal C++ classes.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include "../../bwtc_amd/csrc/entropy_host.hpp"
#include "../../bwtc_amd/csrc/wavelet_gpu_models.hpp"
#include "../../bwtc_amd/csrc/wavelet_host.hpp"
#include "../../bwtc_amd/csrc/wavelet_pipeline.hpp"
#include "../../oracle/bwtc_oracle.h"

using namespace bwtc_hip;

static int failures = 0;
#define CHECK(c, ...) do { if (!(c)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

struct HostJob : WaveletJob {
  std::vector<uint8_t> codes_owner;
  std::vector<uint8_t> out;
  std::vector<uint16_t> w_owner;
};
static bool g_device_models = false;     // blocks arrive modelled (w-words made by the lane functions the GPU runs)

// what wavelet_section_stats_device delivers, computed with plain loops
struct BlockRuns {
  std::vector<uint32_t> sections, first_run, run_start, run_freqs;
  std::vector<uint8_t> run_sym;
  std::vector<std::vector<std::pair<uint32_t, uint32_t> > > dist;
};

static void scanRuns(const std::vector<uint8_t>& bwt, const uint32_t freqs[256], BlockRuns* r) {
  uint32_t sect[256];
  const uint32_t ns = orc_sections(freqs, sect);
  r->sections.assign(sect, sect + ns);
  r->first_run.assign(1, 0);
  r->run_freqs.assign((size_t)ns * 256, 0);
  r->dist.assign(ns, std::vector<std::pair<uint32_t, uint32_t> >());
  size_t beg = 0;
  for (uint32_t s = 0; s < ns; ++s) {
    std::vector<uint32_t> lens;
    for (size_t i = beg; i < beg + sect[s];) {
      size_t j = i;
      while (j < beg + sect[s] && bwt[j] == bwt[i]) ++j;
      r->run_sym.push_back(bwt[i]);
      r->run_start.push_back((uint32_t)i);
      ++r->run_freqs[(size_t)s * 256 + bwt[i]];
      lens.push_back((uint32_t)(j - i));
      i = j;
    }
    std::sort(lens.begin(), lens.end());
    for (size_t i = 0; i < lens.size();) {
      size_t j = i;
      while (j < lens.size() && lens[j] == lens[i]) ++j;
      r->dist[s].push_back(std::make_pair(lens[i], (uint32_t)(j - i)));
      i = j;
    }
    beg += sect[s];
    r->first_run.push_back((uint32_t)r->run_sym.size());
  }
  r->run_start.push_back((uint32_t)bwt.size());
}

static std::vector<uint8_t> text(std::mt19937& rng, size_t n) {
  std::vector<uint8_t> seed(4000), d;
  for (size_t i = 0; i < seed.size(); ++i) seed[i] = (uint8_t)("etaoin shrdlu<>[]&;=\n"[rng() % 21]);
  while (d.size() < n) { d.insert(d.end(), seed.begin(), seed.begin() + 1 + rng() % seed.size()); d.push_back((uint8_t)('a' + rng() % 26)); }
  d.resize(n);
  return d;
}

static void runStream(std::mt19937& rng, size_t blocks, size_t block_size, unsigned threads, uint64_t huge, unsigned sp) {
  std::vector<uint8_t> data = text(rng, blocks * block_size - block_size / 3);     // last block shorter
  HostPipeline pipe(threads, huge);
  uint32_t state = 4;
  std::vector<std::shared_ptr<HostJob> > jobs;
  std::vector<size_t> sizes;
  for (size_t off = 0; off < data.size(); off += block_size) {
    const size_t n = std::min(block_size, data.size() - off);
    std::vector<uint8_t> bwt(data.begin() + off, data.begin() + off + n);
    bwt.push_back(0);
    uint32_t lf[256], n_lf = 0, freqs[256];
    std::memset(freqs, 0, sizeof freqs);
    CHECK(orc_bwt_block(&bwt[0], (uint32_t)n, sp, lf, &n_lf, freqs) == 0, "oracle transform");
    bwt.resize(n);
    BlockRuns runs;
    scanRuns(bwt, freqs, &runs);
    const size_t nsec = runs.sections.size();
    std::vector<bwtc::wavelet::SectionRuns> secs(nsec);
    for (size_t s = 0; s < nsec; ++s) {
      secs[s].symbols = runs.run_sym.data() + runs.first_run[s];
      secs[s].starts = runs.run_start.data() + runs.first_run[s];
      secs[s].n_runs = runs.first_run[s + 1] - runs.first_run[s];
      secs[s].run_freqs = &runs.run_freqs[s * 256];
      secs[s].dist = runs.dist[s].data();
      secs[s].n_dist = runs.dist[s].size();
    }
    std::shared_ptr<HostJob> job(new HostJob());
    job->rank = jobs.size() + 1;
    job->record.assign(6, 0);                                          // WaveletEncoder::writeBlockHeader
    bwtc::writeBWTBlockHeader(lf, n_lf, job->record);
    job->record.push_back((uint8_t)(nsec == 256 ? 0 : nsec));
    for (size_t s = 0; s < nsec; ++s) bwtc::utils::packInteger(runs.sections[s], job->record);
    CHECK(bwtc::wavelet::planStreams(secs, &job->plan), "planStreams");
    CHECK(bwtc::wavelet::expandStreamsOnHost(job->plan, secs, &job->coded_pos, &job->codes_owner), "expandStreamsOnHost");
    job->codes_owner.resize(job->codes_owner.size() + 16);
    job->codes = job->codes_owner.data();
    CHECK(job->prob.reserve(job->coded_pos.back() + 8), "probability buffer");
    job->out.assign(orc_compress_bound(n), 0);
    job->user_out = job->out.data();
    job->user_cap = job->out.size();
    job->fused = std::getenv("BWTC_HIP_FUSED") != nullptr && (jobs.size() % 3) != 1;   // fused engines, with two-stage blocks in between
    if (g_device_models && (jobs.size() % 4) != 2) {                   // every fourth block stays with the host models
      const uint32_t total = job->coded_pos.back();
      std::vector<uint32_t> packed(total / 16 + 8, 0);
      std::memcpy(packed.data(), job->codes_owner.data(), std::min(job->codes_owner.size(), packed.size() * 4));
      std::vector<bwtc::wavelet::gm::Task> tasks;
      std::vector<bwtc::wavelet::gm::Chunk> chunks;
      bwtc::wavelet::gm::buildTasks(job->plan, job->coded_pos.data(), &tasks, &chunks);
      job->w_owner.assign((size_t)total + 64, 0);
      uint32_t st = state;
      CHECK(bwtc::wavelet::gm::modelsOnHostLanes(packed.data(), total, tasks, chunks, &st, job->w_owner.data()), "model lanes");
      job->w = job->w_owner.data();
      job->w_end_state = st;
      job->fused = false;
    }
    state = pipe.queue(job, state, 'B');                               // all blocks under way at once
    jobs.push_back(job);
    sizes.push_back(n);
  }
  std::vector<uint8_t> stream(1, 'B');
  for (size_t k = jobs.size(); k-- > 0;) HostPipeline::wait(*jobs[k]);  // newest first: order of waiting must not matter
  for (size_t k = 0; k < jobs.size(); ++k) {
    std::vector<uint8_t> packed;
    bwtc::utils::packInteger(sizes[k], packed);
    stream.insert(stream.end(), packed.begin(), packed.end());
    stream.push_back(1);
    stream.push_back(0);
    stream.insert(stream.end(), jobs[k]->out.begin(), jobs[k]->out.begin() + jobs[k]->record.size());
  }
  stream.push_back(0);
  std::vector<uint8_t> want(orc_compress_bound(data.size()) + blocks * 8192);
  const size_t wn = orc_compress_B(&data[0], data.size(), block_size, sp, &want[0], want.size());
  CHECK(wn == stream.size() && std::memcmp(&want[0], &stream[0], wn) == 0,
        "%zu blocks of %zu bytes on %u threads (huge >= %llu): stream differs from the oracle's (%zu vs %zu bytes)", blocks, block_size,
        threads, (unsigned long long)huge, stream.size(), wn);
}

// One stream farmed over several pipelines (= contexts, one per GPU in the farm): block
// k goes to pipeline k mod N and joins the stream with the model state the previous block's
// pipeline reported -- what BlockFarm does through bwtc_hip_wavelet_encode_queue.  The stream must
// be the sequential encoder's.
static void runFarmedStream(std::mt19937& rng, size_t blocks, size_t block_size, unsigned n_pipes, unsigned sp) {
  std::vector<uint8_t> data = text(rng, blocks * block_size - block_size / 4);
  std::vector<std::unique_ptr<HostPipeline> > pipes;
  for (unsigned i = 0; i < n_pipes; ++i) pipes.push_back(std::unique_ptr<HostPipeline>(new HostPipeline(2, 40000)));
  uint32_t state = 4;
  std::vector<std::shared_ptr<HostJob> > jobs;
  std::vector<size_t> sizes;
  for (size_t off = 0, k = 0; off < data.size(); off += block_size, ++k) {
    const size_t n = std::min(block_size, data.size() - off);
    std::vector<uint8_t> bwt(data.begin() + off, data.begin() + off + n);
    bwt.push_back(0);
    uint32_t lf[256], n_lf = 0, freqs[256];
    std::memset(freqs, 0, sizeof freqs);
    CHECK(orc_bwt_block(&bwt[0], (uint32_t)n, sp, lf, &n_lf, freqs) == 0, "oracle transform");
    bwt.resize(n);
    BlockRuns runs;
    scanRuns(bwt, freqs, &runs);
    const size_t nsec = runs.sections.size();
    std::vector<bwtc::wavelet::SectionRuns> secs(nsec);
    for (size_t s = 0; s < nsec; ++s) {
      secs[s].symbols = runs.run_sym.data() + runs.first_run[s];
      secs[s].starts = runs.run_start.data() + runs.first_run[s];
      secs[s].n_runs = runs.first_run[s + 1] - runs.first_run[s];
      secs[s].run_freqs = &runs.run_freqs[s * 256];
      secs[s].dist = runs.dist[s].data();
      secs[s].n_dist = runs.dist[s].size();
    }
    std::shared_ptr<HostJob> job(new HostJob());
    job->rank = k / n_pipes + 1;
    job->record.assign(6, 0);
    bwtc::writeBWTBlockHeader(lf, n_lf, job->record);
    job->record.push_back((uint8_t)(nsec == 256 ? 0 : nsec));
    for (size_t s = 0; s < nsec; ++s) bwtc::utils::packInteger(runs.sections[s], job->record);
    CHECK(bwtc::wavelet::planStreams(secs, &job->plan), "planStreams");
    CHECK(bwtc::wavelet::expandStreamsOnHost(job->plan, secs, &job->coded_pos, &job->codes_owner), "expandStreamsOnHost");
    job->codes_owner.resize(job->codes_owner.size() + 16);
    job->codes = job->codes_owner.data();
    CHECK(job->prob.reserve(job->coded_pos.back() + 8), "probability buffer");
    job->out.assign(orc_compress_bound(n), 0);
    job->user_out = job->out.data();
    job->user_cap = job->out.size();
    job->fused = std::getenv("BWTC_HIP_FUSED") != nullptr;
    state = pipes[k % n_pipes]->queue(job, state, 'B');               // the state travels from pipeline to pipeline
    jobs.push_back(job);
    sizes.push_back(n);
  }
  std::vector<uint8_t> stream(1, 'B');
  for (size_t k = 0; k < jobs.size(); ++k) {
    HostPipeline::wait(*jobs[k]);
    std::vector<uint8_t> packed;
    bwtc::utils::packInteger(sizes[k], packed);
    stream.insert(stream.end(), packed.begin(), packed.end());
    stream.push_back(1);
    stream.push_back(0);
    stream.insert(stream.end(), jobs[k]->out.begin(), jobs[k]->out.begin() + jobs[k]->record.size());
  }
  stream.push_back(0);
  std::vector<uint8_t> want(orc_compress_bound(data.size()) + blocks * 8192);
  const size_t wn = orc_compress_B(&data[0], data.size(), block_size, sp, &want[0], want.size());
  CHECK(wn == stream.size() && std::memcmp(&want[0], &stream[0], wn) == 0,
        "%zu blocks over %u pipelines: stream differs from the oracle's (%zu vs %zu bytes)", blocks, n_pipes, stream.size(), wn);
}

int main() {
  std::mt19937 rng(4242);
  runFarmedStream(rng, 9, 150000, 2, 8);           // one stream over two pipelines, then over three
  runFarmedStream(rng, 7, 90000, 3, 1);
  runStream(rng, 7, 300000, 4, 32u << 20, 8);      // lanes only (no group reaches the threshold)
  runStream(rng, 6, 400000, 3, 50000, 8);          // several huge groups per block: paired scalar tasks
  runStream(rng, 5, 200000, 1, 20000, 4);          // one worker thread does everything in turn
  runStream(rng, 12, 60000, 8, 1000, 1);           // many small blocks, nearly everything "huge"
  // range-coder engines: a block's longest chains to the scalar pair engines (threshold lowered so
  // that some qualify), the others to the 16-lane engines; then everything in lanes; then no lanes
  setenv("BWTC_HIP_CODER_LANES", "1", 1);
  setenv("BWTC_HIP_LONG_CHAIN_ELEMENTS", "20000", 1);
  runStream(rng, 9, 250000, 6, 32u << 20, 8);
  setenv("BWTC_HIP_LONG_CHAIN_ELEMENTS", "0", 1);
  runStream(rng, 9, 250000, 6, 32u << 20, 8);
  unsetenv("BWTC_HIP_LONG_CHAIN_ELEMENTS");
  runStream(rng, 7, 300000, 5, 32u << 20, 8);
  unsetenv("BWTC_HIP_CODER_LANES");
  // fused engines: models and range coder in one lane loop, no probability arrays
  setenv("BWTC_HIP_FUSED", "1", 1);
  runStream(rng, 9, 250000, 6, 32u << 20, 8);
  runStream(rng, 14, 70000, 3, 32u << 20, 1);
  runFarmedStream(rng, 8, 120000, 2, 8);
  unsetenv("BWTC_HIP_FUSED");
  // device-modelled blocks: only range coders on the host -- long sections as scalar tasks, the
  // others in the 16-lane engines (where the CPU has them), mixed with host-modelled blocks
  g_device_models = true;
  setenv("BWTC_HIP_LONG_CHAIN_ELEMENTS", "30000", 1);
  runStream(rng, 9, 250000, 6, 32u << 20, 8);
  setenv("BWTC_HIP_LONG_CHAIN_ELEMENTS", "1", 1);                      // every section a scalar task
  runStream(rng, 6, 200000, 4, 32u << 20, 8);
  unsetenv("BWTC_HIP_LONG_CHAIN_ELEMENTS");
  runStream(rng, 12, 120000, 5, 32u << 20, 1);                         // every section in the lanes
  setenv("BWTC_HIP_W_ENGINES", "0", 1);
  runStream(rng, 5, 200000, 3, 32u << 20, 8);                          // no lane engines: sections from a cursor
  unsetenv("BWTC_HIP_W_ENGINES");
  g_device_models = false;
  std::printf(failures ? "%d FAILURES\n" : "host pipeline: all tests passed\n", failures);
  return failures ? 1 : 0;
}
