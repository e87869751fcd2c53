// extern "C" surface of libbwtc_hip.so (include/bwtc_hip.h).
#include "bwt_engine.hpp"
#include "gpu_lanes.hpp"
#include "prepr_host.hpp"
#include <new>
#include "radix_sort.hpp"
#include "bwtc_hip.h"
#include "entropy_host.hpp"
#include "wavelet_host.hpp"
#include "wavelet_gpu_models.hpp"
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <string>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

using namespace bwtc_hip;

struct bwtc_hip_ctx {
  BwtEngine eng;
};
struct bwtc_hip_grammar {
  bwtc::prepr::Grammar g;
};
namespace bwtc_hip {
int pair_replace_device(BwtEngine& e, bwtc::prepr::Grammar& grammar, const u8* d_src, u64 n, u8* d_dst, u64* n_out, u32* replaced);
}

template <typename K>
static int test_sort(bwtc_hip_ctx* ctx, K* keys, uint32_t* vals, uint64_t n, int nbits) {
  if (!ctx || !keys || !vals) return -1;
  BwtEngine& e = ctx->eng;
  if (n > e.cap) return -1;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  K* k0 = static_cast<K*>(e.d_R1);
  K* k1 = static_cast<K*>(e.d_R2);
  BWTC_HIP_TRY(hipMemcpyAsync(k0, keys, n * sizeof(K), hipMemcpyHostToDevice, e.stream));
  BWTC_HIP_TRY(hipMemcpyAsync(e.d_V0, vals, n * 4, hipMemcpyHostToDevice, e.stream));
  K* ks; u32* vs;
  BWTC_HIP_TRY(hipMemsetAsync(e.d_small, 0, 1024 * 4, e.stream));
  e.sort_pairs<K>(k0, k1, e.d_V0, e.d_V1, n, nbits, &ks, &vs, false);
  BWTC_HIP_TRY(hipMemcpyAsync(keys, ks, n * sizeof(K), hipMemcpyDeviceToHost, e.stream));
  BWTC_HIP_TRY(hipMemcpyAsync(vals, vs, n * 4, hipMemcpyDeviceToHost, e.stream));
  BWTC_HIP_TRY(hipMemcpyAsync(e.h_small + 522, e.d_small + 522, 4, hipMemcpyDeviceToHost, e.stream));
  BWTC_HIP_TRY(e.wait());
  if (e.h_small[522]) return -3;
  BWTC_HIP_TRY(hipGetLastError());
  return 0;
}

// how every buffer handed out by bwtc_hip_host_alloc was locked (see lockedHostAlloc)
static std::mutex& host_allocs_mu() { static std::mutex m; return m; }
static std::map<void*, std::pair<int, uint64_t> >& host_allocs() { static std::map<void*, std::pair<int, uint64_t> > m; return m; }

// bwtc_hip_copy_probe: a streaming copy, one 16-byte item per thread.  (Of the forms tried on MI355X -- grid-stride
// loops with 4 or 8 loads in flight, contiguous chunks per workgroup, non-temporal accesses, hipMemcpy device to
// device: 4.1-5.5 TB/s -- this plain one is the fastest, 6.2 TB/s: scripts/dev/copy_probe.cpp.)
static __global__ __launch_bounds__(256) void k_copy_probe(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

extern "C" {

int bwtc_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* bwtc_hip_version(void) { return "bwtc_hip 0.1 (gfx950)"; }

uint64_t bwtc_hip_workspace_bytes(uint32_t max_block_size) {
  return BwtEngine::workspace_bytes(max_block_size);
}

int bwtc_hip_create(int device, uint32_t max_block_size, bwtc_hip_ctx** ctx_out) {
  if (!ctx_out) return -1;
  *ctx_out = nullptr;
  if (max_block_size >= 0x7FFFFFFEu) return -1;   // PrecompressorBlock.cpp:126
  bwtc_hip_ctx* c = new (std::nothrow) bwtc_hip_ctx();
  if (!c) return -2;
  for (int i = 0; i < BwtEngine::kMaxSortEvents; ++i) c->eng.ev_sort[i] = nullptr;
  int rc = c->eng.init(device, max_block_size);
  if (rc != 0) { c->eng.release(); delete c; return rc; }
  *ctx_out = c;
  return 0;
}

void bwtc_hip_destroy(bwtc_hip_ctx* ctx) {
  if (!ctx) return;
  ctx->eng.release();
  delete ctx;
}

void* bwtc_hip_stream(bwtc_hip_ctx* ctx) { return ctx ? (void*)ctx->eng.stream : nullptr; }

int bwtc_hip_get_stats(bwtc_hip_ctx* ctx, bwtc_hip_stats* out) {
  if (!ctx || !out) return -1;
  *out = ctx->eng.stats;
  return 0;
}

int bwtc_hip_set_profiling(bwtc_hip_ctx* ctx, int on) {
  if (!ctx) return -1;
  ctx->eng.probe.enabled = on != 0;
  return 0;
}

int bwtc_hip_get_kernel_timers(bwtc_hip_ctx* ctx, bwtc_hip_kernel_timers* out, int reset) {
  if (!ctx) return -1;
  ScatterProbe& p = ctx->eng.probe;
  if (out) { out->scatter_launches = p.total_launches; out->scatter_bytes = p.total_bytes; out->scatter_ms = p.total_ms; }
  if (reset) { p.total_launches = 0; p.total_bytes = 0; p.total_ms = 0.0; }
  return 0;
}

int bwtc_hip_copy_probe(bwtc_hip_ctx* ctx, uint64_t bytes, int reps, double* gbps) {
  if (!ctx || !gbps || bytes < 16 || reps < 1) return -1;
  BWTC_HIP_TRY(hipSetDevice(ctx->eng.device));
  void *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, bytes) != hipSuccess) return -2;
  if (hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); return -2; }
  hipStream_t st = ctx->eng.stream;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = 0;
  float ms = 0.f;
  const uint64_t n16 = bytes / 16;
  if (hipMemsetAsync(a, 1, bytes, st) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = -3;
  if (!rc) {
    const dim3 grid((unsigned)((n16 + 255) / 256));
    hipLaunchKernelGGL(k_copy_probe, grid, dim3(256), 0, st, (const uint4*)a, (uint4*)b, n16);      // warm
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy_probe, grid, dim3(256), 0, st, (const uint4*)a, (uint4*)b, n16);
    (void)hipEventRecord(e1, st);
    if (hipStreamSynchronize(st) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) rc = -3;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  if (rc) { (void)hipGetLastError(); return rc; }
  *gbps = (double)reps * 2.0 * (double)(n16 * 16) / ((double)ms * 1e-3) / 1e9;
  return 0;
}

void* bwtc_hip_malloc(bwtc_hip_ctx* ctx, uint64_t bytes) {
  if (!ctx || hipSetDevice(ctx->eng.device) != hipSuccess) return nullptr;
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
  return p;
}
void bwtc_hip_free(bwtc_hip_ctx* ctx, void* d_ptr) {
  if (ctx && d_ptr && hipSetDevice(ctx->eng.device) == hipSuccess) (void)hipFree(d_ptr);
}
int bwtc_hip_memcpy_to_device(bwtc_hip_ctx* ctx, void* d_dst, const void* src, uint64_t bytes) {
  if (!ctx || (!d_dst && bytes) || (!src && bytes)) return -1;
  BWTC_HIP_TRY(hipSetDevice(ctx->eng.device));
  if (bytes) BWTC_HIP_TRY(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}
int bwtc_hip_memcpy_to_host(bwtc_hip_ctx* ctx, void* dst, const void* d_src, uint64_t bytes) {
  if (!ctx || (!dst && bytes) || (!d_src && bytes)) return -1;
  BWTC_HIP_TRY(hipSetDevice(ctx->eng.device));
  if (bytes) BWTC_HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
  return 0;
}

void* bwtc_hip_host_alloc(bwtc_hip_ctx* ctx, uint64_t bytes) {
  if (!ctx || hipSetDevice(ctx->eng.device) != hipSuccess) return nullptr;
  int kind = 0;
  void* p = bwtc_hip::lockedHostAlloc(bytes ? bytes : 1, &kind);
  if (!p) return nullptr;
  std::lock_guard<std::mutex> g(host_allocs_mu());
  host_allocs()[p] = std::make_pair(kind, bytes ? bytes : 1);
  return p;
}
void bwtc_hip_host_free(bwtc_hip_ctx* ctx, void* p) {
  if (!ctx || !p || hipSetDevice(ctx->eng.device) != hipSuccess) return;
  std::pair<int, uint64_t> how;
  {
    std::lock_guard<std::mutex> g(host_allocs_mu());
    std::map<void*, std::pair<int, uint64_t> >::iterator it = host_allocs().find(p);
    if (it == host_allocs().end()) return;          // not one of ours
    how = it->second;
    host_allocs().erase(it);
  }
  bwtc_hip::lockedHostFree(p, how.first, static_cast<size_t>(how.second));
}
int bwtc_hip_memcpy_to_device_async(bwtc_hip_ctx* ctx, void* d_dst, const void* src, uint64_t bytes) {
  if (!ctx || (!d_dst && bytes) || (!src && bytes)) return -1;
  BwtEngine& e = ctx->eng;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  if (!e.copy_stream) {
    BWTC_HIP_TRY(hipStreamCreateWithFlags(&e.copy_stream, hipStreamNonBlocking));
    BWTC_HIP_TRY(hipEventCreateWithFlags(&e.ev_copy, hipEventBlockingSync | hipEventDisableTiming));
  }
  if (bytes) BWTC_HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, e.copy_stream));
  return 0;
}
int bwtc_hip_copy_wait(bwtc_hip_ctx* ctx) {
  if (!ctx) return -1;
  BwtEngine& e = ctx->eng;
  if (!e.copy_stream) return 0;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  BWTC_HIP_TRY(hipEventRecord(e.ev_copy, e.copy_stream));
  if (e.pipeline) BWTC_HIP_TRY(e.sleepy_wait(e.ev_copy, BwtEngine::kWaitSlots - 2));     // the 'B' workers need the CPU
  else BWTC_HIP_TRY(hipEventSynchronize(e.ev_copy));
  return 0;
}

uint32_t bwtc_hip_n_lf(uint32_t size, uint32_t starting_points) {
  if (starting_points < 1) starting_points = 1;          // BWTManager.cpp:60-64
  else if (starting_points > 256) starting_points = 256;
  if (size <= 256) return 1;                             // BWTBlock.cpp:104-108
  return starting_points;
}

static int stage_in(BwtEngine& e, const uint8_t* host, uint32_t bytes) {
  if (!bytes) return 0;
  std::memcpy(e.h_stage, host, bytes);
  BWTC_HIP_TRY(hipMemcpyAsync(e.d_in, e.h_stage, bytes, hipMemcpyHostToDevice, e.stream));
  return 0;
}
static int stage_out(BwtEngine& e, uint8_t* host, uint32_t bytes) {
  if (!bytes) return 0;
  BWTC_HIP_TRY(hipMemcpyAsync(e.h_stage, e.d_in, bytes, hipMemcpyDeviceToHost, e.stream));
  BWTC_HIP_TRY(e.wait());
  std::memcpy(host, e.h_stage, bytes);
  return 0;
}

int bwtc_hip_bwt(bwtc_hip_ctx* ctx, uint8_t* T, uint32_t length, uint32_t* lf, uint32_t n_lf,
                 uint32_t* freqs) {
  if (!ctx || !T || !lf || n_lf == 0) return -1;
  if (length <= 1) return 0;                             // divsufsort.c:489
  BwtEngine& e = ctx->eng;
  if ((u64)length > e.cap) return -1;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  int rc = stage_in(e, T, length);
  if (rc) return rc;
  rc = e.transform(e.d_in, e.d_in, length, true, lf, n_lf, freqs);
  if (rc) return rc;
  return stage_out(e, T, length);
}

int bwtc_hip_bwt_block(bwtc_hip_ctx* ctx, uint8_t* block, uint32_t size, uint32_t* lf,
                       uint32_t n_lf, uint32_t* freqs) {
  if (!ctx || (!block && size) || !lf || n_lf == 0) return -1;
  BwtEngine& e = ctx->eng;
  if (size > e.max_block) return -1;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  int rc = stage_in(e, block, size);
  if (rc) return rc;
  rc = e.transform(e.d_in, e.d_in, size, false, lf, n_lf, freqs);
  if (rc) return rc;
  return stage_out(e, block, size);
}

int bwtc_hip_bwt_block_device(bwtc_hip_ctx* ctx, const uint8_t* d_in, uint8_t* d_out,
                              uint32_t size, uint32_t* lf, uint32_t n_lf, uint32_t* freqs) {
  if (!ctx || ((!d_in || !d_out) && size) || !lf || n_lf == 0) return -1;
  BwtEngine& e = ctx->eng;
  if (size > e.max_block) return -1;
  return e.transform(d_in, d_out, size, false, lf, n_lf, freqs);
}

int bwtc_hip_inverse_bwt_block(bwtc_hip_ctx* ctx, uint8_t* block, uint32_t size,
                               const uint32_t* lf, uint32_t n_lf) {
  if (!ctx || (!block && size) || !lf || n_lf == 0) return -1;
  BwtEngine& e = ctx->eng;
  if (size > e.max_block) return -1;
  if (size == 0) return 0;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  int rc = stage_in(e, block, size);
  if (rc) return rc;
  rc = inverse_bwt_device(e, e.d_in, e.d_in, size, lf, n_lf);
  if (rc) return rc;
  return stage_out(e, block, size);
}

int bwtc_hip_inverse_bwt_block_device(bwtc_hip_ctx* ctx, const uint8_t* d_bwt, uint8_t* d_out,
                                      uint32_t size, const uint32_t* lf, uint32_t n_lf) {
  if (!ctx || ((!d_bwt || !d_out) && size) || !lf || n_lf == 0) return -1;
  if (size > ctx->eng.max_block) return -1;
  return inverse_bwt_device(ctx->eng, d_bwt, d_out, size, lf, n_lf);
}

uint64_t bwtc_hip_compress_bound(uint32_t size) { return huffman_compress_bound(size); }

int bwtc_hip_huffman_encode_device(bwtc_hip_ctx* ctx, const uint8_t* d_bwt, uint32_t size,
                                   const uint32_t* lf, uint32_t n_lf, const uint32_t* freqs,
                                   uint8_t* d_out, uint64_t out_cap, uint64_t* out_bytes) {
  if (!ctx || (!d_bwt && size) || !d_out || !lf || !freqs || !out_bytes) return -1;
  if (size > ctx->eng.max_block) return -1;
  return huffman_encode_device(ctx->eng, d_bwt, size, lf, n_lf, freqs, d_out, out_cap, out_bytes);
}

int bwtc_hip_huffman_encode(bwtc_hip_ctx* ctx, const uint8_t* bwt, uint32_t size,
                            const uint32_t* lf, uint32_t n_lf, const uint32_t* freqs,
                            uint8_t* out, uint64_t out_cap, uint64_t* out_bytes) {
  if (!ctx || (!bwt && size) || !out || !lf || !freqs || !out_bytes) return -1;
  BwtEngine& e = ctx->eng;
  if (size > e.max_block) return -1;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  int rc = stage_in(e, bwt, size);
  if (rc) return rc;
  u64 n = 0;
  rc = huffman_encode_device(e, e.d_in, size, lf, n_lf, freqs, e.d_comp,
                             huffman_compress_bound(e.cap), &n);
  if (rc) return rc;
  if (n > out_cap) return -1;
  BWTC_HIP_TRY(hipMemcpyAsync(out, e.d_comp, n, hipMemcpyDeviceToHost, e.stream));
  BWTC_HIP_TRY(e.wait());
  *out_bytes = n;
  return 0;
}

int bwtc_hip_transform_and_encode(bwtc_hip_ctx* ctx, uint8_t* block, uint32_t size,
                                  uint32_t starting_points, uint8_t* out, uint64_t out_cap,
                                  uint64_t* out_bytes) {
  if (!ctx || (!block && size) || !out || !out_bytes) return -1;
  BwtEngine& e = ctx->eng;
  if (size > e.max_block) return -1;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  uint32_t lf[256];
  uint32_t freqs[256];
  std::memset(freqs, 0, sizeof freqs);                    // HuffmanCoders.cpp:53
  const uint32_t n_lf = bwtc_hip_n_lf(size, starting_points);
  int rc = stage_in(e, block, size);
  if (rc) return rc;
  rc = e.transform(e.d_in, e.d_in, size, false, lf, n_lf, freqs);
  if (rc) return rc;
  u64 n = 0;
  rc = huffman_encode_device(e, e.d_in, size, lf, n_lf, freqs, e.d_comp,
                             huffman_compress_bound(e.cap), &n);
  if (rc) return rc;
  if (n > out_cap) return -1;
  BWTC_HIP_TRY(hipMemcpyAsync(out, e.d_comp, n, hipMemcpyDeviceToHost, e.stream));
  rc = stage_out(e, block, size);
  if (rc) return rc;
  *out_bytes = n;
  return 0;
}

int bwtc_hip_wavelet_section_stats(bwtc_hip_ctx* ctx, const uint8_t* bwt, uint32_t size,
                                   const uint32_t* freqs, uint32_t* n_sections,
                                   uint32_t* section_len, uint32_t* run_freqs,
                                   uint64_t* total_runs, uint32_t* dist_offset,
                                   uint32_t* dist_len, uint32_t* dist_cnt, uint32_t dist_cap) {
  if (!ctx || (!bwt && size) || !freqs || !n_sections || !section_len || !run_freqs ||
      !total_runs || !dist_offset || !dist_len || !dist_cnt) return -1;
  BwtEngine& e = ctx->eng;
  if (size > e.max_block) return -1;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  int rc = stage_in(e, bwt, size);
  if (rc) return rc;
  WaveletSectionStats st;
  rc = wavelet_section_stats_device(e, e.d_in, size, freqs, &st);
  if (rc) return rc;
  const uint32_t nsec = (uint32_t)st.sections.size();
  *n_sections = nsec;
  uint32_t at = 0;
  for (uint32_t s = 0; s < nsec; ++s) {
    section_len[s] = st.sections[s];
    total_runs[s] = st.total_runs[s];
    std::memcpy(run_freqs + (size_t)s * 256, &st.run_freqs[(size_t)s * 256], 1024);
    dist_offset[s] = at;
    if ((uint64_t)at + st.dist[s].size() > dist_cap) return -1;
    for (size_t i = 0; i < st.dist[s].size(); ++i) { dist_len[at] = st.dist[s][i].first; dist_cnt[at] = st.dist[s][i].second; ++at; }
  }
  dist_offset[nsec] = at;
  return 0;
}

// CPUs this process may keep busy: the hardware threads, cut down to the cgroup's CPU quota
// (a container with a quota is throttled as a whole, the GPU-feeding thread included, when
// more threads than that run).
static unsigned usable_cpus() {
  unsigned n = std::thread::hardware_concurrency();
  if (n == 0) n = 1;
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32] = {0};
    unsigned long period = 0;
    if (std::fscanf(f, "%31s %lu", quota, &period) == 2 && period > 0 && std::strcmp(quota, "max") != 0) {
      const unsigned long q = std::strtoul(quota, nullptr, 10);
      if (q > 0) n = std::min<unsigned long>(n, std::max<unsigned long>(1, q / period));
    }
    std::fclose(f);
  }
  return n;
}

uint32_t bwtc_hip_host_usable_cpus(void) { return usable_cpus(); }

static unsigned pick_threads(uint32_t threads) {
  if (threads) return threads;
  // all of them: the thread that feeds the GPU sleeps while it waits (BwtEngine::wait), and what
  // it computes between waits is a tenth of a CPU
  return std::min(usable_cpus(), 64u);
}

int bwtc_hip_wavelet_start(bwtc_hip_ctx* ctx, char coder) {
  if (!ctx || !bwtc::wavelet::isWaveletModel(coder)) return -1;
  // blocks begun with _begin that have not joined their stream yet belong to the OLD stream:
  // they join it now, with its state and its model, before either changes
  const int rc = wavelet_join_begun(ctx->eng);
  // whatever became of the old stream's blocks (a failed join is reported, and their _end reports it again), the
  // NEW stream starts from the fresh state: never the old stream's carried state under a new header
  ctx->eng.wavelet_state = 4;
  ctx->eng.wavelet_model = coder;
  ctx->eng.stream_start_error = 0;                    // the caller has the code: the new stream starts clean
  return rc;
}
void bwtc_hip_wavelet_reset(bwtc_hip_ctx* ctx) {      // no result of its own: a failure is kept, the next _begin / _prepare returns it
  const int rc = bwtc_hip_wavelet_start(ctx, 'B');
  if (ctx) ctx->eng.stream_start_error = rc;
}
uint32_t bwtc_hip_wavelet_depth(bwtc_hip_ctx* ctx) { return ctx ? ctx->eng.max_inflight : 0u; }
int bwtc_hip_wavelet_set_depth(bwtc_hip_ctx* ctx, uint32_t depth) {
  if (!ctx || depth == 0) return -1;
  BwtEngine& e = ctx->eng;
  if (e.jobs.size() > depth) return -6;                     // more blocks under way than the new depth allows: collect first
  e.max_inflight = depth;
  // page-locked staging follows the need: what the pools hold beyond the blocks that can still be begun goes back
  const size_t spare = depth + 2 > e.jobs.size() ? depth + 2 - e.jobs.size() : 0;
  e.codes_pool.trim(spare);
  e.w_pool.trim(spare);
  return 0;
}
uint32_t bwtc_hip_wavelet_depth_needed(bwtc_hip_ctx* ctx) { return ctx && ctx->eng.pipeline ? ctx->eng.pipeline->depthNeeded() : 0u; }
int bwtc_hip_wavelet_host_clock(bwtc_hip_ctx* ctx, double* model_seconds, double* coder_seconds, uint64_t* blocks) {
  if (!ctx) return -1;
  const HostPipeline* p = ctx->eng.pipeline;
  if (model_seconds) *model_seconds = p ? p->clock.model_ns.load() * 1e-9 : 0.0;
  if (coder_seconds) *coder_seconds = p ? p->clock.coder_ns.load() * 1e-9 : 0.0;
  if (blocks) *blocks = p ? p->clock.blocks.load() : 0;
  return 0;
}

int bwtc_hip_numa_node(bwtc_hip_ctx* ctx) {
  if (!ctx) return -1;
  char id[64] = {0};
  if (hipDeviceGetPCIBusId(id, sizeof id, ctx->eng.device) != hipSuccess) { (void)hipGetLastError(); return -1; }
  for (char* p = id; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');     // sysfs spells the id in lower case
  const std::string path = std::string("/sys/bus/pci/devices/") + id + "/numa_node";
  int node = -1;
  if (FILE* f = std::fopen(path.c_str(), "r")) { if (std::fscanf(f, "%d", &node) != 1) node = -1; std::fclose(f); }
  return node;
}

int bwtc_hip_host_cpu_slice(int numa_node, uint32_t rank, uint32_t ranks, uint32_t* cpus, uint32_t cap) {
  if (!cpus || ranks == 0 || rank >= ranks) return -1;
  cpu_set_t mine;
  CPU_ZERO(&mine);
  if (sched_getaffinity(0, sizeof mine, &mine) != 0) return -1;
  std::vector<uint32_t> usable;
  for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &mine)) usable.push_back((uint32_t)c);
  if (numa_node >= 0) {
    // "0-15,128-143" -> the node's CPUs; kept only if some of them are ours
    char path[96];
    std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", numa_node);
    // BWTC_HIP_NODE<k>_CPULIST=<file>: a made-up topology (tests of the farm's slicing on boxes with one node)
    char var[48];
    std::snprintf(var, sizeof var, "BWTC_HIP_NODE%d_CPULIST", numa_node);
    const char* fake = std::getenv(var);
    std::vector<uint32_t> on_node;
    if (FILE* f = std::fopen(fake ? fake : path, "r")) {
      unsigned a = 0, b = 0;
      for (;;) {
        if (std::fscanf(f, "%u", &a) != 1) break;
        b = a;
        int ch = std::fgetc(f);
        if (ch == '-') { if (std::fscanf(f, "%u", &b) != 1) break; ch = std::fgetc(f); }
        for (unsigned c = a; c <= b && c < (unsigned)CPU_SETSIZE; ++c) if (CPU_ISSET(c, &mine)) on_node.push_back(c);
        if (ch != ',') break;
      }
      std::fclose(f);
    }
    if (!on_node.empty()) usable.swap(on_node);
  }
  const size_t n = usable.size(), b = n * rank / ranks, e = n * (rank + 1) / ranks;
  uint32_t k = 0;
  for (size_t i = b; i < e && k < cap; ++i) cpus[k++] = usable[i];
  return (int)k;
}

int bwtc_hip_set_worker_cpus(bwtc_hip_ctx* ctx, const uint32_t* cpus, uint32_t n) {
  if (!ctx || (n && !cpus)) return -1;
  ctx->eng.worker_cpus.assign(cpus, cpus + n);
  if (ctx->eng.pipeline) ctx->eng.pipeline->setWorkerCpus(ctx->eng.worker_cpus);
  return 0;
}

int bwtc_hip_wavelet_host_progress(bwtc_hip_ctx* ctx, uint64_t* queued, uint64_t* finished) {
  if (!ctx) return -1;
  const HostPipeline* p = ctx->eng.pipeline;
  if (queued) *queued = p ? p->clock.blocks.load() : 0;
  if (finished) *finished = p ? p->clock.finished.load() : 0;
  return 0;
}

int bwtc_hip_wavelet_latency(bwtc_hip_ctx* ctx, double* mean_seconds) {
  if (!ctx || !mean_seconds) return -1;
  const HostPipeline* p = ctx->eng.pipeline;
  const uint64_t n = p ? p->clock.finished.load() : 0;
  *mean_seconds = n ? p->clock.latency_ns.load() * 1e-9 / (double)n : 0.0;
  return 0;
}

int bwtc_hip_host_staging_bytes(uint64_t* now, uint64_t* peak) {
  if (now) *now = bwtc_hip::PinnedGauge::get().now.load();
  if (peak) *peak = bwtc_hip::PinnedGauge::get().peak.load();
  return 0;
}

int bwtc_hip_wavelet_encode_device_begin(bwtc_hip_ctx* ctx, const uint8_t* d_bwt, uint32_t size,
                                         const uint32_t* lf, uint32_t n_lf, const uint32_t* freqs,
                                         uint32_t threads, uint8_t* out, uint64_t out_cap,
                                         uint64_t* ticket) {
  if (!ctx || (!d_bwt && size) || !lf || !freqs || !out || !ticket) return -1;
  BwtEngine& e = ctx->eng;
  if (size > e.max_block) return -1;
  return wavelet_encode_begin(e, d_bwt, size, lf, n_lf, freqs, pick_threads(threads), out, out_cap, ticket);
}

int bwtc_hip_wavelet_encode_device_prepare(bwtc_hip_ctx* ctx, const uint8_t* d_bwt, uint32_t size,
                                           const uint32_t* lf, uint32_t n_lf, const uint32_t* freqs,
                                           uint32_t threads, uint8_t* out, uint64_t out_cap,
                                           uint64_t* ticket) {
  if (!ctx || (!d_bwt && size) || !lf || !freqs || !out || !ticket) return -1;
  BwtEngine& e = ctx->eng;
  if (size > e.max_block) return -1;
  return wavelet_encode_prepare(e, d_bwt, size, lf, n_lf, freqs, pick_threads(threads), out, out_cap, ticket);
}

int bwtc_hip_wavelet_encode_queue(bwtc_hip_ctx* ctx, uint64_t ticket, uint32_t state_in, uint32_t* state_out) {
  if (!ctx || !state_out || state_in > 7) return -1;
  return wavelet_encode_queue(ctx->eng, ticket, state_in, state_out);
}

int bwtc_hip_wavelet_encode_end(bwtc_hip_ctx* ctx, uint64_t ticket, uint64_t* out_bytes) {
  if (!ctx || !out_bytes) return -1;
  return wavelet_encode_end(ctx->eng, ticket, out_bytes);
}

int bwtc_hip_wavelet_encode_device(bwtc_hip_ctx* ctx, const uint8_t* d_bwt, uint32_t size,
                                   const uint32_t* lf, uint32_t n_lf, const uint32_t* freqs,
                                   uint32_t threads, uint8_t* out, uint64_t out_cap,
                                   uint64_t* out_bytes) {
  uint64_t ticket = 0;
  int rc = bwtc_hip_wavelet_encode_device_begin(ctx, d_bwt, size, lf, n_lf, freqs, threads, out, out_cap, &ticket);
  if (rc) return rc;
  return bwtc_hip_wavelet_encode_end(ctx, ticket, out_bytes);
}

int bwtc_hip_wavelet_encode(bwtc_hip_ctx* ctx, const uint8_t* bwt, uint32_t size,
                            const uint32_t* lf, uint32_t n_lf, const uint32_t* freqs,
                            uint32_t threads, uint8_t* out, uint64_t out_cap, uint64_t* out_bytes) {
  if (!ctx || (!bwt && size)) return -1;
  BwtEngine& e = ctx->eng;
  if (size > e.max_block) return -1;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  int rc = stage_in(e, bwt, size);
  if (rc) return rc;
  return bwtc_hip_wavelet_encode_device(ctx, e.d_in, size, lf, n_lf, freqs, threads, out, out_cap, out_bytes);
}

int bwtc_hip_transform_and_encode_wavelet(bwtc_hip_ctx* ctx, uint8_t* block, uint32_t size,
                                          uint32_t starting_points, uint32_t threads,
                                          uint8_t* out, uint64_t out_cap, uint64_t* out_bytes) {
  if (!ctx || (!block && size) || !out || !out_bytes) return -1;
  BwtEngine& e = ctx->eng;
  if (size > e.max_block) return -1;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  uint32_t lf[256];
  uint32_t freqs[256];
  std::memset(freqs, 0, sizeof freqs);                    // WaveletCoders.cpp:78-79
  const uint32_t n_lf = bwtc_hip_n_lf(size, starting_points);
  int rc = stage_in(e, block, size);
  if (rc) return rc;
  rc = e.transform(e.d_in, e.d_in, size, false, lf, n_lf, freqs);
  if (rc) return rc;
  rc = stage_out(e, block, size);                         // the run scanner reuses the workspace
  if (rc) return rc;
  return bwtc_hip_wavelet_encode_device(ctx, e.d_in, size, lf, n_lf, freqs, threads, out, out_cap, out_bytes);
}

int bwtc_hip_host_wavelet_sections(uint32_t n_sections, const uint32_t* first_run,
                                   const uint8_t* run_sym, const uint32_t* run_start,
                                   const uint32_t* run_freqs, const uint32_t* dist_offset,
                                   const uint32_t* dist_len, const uint32_t* dist_cnt,
                                   uint32_t threads, char coder, uint32_t* state, uint8_t* out,
                                   uint64_t out_cap, uint64_t* out_bytes) {
  if (!first_run || !run_sym || !run_start || !run_freqs || !dist_offset || !dist_len || !dist_cnt ||
      !state || !out || !out_bytes || n_sections > 256 || !bwtc::wavelet::isWaveletModel(coder)) return -1;
  std::vector<std::vector<std::pair<uint32_t, uint32_t> > > dist(n_sections);
  std::vector<bwtc::wavelet::SectionRuns> secs(n_sections);
  for (uint32_t s = 0; s < n_sections; ++s) {
    for (uint32_t i = dist_offset[s]; i < dist_offset[s + 1]; ++i) dist[s].push_back(std::make_pair(dist_len[i], dist_cnt[i]));
    secs[s].symbols = run_sym + first_run[s];
    secs[s].starts = run_start + first_run[s];
    secs[s].n_runs = first_run[s + 1] - first_run[s];
    secs[s].run_freqs = run_freqs + (size_t)s * 256;
    secs[s].dist = dist[s].data();
    secs[s].n_dist = dist[s].size();
  }
  std::vector<bwtc::wavelet::SectionOutput> outs;
  bwtc::wavelet::encodeSections(secs, pick_threads(threads), state, &outs, coder);
  uint64_t n = 0;
  for (uint32_t s = 0; s < n_sections; ++s) {
    if (n + outs[s].bytes.size() > out_cap) return -1;
    std::memcpy(out + n, outs[s].bytes.data(), outs[s].bytes.size());
    n += outs[s].bytes.size();
  }
  *out_bytes = n;
  return 0;
}

int bwtc_hip_host_wavelet_streams(uint32_t n_sections, const uint32_t* first_run,
                                  const uint8_t* run_sym, const uint32_t* run_start,
                                  const uint32_t* run_freqs, const uint32_t* dist_offset,
                                  const uint32_t* dist_len, const uint32_t* dist_cnt,
                                  uint32_t threads, char coder, uint32_t* state, uint8_t* out,
                                  uint64_t out_cap, uint64_t* out_bytes) {
  if (!first_run || !run_sym || !run_start || !run_freqs || !dist_offset || !dist_len || !dist_cnt ||
      !state || !out || !out_bytes || n_sections > 256 || !bwtc::wavelet::isWaveletModel(coder)) return -1;
  std::vector<std::vector<std::pair<uint32_t, uint32_t> > > dist(n_sections);
  std::vector<bwtc::wavelet::SectionRuns> secs(n_sections);
  for (uint32_t s = 0; s < n_sections; ++s) {
    for (uint32_t i = dist_offset[s]; i < dist_offset[s + 1]; ++i) dist[s].push_back(std::make_pair(dist_len[i], dist_cnt[i]));
    secs[s].symbols = run_sym + first_run[s];
    secs[s].starts = run_start + first_run[s];
    secs[s].n_runs = first_run[s + 1] - first_run[s];
    secs[s].run_freqs = run_freqs + (size_t)s * 256;
    secs[s].dist = dist[s].data();
    secs[s].n_dist = dist[s].size();
  }
  bwtc::wavelet::StreamPlan plan;
  if (!bwtc::wavelet::planStreams(secs, &plan)) return -5;
  std::vector<uint32_t> coded_pos;
  std::vector<uint8_t> codes;
  if (!bwtc::wavelet::expandStreamsOnHost(plan, secs, &coded_pos, &codes)) return -3;
  codes.push_back(0);
  std::vector<bwtc::wavelet::SectionOutput> outs;
  bwtc::wavelet::codeStreams(plan, coded_pos.data(), codes.data(), pick_threads(threads), state, &outs, nullptr, coder);
  uint64_t n = 0;
  for (uint32_t s = 0; s < n_sections; ++s) {
    if (n + outs[s].bytes.size() > out_cap) return -1;
    std::memcpy(out + n, outs[s].bytes.data(), outs[s].bytes.size());
    n += outs[s].bytes.size();
  }
  *out_bytes = n;
  return 0;
}

int bwtc_hip_host_wavelet_streams_lanes(uint32_t n_sections, const uint32_t* first_run,
                                        const uint8_t* run_sym, const uint32_t* run_start,
                                        const uint32_t* run_freqs, const uint32_t* dist_offset,
                                        const uint32_t* dist_len, const uint32_t* dist_cnt,
                                        uint32_t threads, char coder, uint32_t* state, uint8_t* out,
                                        uint64_t out_cap, uint64_t* out_bytes) {
  if (!first_run || !run_sym || !run_start || !run_freqs || !dist_offset || !dist_len || !dist_cnt ||
      !state || !out || !out_bytes || n_sections > 256 || coder != 'B') return -1;
  std::vector<std::vector<std::pair<uint32_t, uint32_t> > > dist(n_sections);
  std::vector<bwtc::wavelet::SectionRuns> secs(n_sections);
  for (uint32_t s = 0; s < n_sections; ++s) {
    for (uint32_t i = dist_offset[s]; i < dist_offset[s + 1]; ++i) dist[s].push_back(std::make_pair(dist_len[i], dist_cnt[i]));
    secs[s].symbols = run_sym + first_run[s];
    secs[s].starts = run_start + first_run[s];
    secs[s].n_runs = first_run[s + 1] - first_run[s];
    secs[s].run_freqs = run_freqs + (size_t)s * 256;
    secs[s].dist = dist[s].data();
    secs[s].n_dist = dist[s].size();
  }
  bwtc::wavelet::StreamPlan plan;
  if (!bwtc::wavelet::planStreams(secs, &plan)) return -5;
  std::vector<uint32_t> coded_pos;
  std::vector<uint8_t> codes;
  if (!bwtc::wavelet::expandStreamsOnHost(plan, secs, &coded_pos, &codes)) return -3;
  const uint32_t total = coded_pos.back();
  std::vector<uint32_t> packed(total / 16 + 8, 0);
  std::memcpy(packed.data(), codes.data(), codes.size());
  std::vector<bwtc::wavelet::gm::Task> tasks;
  std::vector<bwtc::wavelet::gm::Chunk> chunks;
  bwtc::wavelet::gm::buildTasks(plan, coded_pos.data(), &tasks, &chunks);
  std::vector<uint16_t> w((size_t)total + 8);
  if (!bwtc::wavelet::gm::modelsOnHostLanes(packed.data(), total, tasks, chunks, state, w.data())) return -7;
  std::vector<bwtc::wavelet::SectionOutput> outs(plan.sections.size());
  bwtc::wavelet::StreamCoder sc(plan, coded_pos.data(), *state, bwtc::wavelet::StreamCoder::Modelled());
  std::atomic<size_t> cursor(0);
  (void)threads;
  sc.codeSectionsW(&cursor, w.data(), &outs);
  uint64_t n = 0;
  for (uint32_t s = 0; s < n_sections; ++s) {
    if (n + outs[s].bytes.size() > out_cap) return -1;
    std::memcpy(out + n, outs[s].bytes.data(), outs[s].bytes.size());
    n += outs[s].bytes.size();
  }
  *out_bytes = n;
  return 0;
}

void bwtc_hip_host_huffman_lengths(const uint64_t* freqs, uint8_t* clen) {
  bwtc::utils::calculateHuffmanLengths(freqs, clen);
}
void bwtc_hip_host_huffman_codes(const uint8_t* clen, uint32_t* code) {
  bwtc::utils::computeHuffmanCodes(clen, code);
}
uint32_t bwtc_hip_host_serialize_shape(const uint8_t* clen, uint8_t* out, uint32_t cap) {
  bool any = false;
  for (int c = 0; c < 256; ++c) any = any || clen[c];
  if (!any) return 0;
  std::vector<uint8_t> v;
  bwtc::serializeShape(clen, v);
  if (v.size() > cap) return 0;
  std::memcpy(out, v.data(), v.size());
  return (uint32_t)v.size();
}
uint32_t bwtc_hip_host_sections(const uint32_t* freqs, uint32_t* section_len) {
  const std::vector<uint32_t> s = bwtc::deduceSections(freqs);
  for (size_t i = 0; i < s.size(); ++i) section_len[i] = s[i];
  return (uint32_t)s.size();
}
uint32_t bwtc_hip_host_bwtblock_header(const uint32_t* lf, uint32_t n_lf, uint8_t* out, uint32_t cap) {
  std::vector<uint8_t> v;
  bwtc::writeBWTBlockHeader(lf, n_lf, v);
  if (v.size() > cap) return 0;
  std::memcpy(out, v.data(), v.size());
  return (uint32_t)v.size();
}

int bwtc_hip_suffix_array(bwtc_hip_ctx* ctx, const uint8_t* T, uint32_t length, uint32_t* sa) {
  if (!ctx || !T || !sa) return -1;
  if (length == 0) return 0;
  BwtEngine& e = ctx->eng;
  if ((u64)length > e.cap) return -1;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  BWTC_HIP_TRY(hipMemcpyAsync(e.d_in, T, length, hipMemcpyHostToDevice, e.stream));
  u32 hist[256];
  int rc = e.load_text(e.d_in, length, length, false, hist);
  if (rc) return rc;
  rc = e.suffix_sort(length, hist, false);
  if (rc) return rc;
  BWTC_HIP_TRY(hipMemcpyAsync(sa, e.d_SA, (u64)length * 4, hipMemcpyDeviceToHost, e.stream));
  BWTC_HIP_TRY(e.wait());
  return 0;
}

int bwtc_hip_test_sort_u32(bwtc_hip_ctx* ctx, uint32_t* keys, uint32_t* vals, uint64_t n, int nbits) {
  return test_sort<u32>(ctx, keys, vals, n, nbits);
}
int bwtc_hip_test_sort_u64(bwtc_hip_ctx* ctx, uint64_t* keys, uint32_t* vals, uint64_t n, int nbits) {
  return test_sort<u64>(ctx, keys, vals, n, nbits);
}

// Test hook of the GPU lane engine (gpu_lanes.hpp): k chains of w-elements, chain j = elements [bounds[j], bounds[j+1]),
// each coded from a fresh BitEncoder -- on GPU lanes (mode 0) or by the host's runChainW (mode 1).  The chains' bytes
// (flush included) come back one after the other in `out`, offsets[j] .. offsets[j+1].
int bwtc_hip_test_gpu_lanes(bwtc_hip_ctx* ctx, const uint16_t* w, uint64_t n, const uint64_t* bounds, uint32_t k, int mode,
                            uint8_t* out, uint64_t out_cap, uint64_t* offsets) {
  if (!ctx || !w || !bounds || !out || !offsets || k == 0 || bounds[k] > n) return -1;
  for (uint32_t j = 0; j < k; ++j) if (bounds[j] > bounds[j + 1]) return -1;       // chains are [bounds[j], bounds[j + 1])
  BWTC_HIP_TRY(hipSetDevice(ctx->eng.device));
  std::vector<std::vector<uint8_t> > outs(k);
  struct Src : bwtc::wavelet::ChainSource {
    const uint16_t* w; const uint64_t* bounds; uint32_t k; std::vector<std::vector<uint8_t> >* outs; std::atomic<uint32_t> at{0}, finished{0};
    bool next(bwtc::wavelet::ChainDesc* d) {
      const uint32_t j = at.fetch_add(1);
      if (j >= k) return false;
      d->codes = nullptr; d->prob = nullptr; d->w = w; d->begin = bounds[j]; d->end = bounds[j + 1]; d->out = &(*outs)[j];
      d->cookie = this;
      return true;
    }
    void done(void*) { ++finished; }
  } src;
  src.w = w; src.bounds = bounds; src.k = k; src.outs = &outs;
  int rc = 0;
  if (mode == 0) {
    uint16_t* d_w = nullptr;
    BWTC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_w), n * 2 + 256));
    if (hipMemcpy(d_w, w, n * 2, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d_w); return -3; }
    const uint16_t* dw = d_w;
    bwtc_hip::GpuLaneStats st;
    rc = bwtc_hip::runCoderLanesGpu(src, ctx->eng.device, [dw](void*) { return dw; }, &st);
    (void)hipFree(d_w);
    if (std::getenv("BWTC_HIP_DEBUG"))
      std::fprintf(stderr, "gpu lanes: %llu chains on lanes, %llu on the host, %llu elements, %llu slices, %llu bytes; %.1f ns per lane step\n", (unsigned long long)st.chains.load(),
                   (unsigned long long)st.host_chains.load(), (unsigned long long)st.elements.load(), (unsigned long long)st.slices.load(), (unsigned long long)st.bytes.load(),
                   st.slice_steps.load() ? (double)st.slice_ns.load() / (double)st.slice_steps.load() : 0.0);
  } else {
    bwtc::wavelet::ChainDesc d;
    while (src.next(&d)) {
      bwtc::wavelet::CoderChain c;
      c.start(d.begin, d.end, d.out);
      bwtc::wavelet::runChainW(c, d.w, c.e);
      c.finish();
      src.done(d.cookie);
    }
  }
  if (rc) return rc;
  if (src.finished.load() != k) return -3;
  uint64_t at = 0;
  for (uint32_t j = 0; j < k; ++j) {
    offsets[j] = at;
    if (at + outs[j].size() > out_cap) return -1;
    std::memcpy(out + at, outs[j].data(), outs[j].size());
    at += outs[j].size();
  }
  offsets[k] = at;
  return 0;
}
int bwtc_hip_test_scan_u32(bwtc_hip_ctx* ctx, uint32_t* data, uint64_t n) {
  if (!ctx || !data) return -1;
  BwtEngine& e = ctx->eng;
  // the scan borrows R1 for the data and R2 for the tile partials
  if (n * 4 > e.cap * 8) return -1;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  u32* d = static_cast<u32*>(e.d_R1);
  BwtEngine::ScanScope scan_scope(e);                // the context's single-launch scan, where it is set up
  BWTC_HIP_TRY(hipMemcpyAsync(d, data, n * 4, hipMemcpyHostToDevice, e.stream));
  exclusive_scan_u32(d, n, static_cast<u32*>(e.d_R2), e.stream);
  BWTC_HIP_TRY(hipMemcpyAsync(data, d, n * 4, hipMemcpyDeviceToHost, e.stream));
  e.h_small[1] = 0;
  if (e.scan_chain.err) BWTC_HIP_TRY(hipMemcpyAsync(e.h_small + 1, e.scan_chain.err, 4, hipMemcpyDeviceToHost, e.stream));
  BWTC_HIP_TRY(e.wait());
  BWTC_HIP_TRY(hipGetLastError());
  return e.h_small[1] ? -3 : 0;
}

// ---- pair-replacing pre-stage (prepr.hip, prepr_host.cpp) ---------------------------------------------
bwtc_hip_grammar* bwtc_hip_grammar_create(void) { return new (std::nothrow) bwtc_hip_grammar(); }
void bwtc_hip_grammar_destroy(bwtc_hip_grammar* g) { delete g; }
uint32_t bwtc_hip_grammar_rules(const bwtc_hip_grammar* g) { return g ? g->g.numberOfRules() : 0; }
uint32_t bwtc_hip_grammar_special_symbols(const bwtc_hip_grammar* g) { return g ? g->g.numberOfSpecialSymbols() : 0; }
int bwtc_hip_grammar_is_special(const bwtc_hip_grammar* g, unsigned symbol) { return g && g->g.isSpecial(static_cast<uint8_t>(symbol)) ? 1 : 0; }
int bwtc_hip_grammar_write(const bwtc_hip_grammar* g, uint8_t* out, uint64_t cap, uint64_t* bytes) {
  if (!g || !out || !bytes) return -1;
  std::vector<uint8_t> v;
  g->g.write(&v);
  if (v.size() > cap) return -1;
  std::memcpy(out, v.data(), v.size());
  *bytes = v.size();
  return 0;
}
int bwtc_hip_grammar_read(bwtc_hip_grammar* g, const uint8_t* in, uint64_t n, uint64_t* consumed) {
  if (!g || !in || !consumed || g->g.numberOfRules() || g->g.numberOfSpecialSymbols()) return -1;
  const size_t used = g->g.read(in, static_cast<size_t>(n));
  if (used == static_cast<size_t>(-1)) return -1;
  *consumed = used;
  return 0;
}
int bwtc_hip_pair_replace_device(bwtc_hip_ctx* ctx, bwtc_hip_grammar* g, const uint8_t* d_src, uint64_t n,
                                 uint8_t* d_dst, uint64_t* n_out, uint32_t* replaced) {
  if (!ctx || !g || !d_src || !d_dst || !n_out || !replaced) return -1;
  return bwtc_hip::pair_replace_device(ctx->eng, g->g, d_src, n, d_dst, n_out, replaced);
}
int bwtc_hip_host_precompress(bwtc_hip_grammar* g, const char* options, uint8_t* block, uint64_t n, uint64_t* n_out);
int bwtc_hip_precompress(bwtc_hip_ctx* ctx, bwtc_hip_grammar* g, const char* options, uint8_t* block, uint64_t n, uint64_t* n_out) {
  if (!ctx || !g || !block || !n_out) return -1;
  *n_out = n;
  if (!options || !options[0] || n < 3) return 0;
  BwtEngine& e = ctx->eng;
  BWTC_HIP_TRY(hipSetDevice(e.device));
  // The device passes index their tiles and output offsets with 32 bits: a precompressor block of 2 GiB and more (the
  // reference's PairReplacer takes them up to 4 GiB: `compress --prepr p -m 2903` and above) takes the host's twin of
  // the same two passes -- same bytes, same grammar.
  if (n >= (1ull << 31) || n / 4096 + 2 > e.cap) return bwtc_hip_host_precompress(g, options, block, n, n_out);
  uint8_t *d_a = nullptr, *d_b = nullptr;
  BWTC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_a), 2 * n + 64));
  if (hipMalloc(reinterpret_cast<void**>(&d_b), 2 * n + 64) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_a); return -2; }
  int rc = hipMemcpy(d_a, block, n, hipMemcpyHostToDevice) == hipSuccess ? 0 : -3;
  uint64_t length = n;
  for (const char* c = options; rc == 0 && *c; ++c) {            // Precompressor.cpp:76-105
    const uint64_t before = length;
    if (*c == 'p') {
      uint64_t now = 0;
      uint32_t replaced = 0;
      rc = bwtc_hip::pair_replace_device(e, g->g, d_a, length, d_b, &now, &replaced);
      if (rc) break;
      std::swap(d_a, d_b);
      length = now;
    }
    if (length == before || length < 3) break;
  }
  if (rc == 0 && hipMemcpy(block, d_a, length, hipMemcpyDeviceToHost) != hipSuccess) rc = -3;
  (void)hipFree(d_a);
  (void)hipFree(d_b);
  if (rc == 0) *n_out = length;
  return rc;
}
int bwtc_hip_host_precompress(bwtc_hip_grammar* g, const char* options, uint8_t* block, uint64_t n, uint64_t* n_out) {
  if (!g || !block || !n_out) return -1;
  *n_out = n;
  if (!options || !options[0] || n < 3) return 0;
  std::vector<uint8_t> other(2 * n + 64);
  std::vector<uint64_t> bf(256), pf(65536);
  uint64_t length = n;
  for (const char* c = options; *c; ++c) {
    const uint64_t before = length;
    if (*c == 'p') {
      bwtc::prepr::pairStatisticsOnHost(block, length, bf.data(), pf.data());
      bwtc::prepr::Replacements r;
      bwtc::prepr::decideReplacements(&g->g, bf.data(), pf.data(), &r);
      if (r.count) {
        length = bwtc::prepr::writeReplacedOnHost(r, block, length, other.data());
        std::memcpy(block, other.data(), length);
      }
    }
    if (length == before || length < 3) break;
  }
  *n_out = length;
  return 0;
}
int bwtc_hip_postprocess(const bwtc_hip_grammar* g, const uint8_t* data, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* n_out) {
  if (!g || (!data && n) || !out || !n_out) return -1;
  std::vector<uint8_t> v;
  if (!bwtc::prepr::postprocess(g->g, data, static_cast<size_t>(n), &v, static_cast<size_t>(cap))) return -1;
  if (!v.empty()) std::memcpy(out, v.data(), v.size());
  *n_out = v.size();
  return 0;
}

}  // extern "C"
