/*
 * bwtc_hip.h -- C ABI of libbwtc_hip.so, the MI355X (gfx950) back-end for the
 * BWTManager -> BWTransform hot path of pjmikkol/bwtc and for the data-parallel half of
 * its 'H' entropy coder.  Plain pointers and sizes only; no exceptions cross this
 * boundary.  Citations are file:line relative to the reference tree.
 *
 * Return convention (mirrors divbwtf, bwtransforms/divsufsort.c:448,513-515):
 *   0 success, -1 bad arguments, -2 out of (device) memory, -3 HIP runtime error.
 *
 * A context owns one device, one stream and a persistent HBM workspace sized for
 * max_block_size, the way the reference back-ends own their per-call workspace
 * (bwtransforms/divsufsort.c:491-493) -- but allocated once, not per block.
 * A context is not thread-safe; use one per worker (= one per GPU in the block farm).
 */
#ifndef BWTC_HIP_H
#define BWTC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bwtc_hip_ctx bwtc_hip_ctx;

/* Statistics of the last transform on a context (all device work, HIP events on the
 * context's stream). */
typedef struct bwtc_hip_stats {
  uint32_t n;                 /* suffixes sorted = block size + 1                          */
  uint32_t rounds;            /* prefix-doubling rounds after the initial 4-byte sort      */
  uint64_t active_sum;        /* sum over rounds of suffixes still in non-singleton groups */
  uint64_t sort_pass_items;   /* sum over all radix passes of items moved                  */
  float    ms_total;          /* whole transform, device time                              */
  float    ms_sort;           /* radix sort passes only                                    */
  /* round 4: which way the block went, and the algorithmic bytes of the kernels that ran      */
  uint32_t route;             /* bit 0: long-key initial sort; bit 1: finisher settled the ties;
                                 bit 2: text rounds ran; bit 3: rank[] completed late for doubling rounds */
  uint32_t finisher_entries;  /* list entries over all finisher passes                      */
  uint64_t alg_bytes;         /* compulsory bytes of the transform's kernels: every array a kernel
                                 reads counted once, every array it writes counted once (SURVEY.md
                                 8(d)'s counting rule applied to the kernel list that actually ran) */
} bwtc_hip_stats;

/* Per-kernel device timing of the dominant kernel (the radix scatter pass), gathered with
 * HIP events on the context's stream while profiling is on.  bytes = ALGORITHMIC bytes:
 * items * (key + value) read once and written once. */
typedef struct bwtc_hip_kernel_timers {
  uint64_t scatter_launches;
  uint64_t scatter_bytes;
  double   scatter_ms;
} bwtc_hip_kernel_timers;

int  bwtc_hip_device_count(void);
const char* bwtc_hip_version(void);

/* Workspace bytes a context for blocks up to max_block_size bytes will allocate. */
uint64_t bwtc_hip_workspace_bytes(uint32_t max_block_size);

/* Create / destroy.  *ctx_out is NULL on failure. */
int  bwtc_hip_create(int device, uint32_t max_block_size, bwtc_hip_ctx** ctx_out);
void bwtc_hip_destroy(bwtc_hip_ctx* ctx);
void* bwtc_hip_stream(bwtc_hip_ctx* ctx);          /* hipStream_t of the context */
int  bwtc_hip_get_stats(bwtc_hip_ctx* ctx, bwtc_hip_stats* out);
int  bwtc_hip_set_profiling(bwtc_hip_ctx* ctx, int on);
int  bwtc_hip_get_kernel_timers(bwtc_hip_ctx* ctx, bwtc_hip_kernel_timers* out, int reset);
/* What this GPU streams: a copy of `bytes` bytes (16 bytes per lane, the library's own kernel) repeated `reps`
 * times on the context's stream, timed with HIP events; *gbps = bytes read + bytes written per second / 1e9.  The
 * denominator bench.py quotes beside the spec peak (not part of the reference's interface: measurement only). */
int  bwtc_hip_copy_probe(bwtc_hip_ctx* ctx, uint64_t bytes, int reps, double* gbps);

/* Device buffers for callers of the *_device entry points that do not link the HIP runtime
 * themselves (allocation on the context's GPU; copies are synchronous). */
void* bwtc_hip_malloc(bwtc_hip_ctx* ctx, uint64_t bytes);
void  bwtc_hip_free(bwtc_hip_ctx* ctx, void* d_ptr);
int   bwtc_hip_memcpy_to_device(bwtc_hip_ctx* ctx, void* d_dst, const void* src, uint64_t bytes);
int   bwtc_hip_memcpy_to_host(bwtc_hip_ctx* ctx, void* dst, const void* d_src, uint64_t bytes);

/* Page-locked host memory and uploads that overlap the context's kernels: a farm worker keeps
 * two device buffers and copies block i+1 in (on the context's own copy stream) while block i
 * is transformed on the compute stream.  _async returns at once (src should come from
 * bwtc_hip_host_alloc; pageable memory still works but is staged by the runtime);
 * bwtc_hip_copy_wait blocks until every copy issued so far on this context has landed.
 * These replace nothing in the reference (its blocks never leave host memory); they are what
 * SURVEY.md 8(e) calls the pinned staging ring of a GPU worker. */
void* bwtc_hip_host_alloc(bwtc_hip_ctx* ctx, uint64_t bytes);
void  bwtc_hip_host_free(bwtc_hip_ctx* ctx, void* p);
int   bwtc_hip_memcpy_to_device_async(bwtc_hip_ctx* ctx, void* d_dst, const void* src, uint64_t bytes);
int   bwtc_hip_copy_wait(bwtc_hip_ctx* ctx);

/* LF powers a block of `size` bytes gets for `starting_points`
 * (BWTManager::setStartingPoints clamp, bwtransforms/BWTManager.cpp:60-64, then
 * BWTBlock::prepareLFpowers, BWTBlock.cpp:104-108). */
uint32_t bwtc_hip_n_lf(uint32_t size, uint32_t starting_points);

/* Raw transform.  Replaces BWTransform::doTransform(byte* begin, uint32 length,
 * std::vector<uint32>& LF, uint32 freqs[256]) (bwtransforms/BWTransform.hpp:53-58) as
 * implemented by Divsufsorter (bwtransforms/Divsufsorter.hpp:60-65) and SAISBWTransform
 * (bwtransforms/SA-IS-bwt.cpp:48-54): T[0..length-1] (host memory, caller already planted
 * the 0 sentinel at T[length-1]) is transformed in place; every position except LF[0] is
 * written; lf[0..n_lf-1] receives the LF powers; freqs (may be NULL) is INCREMENTED. */
int bwtc_hip_bwt(bwtc_hip_ctx* ctx, uint8_t* T, uint32_t length, uint32_t* lf, uint32_t n_lf,
                 uint32_t* freqs);

/* Block-level transform.  Replaces BWTransform::doTransform(BWTBlock&, uint32 freqs[256])
 * (bwtransforms/BWTransform.cpp:52-64): block[0..size-1] (host) is reversed, terminated,
 * transformed and the end-of-block hole filled, all on the device; block[size] is never
 * touched (the reference borrows and restores it).  n_lf as from bwtc_hip_n_lf(). */
int bwtc_hip_bwt_block(bwtc_hip_ctx* ctx, uint8_t* block, uint32_t size, uint32_t* lf,
                       uint32_t n_lf, uint32_t* freqs);

/* Same, device-resident: d_in/d_out are device pointers to `size` bytes (may alias);
 * lf (n_lf words) and freqs (256 words, incremented; may be NULL) are HOST pointers.
 * Work is issued on the context's stream and has completed on return. */
int bwtc_hip_bwt_block_device(bwtc_hip_ctx* ctx, const uint8_t* d_in, uint8_t* d_out,
                              uint32_t size, uint32_t* lf, uint32_t n_lf, uint32_t* freqs);

/* ---- inverse transform ---------------------------------------------------------------- */

/* Replaces InverseBWTransform::doTransform(BWTBlock&) (bwtransforms/InverseBWT.cpp:47-51) and
 * the MTL-SA walk behind it (bwtransforms/MtlSaInverseBWT.cpp:246-362): block[0..size-1]
 * (host) holds a transformed block as the forward path leaves it, lf[0..n_lf-1] its LF powers
 * (lf[0] = end-of-block row); on return block holds the original bytes.  The reference needs
 * block[size] as scratch (InverseBWT.cpp:49); this entry point does not touch it.
 * Returns -4 if an LF power is inconsistent with the data. */
int bwtc_hip_inverse_bwt_block(bwtc_hip_ctx* ctx, uint8_t* block, uint32_t size,
                               const uint32_t* lf, uint32_t n_lf);

/* Same with device pointers (may alias); lf is a HOST array. */
int bwtc_hip_inverse_bwt_block_device(bwtc_hip_ctx* ctx, const uint8_t* d_bwt, uint8_t* d_out,
                                      uint32_t size, const uint32_t* lf, uint32_t n_lf);

/* ---- 'H' entropy coder (HuffmanCoders.cpp) ------------------------------------------- */

/* Upper bound of the bytes one encoded BWT block of `size` input bytes can take. */
uint64_t bwtc_hip_compress_bound(uint32_t size);

/* Encodes an already transformed block.  Replaces the part of
 * HuffmanEncoder::transformAndEncode (HuffmanCoders.cpp:51-61) after the transform:
 * writeBlockHeader (:271-313), encodeData (:119-257) and finishBlock (:259-261).  Output =
 * the complete BWT-block record: 48-bit big-endian length of the rest, BWTBlock header
 * (BWTBlock.cpp:61-86), section count and packed section lengths, then per section packed
 * run count, serialised code shape, Huffman-coded run symbols and gamma-coded run lengths.
 * d_bwt/d_out are device pointers (d_out 4-byte aligned, out_cap bytes); lf/freqs are the
 * HOST arrays the transform returned (freqs is not modified).  *out_bytes = record size. */
int bwtc_hip_huffman_encode_device(bwtc_hip_ctx* ctx, const uint8_t* d_bwt, uint32_t size,
                                   const uint32_t* lf, uint32_t n_lf, const uint32_t* freqs,
                                   uint8_t* d_out, uint64_t out_cap, uint64_t* out_bytes);

/* Same with host buffers (staged through the context's workspace). */
int bwtc_hip_huffman_encode(bwtc_hip_ctx* ctx, const uint8_t* bwt, uint32_t size,
                            const uint32_t* lf, uint32_t n_lf, const uint32_t* freqs,
                            uint8_t* out, uint64_t out_cap, uint64_t* out_bytes);

/* HuffmanEncoder::transformAndEncode(block, bwtm, out) (HuffmanCoders.cpp:51-61) in one
 * call: block (host, `size` bytes, left transformed like the reference leaves it) ->
 * encoded BWT-block record in `out` (host).  starting_points as given to
 * BWTManager::setStartingPoints. */
int bwtc_hip_transform_and_encode(bwtc_hip_ctx* ctx, uint8_t* block, uint32_t size,
                                  uint32_t starting_points, uint8_t* out, uint64_t out_cap,
                                  uint64_t* out_bytes);

/* ---- wavelet coders: run scanner (front-end statistics only) ---------------------------- */

/* utils::calculateRunsAndCharacters (Utils.cpp:128-147) for every section of a transformed
 * block, as the WaveletTree constructor needs it (WaveletTree.hpp:294-308): bwt = host buffer
 * of `size` bytes, freqs = the histogram the transform returned.  Outputs (host):
 *   *n_sections, section_len[256], run_freqs[256*256] ([section][symbol]), total_runs[256],
 *   dist_offset[257] and the (dist_len[i], dist_cnt[i]) pairs of section s in
 *   [dist_offset[s], dist_offset[s+1]), ascending by length (the reference's std::map order).
 * Returns -1 if dist_cap pairs do not suffice. */
int bwtc_hip_wavelet_section_stats(bwtc_hip_ctx* ctx, const uint8_t* bwt, uint32_t size,
                                   const uint32_t* freqs, uint32_t* n_sections,
                                   uint32_t* section_len, uint32_t* run_freqs,
                                   uint64_t* total_runs, uint32_t* dist_offset,
                                   uint32_t* dist_len, uint32_t* dist_cnt, uint32_t dist_cap);

/* WaveletEncoder::transformAndEncode (WaveletCoders.cpp:77-87) for coder 'B' (FSM8 model):
 * block (host, left transformed) -> encoded BWT-block record in `out` (host).  The transform
 * and the run scanner run on the GPU; tree building and the range coder are bit-serial and
 * run on `threads` host threads, one section at a time per thread (0 = all cores).  The
 * coder's model state is carried from call to call like the reference's encoder object
 * carries it from block to block; bwtc_hip_wavelet_reset starts a new stream. */
int bwtc_hip_transform_and_encode_wavelet(bwtc_hip_ctx* ctx, uint8_t* block, uint32_t size,
                                          uint32_t starting_points, uint32_t threads,
                                          uint8_t* out, uint64_t out_cap, uint64_t* out_bytes);
/* Same for an already transformed block (host). */
int bwtc_hip_wavelet_encode(bwtc_hip_ctx* ctx, const uint8_t* bwt, uint32_t size,
                            const uint32_t* lf, uint32_t n_lf, const uint32_t* freqs,
                            uint32_t threads, uint8_t* out, uint64_t out_cap, uint64_t* out_bytes);
/* Same for a transformed block that is resident on the device (the record still goes to host
 * memory: the coder runs there). */
int bwtc_hip_wavelet_encode_device(bwtc_hip_ctx* ctx, const uint8_t* d_bwt, uint32_t size,
                                   const uint32_t* lf, uint32_t n_lf, const uint32_t* freqs,
                                   uint32_t threads, uint8_t* out, uint64_t out_cap,
                                   uint64_t* out_bytes);
/* The same in two halves, so that a caller with several blocks overlaps them (the
 * reference's Compressor::compress loop, Compressor.cpp:103-137, made a pipeline): _begin does
 * the device work of the block (run scanner, tree bit vectors, traversal order, gap flags),
 * advances the encoder's carried model state and queues the block's adaptive models and range
 * coders on the context's worker threads (created by the first call, `threads` of them, 0 =
 * the CPUs the process may use -- affinity and cgroup quota -- up to 64); it returns as soon as the GPU is free for the next block.  _end waits
 * for that block; the record is then in the `out` given to _begin, *out_bytes long.  Records
 * are those of a strictly sequential encoder whatever the overlap.  At most
 * bwtc_hip_wavelet_depth(ctx) blocks (BWTC_HIP_WAVELET_DEPTH, default 16) are under way; a
 * further _begin returns -6 at once: collect the oldest with _end first.  `out` must stay valid
 * until _end (or until bwtc_hip_destroy returns: it lets blocks under way finish);
 * _begin/_end of one context are called from one thread. */
int bwtc_hip_wavelet_encode_device_begin(bwtc_hip_ctx* ctx, const uint8_t* d_bwt, uint32_t size,
                                         const uint32_t* lf, uint32_t n_lf, const uint32_t* freqs,
                                         uint32_t threads, uint8_t* out, uint64_t out_cap,
                                         uint64_t* ticket);
int bwtc_hip_wavelet_encode_end(bwtc_hip_ctx* ctx, uint64_t ticket, uint64_t* out_bytes);
/* _begin in two halves, for ONE stream whose blocks are farmed over several contexts (one per
 * GPU): _prepare does everything that does not depend on earlier blocks -- the
 * whole device half -- and may run on all contexts at once; _queue gives the block its place in
 * the stream: state_in is the main model's carried state after the previous block (4 for the
 * first block of a stream; the reference's WaveletEncoder carries it inside its m_probModel,
 * probmodels/FSM.hpp:196-205), *state_out the state after this block, to be passed to the _queue
 * of the next block on whichever context holds it.  _queue calls of a stream are made in block
 * order; each returns at once.  _begin = _prepare + _queue with the context's own state.
 * With the adaptive models on the device (the default for 'B') _prepare returns when its device
 * half is done -- it leaves the state after the block for each of the eight states the block can
 * start in, so that _queue has nothing to wait for -- and _queue launches the passes that depend
 * on the state; the block joins the context's worker threads when its elements have reached the
 * host.  A block prepared while another prepared block of the same context has not been queued
 * yet takes that one's place in the device workspace: the earlier block's models then run on the
 * worker threads (same bytes). */
int bwtc_hip_wavelet_encode_device_prepare(bwtc_hip_ctx* ctx, const uint8_t* d_bwt, uint32_t size,
                                           const uint32_t* lf, uint32_t n_lf, const uint32_t* freqs,
                                           uint32_t threads, uint8_t* out, uint64_t out_cap,
                                           uint64_t* ticket);
int bwtc_hip_wavelet_encode_queue(bwtc_hip_ctx* ctx, uint64_t ticket, uint32_t state_in,
                                  uint32_t* state_out);
/* Blocks that may be between _begin and _end at once on this context. */
uint32_t bwtc_hip_wavelet_depth(bwtc_hip_ctx* ctx);
/* Lowers (or raises) the number of blocks that may be under way, and gives the page-locked staging buffers the
 * stream no longer needs back to the system: a stream's first blocks run at the default depth, the depth it has
 * shown to need (bwtc_hip_wavelet_depth_needed) is usually half of it, and eight ranks of a node each hold what
 * they keep.  -6: more blocks are under way than `depth` allows (collect some first). */
int bwtc_hip_wavelet_set_depth(bwtc_hip_ctx* ctx, uint32_t depth);
/* Blocks the caller has to keep between _begin and _end for the rate the stream has shown so far: the mean time
 * from a block's _begin to its finished record over the mean time between two _begins, a quarter more, plus two.
 * 0 until four blocks have finished.  A caller that keeps fewer in flight waits in _end; more only hold memory
 * (1.3 GB of page-locked staging per 256 MiB text block under way). */
uint32_t bwtc_hip_wavelet_depth_needed(bwtc_hip_ctx* ctx);
/* Host CPUs for the worker threads of the contexts of one node (one context per GPU):
 * the 'B' coder's host half is memory- and cache-hungry, so a context's workers -- and the
 * page-locked buffers they read, placed by first touch / the caller's policy -- belong on the NUMA
 * node of its GPU, and contexts must not share cores.
 *   _numa_node        NUMA node of the context's GPU (PCI bus id -> /sys/bus/pci/devices/<id>/numa_node),
 *                     -1 when the system does not say;
 *   _host_cpu_slice   host-only: the CPUs this process may use (its affinity mask), cut down to those
 *                     of `numa_node` when that is >= 0 and the node has any, split into `ranks`
 *                     contiguous slices; slice `rank` goes to cpus[0..], the return value is its size
 *                     (0: nothing to hand out, negative: bad arguments).  Callers give rank / ranks
 *                     among the contexts that share the node;
 *   _set_worker_cpus  the context's worker threads (existing ones and those made later) may run on
 *                     exactly these CPUs; n = 0 lifts the restriction.  The calling thread is left alone. */
int bwtc_hip_numa_node(bwtc_hip_ctx* ctx);
int bwtc_hip_host_cpu_slice(int numa_node, uint32_t rank, uint32_t ranks, uint32_t* cpus, uint32_t cap);
int bwtc_hip_set_worker_cpus(bwtc_hip_ctx* ctx, const uint32_t* cpus, uint32_t n);
/* Host time the context's worker threads have spent so far in the two host stages of the wavelet
 * coder (adaptive models, range coders; seconds summed over threads) and the blocks queued. */
int bwtc_hip_wavelet_host_clock(bwtc_hip_ctx* ctx, double* model_seconds, double* coder_seconds,
                                uint64_t* blocks);
/* Blocks that have joined the host half so far (*queued) and blocks whose record the worker
 * threads have finished (*finished): what a caller needs to tell the rate the host half sustains
 * from the rate at which blocks are begun. */
int bwtc_hip_wavelet_host_progress(bwtc_hip_ctx* ctx, uint64_t* queued, uint64_t* finished);
/* Mean time a block has been under way so far: from the start of its device half (_begin / _prepare)
 * to its finished record.  Blocks under way needed = that time / the time per block of the caller's
 * loop; a context keeps the limit it was created with (BWTC_HIP_WAVELET_DEPTH). */
int bwtc_hip_wavelet_latency(bwtc_hip_ctx* ctx, double* mean_seconds);
/* Host staging memory (page-locked where the system allows) held by this process for blocks under
 * way -- packed streams and w-elements of every context -- now and at its highest so far: what a
 * deployment has to provide per rank for the depth it runs (no reference counterpart; the reference
 * holds one block, Compressor.cpp:100-108). */
int bwtc_hip_host_staging_bytes(uint64_t* now, uint64_t* peak);
/* CPUs this process may keep busy: the hardware threads, cut down to the cgroup's CPU quota -- what
 * `threads` = 0 means in the calls above.  A caller that runs several contexts splits this number
 * between them (the reference is single-threaded, Compressor.cpp:67-70). */
uint32_t bwtc_hip_host_usable_cpus(void);
void bwtc_hip_wavelet_reset(bwtc_hip_ctx* ctx);
/* A new wavelet stream with the main probability model of coder letter `coder`
 * (WaveletEncoder(char), WaveletCoders.hpp:52; giveProbabilityModel,
 * probmodels/ProbabilityModel.cpp:47-76): 'B' FSM8 (what bwtc_hip_wavelet_reset selects),
 * 'b' FSM<6, EvenIntervalPredictor<4>>, 'u' EvenIntervalPredictor<4>.  -1 for other letters:
 * 'm' / 'M' (SimpleMarkov) index one entry past their history table in the reference
 * (:91-93 vs :110-118), their output is undefined and they are not offered. */
int bwtc_hip_wavelet_start(bwtc_hip_ctx* ctx, char coder);

/* Host half of the 'B' coder alone (no device work): the sections' runs as the GPU scanner
 * delivers them -> the concatenated section payloads (packed bitsInRoot, tree shape, range-coded
 * bytes per section).  first_run[n_sections+1] delimits every section's runs inside
 * run_sym[] / run_start[] (run_start has one extra entry: the end of the last run);
 * run_freqs[n_sections*256]; the run-length distribution of section s is the pairs
 * [dist_offset[s], dist_offset[s+1]).  *state is the model state carried in and out. */
int bwtc_hip_host_wavelet_sections(uint32_t n_sections, const uint32_t* first_run,
                                   const uint8_t* run_sym, const uint32_t* run_start,
                                   const uint32_t* run_freqs, const uint32_t* dist_offset,
                                   const uint32_t* dist_len, const uint32_t* dist_cnt,
                                   uint32_t threads, char coder, uint32_t* state, uint8_t* out,
                                   uint64_t out_cap, uint64_t* out_bytes);

/* Same arguments and same bytes as bwtc_hip_host_wavelet_sections, computed the way the device
 * path does it: the tree shapes are planned from the statistics alone, every run is expanded
 * into (tree node, bit) steps that are sorted into coding order, and the coder only sees the
 * finished streams.  The expansion is done with plain host loops here (no device work), so
 * the planning and stream-coding halves can be checked without a GPU.  -5: a shape the
 * stream path does not take (the library then uses the tree builder of ..._sections). */
int bwtc_hip_host_wavelet_streams(uint32_t n_sections, const uint32_t* first_run,
                                  const uint8_t* run_sym, const uint32_t* run_start,
                                  const uint32_t* run_freqs, const uint32_t* dist_offset,
                                  const uint32_t* dist_len, const uint32_t* dist_cnt,
                                  uint32_t threads, char coder, uint32_t* state, uint8_t* out,
                                  uint64_t out_cap, uint64_t* out_bytes);
/* The same, with the adaptive models run the way the GPU runs them (wavelet_gpu_models.hpp: state
 * scan, slot space, bracketed chains, w-elements) -- lane by lane on the host, for tests without a
 * GPU.  Coder 'B' only.  -7: the passes raised their error flag. */
int bwtc_hip_host_wavelet_streams_lanes(uint32_t n_sections, const uint32_t* first_run,
                                  const uint8_t* run_sym, const uint32_t* run_start,
                                  const uint32_t* run_freqs, const uint32_t* dist_offset,
                                  const uint32_t* dist_len, const uint32_t* dist_cnt,
                                  uint32_t threads, char coder, uint32_t* state, uint8_t* out,
                                  uint64_t out_cap, uint64_t* out_bytes);

/* Host-only pieces of the 'H' coder (no device work; usable without a GPU).  They are the
 * small-table steps the encoder runs between its device passes, exported so the host logic
 * can be checked on its own:
 *   lengths : utils::calculateHuffmanLengths (Utils.cpp:408-473), freqs[256] -> clen[256]
 *   codes   : utils::computeHuffmanCodes (Utils.cpp:180-202)
 *   shape   : HuffmanEncoder::serializeShape, byte padded (HuffmanCoders.cpp:63-86,181-192);
 *             returns bytes written (<= cap) or 0
 *   sections: section heuristic of writeBlockHeader (HuffmanCoders.cpp:282-296); returns count
 *   header  : BWTBlock::writeHeader (BWTBlock.cpp:61-86); returns bytes written */
void     bwtc_hip_host_huffman_lengths(const uint64_t* freqs, uint8_t* clen);
void     bwtc_hip_host_huffman_codes(const uint8_t* clen, uint32_t* code);
uint32_t bwtc_hip_host_serialize_shape(const uint8_t* clen, uint8_t* out, uint32_t cap);
uint32_t bwtc_hip_host_sections(const uint32_t* freqs, uint32_t* section_len);
uint32_t bwtc_hip_host_bwtblock_header(const uint32_t* lf, uint32_t n_lf, uint8_t* out, uint32_t cap);

/* Synthetic inputs of the BASELINE.json configurations (no device
 * work): kind 'r' random bytes (C1), 'd' uniform ACGT (C2), 't' token text (C3/C4/C5), generated
 * from splitmix64(seed) -- the same bytes as bwtc_amd/synth.py.  -1 for another kind. */
int bwtc_hip_synth(char kind, uint64_t seed, uint64_t size, uint8_t* out);

/* Suffix array of T[0..length-1] under "proper prefix sorts first"
 * (test/SaisTest.cpp:45-53); sa is a host buffer of `length` words.  Test hook for the
 * property the reference checks in test/SaisTest.cpp:55-70. */
int bwtc_hip_suffix_array(bwtc_hip_ctx* ctx, const uint8_t* T, uint32_t length, uint32_t* sa);

/* Unit-test hooks for the sort/scan primitives (host buffers, in place). */
int bwtc_hip_test_sort_u32(bwtc_hip_ctx* ctx, uint32_t* keys, uint32_t* vals, uint64_t n, int nbits);
int bwtc_hip_test_sort_u64(bwtc_hip_ctx* ctx, uint64_t* keys, uint32_t* vals, uint64_t n, int nbits);
int bwtc_hip_test_scan_u32(bwtc_hip_ctx* ctx, uint32_t* data, uint64_t n);
/* Test hook: k range-coder chains over w-elements (w = bit << 15 | probability of the coded bit), chain j = elements
 * [bounds[j], bounds[j+1]), on the GPU lane engine (mode 0; BitEncoder, BitCoders.cpp:59-113, one lane per chain) or by the
 * host's scalar loop (mode 1); bytes of chain j = out[offsets[j] .. offsets[j+1]). */
int bwtc_hip_test_gpu_lanes(bwtc_hip_ctx* ctx, const uint16_t* w, uint64_t n, const uint64_t* bounds, uint32_t k, int mode,
                            uint8_t* out, uint64_t out_cap, uint64_t* offsets);

/* ---- pair-replacing pre-stage, `--prepr p...` ------------------------------------------
 * Replaces preprocessors/PairReplacer.cpp (analyseData :53-63, decideReplacements :402-484,
 * writeReplacedVersion :330-400), Grammar.cpp (writeGrammar :309-320, readGrammar :198-307),
 * Precompressor::precompress (Precompressor.cpp:62-121) and Postprocessor::uncompress (Postprocessor.cpp:112-133).
 * A grammar object is what a PrecompressorBlock carries (PrecompressorBlock.hpp:70): it is updated by every
 * round over the block and written into the block's header (PrecompressorBlock.cpp:64-90). */
typedef struct bwtc_hip_grammar bwtc_hip_grammar;
bwtc_hip_grammar* bwtc_hip_grammar_create(void);                       /* Grammar::Grammar: no rules */
void     bwtc_hip_grammar_destroy(bwtc_hip_grammar* g);
uint32_t bwtc_hip_grammar_rules(const bwtc_hip_grammar* g);            /* Grammar::numberOfRules          */
uint32_t bwtc_hip_grammar_special_symbols(const bwtc_hip_grammar* g);  /* Grammar::numberOfSpecialSymbols */
int      bwtc_hip_grammar_is_special(const bwtc_hip_grammar* g, unsigned symbol);
/* Grammar::writeGrammar into out (cap bytes): 0, *bytes = size; -1 when it does not fit */
int      bwtc_hip_grammar_write(const bwtc_hip_grammar* g, uint8_t* out, uint64_t cap, uint64_t* bytes);
/* Grammar::readGrammar into an EMPTY grammar: 0, *consumed = bytes read; -1 when the input is cut short */
int      bwtc_hip_grammar_read(bwtc_hip_grammar* g, const uint8_t* in, uint64_t n, uint64_t* consumed);
/* One PairReplacer (analyseData + decideReplacements + writeReplacedVersion) over a device-resident text of
 * 3 <= n < 2^31 bytes: d_src -> d_dst (room for 2 n bytes; must not overlap), *n_out = the new length,
 * *replaced = pairs replaced (0: the text is copied unchanged). */
int      bwtc_hip_pair_replace_device(bwtc_hip_ctx* ctx, bwtc_hip_grammar* g, const uint8_t* d_src, uint64_t n,
                                      uint8_t* d_dst, uint64_t* n_out, uint32_t* replaced);
/* Precompressor::precompress over a host block, in place: one round per letter of `options` ('p'), stopped by the
 * first round that does not shorten the block; *n_out = the new length.  Blocks shorter than three bytes are left
 * alone (the reference asserts length > 2). */
int      bwtc_hip_precompress(bwtc_hip_ctx* ctx, bwtc_hip_grammar* g, const char* options, uint8_t* block, uint64_t n,
                              uint64_t* n_out);
/* the same with both sweeps on the calling thread (no device; the CPU test suite, and the mirror's blocks below 64 KiB) */
int      bwtc_hip_host_precompress(bwtc_hip_grammar* g, const char* options, uint8_t* block, uint64_t n, uint64_t* n_out);
/* Postprocessor::uncompress: data (n bytes) expanded into out (cap bytes); -1 when it does not fit */
int      bwtc_hip_postprocess(const bwtc_hip_grammar* g, const uint8_t* data, uint64_t n, uint8_t* out, uint64_t cap,
                              uint64_t* n_out);

#ifdef __cplusplus
}
#endif
#endif
