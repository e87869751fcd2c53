/*
 * oracle/bwtc_oracle.h -- CPU restatement of the bwtc BWTManager/BWTransform hot path
 * and of the 'H' (Huffman) entropy coder + stream framing around it.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bwtc_amd/ links, imports or executes this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and
 * only as the checker.  Pinned against: the reference's own libdivsufsort compiled from
 * /root/reference (oracle/_ref, see oracle/Makefile), the golden vectors recorded from the
 * reference in SURVEY.md 8c (tests/golden/), and the known-answer asserts of the
 * reference's test/UtilsTest.cpp.
 */
#ifndef BWTC_ORACLE_H
#define BWTC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- BWT ---------------------------------------- */
uint32_t orc_n_lf(uint32_t size, uint32_t starting_points);
int  orc_suffix_array(const uint8_t *T, uint32_t n, uint32_t *SA);
int  orc_bwt_raw(uint8_t *T, uint32_t n, uint32_t *lf, uint32_t n_lf, uint32_t *freqs);
int  orc_bwt_block(uint8_t *block, uint32_t size, uint32_t starting_points,
                   uint32_t *lf, uint32_t *n_lf_out, uint32_t *freqs);
int  orc_inverse_bwt_block(uint8_t *block, uint32_t size, const uint32_t *lf, uint32_t n_lf);

/* ---- utils (Utils.cpp / Utils.hpp) ----------------------------------------- */
uint64_t orc_pack_integer(uint64_t v, int *bytes_needed);
uint64_t orc_run_frequencies(uint64_t *run_freqs, const uint8_t *src, size_t len);
uint64_t orc_run_frequencies_store(uint64_t *run_freqs, uint8_t *runseq, uint32_t *runlen,
                                   const uint8_t *src, size_t len);
/* utils::calculateRunsAndCharacters: returns total runs; run_freqs[256] incremented; the run
 * length distribution comes back as up to cap (length, count) pairs ascending by length
 * (*n_pairs = number written, or cap+1.. if it did not fit) */
uint64_t orc_runs_and_characters(uint64_t *run_freqs, const uint8_t *src, size_t len,
                                 uint32_t *dist_len, uint32_t *dist_cnt, size_t cap, size_t *n_pairs);
int  orc_huffman_lengths(const uint64_t freqs[256], uint32_t clen[256]);
void orc_huffman_codes(const uint32_t clen[256], uint32_t code[256]);
/* bit strings are returned one bit per byte (0/1) for easy comparison with the
 * reference's vector<bool> known answers; return value = number of bits */
size_t orc_binary_code(size_t n, size_t lo, size_t hi, uint8_t *bits);
size_t orc_binary_interpolative_code(const uint32_t *list, size_t count, size_t max_value,
                                     uint8_t *bits);
size_t orc_serialize_shape(const uint32_t clen[256], uint8_t *bits);
uint32_t orc_sections(const uint32_t freqs[256], uint32_t section_len[256]);

/* ---- 'H' coder + framing (HuffmanCoders.cpp, BWTBlock.cpp, Compressor.cpp) -- */
size_t orc_write_bwtblock_header(const uint32_t *lf, uint32_t n_lf, uint8_t *out);
/* encodes an already-transformed block; returns bytes written incl. the 6 length bytes */
size_t orc_huffman_encode_block(const uint8_t *bwt, uint32_t size, const uint32_t *lf,
                                uint32_t n_lf, const uint32_t freqs[256], uint8_t *out,
                                size_t out_cap);
/* full stream: 'H' + pblocks + 0x00.  block_size = bwt block size (Compressor.cpp:77-79) */
size_t orc_compress_H(const uint8_t *in, size_t size, size_t block_size,
                      uint32_t starting_points, uint8_t *out, size_t out_cap);
/* returns decoded size, or (size_t)-1 on malformed input */
size_t orc_decompress_H(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap);
size_t orc_compress_bound(size_t size);

/* ---- 'B' wavelet coder (oracle/wavelet_oracle.c; WaveletCoders.cpp, WaveletTree.hpp) ---- */
size_t orc_wavelet_encode_block(const uint8_t *bwt, uint32_t size, const uint32_t *lf, uint32_t n_lf,
                                const uint32_t freqs[256], uint8_t *out, size_t out_cap);
size_t orc_compress_B(const uint8_t *in, size_t size, size_t block_size, uint32_t starting_points,
                      uint8_t *out, size_t out_cap);
/* the same with the main model chosen by the coder letter (giveProbabilityModel,
 * probmodels/ProbabilityModel.cpp:47-76): 'B' (and any other letter) FSM8, 'b' FSM<6,
 * EvenIntervalPredictor<4>>, 'u' EvenIntervalPredictor<4>.  'm' / 'M' return 0: the
 * reference's SimpleMarkov indexes one past its history table, its output is undefined.
 * No reference-produced vector exists for 'b' and 'u' (parity unpinned for those letters). */
size_t orc_wavelet_encode_block_with(char coder, const uint8_t *bwt, uint32_t size, const uint32_t *lf,
                                     uint32_t n_lf, const uint32_t freqs[256], uint8_t *out, size_t out_cap);
size_t orc_compress_wavelet(char coder, const uint8_t *in, size_t size, size_t block_size,
                            uint32_t starting_points, uint8_t *out, size_t out_cap);

/* Decompressor::decompress over a stream made by a wavelet coder ('B', 'b', 'u'): a literal
 * restatement of WaveletDecoder::decodeBlock, WaveletTree::readShape / decodeTreeBF / message and
 * BitDecoder (wavelet_oracle.c, second half).  Returns the decoded size or (size_t)-1 on
 * malformed input; bwt_out (may be NULL) receives the transformed bytes of the blocks. */
size_t orc_decompress_wavelet(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap,
                              uint8_t *bwt_out);

/* ---- pair-replacing pre-stage, `--prepr p...` (oracle/prepr_oracle.cpp; preprocessors/PairReplacer.cpp,
 * Grammar.cpp, FrequencyTable.cpp, Precompressor.cpp, Postprocessor.cpp) ---- */
typedef struct orc_grammar orc_grammar;
orc_grammar *orc_grammar_new(void);
void     orc_grammar_free(orc_grammar *g);
uint32_t orc_grammar_rules(const orc_grammar *g);
uint32_t orc_grammar_specials(const orc_grammar *g);
int      orc_grammar_is_special(const orc_grammar *g, unsigned symbol);
void     orc_grammar_frequencies(const orc_grammar *g, uint64_t f[256]);
size_t   orc_grammar_write(const orc_grammar *g, uint8_t *out, size_t cap);       /* Grammar::writeGrammar */
size_t   orc_grammar_read(orc_grammar *g, const uint8_t *in, size_t n);           /* Grammar::readGrammar: bytes consumed */
size_t   orc_grammar_span(const uint8_t *in, size_t n);                           /* bytes a serialised grammar occupies */
void     orc_grammar_add_rule(orc_grammar *g, unsigned variable, unsigned first, unsigned second);
size_t   orc_grammar_expand_alphabet(orc_grammar *g, const uint8_t *freed, size_t n_freed, const uint8_t *specials,
                                     size_t n_specials, uint16_t *next_pairs);
/* PairReplacer::analyseData: byte counts (grammar's included) and the 65536 pair counts */
void     orc_pair_statistics(const orc_grammar *g, const uint8_t *data, size_t n, uint64_t freqs[256], uint64_t *pair_freqs);
/* analyseData + decideReplacements + writeReplacedVersion: returns the replacements made; out needs 2 n bytes */
size_t   orc_pair_replace_round(orc_grammar *g, const uint8_t *data, size_t n, uint8_t *out, size_t *out_n);
size_t   orc_precompress(orc_grammar *g, const char *options, uint8_t *data, size_t n);   /* in place, new length */
size_t   orc_postprocess(const orc_grammar *g, const uint8_t *data, size_t n, uint8_t *out, size_t cap);
size_t   orc_postprocess_bytes(const uint8_t *gram, size_t gram_n, const uint8_t *data, size_t n, uint8_t *out, size_t cap);
/* Compressor::compress with the pre-stage: mem_limit in bytes as Compressor.cpp:77-97 uses it */
size_t   orc_compress_prepr(char coder, const char *options, const uint8_t *in, size_t size, uint64_t mem_limit,
                            uint32_t starting_points, uint8_t *out, size_t out_cap);
void    *orc_wavelet_stream_new(char coder);
size_t   orc_wavelet_stream_encode_block(void *ws, const uint8_t *bwt, uint32_t size, const uint32_t *lf,
                                         uint32_t n_lf, const uint32_t freqs[256], uint8_t *out);
void     orc_wavelet_stream_free(void *ws);

/* restatements of what test/WaveletTest.cpp pins with known answers (see wavelet_oracle.c) */
void   orc_min_heap_order(const int *values, const uint64_t *weights, size_t n, int *out);
size_t orc_create_huffman_shape(const uint64_t run_freqs[256], uint32_t *symbols, uint32_t *depths,
                                uint8_t *code_bits, uint32_t code_len[256]);
size_t orc_gamma_code(size_t integer, uint8_t *bits);
size_t orc_fixed_integer_code(uint32_t x, uint32_t w, uint8_t *bits);
void   orc_wavelet_symbol_codes(const uint64_t run_freqs[256], uint8_t *code_bits, uint32_t code_len[256]);

#ifdef __cplusplus
}
#endif
#endif
