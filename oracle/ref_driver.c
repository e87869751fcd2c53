/*
 * oracle/ref_driver.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin C driver around the reference's own, unmodified libdivsufsort sources
 * (/root/reference/bwtransforms/{divsufsort,sssort,trsort}.c), which are compiled
 * where they lie by oracle/Makefile into oracle/_ref/libbwtc_ref.so.  No reference
 * source is copied into this repository.
 *
 * The reference's C++ wrapper around divbwtf (bwtransforms/BWTransform.cpp:52-64 and
 * bwtransforms/Divsufsorter.hpp:60-65) cannot be compiled here because it pulls in
 * Boost through globaldefs.hpp; the six lines it contributes (reverse, plant the 0
 * sentinel, call divbwtf in place on size+1 bytes, fill the end-of-block hole, restore
 * the borrowed byte) are restated below with the line each one follows.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* prototype as declared in /root/reference/bwtransforms/divsufsort.h (bwtc-modified) */
int32_t divbwtf(const uint8_t *T, uint8_t *U, int32_t *A, int32_t n,
                unsigned *LFpowers, unsigned nLFpowers, unsigned freqs[256]);

/* Raw transform: bwtransforms/Divsufsorter.hpp:60-65.  T[0..n-1] in place. */
int ref_bwt_raw(uint8_t *T, uint32_t n, uint32_t *lf, uint32_t n_lf, uint32_t *freqs)
{
    return (int)divbwtf(T, T, NULL, (int32_t)n, lf, n_lf, freqs);
}

/* BWTBlock::prepareLFpowers (BWTBlock.cpp:104-108) after the clamp of
 * BWTManager::setStartingPoints (bwtransforms/BWTManager.cpp:60-64). */
uint32_t ref_n_lf(uint32_t size, uint32_t starting_points)
{
    if (starting_points < 1) starting_points = 1;
    else if (starting_points > 256) starting_points = 256;
    if (size <= 256 || starting_points == 0) return 1;
    return starting_points;
}

/* Block-level transform: BWTransform::doTransform(BWTBlock&, freqs),
 * bwtransforms/BWTransform.cpp:52-64.  block must have size+1 bytes allocated. */
int ref_bwt_block(uint8_t *block, uint32_t size, uint32_t starting_points,
                  uint32_t *lf, uint32_t *n_lf_out, uint32_t *freqs)
{
    uint32_t n_lf = ref_n_lf(size, starting_points);
    uint32_t i;
    uint8_t next;
    int r;
    for (i = 0; i < size / 2; ++i) {                 /* :53 std::reverse */
        uint8_t t = block[i]; block[i] = block[size - 1 - i]; block[size - 1 - i] = t;
    }
    next = block[size];                              /* :54 */
    block[size] = 0;                                 /* :55 */
    r = ref_bwt_raw(block, size + 1, lf, n_lf, freqs); /* :57 */
    block[lf[0]] = block[size];                      /* :60 */
    block[size] = next;                              /* :63 */
    *n_lf_out = n_lf;
    return r;
}
