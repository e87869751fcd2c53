/*
 * oracle/bwtc_oracle.c -- CPU restatement of the bwtc hot path.  See bwtc_oracle.h.
 *
 * TEST INFRASTRUCTURE ONLY: the product (bwtc_amd/) never links, imports or runs this.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * The suffix sorter is an independent plain prefix-doubling sorter: the reference's sorter
 * (libdivsufsort, 2400 lines) is not restated -- it is compiled as-is into oracle/_ref and
 * this file is pinned against it (tests/test_oracle.py), the same way the reference's own
 * test/LFpowersTest.cpp:49-181 checks its transform against an independently built SA.
 */
#include "bwtc_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* BWT                                                                        */
/* ------------------------------------------------------------------------- */

/* BWTManager::setStartingPoints clamp (bwtransforms/BWTManager.cpp:60-64) followed by
 * BWTBlock::prepareLFpowers (BWTBlock.cpp:104-108). */
uint32_t orc_n_lf(uint32_t size, uint32_t starting_points)
{
    if (starting_points < 1) starting_points = 1;
    else if (starting_points > 256) starting_points = 256;
    if (size <= 256) return 1;
    return starting_points;
}

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

/* Suffix array of T[0..n-1] under the order the reference's tests spell out
 * (test/SaisTest.cpp:45-53): plain lexicographic, a proper prefix sorts first.
 * Prefix doubling: after a round with offset h every group of equal rank shares its
 * first 2h characters.  key2 = 0 for the one suffix whose h-successor is the (empty)
 * suffix n, rank+1 otherwise.  O(n log n) per round; meant for test sizes. */
int orc_suffix_array(const uint8_t *T, uint32_t n, uint32_t *SA)
{
    uint32_t *rank, *nrank, i, h;
    uint64_t *tmp;
    uint32_t cnt[257];
    if (n == 0) return 0;
    rank = (uint32_t *)malloc((size_t)n * 4);
    nrank = (uint32_t *)malloc((size_t)n * 4);
    tmp = (uint64_t *)malloc((size_t)n * 8);
    if (!rank || !nrank || !tmp) { free(rank); free(nrank); free(tmp); return -2; }
    memset(cnt, 0, sizeof cnt);
    for (i = 0; i < n; ++i) cnt[T[i] + 1]++;
    for (i = 0; i < 256; ++i) cnt[i + 1] += cnt[i];
    {
        uint32_t start[256];
        memcpy(start, cnt, sizeof start);
        for (i = 0; i < n; ++i) { rank[i] = start[T[i]]; SA[cnt[T[i]]++] = i; }
    }
    for (h = 1;; h *= 2) {
        uint32_t j = 0;
        int done = 1;
        while (j < n) {
            uint32_t g = rank[SA[j]], e = j + 1, k, head;
            while (e < n && rank[SA[e]] == g) ++e;
            if (e - j == 1) { nrank[SA[j]] = j; j = e; continue; }
            for (k = j; k < e; ++k) {
                uint32_t s = SA[k];
                uint64_t key2 = ((uint64_t)s + h < n) ? (uint64_t)rank[s + h] + 1 : 0;
                tmp[k] = (key2 << 32) | s;
            }
            qsort(tmp + j, e - j, 8, cmp_u64);
            head = j;
            for (k = j; k < e; ++k) {
                if (k > j && (tmp[k] >> 32) != (tmp[k - 1] >> 32)) head = k;
                SA[k] = (uint32_t)tmp[k];
                nrank[SA[k]] = head;
                if (k > j && head != k) done = 0;
            }
            j = e;
        }
        { uint32_t *t = rank; rank = nrank; nrank = t; }
        if (done || h >= n) break;
    }
    free(rank); free(nrank); free(tmp);
    return 0;
}

/* Raw transform = what Divsufsorter::doTransform(byte*,uint32,LF,freqs)
 * (bwtransforms/Divsufsorter.hpp:60-65 -> divbwtf, bwtransforms/divsufsort.c:480-522)
 * leaves behind: U[i] = T[SA[i]-1] for every i except pidx (:506-512, the slot of
 * suffix 0 is left untouched), freqs[U[i]] incremented for those i, LF[0] = pidx
 * (:498-503) and LF[k] = rank of suffix n - k*(n/nLF) (construct_BWT, :337-338,350,381,390;
 * equivalently test/LFpowersTest.cpp:116-135).  n <= 1 edge case: :489. */
int orc_bwt_raw(uint8_t *T, uint32_t n, uint32_t *lf, uint32_t n_lf, uint32_t *freqs)
{
    uint32_t *SA, *ISA = NULL, i, pidx = 0, x;
    uint8_t *U;
    if (!T || !lf || n_lf == 0) return -1;
    if (n <= 1) return (int)n;
    SA = (uint32_t *)malloc((size_t)n * 4);
    U = (uint8_t *)malloc(n);
    if (n_lf > 1) ISA = (uint32_t *)malloc((size_t)n * 4);
    if (!SA || !U || (n_lf > 1 && !ISA)) { free(SA); free(U); free(ISA); return -2; }
    if (orc_suffix_array(T, n, SA) != 0) { free(SA); free(U); free(ISA); return -2; }
    for (i = 0; i < n; ++i) {
        if (SA[i] == 0) { pidx = i; U[i] = T[i]; }
        else U[i] = T[SA[i] - 1];
        if (ISA) ISA[SA[i]] = i;
    }
    for (i = 0; i < n; ++i) {
        if (i == pidx) continue;
        T[i] = U[i];
        if (freqs) ++freqs[U[i]];
    }
    lf[0] = pidx;
    x = n / n_lf;
    for (i = 1; i < n_lf; ++i) lf[i] = ISA[n - i * x];
    free(SA); free(U); free(ISA);
    return (int)pidx;
}

/* BWTManager::doTransform(block,freqs) (bwtransforms/BWTManager.cpp:53-58) ->
 * BWTransform::doTransform(BWTBlock&,freqs) (bwtransforms/BWTransform.cpp:52-64).
 * block needs size+1 bytes; block[size] is preserved. */
int orc_bwt_block(uint8_t *block, uint32_t size, uint32_t starting_points,
                  uint32_t *lf, uint32_t *n_lf_out, uint32_t *freqs)
{
    uint32_t n_lf = orc_n_lf(size, starting_points), i;
    uint8_t next;
    int r;
    for (i = 0; i < size / 2; ++i) {                       /* :53 */
        uint8_t t = block[i]; block[i] = block[size - 1 - i]; block[size - 1 - i] = t;
    }
    next = block[size];                                    /* :54 */
    block[size] = 0;                                       /* :55 */
    r = orc_bwt_raw(block, size + 1, lf, n_lf, freqs);     /* :57 */
    if (r < 0) return r;
    if (size == 0) lf[0] = 0;
    block[lf[0]] = block[size];                            /* :60 */
    block[size] = next;                                    /* :63 */
    if (n_lf_out) *n_lf_out = n_lf;
    return 0;
}

/* Inverse: InverseBWTransform::doTransform(BWTBlock&) (bwtransforms/InverseBWT.cpp:47-51)
 * puts data[LF[0]] back at *end and inverts the N = size+1 row transform with
 * eob = LF[0].  The LF walk is the plain one of test/LFpowersTest.cpp:85-105 (the EOB
 * counts as the smallest symbol and maps to row 0); the reference's production inverse
 * (MtlSaInverseBWT.cpp) computes the same permutation two characters at a time.
 * Also checks every LF power against the walk; returns -3 on an inconsistent one. */
int orc_inverse_bwt_block(uint8_t *block, uint32_t size, const uint32_t *lf, uint32_t n_lf)
{
    uint32_t n = size + 1, i, p, x, pidx;
    uint32_t count[257];
    uint32_t *LF;
    uint8_t *L, *out;
    int rc = 0;
    if (size == 0) return 0;
    pidx = lf[0];
    if (pidx > size) return -1;
    L = (uint8_t *)malloc(n);
    out = (uint8_t *)malloc(size);
    LF = (uint32_t *)malloc((size_t)n * 4);
    if (!L || !out || !LF) { free(L); free(out); free(LF); return -2; }
    memcpy(L, block, size);
    L[size] = block[pidx];                                 /* InverseBWT.cpp:49 */
    memset(count, 0, sizeof count);
    count[0] = 1;
    for (i = 0; i < n; ++i) if (i != pidx) count[L[i] + 1]++;
    for (i = 0; i < 256; ++i) count[i + 1] += count[i];
    for (i = 0; i < n; ++i) if (i != pidx) LF[i] = count[L[i]]++;
    LF[pidx] = 0;
    x = n / n_lf;
    /* row 0 is suffix n-1 (the sentinel alone); LF^j(pidx) = row of suffix n-j */
    p = 0;
    for (i = 0; i < size; ++i) {
        uint32_t j = i + 1;              /* p is the row of suffix n-j */
        if (n_lf > 1 && j % x == 0 && j / x < n_lf && lf[j / x] != p) rc = -3;
        if (p == pidx) { rc = -4; break; }
        out[i] = L[p];
        p = LF[p];
    }
    if (rc == 0 && p != pidx) rc = -4;
    if (rc == 0 || rc == -3) memcpy(block, out, size);
    free(L); free(out); free(LF);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Utils                                                                      */
/* ------------------------------------------------------------------------- */

/* utils::packInteger, Utils.cpp:89-101 */
uint64_t orc_pack_integer(uint64_t integer, int *bytes_needed)
{
    uint64_t result = 0;
    int i;
    for (i = 0; integer; ++i) {
        result |= ((integer & 0x7F) << (i * 8));
        integer >>= 7;
        if (integer) result |= ((uint64_t)0x80 << (i * 8));
    }
    if (i == 0) ++i;
    *bytes_needed = i;
    return result;
}

/* writePackedInteger, HuffmanCoders.cpp:316-322 / Utils.cpp:475-481 */
static size_t put_packed(uint64_t v, uint8_t *out)
{
    int bytes, i;
    uint64_t p = orc_pack_integer(v, &bytes);
    for (i = 0; i < bytes; ++i) { out[i] = (uint8_t)(p & 0xFF); p >>= 8; }
    return (size_t)bytes;
}

/* utils::calculateRunFrequencies, Utils.cpp:116-126 */
uint64_t orc_run_frequencies(uint64_t *run_freqs, const uint8_t *src, size_t len)
{
    uint64_t runs = 0;
    size_t i = 0;
    while (i < len) {
        size_t j = i + 1;
        while (j < len && src[j] == src[i]) ++j;
        ++run_freqs[src[i]];
        ++runs;
        i = j;
    }
    return runs;
}

/* utils::calculateRunFrequenciesAndStoreRuns, Utils.cpp:150-170 */
uint64_t orc_run_frequencies_store(uint64_t *run_freqs, uint8_t *runseq, uint32_t *runlen,
                                   const uint8_t *src, size_t len)
{
    uint64_t runs = 0;
    size_t i = 0;
    while (i < len) {
        size_t j = i + 1;
        while (j < len && src[j] == src[i]) ++j;
        ++run_freqs[src[i]];
        runseq[runs] = src[i];
        runlen[runs] = (uint32_t)(j - i);
        ++runs;
        i = j;
    }
    return runs;
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* utils::calculateRunsAndCharacters, Utils.cpp:128-147: maximal runs of the section, runs per
 * symbol, and the std::map<run length, count> the WaveletTree constructor consumes
 * (WaveletTree.hpp:294-308), here as pairs in the map's iteration order. */
uint64_t orc_runs_and_characters(uint64_t *run_freqs, const uint8_t *src, size_t len,
                                 uint32_t *dist_len, uint32_t *dist_cnt, size_t cap, size_t *n_pairs)
{
    uint64_t runs = 0;
    size_t i = 0, k = 0, np = 0;
    uint32_t *lens = (uint32_t *)malloc((len ? len : 1) * sizeof(uint32_t));
    while (i < len) {
        size_t j = i + 1;
        while (j < len && src[j] == src[i]) ++j;
        ++run_freqs[src[i]];
        lens[runs++] = (uint32_t)(j - i);
        i = j;
    }
    qsort(lens, runs, sizeof(uint32_t), cmp_u32);
    while (k < runs) {
        size_t j = k;
        while (j < runs && lens[j] == lens[k]) ++j;
        if (np < cap) { dist_len[np] = lens[k]; dist_cnt[np] = (uint32_t)(j - k); }
        ++np;
        k = j;
    }
    free(lens);
    *n_pairs = np;
    return runs;
}

typedef struct { uint64_t w; uint32_t sym; } wsym;
static int cmp_wsym(const void *a, const void *b)
{
    const wsym *x = (const wsym *)a, *y = (const wsym *)b;
    if (x->w != y->w) return (x->w > y->w) - (x->w < y->w);
    return (x->sym > y->sym) - (x->sym < y->sym);
}

/* utils::calculateHuffmanLengths(codeLengths, freqs, 256) (Utils.cpp:408-418) +
 * utils::calculateCodeLengths (Utils.cpp:422-473): Moffat-Katajainen in-place
 * minimum-redundancy code lengths over the (freq, symbol)-sorted nonzero entries.
 * Returns the number of symbols with a code. */
int orc_huffman_lengths(const uint64_t freqs[256], uint32_t clen[256])
{
    wsym v[256];
    uint64_t f[256];
    size_t n = 0, i, s = 0, r = 0, t;
    memset(clen, 0, 256 * sizeof(uint32_t));
    for (i = 0; i < 256; ++i) if (freqs[i]) { v[n].w = freqs[i]; v[n].sym = (uint32_t)i; ++n; }
    if (n == 0) return 0;
    if (n == 1) { clen[v[0].sym] = 1; return 1; }                 /* :426-429 */
    qsort(v, n, sizeof(wsym), cmp_wsym);                          /* :433 */
    for (i = 0; i < n; ++i) f[i] = v[i].w;
    for (t = 0; t < n - 1; ++t) {                                 /* :440-455 */
        if (s >= n || (r < t && f[r] < f[s])) { f[t] = f[r]; f[r++] = t; }
        else f[t] = f[s++];
        if (s >= n || (r < t && f[r] < f[s])) { f[t] += f[r]; f[r++] = t; }
        else f[t] += f[s++];
    }
    f[n - 2] = 0;                                                 /* :457-460 */
    for (long k = (long)n - 3; k >= 0; --k) f[k] = f[f[k]] + 1;
    {
        long a = 1, u = 0, x = (long)n - 1, tt = (long)n - 2;     /* :461-470 */
        uint64_t depth = 0;
        while (a > 0) {
            while (tt >= 0 && f[tt] == depth) { ++u; --tt; }
            while (a > u) { f[x] = depth; --x; --a; }
            a = 2 * u;
            ++depth;
            u = 0;
        }
    }
    for (i = 0; i < n; ++i) clen[v[i].sym] = (uint32_t)f[i];
    return (int)n;
}

/* utils::computeHuffmanCodes, Utils.cpp:180-202 */
void orc_huffman_codes(const uint32_t clen[256], uint32_t code[256])
{
    uint32_t lengths_count[256], start[256], first = 0;
    int k, max_len = 0;
    memset(lengths_count, 0, sizeof lengths_count);
    for (k = 0; k < 256; ++k) ++lengths_count[clen[k]];
    for (k = 0; k < 256; ++k) if (lengths_count[k] > 0) max_len = k;
    memset(start, 0, sizeof start);
    for (k = max_len; k >= 0; --k) {
        start[k] = first;
        first = (first + lengths_count[k]) >> 1;
    }
    memset(code, 0, 256 * sizeof(uint32_t));
    for (k = 0; k < 256; ++k) if (clen[k] > 0) code[k] = start[clen[k]]++;
}

static unsigned log_floor(size_t n) { unsigned l = 0; while (n > 1) { n >>= 1; ++l; } return l; }
/* utils::logCeiling, Utils.hpp:122-126 */
static unsigned log_ceiling(size_t n)
{
    unsigned l = log_floor(n);
    return (((size_t)1 << l) < n) ? l + 1 : l;
}
static size_t push_bits(uint8_t *bits, size_t pos, uint64_t v, unsigned nbits)
{
    unsigned i;
    for (i = 1; i <= nbits; ++i) bits[pos++] = (uint8_t)((v >> (nbits - i)) & 1);
    return pos;
}

/* utils::binaryCode, Utils.hpp:239-252 */
static size_t binary_code(size_t n, size_t lo, size_t hi, uint8_t *bits, size_t pos)
{
    size_t range_len = hi - lo + 1, short_cw, long_cw2;
    unsigned code_len;
    if (range_len == 1) return pos;
    code_len = log_ceiling(range_len);
    short_cw = ((size_t)1 << code_len) - range_len;
    long_cw2 = (range_len - short_cw) / 2;
    if (n - lo < long_cw2) return push_bits(bits, pos, n - lo, code_len);
    if (n - lo < long_cw2 + short_cw) return push_bits(bits, pos, n - lo, code_len - 1);
    return push_bits(bits, pos, n - lo - short_cw, code_len);
}
size_t orc_binary_code(size_t n, size_t lo, size_t hi, uint8_t *bits)
{
    return binary_code(n, lo, hi, bits, 0);
}

/* utils::binaryInterpolativeCode, Utils.hpp:263-281 (indices are inclusive) */
static size_t bic(const uint32_t *list, size_t begin, size_t end, size_t lo, size_t hi,
                  uint8_t *bits, size_t pos)
{
    size_t h, half;
    if (begin > end) return pos;
    if (end - begin == hi - lo) return pos;
    if (begin == end) return binary_code(list[begin], lo, hi, bits, pos);
    h = (end - begin) / 2;
    half = begin + h;
    pos = binary_code(list[half], lo + h, hi + half - end, bits, pos);
    if (half > begin) pos = bic(list, begin, half - 1, lo, list[half] - 1, bits, pos);
    return bic(list, half + 1, end, list[half] + 1, hi, bits, pos);
}
/* Utils.hpp:292-297 */
size_t orc_binary_interpolative_code(const uint32_t *list, size_t count, size_t max_value,
                                     uint8_t *bits)
{
    return bic(list, 0, count - 1, 0, max_value, bits, 0);
}

/* HuffmanEncoder::serializeShape, HuffmanCoders.cpp:63-86.  bits: one bit per byte,
 * capacity >= 8192. */
size_t orc_serialize_shape(const uint32_t clen[256], uint8_t *bits)
{
    uint32_t symbols[256];
    size_t ns = 0, pos = 0, i, max_len = 0;
    int bytes_in_longest;
    uint64_t packed;
    for (i = 0; i < 256; ++i) if (clen[i] > 0) {
        symbols[ns++] = (uint32_t)i;
        if (clen[i] > max_len) max_len = clen[i];
    }
    pos = push_bits(bits, pos, symbols[ns - 1], 8);
    pos = push_bits(bits, pos, ns, 8);
    packed = orc_pack_integer(max_len, &bytes_in_longest);
    pos = push_bits(bits, pos, packed, (unsigned)bytes_in_longest * 8);
    pos = bic(symbols, 0, ns - 1, 0, symbols[ns - 1], bits, pos);
    for (i = 0; i < ns; ++i) {                     /* utils::unaryCode, Utils.hpp:398-402 */
        size_t n = max_len - clen[symbols[i]] + 1;
        while (n-- > 1) bits[pos++] = 0;
        bits[pos++] = 1;
    }
    return pos;
}

/* Section heuristic of HuffmanEncoder::writeBlockHeader (HuffmanCoders.cpp:282-296),
 * identical in WaveletEncoder::writeBlockHeader (WaveletCoders.cpp:184-198). */
uint32_t orc_sections(const uint32_t freqs[256], uint32_t section_len[256])
{
    uint32_t n = 0, i;
    uint64_t sum = 0;
    for (i = 0; i < 256; ++i) {
        sum += freqs[i];
        if (sum >= 10000) { section_len[n++] = (uint32_t)sum; sum = 0; }
    }
    if (sum != 0) {
        if (n > 0) section_len[n - 1] += (uint32_t)sum;
        else section_len[n++] = (uint32_t)sum;
    }
    return n;
}

/* ------------------------------------------------------------------------- */
/* 'H' coder + framing                                                        */
/* ------------------------------------------------------------------------- */

/* BWTBlock::writeHeader, BWTBlock.cpp:61-86 */
size_t orc_write_bwtblock_header(const uint32_t *lf, uint32_t n_lf, uint8_t *out)
{
    size_t bytes = 1;
    uint8_t s = (uint8_t)(n_lf - 1);
    int bits_left = 8, j;
    uint32_t i;
    out[0] = s;
    for (i = 0; i < n_lf; ++i) {
        for (j = 30; j >= 0; --j) {
            s = (uint8_t)((s << 1) | ((lf[i] >> j) & 1));
            if (--bits_left == 0) { out[bytes++] = s; bits_left = 8; }
        }
    }
    if (bits_left < 8) out[bytes++] = (uint8_t)(s << bits_left);
    return bytes;
}

typedef struct { uint8_t *p; size_t n; uint64_t buf; int nb; } bitw;
static void bw_put(bitw *w, uint64_t v, int nbits)   /* HuffmanCoders.cpp:200-212 */
{
    while (w->nb + nbits > 64) { w->nb -= 8; w->p[w->n++] = (uint8_t)(w->buf >> w->nb); }
    w->buf = (nbits >= 64) ? v : ((w->buf << nbits) | v);
    w->nb += nbits;
}
static void bw_flush(bitw *w)                        /* HuffmanCoders.cpp:214-226 */
{
    while (w->nb >= 8) { w->nb -= 8; w->p[w->n++] = (uint8_t)(w->buf >> w->nb); }
    if (w->nb > 0) { w->p[w->n++] = (uint8_t)((w->buf << (8 - w->nb)) & 0xFF); }
    w->nb = 0; w->buf = 0;
}

/* HuffmanEncoder::transformAndEncode minus the transform (HuffmanCoders.cpp:51-61):
 * writeBlockHeader (:271-313), encodeData (:119-257), finishBlock (:259-261 ->
 * Streams.cpp:106-117 big-endian 48-bit length that excludes its own 6 bytes). */
size_t orc_huffman_encode_block(const uint8_t *bwt, uint32_t size, const uint32_t *lf,
                                uint32_t n_lf, const uint32_t freqs[256], uint8_t *out,
                                size_t out_cap)
{
    uint32_t sect[256], ns, si;
    size_t pos = 6, beg = 0, i;
    uint8_t *runseq = (uint8_t *)malloc(size ? size : 1);
    uint32_t *runlen = (uint32_t *)malloc((size_t)(size ? size : 1) * 4);
    uint8_t *shape = (uint8_t *)malloc(16384);
    uint64_t length;
    (void)out_cap;
    if (!runseq || !runlen || !shape) { free(runseq); free(runlen); free(shape); return 0; }
    memset(out, 0, 6);
    pos += orc_write_bwtblock_header(lf, n_lf, out + pos);
    ns = orc_sections(freqs, sect);
    out[pos++] = (uint8_t)(ns == 256 ? 0 : ns);
    for (si = 0; si < ns; ++si) pos += put_packed(sect[si], out + pos);
    for (si = 0; si < ns; ++si) {
        uint64_t rf[256], n_runs, k;
        uint32_t clen[256], code[256];
        size_t nbits;
        bitw w;
        if (sect[si] == 0) continue;
        memset(rf, 0, sizeof rf);
        n_runs = orc_run_frequencies_store(rf, runseq, runlen, bwt + beg, sect[si]);
        orc_huffman_lengths(rf, clen);
        pos += put_packed(n_runs, out + pos);                      /* :174-178 */
        nbits = orc_serialize_shape(clen, shape);                  /* :180-192 */
        for (i = 0; i < nbits;) {
            uint8_t b = 0; size_t j = 0;
            for (; j < 8 && i < nbits; ++i, ++j) b = (uint8_t)((b << 1) | shape[i]);
            if (j < 8) b = (uint8_t)(b << (8 - j));
            out[pos++] = b;
        }
        orc_huffman_codes(clen, code);                             /* :195-196 */
        w.p = out; w.n = pos; w.buf = 0; w.nb = 0;
        for (k = 0; k < n_runs; ++k) bw_put(&w, code[runseq[k]], (int)clen[runseq[k]]);
        bw_flush(&w);
        for (k = 0; k < n_runs; ++k)                               /* :229-251 */
            bw_put(&w, runlen[k], (int)log_floor(runlen[k]) * 2 + 1);
        bw_flush(&w);
        pos = w.n;
        beg += sect[si];
    }
    length = pos - 6;
    for (i = 0; i < 6; ++i) out[i] = (uint8_t)(length >> (8 * (5 - i)));
    free(runseq); free(runlen); free(shape);
    return pos;
}

size_t orc_compress_bound(size_t size) { return size * 2 + (size / 10000 + 2) * 1200 + 4096; }

/* Compressor::compress (Compressor.cpp:65-118) with no precompression: global header
 * (:55-58), one precompressor block per read of block_size bytes (pbBlockSize ==
 * bwtBlockSize when prepr is empty, :81), its header = packed original size, packed
 * #slices (PrecompressorBlock.cpp:64-90) + empty grammar = one 0x00 byte
 * (preprocessors/Grammar.cpp:310-320), the slices, and the 0x00 terminator (:115). */
size_t orc_compress_H(const uint8_t *in, size_t size, size_t block_size,
                      uint32_t starting_points, uint8_t *out, size_t out_cap)
{
    size_t pos = 0, off = 0;
    out[pos++] = 'H';
    while (off < size) {
        size_t bs = size - off < block_size ? size - off : block_size;
        uint8_t *blk = (uint8_t *)malloc(bs + 1);
        uint32_t lf[256], n_lf = 0, freqs[256];
        memcpy(blk, in + off, bs);
        blk[bs] = 0;
        pos += put_packed(bs, out + pos);
        pos += put_packed(1, out + pos);
        out[pos++] = 0;
        memset(freqs, 0, sizeof freqs);
        orc_bwt_block(blk, (uint32_t)bs, starting_points, lf, &n_lf, freqs);
        pos += orc_huffman_encode_block(blk, (uint32_t)bs, lf, n_lf, freqs, out + pos,
                                        out_cap - pos);
        free(blk);
        off += bs;
    }
    out[pos++] = 0;
    return pos;
}

typedef struct { const uint8_t *p; size_t n, pos; int bit; } bitr;
static int br_bit(bitr *r)
{
    int b;
    if (r->pos >= r->n) return 0;
    b = (r->p[r->pos] >> (7 - r->bit)) & 1;
    if (++r->bit == 8) { r->bit = 0; ++r->pos; }
    return b;
}
static void br_flush(bitr *r) { if (r->bit) { r->bit = 0; ++r->pos; } }
static uint64_t br_bits(bitr *r, int n) { uint64_t v = 0; while (n--) v = (v << 1) | (uint64_t)br_bit(r); return v; }
static uint64_t br_packed(bitr *r)       /* utils::readPackedInteger, Utils.hpp:164-178 */
{
    uint64_t res = 0, rd = 0xff; int j = 0;
    while (rd & 0x80) { rd = br_bits(r, 8); res |= (rd & 0x7f) << j; j += 7; }
    return res;
}
/* utils::binaryDecode, Utils.hpp:299-317 */
static size_t binary_decode(bitr *r, size_t lo, size_t hi)
{
    size_t range_len = hi - lo + 1, short_cw, long_cw2, result = 0;
    unsigned code_len, i;
    if (range_len == 1) return lo;
    code_len = log_ceiling(range_len);
    short_cw = ((size_t)1 << code_len) - range_len;
    long_cw2 = (range_len - short_cw) / 2;
    for (i = 0; i + 1 < code_len; ++i) result = (result << 1) | (size_t)br_bit(r);
    if (result >= long_cw2) return result + lo;
    result = (result << 1) | (size_t)br_bit(r);
    if (result < long_cw2) return result + lo;
    return result + lo + short_cw;
}
/* utils::binaryInterpolativeDecode, Utils.hpp:342-360 */
static void bic_decode(bitr *r, uint32_t *list, size_t *n, size_t lo, size_t hi, size_t elements)
{
    size_t h, rr, mid;
    if (elements == 0) return;
    if (elements == hi - lo + 1) { size_t i; for (i = lo; i <= hi; ++i) list[(*n)++] = (uint32_t)i; return; }
    h = (elements - 1) / 2;
    rr = elements / 2 - h;
    mid = binary_decode(r, lo + h, hi - h - rr);
    bic_decode(r, list, n, lo, mid - 1, h);
    list[(*n)++] = (uint32_t)mid;
    bic_decode(r, list, n, mid + 1, hi, elements - h - 1);
}

/* Decompressor::decompress (Decompressor.cpp:58-94) for coder 'H':
 * HuffmanDecoder::decodeBlock (HuffmanCoders.cpp:338-616, decoded here bit by bit from
 * the canonical code instead of through its lookup tables), deserializeShape (:88-117),
 * BWTBlock::readHeader (BWTBlock.cpp:88-102), then the inverse BWT. */
size_t orc_decompress_H(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_cap)
{
    bitr r; size_t opos = 0, dpos = 0, dcap = 0, gram_n = 0;
    const uint8_t *gram = NULL;
    uint8_t *aside = NULL, *dst = NULL;
    r.p = in; r.n = in_size; r.pos = 0; r.bit = 0;
    if (in_size < 2 || in[0] != 'H') return (size_t)-1;
    r.pos = 1;
    for (;;) {
        uint64_t orig = br_packed(&r), slices, s;
        if (orig == 0) break;
        slices = br_packed(&r);
        /* the grammar (PrecompressorBlock::readBlockHeader, PrecompressorBlock.cpp:97-108): one 0x00 byte when the
         * pre-stage was off; with rules the slices are decoded aside and expanded by the Postprocessor */
        {
            const size_t glen = r.pos <= r.n ? orc_grammar_span(in + r.pos, r.n - r.pos) : (size_t)-1;
            if (glen == (size_t)-1) return (size_t)-1;
            gram = in + r.pos; gram_n = glen;
            r.pos += glen;
        }
        aside = NULL; dst = out + opos; dcap = out_cap - opos; dpos = 0;
        if (!(gram_n == 1 && gram[0] == 0)) {
            dcap = 2 * (size_t)orig + 64;
            aside = (uint8_t *)malloc(dcap);
            if (!aside) return (size_t)-1;
            dst = aside;
        }
        for (s = 0; s < slices; ++s) {
            uint64_t clen48 = br_bits(&r, 48);
            size_t start = r.pos, bsize = 0, done = 0;
            uint32_t n_lf = (uint32_t)br_bits(&r, 8) + 1, lf[256], i, nsec;
            uint64_t sect[256];
            uint8_t *blk;
            for (i = 0; i < n_lf; ++i) lf[i] = (uint32_t)br_bits(&r, 31);
            br_flush(&r);
            nsec = (uint32_t)br_bits(&r, 8); if (nsec == 0) nsec = 256;
            for (i = 0; i < nsec; ++i) { sect[i] = br_packed(&r); bsize += sect[i]; }
            if (dpos + bsize > dcap) { free(aside); return (size_t)-1; }
            blk = dst + dpos;
            for (i = 0; i < nsec; ++i) {
                uint64_t n_runs, k;
                uint32_t clen[256], code[256], alphabet[256];
                size_t max_sym, nsym, max_len = 0, na = 0, rd = 0xff; int j = 0;
                uint8_t *runseq;
                if (sect[i] == 0) continue;
                n_runs = br_packed(&r);
                memset(clen, 0, sizeof clen);
                max_sym = br_bits(&r, 8); nsym = br_bits(&r, 8); if (nsym == 0) nsym = 256;
                while (rd & 0x80) { rd = br_bits(&r, 8); max_len |= (rd & 0x7f) << j; j += 7; }
                bic_decode(&r, alphabet, &na, 0, max_sym, nsym);
                for (k = 0; k < nsym; ++k) {
                    size_t n = 1; while (!br_bit(&r)) { ++n; if (r.pos >= r.n) return (size_t)-1; }
                    clen[alphabet[k]] = (uint32_t)(max_len - n + 1);
                }
                br_flush(&r);
                orc_huffman_codes(clen, code);
                runseq = (uint8_t *)malloc(n_runs ? n_runs : 1);
                for (k = 0; k < n_runs; ++k) {
                    uint32_t v = 0, len = 0; int sym = -1, c;
                    while (sym < 0) {
                        v = (v << 1) | (uint32_t)br_bit(&r); ++len;
                        if (len > 64 || r.pos > r.n) { free(runseq); return (size_t)-1; }
                        for (c = 0; c < 256; ++c) if (clen[c] == len && code[c] == v) { sym = c; break; }
                    }
                    runseq[k] = (uint8_t)sym;
                }
                br_flush(&r);
                for (k = 0; k < n_runs; ++k) {
                    int zeros = 0; uint64_t v;
                    while (!br_bit(&r)) { ++zeros; if (r.pos >= r.n) { free(runseq); return (size_t)-1; } }
                    v = br_bits(&r, zeros) | ((uint64_t)1 << zeros);
                    if (done + v > bsize) { free(runseq); return (size_t)-1; }
                    memset(blk + done, runseq[k], v);
                    done += v;
                }
                br_flush(&r);
                free(runseq);
            }
            if (done != bsize || r.pos - start != clen48) return (size_t)-1;
            if (orc_inverse_bwt_block(blk, (uint32_t)bsize, lf, n_lf) != 0) return (size_t)-1;
            dpos += bsize;
        }
        if (aside) {
            const size_t got = orc_postprocess_bytes(gram, gram_n, aside, dpos, out + opos, out_cap - opos);
            free(aside);
            if (got == (size_t)-1 || got != orig) return (size_t)-1;
            opos += got;
        } else {
            opos += dpos;
        }
    }
    return opos;
}
