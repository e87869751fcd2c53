/*
 * oracle/wavelet_oracle.c -- CPU restatement of the reference's wavelet coder 'B'
 * (WaveletEncoder + WaveletTree<vector<bool>> + BitEncoder + the 'B' probability models).
 *
 * TEST INFRASTRUCTURE ONLY.  Literal restatement, quirks included:
 * every function cites the reference lines it follows.  Both OPTIMIZED_INTEGER_CODE and
 * SEMI_FIXED_CODE are defined in the reference (WaveletTree.hpp:48-49); only those paths are
 * restated.  Where the reference runs into undefined behaviour (bitsForIntegers walking a
 * reverse iterator past rend(), WaveletTree.hpp:1606-1613) the candidate is dropped here.
 * Pin: the 36-byte 'B' stream recorded from the reference (tests/golden/streams.json).
 * The letters 'b' and 'u' (other main models, same everything else) have NO reference-produced
 * vector here: parity unpinned for them.  'm' / 'M' are not restated (undefined behaviour in
 * the reference, see model_init).
 */
#include "bwtc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- bit vectors / tree --- */
typedef struct { uint8_t *b; size_t n, cap; } bitvec;
static void bv_push(bitvec *v, int bit)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 16; v->b = (uint8_t *)realloc(v->b, v->cap); }
    v->b[v->n++] = (uint8_t)(bit != 0);
}
static void bv_append(bitvec *d, const bitvec *s) { size_t i; for (i = 0; i < s->n; ++i) bv_push(d, s->b[i]); }
static void bv_free(bitvec *v) { free(v->b); v->b = NULL; v->n = v->cap = 0; }

/* TreeNode, WaveletTree.hpp:53-76 */
typedef struct { int left, right; int has_symbol; uint32_t symbol; bitvec bv; } node;
typedef struct { node *v; int n, cap; } nodepool;
static int np_new(nodepool *p, int has_symbol, uint32_t symbol)
{
    if (p->n == p->cap) { p->cap = p->cap ? p->cap * 2 : 64; p->v = (node *)realloc(p->v, (size_t)p->cap * sizeof(node)); }
    memset(&p->v[p->n], 0, sizeof(node));
    p->v[p->n].left = p->v[p->n].right = -1;
    p->v[p->n].has_symbol = has_symbol;
    p->v[p->n].symbol = symbol;
    return p->n++;
}

typedef struct { uint64_t first; uint32_t second; } pair;     /* std::pair<uint64, uint32> */
static int cmp_pair(const void *a, const void *b)
{
    const pair *x = (const pair *)a, *y = (const pair *)b;
    if (x->first != y->first) return (x->first > y->first) - (x->first < y->first);
    return (x->second > y->second) - (x->second < y->second);
}

/* static assignPrefixCodes(lengths, node, elem, bits), WaveletTree.hpp:1446-1482 */
static size_t assign_prefix_codes(nodepool *p, const pair *lengths, size_t n, int nd, size_t elem, size_t bits)
{
    if (elem >= n) return elem;
    if (bits == lengths[elem].first - 1) {
        if (p->v[nd].left < 0) {
            int c = np_new(p, 1, lengths[elem].second);
            p->v[nd].left = c;
            elem = assign_prefix_codes(p, lengths, n, nd, elem + 1, bits);
        } else {
            int c = np_new(p, 1, lengths[elem].second);
            p->v[nd].right = c;
            ++elem;
        }
        return elem;
    }
    if (p->v[nd].left < 0) {
        int c = np_new(p, 0, 0);
        p->v[nd].left = c;
        elem = assign_prefix_codes(p, lengths, n, c, elem, bits + 1);
    }
    if (elem < n) {
        int c = np_new(p, 0, 0);
        p->v[nd].right = c;
        elem = assign_prefix_codes(p, lengths, n, c, elem, bits + 1);
    }
    return elem;
}

/* utils::calculateCodeLengths(codeLengths, 0, sorted), Utils.cpp:422-473 */
static void calculate_code_lengths(pair *cl, size_t n, int sorted)
{
    uint64_t *f;
    size_t s = 0, r = 0, t, i;
    if (n == 1) { cl[0].first = 1; return; }
    if (!sorted) qsort(cl, n, sizeof(pair), cmp_pair);
    f = (uint64_t *)malloc(n * sizeof(uint64_t));
    for (i = 0; i < n; ++i) f[i] = cl[i].first;
    for (t = 0; t < n - 1; ++t) {
        if (s >= n || (r < t && f[r] < f[s])) { f[t] = f[r]; f[r++] = t; } else f[t] = f[s++];
        if (s >= n || (r < t && f[r] < f[s])) { f[t] += f[r]; f[r++] = t; } else f[t] += f[s++];
    }
    f[n - 2] = 0;
    for (long k = (long)n - 3; k >= 0; --k) f[k] = f[f[k]] + 1;
    {
        long a = 1, u = 0, x = (long)n - 1, tt = (long)n - 2;
        uint64_t depth = 0;
        while (a > 0) {
            while (tt >= 0 && f[tt] == depth) { ++u; --tt; }
            while (a > u) { f[x] = depth; --x; --a; }
            a = 2 * u; ++depth; u = 0;
        }
    }
    for (i = 0; i < n; ++i) cl[i].first = f[i];
    free(f);
}

static unsigned log_floor64(uint64_t n) { unsigned l = 0; while (n > 1) { n >>= 1; ++l; } return l; }
/* lengthOfFixedCode, WaveletTree.hpp:528-533 */
static uint32_t length_of_fixed_code(uint32_t x, uint32_t w)
{
    const uint64_t wpow = (uint64_t)1 << w;
    const size_t B = log_floor64((uint64_t)x - 1 + wpow) - w;
    return (uint32_t)(2 * B + w + 1);
}
/* fixedIntegerCode, WaveletTree.hpp:514-526 */
static void fixed_integer_code(bitvec *bits, uint32_t x, uint32_t w)
{
    const uint64_t wpow = (uint64_t)1 << w;
    const size_t B = log_floor64((uint64_t)x - 1 + wpow) - w;
    size_t i;
    uint64_t y;
    int k;
    for (i = 0; i < B; ++i) bv_push(bits, 1);
    bv_push(bits, 0);
    y = (uint64_t)x - (1 + ((((uint64_t)1 << B) - 1) << w));      /* fixedIntegerCodeTranslation */
    for (k = (int)(w + B) - 1; k >= 0; --k) bv_push(bits, (int)((y >> k) & 1));
}

/* bitsForIntegers, WaveletTree.hpp:1552-1641.  out/n_out = codeLengths (appended to). */
static uint64_t bits_for_integers(uint32_t w, uint32_t depth, const pair *ifq, size_t n_ifq,
                                  pair *out, size_t *n_out, uint64_t total_freq, int *ub)
{
    uint64_t total = 0, min_freq = 0;
    size_t i, n = *n_out;
    uint8_t *removed;
    /* std::map<uint64, uint32> notRemoved: key = frequency, later insertions overwrite */
    pair *not_removed;
    size_t n_nr = 0;
    if (n_ifq < depth) return 0;
    if (depth == 0) {
        for (i = 0; i < n_ifq; ++i) total += ifq[i].first * length_of_fixed_code(ifq[i].second, w);
        return total;
    }
    removed = (uint8_t *)calloc(n_ifq, 1);
    not_removed = (pair *)malloc((n_ifq + 1) * sizeof(pair));
    for (i = 0; i < n_ifq; ++i) {
        const double optimal = log((double)total_freq / (double)ifq[i].first);
        const double price = (double)ifq[i].first * (double)(depth + length_of_fixed_code(ifq[i].second, w));
        if ((double)ifq[i].first * optimal + 64 < price && ifq[i].first >= 10) {
            removed[i] = 1;
            out[n++] = ifq[i];
            if (min_freq == 0 || min_freq > ifq[i].first) min_freq = ifq[i].first;
        } else {
            size_t k;
            for (k = 0; k < n_nr; ++k) if (not_removed[k].first == ifq[i].first) break;
            not_removed[k].first = ifq[i].first;
            not_removed[k].second = (uint32_t)i;
            if (k == n_nr) ++n_nr;
        }
    }
    if (depth > n) {
        const int need = (int)(depth - n);
        int k;
        qsort(not_removed, n_nr, sizeof(pair), cmp_pair);          /* map order: ascending key */
        if ((size_t)need > n_nr) { *ub = 1; free(removed); free(not_removed); return 0; }
        for (k = 0; k < need; ++k) {                               /* reverse iteration */
            const pair *it = &not_removed[n_nr - 1 - (size_t)k];
            removed[it->second] = 1;
            out[n++] = ifq[it->second];
            if (min_freq == 0 || min_freq > it->first) min_freq = it->first;
        }
        out[n].first = min_freq; out[n].second = 0; ++n;
        qsort(out, n, sizeof(pair), cmp_pair);
    } else {
        out[n].first = min_freq; out[n].second = 0; ++n;
    }
    calculate_code_lengths(out, n, 1);
    qsort(out, n, sizeof(pair), cmp_pair);
    {
        long j = (long)n - 1;
        uint32_t real_depth;
        while (out[j].second != 0) --j;
        while (j > 0) {
            if (out[j - 1].first >= depth) { uint32_t t = out[j].second; out[j].second = out[j - 1].second; out[j - 1].second = t; --j; }
            else break;
        }
        real_depth = out[j].second;                                /* sic: the symbol, i.e. 0 */
        for (i = 0; i < n_ifq; ++i) {
            if (removed[i]) {
                size_t k; uint64_t hl = 0;
                for (k = 0; k < n; ++k) if (out[k].second == ifq[i].second) hl = out[k].first;
                total += ifq[i].first * hl;
            } else {
                total += ifq[i].first * (real_depth + length_of_fixed_code(ifq[i].second, w));
            }
        }
    }
    free(removed); free(not_removed);
    *n_out = n;
    return total;
}

/* findParametersForSemiFixedCodes, WaveletTree.hpp:1644-1680 */
static uint32_t find_parameters(pair *ifq, size_t *n_ifq, uint64_t total_freq)
{
    const size_t n = *n_ifq;
    pair *code_lengths = (pair *)malloc((n + 2) * sizeof(pair));
    pair *tmp = (pair *)malloc((n + 2) * sizeof(pair));
    size_t n_cl = 0;
    uint32_t best_w = 0, w, depth;
    uint64_t least;
    int ub = 0;
    qsort(ifq, n, sizeof(pair), cmp_pair);
    least = bits_for_integers(0, 0, ifq, n, code_lengths, &n_cl, total_freq, &ub);
    for (w = 1; w < 16; ++w) {
        const uint64_t bits = bits_for_integers(w, 0, ifq, n, code_lengths, &n_cl, total_freq, &ub);
        if (bits < least) best_w = w;                              /* sic: least is not updated */
    }
    for (w = 0; w < 16; ++w) {
        for (depth = 1; depth <= 3; ++depth) {
            size_t n_tmp = 0;
            uint64_t bits;
            ub = 0;
            bits = bits_for_integers(w, depth, ifq, n, tmp, &n_tmp, total_freq, &ub);
            if (ub) continue;                                      /* reference: undefined behaviour */
            if (bits == 0) break;
            if (bits < least) { memcpy(code_lengths, tmp, n_tmp * sizeof(pair)); n_cl = n_tmp; least = bits; best_w = w; }
        }
    }
    memcpy(ifq, code_lengths, n_cl * sizeof(pair));
    *n_ifq = n_cl;
    free(code_lengths); free(tmp);
    return best_w;
}

/* ------------------------------------------------------------------------- the tree --- */
typedef struct {
    nodepool pool;
    int root;
    bitvec codes[256];                      /* m_codes */
    uint32_t W;                             /* m_W */
    /* m_integerCodes (std::map<uint32, BitVector>), kept sorted by key */
    uint32_t *ic_key; bitvec *ic_code; size_t n_ic;
} wtree;

static void collect_codes_sym(const nodepool *p, int nd, bitvec *path, bitvec *codes)
{                                                                   /* WaveletTree.hpp:1533-1550 */
    if (p->v[nd].left < 0 && p->v[nd].right < 0) { codes[p->v[nd].symbol].n = 0; bv_append(&codes[p->v[nd].symbol], path); }
    if (p->v[nd].left >= 0) { bv_push(path, 0); collect_codes_sym(p, p->v[nd].left, path, codes); --path->n; }
    if (p->v[nd].right >= 0) { bv_push(path, 1); collect_codes_sym(p, p->v[nd].right, path, codes); --path->n; }
}
static void ic_set(wtree *t, uint32_t key, const bitvec *code)
{
    size_t i, pos = t->n_ic;
    for (i = 0; i < t->n_ic; ++i) { if (t->ic_key[i] == key) { t->ic_code[i].n = 0; bv_append(&t->ic_code[i], code); return; } if (t->ic_key[i] > key) { pos = i; break; } }
    t->ic_key = (uint32_t *)realloc(t->ic_key, (t->n_ic + 1) * sizeof(uint32_t));
    t->ic_code = (bitvec *)realloc(t->ic_code, (t->n_ic + 1) * sizeof(bitvec));
    memmove(t->ic_key + pos + 1, t->ic_key + pos, (t->n_ic - pos) * sizeof(uint32_t));
    memmove(t->ic_code + pos + 1, t->ic_code + pos, (t->n_ic - pos) * sizeof(bitvec));
    t->ic_key[pos] = key; memset(&t->ic_code[pos], 0, sizeof(bitvec));
    bv_append(&t->ic_code[pos], code);
    ++t->n_ic;
}
static const bitvec *ic_find(const wtree *t, uint32_t key)
{
    size_t i; for (i = 0; i < t->n_ic; ++i) if (t->ic_key[i] == key) return &t->ic_code[i];
    return NULL;
}
static void collect_codes_int(wtree *t, const nodepool *p, int nd, bitvec *path)
{
    if (p->v[nd].left < 0 && p->v[nd].right < 0) ic_set(t, p->v[nd].symbol, path);
    if (p->v[nd].left >= 0) { bv_push(path, 0); collect_codes_int(t, p, p->v[nd].left, path); --path->n; }
    if (p->v[nd].right >= 0) { bv_push(path, 1); collect_codes_int(t, p, p->v[nd].right, path); --path->n; }
}

/* pushBits(bits) from the root, WaveletTree.hpp:1184-1199 */
static int push_code(wtree *t, const bitvec *bits)
{
    int nd = t->root; size_t i;
    for (i = 0; i < bits->n; ++i) { bv_push(&t->pool.v[nd].bv, bits->b[i]); nd = bits->b[i] ? t->pool.v[nd].right : t->pool.v[nd].left; }
    return nd;
}
/* pushBits(node, bits, symbol), WaveletTree.hpp:1228-1249 */
static void push_bits_sym(wtree *t, int nd, const bitvec *bits, uint32_t symbol)
{
    size_t i;
    for (i = 0; i + 1 < bits->n; ++i) {
        bv_push(&t->pool.v[nd].bv, bits->b[i]);
        if (bits->b[i]) { if (t->pool.v[nd].right < 0) { int c = np_new(&t->pool, 0, 0); t->pool.v[nd].right = c; } nd = t->pool.v[nd].right; }
        else { if (t->pool.v[nd].left < 0) { int c = np_new(&t->pool, 0, 0); t->pool.v[nd].left = c; } nd = t->pool.v[nd].left; }
    }
    bv_push(&t->pool.v[nd].bv, bits->b[bits->n - 1]);
    if (bits->b[bits->n - 1] && t->pool.v[nd].right < 0) { int c = np_new(&t->pool, 1, symbol); t->pool.v[nd].right = c; }
    else if (!bits->b[bits->n - 1] && t->pool.v[nd].left < 0) { int c = np_new(&t->pool, 1, symbol); t->pool.v[nd].left = c; }
}

/* WaveletTree(const byte*, size_t), WaveletTree.hpp:294-360 */
static void wtree_build(wtree *t, const uint8_t *src, size_t length)
{
    uint64_t run_freqs[256];
    uint32_t *dl = (uint32_t *)malloc((length + 1) * sizeof(uint32_t));
    uint32_t *dc = (uint32_t *)malloc((length + 1) * sizeof(uint32_t));
    size_t n_pairs = 0, i, n_sym = 0;
    uint64_t total_runs;
    pair cl[256];
    bitvec path = {0, 0, 0};
    memset(t, 0, sizeof *t);
    memset(run_freqs, 0, sizeof run_freqs);
    total_runs = orc_runs_and_characters(run_freqs, src, length, dl, dc, length + 1, &n_pairs);
    /* calculateHuffmanLengths + assignPrefixCodes(lengths): sort by (length, symbol), :1440-1444 */
    for (i = 0; i < 256; ++i) if (run_freqs[i]) { cl[n_sym].first = run_freqs[i]; cl[n_sym].second = (uint32_t)i; ++n_sym; }
    calculate_code_lengths(cl, n_sym, 0);
    qsort(cl, n_sym, sizeof(pair), cmp_pair);
    t->root = np_new(&t->pool, 0, 0);
    assign_prefix_codes(&t->pool, cl, n_sym, t->root, 0, 0);
    {   /* integer codes, :312-347 */
        pair *ifq = (pair *)malloc((n_pairs + 2) * sizeof(pair));
        size_t n_ifq = n_pairs;
        nodepool ip = {0, 0, 0};
        int iroot;
        for (i = 0; i < n_pairs; ++i) { ifq[i].first = dc[i]; ifq[i].second = dl[i]; }
        t->W = find_parameters(ifq, &n_ifq, total_runs);
        iroot = np_new(&ip, 0, 0);
        if (n_ifq > 0) {
            assign_prefix_codes(&ip, ifq, n_ifq, iroot, 0, 0);
            collect_codes_int(t, &ip, iroot, &path);
        } else {
            bitvec empty = {0, 0, 0};
            ic_set(t, 0, &empty);
        }
        for (i = 0; i < (size_t)ip.n; ++i) bv_free(&ip.v[i].bv);
        free(ip.v); free(ifq);
    }
    path.n = 0;
    collect_codes_sym(&t->pool, t->root, &path, t->codes);
    /* pushMessage / pushRun, :1484-1494, :1251-1275 */
    {
        size_t a = 0;
        bitvec code = {0, 0, 0};
        while (a < length) {
            size_t b = a + 1;
            int leaf;
            const bitvec *ic;
            while (b < length && src[b] == src[a]) ++b;
            leaf = push_code(t, &t->codes[src[a]]);
            ic = ic_find(t, (uint32_t)(b - a));
            if (!ic) {
                code.n = 0;
                bv_append(&code, ic_find(t, 0));
                fixed_integer_code(&code, (uint32_t)(b - a), t->W);
                push_bits_sym(t, leaf, &code, (uint32_t)(b - a));
            } else {
                push_bits_sym(t, leaf, ic, (uint32_t)(b - a));
            }
            a = b;
        }
        bv_free(&code);
    }
    bv_free(&path); free(dl); free(dc);
}
static void wtree_free(wtree *t)
{
    int i; size_t k;
    for (i = 0; i < t->pool.n; ++i) bv_free(&t->pool.v[i].bv);
    free(t->pool.v);
    for (i = 0; i < 256; ++i) bv_free(&t->codes[i]);
    for (k = 0; k < t->n_ic; ++k) bv_free(&t->ic_code[k]);
    free(t->ic_key); free(t->ic_code);
}

/* ------------------------------------------------------------------------ treeShape --- */
static void push_bits_msb(bitvec *v, uint64_t n, unsigned bits) { unsigned i; for (i = 1; i <= bits; ++i) bv_push(v, (int)((n >> (bits - i)) & 1)); }
static void push_bits_rev(bitvec *v, uint64_t n, unsigned bits) { unsigned i; for (i = 0; i < bits; ++i) { bv_push(v, (int)(n & 1)); n >>= 1; } }
static void unary_code(bitvec *v, size_t n) { while (n-- > 1) bv_push(v, 0); bv_push(v, 1); }
static void interp(bitvec *v, const uint32_t *list, size_t count, size_t lo, size_t hi)
{
    uint8_t *tmp = (uint8_t *)malloc(64 * (count + 1) + 64);
    size_t nb, i;
    /* orc_binary_interpolative_code codes with lo = 0; the integer list uses lo = 0 too (:600) */
    (void)lo;
    nb = orc_binary_interpolative_code(list, count, hi, tmp);
    for (i = 0; i < nb; ++i) bv_push(v, tmp[i]);
    free(tmp);
}
/* treeShape, WaveletTree.hpp:553-635 */
static void tree_shape(const wtree *t, bitvec *vec)
{
    uint32_t symbols[256];
    size_t ns = 0, max_len = 0, i;
    int bytes;
    uint64_t packed;
    for (i = 0; i < 256; ++i) if (t->codes[i].n > 0) { symbols[ns++] = (uint32_t)i; if (t->codes[i].n > max_len) max_len = t->codes[i].n; }
    push_bits_msb(vec, symbols[ns - 1], 8);
    push_bits_msb(vec, ns, 8);
    packed = orc_pack_integer(max_len, &bytes);
    push_bits_rev(vec, packed, 8u * (unsigned)bytes);
    interp(vec, symbols, ns, 0, symbols[ns - 1]);
    for (i = 0; i < ns; ++i) unary_code(vec, max_len - t->codes[symbols[i]].n + 1);
    {
        max_len = 0;
        for (i = 0; i < t->n_ic; ++i) if (t->ic_code[i].n > max_len) max_len = t->ic_code[i].n;
        packed = orc_pack_integer(t->ic_key[t->n_ic - 1], &bytes);
        push_bits_rev(vec, packed, 8u * (unsigned)bytes);
        if (packed > 0) {
            packed = orc_pack_integer(t->n_ic, &bytes);
            push_bits_rev(vec, packed, 8u * (unsigned)bytes);
            packed = orc_pack_integer(max_len, &bytes);
            push_bits_rev(vec, packed, 8u * (unsigned)bytes);
            interp(vec, t->ic_key, t->n_ic, 0, t->ic_key[t->n_ic - 1]);
            for (i = 0; i < t->n_ic; ++i) unary_code(vec, max_len - t->ic_code[i].n + 1);
        }
        for (int k = 3; k >= 0; --k) bv_push(vec, (int)((t->W >> k) & 1));
    }
}

/* --------------------------------------------------------- models and the range coder --- */
/* UnbiasedPredictor<Min, Delay, Initial>, probmodels/BitPredictors.hpp:37-65 (Probability =
 * uint16, scale 4096) */
typedef struct { uint16_t p, min, init; unsigned delay; } upred;
static void up_init(upred *u, uint16_t min, unsigned delay, uint16_t init) { u->min = min; u->delay = delay; u->init = init; u->p = init; }
static void up_update(upred *u, int bit)
{
    const uint16_t maxp = (uint16_t)(4096 - u->min);
    if (bit) u->p = (uint16_t)(u->p + ((maxp - u->p) >> u->delay));
    else u->p = (uint16_t)(u->p - ((u->p - u->min) >> u->delay));
}
/* generic nextState<states>, probmodels/FSM.hpp:42-54, and the <3> specialisation :61-67 */
static uint32_t next_state(uint32_t states, uint32_t cur, int bit)
{
    if (states == 3) {
        if (cur == 1) return bit ? 2 : 0;
        if (cur == 2 && bit) return 2;
        if (cur == 0 && !bit) return 0;
        return 1;
    }
    if (bit) return cur >= states / 2 ? (cur + 1 < states - 1 ? cur + 1 : states - 1) : states / 2;
    return cur < states / 2 ? (cur > 0 ? cur - 1 : 0) : (states - 1) / 2;
}
/* EvenIntervalPredictor<M>, probmodels/BitPredictors.hpp:95-125: the probability steps by
 * 4096/M and stays strictly inside (0, 4096) */
static void ei_update(upred *u, int bit, uint16_t interval)
{
    if (bit) { if ((uint32_t)u->p + interval < 4096u) u->p = (uint16_t)(u->p + interval); }
    else if (interval < u->p) u->p = (uint16_t)(u->p - interval);
}
/* model = FSM<N, Predictor> (FSM.hpp:81-110), FSM8 (:168-227) or a lone predictor.  For FSM8
 * states 4..7 are InversePredictors (BitPredictors.hpp:300-319) of z1..z4 and resetModel()
 * keeps the current state; a lone predictor has no state machine (ProbabilityModel::updateState
 * is a no-op, probmodels/ProbabilityModel.hpp:48-50).  even = EvenIntervalPredictor<4> states
 * instead of UnbiasedPredictors. */
typedef struct { int is_fsm8, even; uint32_t n, cur; upred st[8]; } model;
static void model_reset(model *m)
{
    uint32_t i;
    for (i = 0; i < m->n; ++i) m->st[i].p = m->st[i].init;
    if (!m->is_fsm8) m->cur = m->n / 2;
}
static uint16_t model_p1(const model *m)
{
    if (m->is_fsm8 && m->cur >= 4) return (uint16_t)(4096 - m->st[m->cur].p);
    return m->st[m->cur].p;
}
static void model_update_state(model *m, int bit) { if (m->n > 1) m->cur = next_state(m->n, m->cur, bit); }
static void model_update(model *m, int bit)
{
    if (m->even) ei_update(&m->st[m->cur], bit, 1024);
    else if (m->is_fsm8 && m->cur >= 4) up_update(&m->st[m->cur], !bit);
    else up_update(&m->st[m->cur], bit);
    model_update_state(m, bit);
}
/* giveProbabilityModel(choice), probmodels/ProbabilityModel.cpp:47-76.  'm' and 'M'
 * (SimpleMarkov) are not restated: their history table is one entry short of the contexts it
 * is indexed with (:91-93 vs :110-118), so their output is undefined.  Returns 0 for them. */
static int model_init(char choice, model *pm, model *gm, model *gapm)
{
    uint32_t i;
    memset(pm, 0, sizeof *pm); memset(gm, 0, sizeof *gm); memset(gapm, 0, sizeof *gapm);
    if (choice == 'm' || choice == 'M') return 0;
    if (choice == 'u') {                                           /* EvenIntervalPredictor<4> */
        pm->even = 1; pm->n = 1; pm->cur = 0;
        up_init(&pm->st[0], 0, 0, 2048);
    } else if (choice == 'b') {                                    /* FSM<6, EvenIntervalPredictor<4> > */
        pm->even = 1; pm->n = 6; pm->cur = 3;
        for (i = 0; i < 6; ++i) up_init(&pm->st[i], 0, 0, 2048);
    } else {                                                       /* 'B' and default: FSM8<Z4,Z3,Z2,Z1>, states z4 z3 z2 z1 o1 o2 o3 o4 */
        pm->is_fsm8 = 1; pm->n = 8; pm->cur = 4;
        up_init(&pm->st[0], 2, 4, 2400); up_init(&pm->st[1], 2, 5, 2300);
        up_init(&pm->st[2], 2, 5, 2200); up_init(&pm->st[3], 2, 5, 2100);
        up_init(&pm->st[4], 2, 5, 2100); up_init(&pm->st[5], 2, 5, 2200);
        up_init(&pm->st[6], 2, 5, 2300); up_init(&pm->st[7], 2, 4, 2400);
    }
    gm->n = 3; gm->cur = 1;                                        /* giveModelForIntegerCodes :38-41 */
    for (i = 0; i < 3; ++i) up_init(&gm->st[i], 100, 5, 2048);
    gapm->n = 4; gapm->cur = 2;                                    /* giveModelForGaps :43-45 */
    for (i = 0; i < 4; ++i) up_init(&gapm->st[i], 2, 5, 2048);
    return 1;
}

/* BitEncoder, BitCoders.cpp:59-113 */
typedef struct { uint32_t low, high; uint8_t *out; size_t n; } rcoder;
static void rc_encode(rcoder *c, int bit, uint16_t p1)
{
    const uint32_t range_size = c->high - c->low - 1;
    const uint32_t high_bits = range_size >> 12, low_bits = range_size & 4095u;
    const uint32_t split = c->low + high_bits * p1 + ((low_bits * p1 + 2048u) >> 12);
    if (bit) c->high = split; else c->low = split + 1;
    while (((c->low ^ c->high) & 0xFF000000u) == 0) {
        c->out[c->n++] = (uint8_t)(c->low >> 24);
        c->low <<= 8;
        c->high = (c->high << 8) + 255;
    }
}
static void rc_finish(rcoder *c)
{
    c->out[c->n++] = (uint8_t)(c->low >> 24);
    c->out[c->n++] = 255; c->out[c->n++] = 255; c->out[c->n++] = 255;
    c->low = 0; c->high = 0xFFFFFFFFu;
}

/* ---------------------------------------------------------------------- encodeTreeBF --- */
typedef struct { int nd; bitvec gaps; } inode;                     /* InternalNode :652 */
/* WaveletTree.hpp:637-809 */
static void encode_tree_bf(wtree *t, rcoder *enc, model *pm, model *gm, model *gapm)
{
    nodepool *p = &t->pool;
    inode *queue = (inode *)calloc((size_t)p->n + 4, sizeof(inode));
    int *icn = (int *)malloc(((size_t)p->n * 4 + 16) * sizeof(int));   /* integerCodeNodes list */
    size_t qh = 0, qt = 0, ih = 0, it = 0, i;
    {
        inode left, right;
        const node *root = &p->v[t->root];
        int prev = !root->bv.b[0];
        memset(&left, 0, sizeof left); memset(&right, 0, sizeof right);
        for (i = 0; i < root->bv.n; ++i) {
            const int bit = root->bv.b[i];
            rc_encode(enc, bit, model_p1(pm));
            model_update(pm, bit);
            bv_push(bit ? &right.gaps : &left.gaps, prev != bit);
            prev = bit;
        }
        if (root->left >= 0) {
            if (p->v[root->left].has_symbol) { icn[it++] = root->left; bv_free(&left.gaps); }
            else { left.nd = root->left; queue[qt++] = left; }
        } else bv_free(&left.gaps);
        if (root->right >= 0) {
            if (p->v[root->right].has_symbol) { icn[it++] = root->right; bv_free(&right.gaps); }
            else { right.nd = root->right; queue[qt++] = right; }
        } else bv_free(&right.gaps);
    }
    while (qh < qt) {
        inode left, right, *nodeq = &queue[qh];
        const node *nd = &p->v[nodeq->nd];
        int prev;
        memset(&left, 0, sizeof left); memset(&right, 0, sizeof right);
        model_reset(pm); model_reset(gapm);
        prev = !nd->bv.b[0];
        if (p->v[nd->left].has_symbol || p->v[nd->right].has_symbol) {
            if (p->v[nd->left].has_symbol && p->v[nd->right].has_symbol) {
                for (i = 0; i < nd->bv.n; ++i) {
                    int bit;
                    if (!nodeq->gaps.b[i]) continue;
                    bit = nd->bv.b[i];
                    rc_encode(enc, bit, model_p1(gapm));
                    model_update(gapm, bit);
                }
                icn[it++] = nd->left; icn[it++] = nd->right;
            } else if (p->v[nd->left].has_symbol) {
                right.nd = nd->right;
                for (i = 0; i < nd->bv.n; ++i) {
                    const int bit = nd->bv.b[i];
                    if (bit) bv_push(&right.gaps, prev != bit || nodeq->gaps.b[i]);
                    if (prev || nodeq->gaps.b[i]) {
                        if (nodeq->gaps.b[i]) {
                            rc_encode(enc, bit, model_p1(gapm));
                            model_update(gapm, bit);
                            model_update_state(pm, bit);
                        } else {
                            rc_encode(enc, bit, model_p1(pm));
                            model_update(pm, bit);
                        }
                    }
                    prev = bit;
                }
                queue[qt++] = right;
                icn[it++] = nd->left;
            }
        } else {
            for (i = 0; i < nd->bv.n; ++i) {
                const int bit = nd->bv.b[i];
                if (nodeq->gaps.b[i]) {
                    rc_encode(enc, bit, model_p1(gapm));
                    model_update(gapm, bit);
                    model_update_state(pm, bit);
                } else {
                    rc_encode(enc, bit, model_p1(pm));
                    model_update(pm, bit);
                }
                bv_push(bit ? &right.gaps : &left.gaps, prev != bit || nodeq->gaps.b[i]);
                prev = bit;
            }
            left.nd = nd->left; queue[qt++] = left;
            right.nd = nd->right; queue[qt++] = right;
        }
        bv_free(&nodeq->gaps);
        ++qh;
    }
    {   /* synchronised integer-coding phase, :767-804 */
        int *lefts = (int *)malloc(((size_t)p->n + 4) * sizeof(int));
        int *rights = (int *)malloc(((size_t)p->n + 4) * sizeof(int));
        size_t nl = 0, nr = 0;
        while (ih < it || nl || nr) {
            for (i = 0; i < nl; ++i) icn[it++] = lefts[i];
            for (i = 0; i < nr; ++i) icn[it++] = rights[i];
            nl = nr = 0;
            model_reset(gm);
            while (ih < it) {
                const node *nd = &p->v[icn[ih++]];
                for (i = 0; i < nd->bv.n; ++i) { rc_encode(enc, nd->bv.b[i], model_p1(gm)); model_update(gm, nd->bv.b[i]); }
                if (nd->left >= 0 && (!p->v[nd->left].has_symbol || p->v[nd->left].symbol == 0)) lefts[nl++] = nd->left;
                if (nd->right >= 0 && (!p->v[nd->right].has_symbol || p->v[nd->right].symbol == 0)) rights[nr++] = nd->right;
            }
        }
        free(lefts); free(rights);
    }
    free(queue); free(icn);
}

/* ------------------------------------------------------------------- block and stream --- */
static size_t put_packed_w(uint64_t v, uint8_t *out)
{
    int bytes, i; uint64_t pk = orc_pack_integer(v, &bytes);
    for (i = 0; i < bytes; ++i) { out[i] = (uint8_t)(pk & 0xFF); pk >>= 8; }
    return (size_t)bytes;
}

/* persistent coder state across sections and blocks, as WaveletEncoder holds it */
typedef struct { model pm, gm, gapm; rcoder rc; } wstate;

/* WaveletEncoder::writeBlockHeader + encodeData + finishBlock for one transformed block
 * (WaveletCoders.cpp:173-219, :112-157, :159-163) */
static size_t wavelet_encode_block(wstate *ws, const uint8_t *bwt, uint32_t size, const uint32_t *lf,
                                   uint32_t n_lf, const uint32_t freqs[256], uint8_t *out)
{
    uint32_t sect[256], ns, si;
    size_t pos = 6, beg = 0, i;
    uint64_t length;
    memset(out, 0, 6);
    pos += orc_write_bwtblock_header(lf, n_lf, out + pos);
    ns = orc_sections(freqs, sect);
    out[pos++] = (uint8_t)(ns == 256 ? 0 : ns);
    for (si = 0; si < ns; ++si) pos += put_packed_w(sect[si], out + pos);
    for (si = 0; si < ns; ++si) {
        wtree t;
        bitvec shape = {0, 0, 0};
        if (sect[si] == 0) continue;
        wtree_build(&t, bwt + beg, sect[si]);
        pos += put_packed_w(t.pool.v[t.root].bv.n, out + pos);              /* bitsInRoot */
        tree_shape(&t, &shape);
        for (i = 0; i < shape.n;) {
            uint8_t b = 0; size_t j = 0;
            for (; j < 8 && i < shape.n; ++i, ++j) b = (uint8_t)((b << 1) | shape.b[i]);
            if (j < 8) b = (uint8_t)(b << (8 - j));
            out[pos++] = b;
        }
        ws->rc.out = out; ws->rc.n = pos;
        encode_tree_bf(&t, &ws->rc, &ws->pm, &ws->gm, &ws->gapm);
        /* endContextBlock, WaveletCoders.cpp:62-68 */
        model_reset(&ws->pm); model_reset(&ws->gm); model_reset(&ws->gapm);
        rc_finish(&ws->rc);
        pos = ws->rc.n;
        beg += sect[si];
        bv_free(&shape);
        wtree_free(&t);
    }
    length = pos - 6;
    for (i = 0; i < 6; ++i) out[i] = (uint8_t)(length >> (8 * (5 - i)));
    return pos;
}

size_t orc_wavelet_encode_block_with(char coder, const uint8_t *bwt, uint32_t size, const uint32_t *lf,
                                     uint32_t n_lf, const uint32_t freqs[256], uint8_t *out, size_t out_cap)
{
    wstate ws;
    (void)out_cap;
    if (!model_init(coder, &ws.pm, &ws.gm, &ws.gapm)) return 0;
    ws.rc.low = 0; ws.rc.high = 0xFFFFFFFFu;
    return wavelet_encode_block(&ws, bwt, size, lf, n_lf, freqs, out);
}

size_t orc_wavelet_encode_block(const uint8_t *bwt, uint32_t size, const uint32_t *lf, uint32_t n_lf,
                                const uint32_t freqs[256], uint8_t *out, size_t out_cap)
{
    return orc_wavelet_encode_block_with('B', bwt, size, lf, n_lf, freqs, out, out_cap);
}

/* Compressor::compress with a wavelet coder and no precompression (Compressor.cpp:65-118);
 * the coder letter is the global header byte and selects the main model */
size_t orc_compress_wavelet(char coder, const uint8_t *in, size_t size, size_t block_size,
                            uint32_t starting_points, uint8_t *out, size_t out_cap)
{
    size_t pos = 0, off = 0;
    wstate ws;
    if (!model_init(coder, &ws.pm, &ws.gm, &ws.gapm)) return 0;
    ws.rc.low = 0; ws.rc.high = 0xFFFFFFFFu;
    (void)out_cap;
    out[pos++] = (uint8_t)coder;
    while (off < size) {
        size_t bs = size - off < block_size ? size - off : block_size;
        uint8_t *blk = (uint8_t *)malloc(bs + 1);
        uint32_t lf[256], n_lf = 0, freqs[256];
        memcpy(blk, in + off, bs);
        blk[bs] = 0;
        pos += put_packed_w(bs, out + pos);
        pos += put_packed_w(1, out + pos);
        out[pos++] = 0;
        memset(freqs, 0, sizeof freqs);
        orc_bwt_block(blk, (uint32_t)bs, starting_points, lf, &n_lf, freqs);
        pos += wavelet_encode_block(&ws, blk, (uint32_t)bs, lf, n_lf, freqs, out + pos);
        free(blk);
        off += bs;
    }
    out[pos++] = 0;
    return pos;
}

size_t orc_compress_B(const uint8_t *in, size_t size, size_t block_size, uint32_t starting_points,
                      uint8_t *out, size_t out_cap)
{
    return orc_compress_wavelet('B', in, size, block_size, starting_points, out, out_cap);
}
