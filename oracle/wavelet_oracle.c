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

/* one WaveletEncoder object for streams that are framed elsewhere (prepr_oracle.cpp) */
void *orc_wavelet_stream_new(char coder)
{
    wstate *ws = (wstate *)malloc(sizeof(wstate));
    if (!ws) return NULL;
    if (!model_init(coder, &ws->pm, &ws->gm, &ws->gapm)) { free(ws); return NULL; }
    ws->rc.low = 0; ws->rc.high = 0xFFFFFFFFu;
    return ws;
}
size_t orc_wavelet_stream_encode_block(void *ws, const uint8_t *bwt, uint32_t size, const uint32_t *lf,
                                       uint32_t n_lf, const uint32_t freqs[256], uint8_t *out)
{
    return wavelet_encode_block((wstate *)ws, bwt, size, lf, n_lf, freqs, out);
}
void orc_wavelet_stream_free(void *ws) { free(ws); }

size_t orc_compress_B(const uint8_t *in, size_t size, size_t block_size, uint32_t starting_points,
                      uint8_t *out, size_t out_cap)
{
    return orc_compress_wavelet('B', in, size, block_size, starting_points, out, out_cap);
}

/* ============================================================================================
 * DECODER.  Literal restatement of WaveletDecoder::decodeBlock / readBlockHeader
 * (WaveletCoders.cpp:231-306), WaveletTree::readShape (WaveletTree.hpp:403-500),
 * decodeTreeBF (:857-1165, OPTIMIZED_INTEGER_CODE + SEMI_FIXED_CODE branches), message
 * (:1277-1378), BitDecoder (BitCoders.cpp:115-148), the InStream bit reads (Streams.hpp:120-137)
 * and Decompressor::decompress (Decompressor.cpp:57-100, no grammar).  It shares NOTHING with
 * the encoder above except the models, the node pool and assignPrefixCodes -- the same things
 * the reference's decoder shares with its encoder -- so product streams that this function
 * decodes back to the input are streams the reference's algorithm accepts.
 * ========================================================================================== */
typedef struct { const uint8_t *p; size_t n, pos; uint16_t buffer; uint8_t bits_in_buffer; int overrun; } instream;
static unsigned in_fetch(instream *s) { if (s->pos < s->n) return s->p[s->pos++]; s->overrun = 1; ++s->pos; return 0; }
static int in_read_bit(instream *s)                                /* Streams.hpp:120-126 */
{
    if (s->bits_in_buffer == 0) { s->buffer = (uint16_t)in_fetch(s); s->bits_in_buffer = 8; }
    return (s->buffer >> --s->bits_in_buffer) & 1;
}
static unsigned in_read_byte(instream *s)                          /* Streams.hpp:128-133 */
{
    const unsigned next = in_fetch(s);
    s->buffer = (uint16_t)((s->buffer << 8) | next);
    return (s->buffer >> s->bits_in_buffer) & 0xffu;
}
static void in_flush(instream *s) { s->bits_in_buffer = 0; }       /* Streams.hpp:135-137 */
static uint64_t in_read48(instream *s) { uint64_t r = 0; int i; for (i = 0; i < 6; ++i) r = (r << 8) | in_fetch(s); return r; }   /* Streams.cpp:119-126 */

/* utils::readPackedIntegerRev, Utils.hpp:148-162 */
static size_t read_packed_rev(instream *in, size_t *bytes_read)
{
    size_t read = 0xff, result = 0, j = 0, i;
    *bytes_read = 0;
    while (read & 0x80) {
        read = 0;
        for (i = 0; i < 8; ++i) read |= ((size_t)in_read_bit(in) << i);
        result |= ((read & 0x7f) << j);
        j += 7;
        ++*bytes_read;
        if (in->overrun) break;
    }
    return result;
}
/* utils::readPackedInteger, Utils.hpp:164-178 (MSB-first bits) */
static size_t read_packed(instream *in)
{
    size_t read = 0xff, result = 0, j = 0; int i;
    while (read & 0x80) {
        read = 0;
        for (i = 7; i >= 0; --i) read |= ((size_t)in_read_bit(in) << i);
        result |= ((read & 0x7f) << j);
        j += 7;
        if (in->overrun) break;
    }
    return result;
}
static unsigned log_ceiling(size_t n) { unsigned l = 0; size_t v = 1; while (v < n) { v <<= 1; ++l; } return l; }   /* utils::logCeiling */
/* utils::binaryDecode(input, lo, hi, bitsRead), Utils.hpp:321-341 */
static size_t binary_decode(instream *in, size_t lo, size_t hi)
{
    const size_t range_len = hi - lo + 1;
    unsigned code_length; size_t short_cw, long_cw2, result = 0; int i;
    if (range_len == 1) return lo;
    code_length = log_ceiling(range_len);
    short_cw = ((size_t)1 << code_length) - range_len;
    long_cw2 = (range_len - short_cw) / 2;
    for (i = 0; i < (int)code_length - 1; ++i) result = (result << 1) | (size_t)in_read_bit(in);
    if (result >= long_cw2) return result + lo;
    result = (result << 1) | (size_t)in_read_bit(in);
    if (result < long_cw2) return result + lo;
    return result + lo + short_cw;
}
/* utils::binaryInterpolativeDecode(list, input, lo, hi, elements), Utils.hpp:343-362 */
static void interp_decode(uint32_t *list, size_t *n_list, instream *in, size_t lo, size_t hi, size_t elements)
{
    size_t h, r, mid;
    if (elements == 0 || in->overrun) return;
    if (elements == hi - lo + 1) { size_t i; for (i = lo; i <= hi; ++i) list[(*n_list)++] = (uint32_t)i; return; }
    h = (elements - 1) / 2;
    r = elements / 2 - h;
    mid = binary_decode(in, lo + h, hi - h - r);
    interp_decode(list, n_list, in, lo, mid - 1, h);
    list[(*n_list)++] = (uint32_t)mid;
    interp_decode(list, n_list, in, mid + 1, hi, elements - h - 1);
}
static size_t unary_decode(instream *in) { size_t n = 1; while (!in_read_bit(in)) { ++n; if (in->overrun) break; } return n; }   /* Utils.hpp:411-416 */

/* the decoder's tree: main pool (m_root...) + the integer code tree (m_integerCodeTree) in a pool of its own */
typedef struct {
    nodepool pool; int root;
    nodepool ipool; int iroot;            /* m_integerCodeTree */
    size_t n_integer_codes;               /* m_integerCodes.size() */
    uint32_t W;
} dtree;

static size_t count_leaves(const nodepool *p, int nd)
{
    size_t c = 0;
    if (p->v[nd].left < 0 && p->v[nd].right < 0) return 1;
    if (p->v[nd].left >= 0) c += count_leaves(p, p->v[nd].left);
    if (p->v[nd].right >= 0) c += count_leaves(p, p->v[nd].right);
    return c;
}

/* WaveletTree::readShape, WaveletTree.hpp:403-500.  Returns 0 on malformed input. */
static int read_shape(dtree *t, instream *in)
{
    size_t max_sym = in_read_byte(in), symbols = in_read_byte(in), bytes_read = 0, max_len, i, na = 0;
    uint32_t alphabet[257];
    pair cl[256];
    if (symbols == 0) symbols = 256;
    max_len = read_packed_rev(in, &bytes_read);
    if (symbols > max_sym + 1) return 0;
    interp_decode(alphabet, &na, in, 0, max_sym, symbols);
    if (na != symbols || in->overrun) return 0;
    for (i = 0; i < symbols; ++i) {
        const size_t n = unary_decode(in);
        if (n > max_len) return 0;
        cl[i].first = max_len - n + 1; cl[i].second = alphabet[i];
    }
    qsort(cl, symbols, sizeof(pair), cmp_pair);                    /* assignPrefixCodes(lengths) sorts, :1440-1444 */
    t->root = np_new(&t->pool, 0, 0);
    assign_prefix_codes(&t->pool, cl, symbols, t->root, 0, 0);
    {
        const size_t longest_run = read_packed_rev(in, &bytes_read);
        if (longest_run > 0) {
            uint32_t *integers; pair *icl; size_t ni = 0;
            symbols = read_packed_rev(in, &bytes_read);
            max_len = read_packed_rev(in, &bytes_read);
            if (in->overrun || symbols == 0 || symbols > longest_run + 1) return 0;
            integers = (uint32_t *)malloc((symbols + 1) * sizeof(uint32_t));
            icl = (pair *)malloc((symbols + 1) * sizeof(pair));
            interp_decode(integers, &ni, in, 0, longest_run, symbols);
            if (ni != symbols || in->overrun) { free(integers); free(icl); return 0; }
            for (i = 0; i < symbols; ++i) {
                const size_t n = unary_decode(in);
                if (n > max_len) { free(integers); free(icl); return 0; }
                icl[i].first = max_len + 1 - n; icl[i].second = integers[i];
            }
            qsort(icl, symbols, sizeof(pair), cmp_pair);
            t->iroot = np_new(&t->ipool, 0, 0);
            assign_prefix_codes(&t->ipool, icl, symbols, t->iroot, 0, 0);
            t->n_integer_codes = count_leaves(&t->ipool, t->iroot);   /* collectCodes: one map entry per leaf symbol */
            free(integers); free(icl);
        } else {
            t->iroot = np_new(&t->ipool, 1, 0);                    /* new TreeNode(0): a leaf with symbol 0 */
            t->n_integer_codes = 1;
        }
        t->W = 0;
        for (i = 0; i < 4; ++i) t->W = (t->W << 1) | (uint32_t)in_read_bit(in);
    }
    return !in->overrun;
}

/* BitDecoder, BitCoders.cpp:115-148 */
typedef struct { uint32_t low, high, next; instream *in; } rdecoder;
static void rd_start(rdecoder *d)
{
    d->low = 0; d->high = 0xFFFFFFFFu;
    d->next = in_read_byte(d->in);
    d->next = (d->next << 8) + in_read_byte(d->in);
    d->next = (d->next << 8) + in_read_byte(d->in);
    d->next = (d->next << 8) + in_read_byte(d->in);
}
static int rd_decode(rdecoder *d, uint16_t p1)
{
    const uint32_t range_size = d->high - d->low - 1;              /* Split(), BitCoders.cpp:40-57 */
    const uint32_t high_bits = range_size >> 12, low_bits = range_size & 4095u;
    const uint32_t split = d->low + high_bits * p1 + ((low_bits * p1 + 2048u) >> 12);
    const int bit = d->next <= split;
    if (bit) d->high = split; else d->low = split + 1;
    while (((d->low ^ d->high) & 0xFF000000u) == 0) {
        d->low <<= 8;
        d->high = (d->high << 8) + 255;
        d->next = (d->next << 8) + in_read_byte(d->in);
    }
    return bit;
}

/* IntegerNode (SEMI_FIXED_CODE form), WaveletTree.hpp:838-850.  int_node = -1 is the null pointer. */
typedef struct { int nd; int int_node; size_t bits; uint32_t leading_ones; uint8_t code_status; } intnode;
typedef struct { intnode *v; size_t n, cap, head; } intlist;
static void il_push(intlist *l, intnode x)
{
    if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 64; l->v = (intnode *)realloc(l->v, l->cap * sizeof(intnode)); }
    l->v[l->n++] = x;
}
static intnode mk_intnode(int nd, int int_node, size_t bits, uint32_t lo, uint8_t st)
{
    intnode x; x.nd = nd; x.int_node = int_node; x.bits = bits; x.leading_ones = lo; x.code_status = st; return x;
}

/* decodeTreeBF, WaveletTree.hpp:857-1165 */
static int decode_tree_bf(dtree *t, size_t root_size, rdecoder *dec, model *pm, model *gm, model *gapm)
{
    nodepool *p = &t->pool;
    const nodepool *ip = &t->ipool;
    inode *queue = NULL; size_t qh = 0, qt = 0, qcap = 0;
    intlist icn = {0, 0, 0, 0};
    size_t i;
#define QPUSH(x) do { if (qt == qcap) { qcap = qcap ? qcap * 2 : 64; queue = (inode *)realloc(queue, qcap * sizeof(inode)); } queue[qt++] = (x); } while (0)
    if (root_size == 0) return 0;
    {   /* root, :874-924 */
        inode left, right;
        int prev;
        memset(&left, 0, sizeof left); memset(&right, 0, sizeof right);
        prev = rd_decode(dec, model_p1(pm));
        model_update(pm, prev);
        bv_push(&p->v[t->root].bv, prev);
        if (prev) bv_push(&right.gaps, 1); else bv_push(&left.gaps, 1);
        for (i = 1; i < root_size; ++i) {
            const int bit = rd_decode(dec, model_p1(pm));
            model_update(pm, bit);
            bv_push(&p->v[t->root].bv, bit);
            bv_push(bit ? &right.gaps : &left.gaps, prev != bit);
            prev = bit;
            if (dec->in->overrun) return 0;
        }
        if (p->v[t->root].left < 0) return 0;                      /* "Left node has to always exist" */
        if (p->v[p->v[t->root].left].has_symbol) { il_push(&icn, mk_intnode(p->v[t->root].left, t->iroot, left.gaps.n, 0, 0)); bv_free(&left.gaps); }
        else { left.nd = p->v[t->root].left; QPUSH(left); }
        if (right.gaps.n > 0) {
            if (p->v[t->root].right < 0) return 0;
            if (p->v[p->v[t->root].right].has_symbol) { il_push(&icn, mk_intnode(p->v[t->root].right, t->iroot, right.gaps.n, 0, 0)); bv_free(&right.gaps); }
            else { right.nd = p->v[t->root].right; QPUSH(right); }
        } else bv_free(&right.gaps);
    }
    while (qh < qt) {                                              /* internal nodes, :927-1021 */
        inode left, right, node = queue[qh];
        const int nl = p->v[node.nd].left, nr = p->v[node.nd].right;
        memset(&left, 0, sizeof left); memset(&right, 0, sizeof right);
        model_reset(pm); model_reset(gapm);
        if (nl < 0 || nr < 0) return 0;                            /* "Node must have both left and right child" */
        if (p->v[nl].has_symbol || p->v[nr].has_symbol) {
            if (p->v[nl].has_symbol && p->v[nr].has_symbol) {
                size_t ones = 0; int prev = 1;
                for (i = 0; i < node.gaps.n; ++i) {
                    if (!node.gaps.b[i]) prev = !prev;
                    else { prev = rd_decode(dec, model_p1(gapm)); model_update(gapm, prev); }
                    bv_push(&p->v[node.nd].bv, prev);
                    if (prev) ++ones;
                }
                il_push(&icn, mk_intnode(nl, t->iroot, node.gaps.n - ones, 0, 0));
                il_push(&icn, mk_intnode(nr, t->iroot, ones, 0, 0));
            } else {
                int prev = 1;
                if (p->v[nr].has_symbol) return 0;                 /* assert(!node.first->m_right->m_hasSymbol) */
                for (i = 0; i < node.gaps.n; ++i) {
                    int bit;
                    if (!node.gaps.b[i] && !prev) bit = 1;
                    else if (node.gaps.b[i]) { bit = rd_decode(dec, model_p1(gapm)); model_update(gapm, bit); model_update_state(pm, bit); }
                    else { bit = rd_decode(dec, model_p1(pm)); model_update(pm, bit); }
                    bv_push(&p->v[node.nd].bv, bit);
                    if (bit) bv_push(&right.gaps, prev != bit || node.gaps.b[i]);
                    prev = bit;
                }
                il_push(&icn, mk_intnode(nl, t->iroot, node.gaps.n - right.gaps.n, 0, 0));
                right.nd = nr; QPUSH(right);
            }
        } else {
            int prev = 1;
            for (i = 0; i < node.gaps.n; ++i) {
                int bit;
                if (node.gaps.b[i]) { bit = rd_decode(dec, model_p1(gapm)); model_update(gapm, bit); model_update_state(pm, bit); }
                else { bit = rd_decode(dec, model_p1(pm)); model_update(pm, bit); }
                bv_push(&p->v[node.nd].bv, bit);
                bv_push(bit ? &right.gaps : &left.gaps, prev != bit || node.gaps.b[i]);
                prev = bit;
            }
            left.nd = nl; QPUSH(left);
            right.nd = nr; QPUSH(right);
        }
        bv_free(&queue[qh].gaps);
        ++qh;
        if (dec->in->overrun) return 0;
    }
    {   /* integer-code nodes, :1024-1163 */
        intlist lefts = {0, 0, 0, 0}, rights = {0, 0, 0, 0};
        while (icn.head < icn.n || lefts.n || rights.n) {
            model_reset(gm);
            while (icn.head < icn.n) {
                intnode node = icn.v[icn.head++];
                size_t ones = 0;
                if (node.int_node >= 0 && ip->v[node.int_node].has_symbol) {
                    p->v[node.nd].has_symbol = 1;
                    if (t->n_integer_codes > 1) p->v[node.nd].symbol = ip->v[node.int_node].symbol;
                    if (ip->v[node.int_node].symbol != 0) continue;
                }
                for (i = 0; i < node.bits; ++i) {
                    const int bit = rd_decode(dec, model_p1(gm));
                    model_update(gm, bit);
                    if (bit) ++ones;
                    bv_push(&p->v[node.nd].bv, bit);
                }
                if (dec->in->overrun) return 0;
                if (node.bits > ones) {
                    intnode ln;
                    const int il = node.int_node >= 0 ? ip->v[node.int_node].left : -1;
                    if (p->v[node.nd].left < 0) { int c = np_new(p, 0, 0); p->v[node.nd].left = c; }
                    ln = mk_intnode(p->v[node.nd].left, il, node.bits - ones, node.leading_ones, node.code_status);
                    if (node.int_node < 0 || il < 0) {
                        if (node.code_status == 0) { ln.code_status = 2; ln.leading_ones = t->W; }
                        else if (node.code_status == 1) { ln.code_status = 2; ln.leading_ones += t->W; }
                        else --ln.leading_ones;
                    }
                    if (ln.code_status != 2 || ln.leading_ones > 0) il_push(&lefts, ln);
                }
                if (ones > 0) {
                    intnode rn;
                    const int ir = node.int_node >= 0 ? ip->v[node.int_node].right : -1;
                    if (p->v[node.nd].right < 0) { int c = np_new(p, 0, 0); p->v[node.nd].right = c; }
                    rn = mk_intnode(p->v[node.nd].right, ir, ones, node.leading_ones, node.code_status);
                    if (node.int_node < 0 || ir < 0) {
                        if (node.code_status == 0) { rn.code_status = 1; ++rn.leading_ones; }
                        else if (node.code_status == 1) ++rn.leading_ones;
                        else --rn.leading_ones;
                    }
                    if (rn.code_status != 2 || rn.leading_ones > 0) il_push(&rights, rn);
                }
            }
            for (i = 0; i < lefts.n; ++i) il_push(&icn, lefts.v[i]);    /* splice(end, left); splice(end, right) */
            for (i = 0; i < rights.n; ++i) il_push(&icn, rights.v[i]);
            lefts.n = rights.n = 0;
        }
        free(lefts.v); free(rights.v);
    }
    free(icn.v); free(queue);
#undef QPUSH
    return 1;
}

/* message, WaveletTree.hpp:1277-1378.  bitsSeen (a std::map keyed by node) is an array over the
 * pool.  Returns the bytes written, or (size_t)-1 when the tree does not describe `cap` bytes. */
static size_t tree_message(const dtree *t, uint8_t *out, size_t cap)
{
    const nodepool *p = &t->pool;
    size_t *seen = (size_t *)calloc((size_t)p->n + 1, sizeof(size_t));
    const size_t msg_size = p->v[t->root].bv.n;
    size_t len = 0, j, k;
#define BITAT(nd, i) ((nd) < 0 || (i) >= p->v[nd].bv.n ? (bad = 1, 0) : p->v[nd].bv.b[i])
    int bad = 0;
    for (j = 0; j < msg_size && !bad; ++j) {
        int bit, nd = t->root;
        size_t i = j, run_length = 0;
        uint8_t symbol;
        do {
            bit = BITAT(nd, i);
            if (bit) i = seen[nd]++; else i = i - seen[nd];
            nd = bit ? p->v[nd].right : p->v[nd].left;
        } while (!bad && nd >= 0 && !p->v[nd].has_symbol);
        if (bad || nd < 0) { bad = 1; break; }
        symbol = (uint8_t)p->v[nd].symbol;
        if (!t->ipool.v[t->iroot].has_symbol) {                    /* not plain gamma codes */
            do {
                bit = BITAT(nd, i);
                if (bit) { i = seen[nd]++; nd = p->v[nd].right; }
                else { i = i - seen[nd]; nd = p->v[nd].left; }
            } while (!bad && nd >= 0 && !p->v[nd].has_symbol);
            if (bad || nd < 0) { bad = 1; break; }
            run_length = p->v[nd].symbol;
        }
        if (run_length == 0) {
            size_t leading_ones = 0;
            bit = BITAT(nd, i);
            while (bit && !bad) {
                ++leading_ones;
                i = seen[nd]++;
                nd = bit ? p->v[nd].right : p->v[nd].left;
                bit = BITAT(nd, i);
            }
            for (k = 0; k < leading_ones + t->W && !bad; ++k) {
                run_length <<= 1;
                if (bit) i = seen[nd]++; else i = i - seen[nd];
                nd = bit ? p->v[nd].right : p->v[nd].left;
                bit = BITAT(nd, i);
                run_length |= (size_t)(bit ? 1 : 0);
            }
            run_length += 1 + ((((size_t)1 << leading_ones) - 1) << t->W);   /* fixedIntegerCodeTranslation */
        }
        if (bad || len + run_length > cap) { bad = 1; break; }
        memset(out + len, symbol, run_length);
        len += run_length;
    }
#undef BITAT
    free(seen);
    return bad ? (size_t)-1 : len;
}

static void dtree_free(dtree *t)
{
    int i;
    for (i = 0; i < t->pool.n; ++i) bv_free(&t->pool.v[i].bv);
    free(t->pool.v);
    for (i = 0; i < t->ipool.n; ++i) bv_free(&t->ipool.v[i].bv);
    free(t->ipool.v);
}

/* WaveletDecoder::readPackedInteger + utils::unpackInteger, WaveletCoders.cpp:290-306, Utils.cpp:103-118 */
static uint64_t read_packed_bytes(instream *in)
{
    uint64_t v = 0; int shift = 0, more = 1;
    while (more && !in->overrun) {
        const unsigned b = in_read_byte(in);
        more = (b & 0x80) != 0;
        v |= (uint64_t)(b & 0x7f) << shift;
        shift += 7;
    }
    return v;
}

/* Decompressor::decompress (Decompressor.cpp:57-100) over a stream made by a wavelet coder: global
 * header letter -> model (giveEntropyDecoder / giveProbabilityModel), precompressor block headers
 * with an empty grammar, WaveletDecoder::decodeBlock per slice, inverse transform.  Returns the
 * decoded size, or (size_t)-1 on malformed input.  If bwt_out is not NULL the transformed bytes
 * of the blocks (before the inverse transform) are copied there as well. */
size_t orc_decompress_wavelet(const uint8_t *in_bytes, size_t in_size, uint8_t *out, size_t out_cap, uint8_t *bwt_out)
{
    instream in;
    model pm, gm, gapm;
    rdecoder dec;
    size_t opos = 0, bpos = 0;
    memset(&in, 0, sizeof in);
    in.p = in_bytes; in.n = in_size;
    if (in_size < 2) return (size_t)-1;
    if (!model_init((char)in_read_byte(&in), &pm, &gm, &gapm)) return (size_t)-1;
    dec.in = &in;
    for (;;) {
        const size_t orig = read_packed(&in);
        size_t slices, s;
        if (in.overrun) return (size_t)-1;
        if (orig == 0) break;
        const uint8_t *gram;
        size_t gram_n, dpos = 0, dcap;
        uint8_t *aside = NULL, *dst;
        slices = read_packed(&in);
        /* the grammar (PrecompressorBlock.cpp:97-108): one 0x00 byte when the pre-stage was off; with rules the
         * slices are decoded aside and expanded by the Postprocessor (Decompressor.cpp:82-88) */
        gram_n = in.pos <= in.n ? orc_grammar_span(in.p + in.pos, in.n - in.pos) : (size_t)-1;
        if (gram_n == (size_t)-1) return (size_t)-1;
        gram = in.p + in.pos;
        in.pos += gram_n;
        dst = out + opos; dcap = out_cap - opos;
        if (!(gram_n == 1 && gram[0] == 0)) {
            dcap = 2 * orig + 64;
            aside = (uint8_t *)malloc(dcap);
            if (!aside) return (size_t)-1;
            dst = aside;
        }
        for (s = 0; s < slices; ++s) {
            const uint64_t clen48 = in_read48(&in);                /* readBlockHeader, :231-244 */
            const size_t start = in.pos;
            uint32_t n_lf = in_read_byte(&in) + 1, lf[256], i, nsec, j;
            uint64_t sect[256];
            size_t bsize = 0, len = 0;
            uint8_t *blk = dst + dpos;
            for (i = 0; i < n_lf; ++i) { uint32_t pos = 0; for (j = 0; j < 31; ++j) pos = (pos << 1) | (uint32_t)in_read_bit(&in); lf[i] = pos; }
            in_flush(&in);                                         /* BWTBlock::readHeader, BWTBlock.cpp:88-102 */
            nsec = in_read_byte(&in); if (nsec == 0) nsec = 256;
            for (i = 0; i < nsec; ++i) { sect[i] = read_packed_bytes(&in); bsize += sect[i]; }
            if (in.overrun || dpos + bsize > dcap) { free(aside); return (size_t)-1; }
            for (i = 0; i < nsec; ++i) {                           /* decodeBlock, :246-288 */
                dtree t;
                size_t root_size, clen;
                int ok;
                if (sect[i] == 0) continue;
                root_size = (size_t)read_packed_bytes(&in);
                memset(&t, 0, sizeof t);
                ok = read_shape(&t, &in);
                if (ok) {
                    in_flush(&in);
                    rd_start(&dec);
                    ok = decode_tree_bf(&t, root_size, &dec, &pm, &gm, &gapm);
                }
                clen = ok ? tree_message(&t, blk + len, bsize - len) : (size_t)-1;
                dtree_free(&t);
                if (clen == (size_t)-1 || clen != sect[i]) return (size_t)-1;
                len += clen;
                model_reset(&pm); model_reset(&gm); model_reset(&gapm);   /* endContextBlock, :70-75 */
            }
            if (len != bsize || in.pos - start != clen48) return (size_t)-1;
            if (bwt_out) { memcpy(bwt_out + bpos, blk, bsize); bpos += bsize; }
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

/* ============================================================================================
 * Pieces the reference's own unit tests pin with known answers (test/WaveletTest.cpp:46-74
 * HeapTest1-2, :116-198 HuffmanShape1-4, :427-449 GammaCodes), restated so that those answers
 * (tests/golden/wavelet_known_answers.json) can be checked here.  createHuffmanShape is NOT what
 * the WaveletTree constructor uses (that is calculateHuffmanLengths + assignPrefixCodes, above);
 * the shapes' DEPTHS are what both must agree on, and tests/test_oracle.py checks exactly that.
 * ========================================================================================== */
/* MinimumHeap<T>, WaveletTree.hpp:1386-1438: (value, weight) pairs, sift-up on insert with a
 * strict '>' test, heapify preferring the RIGHT child on equal weights ((wl < wr) ? left : right) */
typedef struct { int value; size_t weight; } hpair;
typedef struct { hpair *v; size_t n, cap; } minheap;
static void mh_insert(minheap *h, int value, size_t weight)
{
    size_t index = h->n;
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 16; h->v = (hpair *)realloc(h->v, h->cap * sizeof(hpair)); }
    ++h->n;
    while (index > 0 && h->v[(index - 1) / 2].weight > weight) { h->v[index] = h->v[(index - 1) / 2]; index = (index - 1) / 2; }
    h->v[index].value = value; h->v[index].weight = weight;
}
static void mh_heapify(minheap *h, size_t index)
{
    const size_t l = 2 * index + 1, r = 2 * index + 2;
    if (h->n > r) {
        const size_t smaller = (h->v[l].weight < h->v[r].weight) ? l : r;
        if (h->v[smaller].weight < h->v[index].weight) { hpair t = h->v[index]; h->v[index] = h->v[smaller]; h->v[smaller] = t; mh_heapify(h, smaller); }
    } else if (h->n == r && h->v[l].weight < h->v[index].weight) {
        hpair t = h->v[index]; h->v[index] = h->v[l]; h->v[l] = t;
    }
}
static hpair mh_delete_min(minheap *h)
{
    hpair min = h->v[0];
    h->v[0] = h->v[h->n - 1];
    --h->n;
    mh_heapify(h, 0);
    return min;
}
/* inserts (values[i], weights[i]) in order, then deletes everything: out[] = values in deletion order */
void orc_min_heap_order(const int *values, const uint64_t *weights, size_t n, int *out)
{
    minheap h = {0, 0, 0}; size_t i;
    for (i = 0; i < n; ++i) mh_insert(&h, values[i], (size_t)weights[i]);
    for (i = 0; i < n; ++i) out[i] = mh_delete_min(&h).value;
    free(h.v);
}
static void shape_walk(const nodepool *p, int nd, int depth, uint8_t *path, uint32_t *symbols, uint32_t *depths,
                       uint8_t *code_bits, uint32_t *code_len, size_t *n)
{                                                                  /* checkHuffmanShape / collectCodes order: left first */
    if (p->v[nd].left < 0 && p->v[nd].right < 0) {
        symbols[*n] = p->v[nd].symbol; depths[*n] = (uint32_t)depth;
        memcpy(code_bits + 64 * p->v[nd].symbol, path, (size_t)depth); code_len[p->v[nd].symbol] = (uint32_t)depth;
        ++*n; return;
    }
    if (p->v[nd].left >= 0) { path[depth] = 0; shape_walk(p, p->v[nd].left, depth + 1, path, symbols, depths, code_bits, code_len, n); }
    if (p->v[nd].right >= 0) { path[depth] = 1; shape_walk(p, p->v[nd].right, depth + 1, path, symbols, depths, code_bits, code_len, n); }
}
/* createHuffmanShape(runFreqs) + collectCodes, WaveletTree.hpp:1496-1550: leaves left to right
 * (symbols[], depths[]), and code_bits[64 * symbol + i] / code_len[symbol].  Returns the leaves. */
size_t orc_create_huffman_shape(const uint64_t run_freqs[256], uint32_t *symbols, uint32_t *depths,
                                uint8_t *code_bits, uint32_t code_len[256])
{
    nodepool p = {0, 0, 0}; minheap h = {0, 0, 0};
    uint8_t path[300]; size_t n = 0; int b, root;
    memset(code_len, 0, 256 * sizeof(uint32_t));
    for (b = 0; b < 256; ++b) if (run_freqs[b] != 0) mh_insert(&h, np_new(&p, 1, (uint32_t)b), (size_t)run_freqs[b]);
    if (h.n == 0) { free(p.v); free(h.v); return 0; }
    if (h.n == 1) {                                                /* TreeNode(left = the only leaf, right = 0) */
        const int leaf = mh_delete_min(&h).value, r = np_new(&p, 0, 0);
        p.v[r].left = leaf;
        mh_insert(&h, r, 1);
    }
    while (h.n > 1) {
        const hpair p1 = mh_delete_min(&h), p2 = mh_delete_min(&h);
        const int nd = np_new(&p, 0, 0);
        p.v[nd].left = p2.value; p.v[nd].right = p1.value;        /* TreeNode(p2.first, p1.first) */
        mh_insert(&h, nd, p1.weight + p2.weight);
    }
    root = mh_delete_min(&h).value;
    shape_walk(&p, root, 0, path, symbols, depths, code_bits, code_len, &n);
    free(p.v); free(h.v);
    return n;
}
/* gammaCode, WaveletTree.hpp:541-551 (one bit per byte) */
size_t orc_gamma_code(size_t integer, uint8_t *bits)
{
    const int lg = (int)log_floor64(integer); size_t n = 0; int i;
    for (i = 0; i < lg; ++i) bits[n++] = 1;
    bits[n++] = 0;
    for (i = lg - 1; i >= 0; --i) bits[n++] = (uint8_t)((integer >> i) & 1);
    return n;
}
/* fixedIntegerCode(bits, x, w), the code the SEMI_FIXED_CODE path actually pushes (:514-526);
 * "when W=0, the code is standard gamma-code" (:512) */
size_t orc_fixed_integer_code(uint32_t x, uint32_t w, uint8_t *bits)
{
    bitvec v = {0, 0, 0}; size_t n;
    fixed_integer_code(&v, x, w);
    n = v.n; memcpy(bits, v.b, n); bv_free(&v);
    return n;
}
/* the code lengths the WaveletTree constructor gives the symbols (calculateHuffmanLengths +
 * assignPrefixCodes, :302-307): code_bits[64 * symbol + i] / code_len[symbol] */
void orc_wavelet_symbol_codes(const uint64_t run_freqs[256], uint8_t *code_bits, uint32_t code_len[256])
{
    pair cl[256]; size_t n_sym = 0, i; nodepool p = {0, 0, 0}; int root;
    bitvec codes[256], path = {0, 0, 0};
    memset(codes, 0, sizeof codes); memset(code_len, 0, 256 * sizeof(uint32_t));
    for (i = 0; i < 256; ++i) if (run_freqs[i]) { cl[n_sym].first = run_freqs[i]; cl[n_sym].second = (uint32_t)i; ++n_sym; }
    if (!n_sym) return;
    calculate_code_lengths(cl, n_sym, 0);
    qsort(cl, n_sym, sizeof(pair), cmp_pair);
    root = np_new(&p, 0, 0);
    assign_prefix_codes(&p, cl, n_sym, root, 0, 0);
    collect_codes_sym(&p, root, &path, codes);
    for (i = 0; i < 256; ++i) { code_len[i] = (uint32_t)codes[i].n; if (codes[i].n) memcpy(code_bits + 64 * i, codes[i].b, codes[i].n); bv_free(&codes[i]); }
    bv_free(&path); free(p.v);
}
