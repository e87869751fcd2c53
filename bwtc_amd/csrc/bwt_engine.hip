// BwtEngine implementation: kernels + host orchestration.  See bwt_engine.hpp.
#include "bwt_engine.hpp"
#include "radix_sort.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <ctime>
#include <sys/prctl.h>

namespace bwtc_hip {

// ---------------------------------------------------------------------------------------
// K1  load (+reverse) + 256-bin histogram.  T[j] = src[ncopy-1-j] (reverse) or src[j] for
//     j < ncopy; T[ncopy..padded) = 0.  The histogram is over the ncopy source bytes
//     (BWTransform.cpp:53-55 + divsufsort.c:506-512: the sentinel is never counted).
//     One workgroup = 4096 destination bytes: the source window is staged in LDS with
//     16-byte loads, every thread assembles one aligned 16-byte store.  Histogram in LDS,
//     16 interleaved copies per bin (copy = lane & 15) so that skewed inputs (text, DNA,
//     all-equal blocks) do not serialise on one address; one global atomic per non-empty
//     bin per workgroup.
// ---------------------------------------------------------------------------------------
constexpr int kLoadTPB = 256;
constexpr int kLoadTile = kLoadTPB * 16;
constexpr u32 kTextPad = 128;  // zero bytes guaranteed after T[n-1] (a code key looks at up to 74 characters of a suffix)

__global__ __launch_bounds__(kLoadTPB) void k_load_hist(const u8* __restrict__ src,
                                                        u8* __restrict__ T, u32 ncopy,
                                                        u32 padded, int reverse, int aligned,
                                                        u32* __restrict__ freqs) {
  __shared__ __attribute__((aligned(16))) u8 s_src[kLoadTile + 48];   // + slack: the fast path reads one word past its 16 bytes
  __shared__ u32 s_hist[256 * 16];
  const u32 tid = threadIdx.x;
  for (u32 i = tid; i < 256 * 16; i += kLoadTPB) s_hist[i] = 0;
  // A workgroup walks many tiles and flushes its histogram once: one flush per 4 KiB tile was
  // 65 000 global atomics on each of the 256 counters for a 256 MiB block, and that serialisation
  // -- not the bytes -- was this kernel's time.
  const u32 ntiles = (padded + (u32)kLoadTile - 1u) / (u32)kLoadTile;
  for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  __syncthreads();                                       // the previous tile's readers of s_src are done
  const u32 d0 = tile * kLoadTile;                       // first destination byte
  const u32 dend = min(d0 + (u32)kLoadTile, padded);     // one past the last one
  // source bytes this tile needs: [a_lo, a_hi)
  u32 a_lo = 0, a_hi = 0;
  if (d0 < ncopy) {
    const u32 last = min(dend, ncopy);                   // destinations [d0, last) have a source
    if (reverse) { a_lo = ncopy - last; a_hi = ncopy - d0; }
    else { a_lo = d0; a_hi = last; }
  }
  const u32 w_lo = a_lo & ~15u;
  if (a_hi > a_lo) {
    const u32 nchunks = (a_hi - w_lo + 15u) / 16u;
    for (u32 c = tid; c < nchunks; c += kLoadTPB) {
      const u32 a = w_lo + 16u * c;
      if (aligned && a + 16u <= ncopy) {
        *reinterpret_cast<uint4*>(&s_src[16u * c]) = *reinterpret_cast<const uint4*>(src + a);
      } else {
        for (u32 b = 0; b < 16u; ++b) s_src[16u * c + b] = (a + b < ncopy) ? src[a + b] : (u8)0;
      }
    }
  }
  __syncthreads();
  const u32 j0 = d0 + tid * 16u;
  if (j0 < dend) {
    u32 wds[4] = {0, 0, 0, 0};
    const u32 copy = tid & 15u;
    if (j0 + 16u <= ncopy) {
      // all sixteen bytes have a source: five aligned LDS words, shifted into place (byte reads of
      // LDS made this kernel latency bound); reversed = the same bytes, words and bytes swapped
      const u32 off = (reverse ? (ncopy - 16u - j0) : j0) - w_lo;
      const u32* s32 = reinterpret_cast<const u32*>(s_src) + (off >> 2);
      const u32 sh = (off & 3u) * 8u;
      u32 d[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) d[q] = s32[q];
      u32 w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = sh ? ((d[q] >> sh) | (d[q + 1] << (32u - sh))) : d[q];
#pragma unroll
      for (int q = 0; q < 4; ++q) wds[q] = reverse ? __builtin_bswap32(w[3 - q]) : w[q];
#pragma unroll
      for (int e = 0; e < 16; ++e) atomicAdd(&s_hist[((wds[e >> 2] >> (8 * (e & 3))) & 255u) * 16u + copy], 1u);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const u32 j = j0 + e;
        if (j < ncopy) {
          const u32 a = reverse ? (ncopy - 1u - j) : j;
          const u32 c = s_src[a - w_lo];
          wds[e >> 2] |= c << (8 * (e & 3));
          atomicAdd(&s_hist[c * 16u + copy], 1u);
        }
      }
    }
    *reinterpret_cast<uint4*>(T + j0) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  }
  }                                                      // tiles
  __syncthreads();
  u32 c = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) c += s_hist[tid * 16u + r];
  if (c) atomicAdd(&freqs[tid], c);
}

// ---------------------------------------------------------------------------------------
// K2  initial keys.  The alphabet is remapped to dense codes 0..sigma-1 (order preserving,
//     byte 0 -> code 0) and the key of suffix i is its first k characters as a base-sigma
//     number, most significant first (sigma^k <= 2^bits of K; mixed-radix packing wastes no
//     fractional bits: 10 characters of a 45-symbol text fit 55 bits = 7 radix passes);
//     characters past the end read the zero padding.  Slot j holds suffix
//     i = n-1-j (descending!).  Feeding the stable sort in descending suffix order puts,
//     inside every group of equal keys, the suffixes shorter than k first and
//     shortest-first -- exactly "proper prefix sorts first" -- without spending key bits.
// ---------------------------------------------------------------------------------------
// Dense codes of T[i_base .. i_base + count) into s_code, for a workgroup of 256 threads and
// count <= 2040: aligned words of T, at most two per thread and both in flight, instead of a loop
// of byte loads (each a memory latency of its own).  T is zero padded (and its region longer still).
__device__ __forceinline__ void stage_codes(const u8* __restrict__ T, const u8* s_lut, u8* s_code,
                                            u32 i_base, u32 count) {
  const u32 a0 = i_base & ~3u, sh = i_base - a0;
  const u32 nw = (sh + count + 3u) / 4u;
  u32 w[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const u32 x = threadIdx.x + 256u * e;
    w[e] = x < nw ? *reinterpret_cast<const u32*>(T + a0 + 4u * x) : 0u;
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const u32 x = threadIdx.x + 256u * e;
    if (x < nw) {
#pragma unroll
      for (u32 b = 0; b < 4u; ++b) {
        const u32 t = 4u * x + b - sh;                    // wraps for the bytes before i_base
        if (t < count) s_code[t] = s_lut[(w[e] >> (8u * b)) & 255u];
      }
    }
  }
}

template <typename K>
__global__ __launch_bounds__(256) void k_make_keys(const u8* __restrict__ T,
                                                   const u8* __restrict__ lut,
                                                   K* __restrict__ keys, u32* __restrict__ idx,
                                                   u32 n, int k, u32 sigma, K top, u8* __restrict__ plane,
                                                   int idx_hi) {
  // A workgroup makes 1024 consecutive slots.  The dense codes of its stretch of T are staged in
  // LDS once; a thread builds four consecutive suffixes' keys -- the first from k characters, the
  // next three by rolling one character out and one in (top = sigma^(k-1)), k+3 LDS bytes instead
  // of 4k -- and parks them in LDS, from where consecutive lanes write consecutive slots (a thread
  // storing its own four keys would touch a cache line per lane).
  constexpr u32 kTile = 1024;
  __shared__ u8 s_lut[256];
  __shared__ u8 s_code[kTile + 64];
  __shared__ K s_key[kTile];
  s_lut[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const u32 J0 = blockIdx.x * kTile;                // first slot of the tile
  if (J0 >= n) return;
  const u32 cntw = min(kTile, n - J0);              // its slots
  const u32 i_base = n - J0 - cntw;                 // smallest suffix of the tile (its last slot)
  stage_codes(T, s_lut, s_code, i_base, cntw + (u32)k);
  __syncthreads();
  const u32 o = 4u * threadIdx.x;                   // this thread's suffixes: i_base + o .. + 3
  if (o < cntw) {
    const u32 cnt = min(4u, cntw - o);
    K key = 0;
    for (int t = 0; t < k; ++t) key = (K)(key * (K)sigma) + (K)s_code[o + t];
    for (u32 s = 0; s < cnt; ++s) {
      s_key[cntw - 1u - (o + s)] = key;             // slot order is descending suffix order
      key = (K)((key - (K)s_code[o + s] * top) * (K)sigma) + (K)s_code[o + s + k];
    }
  }
  __syncthreads();
  for (u32 j = threadIdx.x; j < cntw; j += 256u) {
    K key = s_key[j];
    // idx_hi: the upper bits of the suffix number ride in bits 48.. of the (at most 48-bit) key
    if (sizeof(K) == 8 && idx_hi) key |= (K)((u64)((n - 1u - (J0 + j)) >> 16) << 48);
    keys[J0 + j] = key;
    if (idx) idx[J0 + j] = n - 1u - (J0 + j);        // nullptr: the sort's first pass makes the values up
    if (plane) plane[J0 + j] = (u8)s_key[j];         // the first radix pass's digits (radix_sort.hpp, planes)
  }
}

// ---------------------------------------------------------------------------------------
// K2g initial keys from dense g-gram codes.  A text uses few of the sigma^g possible g-grams
//     (the 256 MiB synthetic text: 40 symbols, 2.56 M possible 4-grams, some ten thousand
//     present), so numbering the PRESENT grams in order packs g characters into
//     b = log2(present) bits instead of g log2(sigma): 12 characters of that text take 48 key bits
//     = 6 radix passes instead of 64 bits = 8 passes, and the order of the keys is the same
//     (the numbering is monotone in the gram's value, which is its characters as a base-sigma
//     number).  Small passes before the keys: mark the grams that occur in a byte map, pack it
//     into a bitmap, and one workgroup turns that into a rank table (word, grams before it) --
//     640 KB for the text alphabet's 4-grams, resident in L2, where a table of codes indexed by
//     the gram's value spread the few ten thousand present grams over 3.5 MB of cache lines; the
//     key maker numbers a gram with one 16-byte lookup and a popcount.  Grams that start in the
//     padding are the all-zero gram, which position n-1 (the terminator and padding) marks.
// ---------------------------------------------------------------------------------------
constexpr u32 kGramMaxU = 1u << 22;         // largest gram universe: 64 Ki bitmap words, a table of 1 MiB
constexpr u32 kGramTile = 4096;
constexpr u32 kGramMinN = 1u << 22;         // smaller blocks keep the plain base-sigma key

// Marks the grams that occur in a byte map: a look at the byte, a plain store of 1 where it is
// still 0 (every writer stores the same value; a bitmap would need an atomic OR per new bit, and
// with the lookups served by each CU's own L1 the same hot words were OR-ed from everywhere:
// three times slower).
__global__ __launch_bounds__(256) void k_gram_mark(const u8* __restrict__ T, const u8* __restrict__ lut,
                                                   u32 n, int g, u32 sigma, u32 top,
                                                   u8* __restrict__ present) {
  __shared__ u8 s_lut[256];
  __shared__ __attribute__((aligned(16))) u8 s_code[kGramTile + 64];
  s_lut[threadIdx.x] = lut[threadIdx.x];
  const u32 ntiles = (n + kGramTile - 1u) / kGramTile;
  for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    const u32 base = tile * kGramTile;
    {
      const uint4 v = *reinterpret_cast<const uint4*>(T + base + 16u * threadIdx.x);   // T's region is padded by a tile
      const u32 w[4] = {v.x, v.y, v.z, v.w};
      u32 c[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        c[q] = (u32)s_lut[w[q] & 255u] | (u32)s_lut[(w[q] >> 8) & 255u] << 8 |
               (u32)s_lut[(w[q] >> 16) & 255u] << 16 | (u32)s_lut[w[q] >> 24] << 24;
      *reinterpret_cast<uint4*>(&s_code[16u * threadIdx.x]) = make_uint4(c[0], c[1], c[2], c[3]);
      if (threadIdx.x < 64u) s_code[kGramTile + threadIdx.x] = s_lut[T[base + kGramTile + threadIdx.x]];
    }
    __syncthreads();
    const u32 o = 16u * threadIdx.x;
    if (base + o < n) {
      u32 v = 0;
      for (int t = 0; t < g; ++t) v = v * sigma + s_code[o + t];
      const u32 cnt = min(16u, n - (base + o));
      u32 vs[16];
      u8 seen[16];
#pragma unroll
      for (u32 e = 0; e < 16u; ++e) {
        vs[e] = e < cnt ? v : 0u;
        v = (v - (u32)s_code[o + e] * top) * sigma + s_code[o + e + g];
      }
#pragma unroll
      for (u32 e = 0; e < 16u; ++e) seen[e] = present[vs[e]];     // sixteen independent lookups, then the few stores
#pragma unroll
      for (u32 e = 0; e < 16u; ++e) if (e < cnt && !seen[e]) present[vs[e]] = 1;
    }
  }
}

// 64 bytes of the map (each 0 or 1) -> one bitmap word
__global__ __launch_bounds__(256) void k_gram_pack(const u8* __restrict__ present, u32 nwords,
                                                   unsigned long long* __restrict__ bits) {
  const u32 w = blockIdx.x * 256u + threadIdx.x;
  if (w >= nwords) return;                              // the map is allocated and cleared in whole words
  const uint4* p = reinterpret_cast<const uint4*>(present + 64ull * w);
  unsigned long long out = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint4 v = p[q];
    const u32 x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)     // four bytes b0..b3 -> b0 | b1 << 1 | b2 << 2 | b3 << 3 in the product's top byte
      out |= (unsigned long long)((x[k] * 0x01020408u) >> 24 & 15u) << (16 * q + 4 * k);
  }
  bits[w] = out;
}

// One workgroup: the bitmap's words with the number of present grams before each ->
// table[w] = {bits low, bits high, grams before, 0}; the code of gram v is
// table[v >> 6].z + popcount(bits below bit v & 63).  At most 65 536 words.
__global__ __launch_bounds__(1024) void k_gram_table(const unsigned long long* __restrict__ present, u32 nwords,
                                                     uint4* __restrict__ table, u32* __restrict__ distinct) {
  __shared__ u32 scratch[1024 / kWave + 1];
  const u32 per = ((nwords + 1023u) / 1024u + 7u) & ~7u;   // consecutive words per thread (<= 64), in batches of eight loads
  const u32 w0 = threadIdx.x * per;
  u32 mine = 0;
  for (u32 i = 0; i < per; i += 8u) {
    unsigned long long b[8];
#pragma unroll
    for (u32 k = 0; k < 8u; ++k) b[k] = w0 + i + k < nwords ? present[w0 + i + k] : 0ull;
#pragma unroll
    for (u32 k = 0; k < 8u; ++k) mine += (u32)__popcll(b[k]);
  }
  u32 total;
  u32 run = block_scan_excl_add<1024>(mine, scratch, &total);
  for (u32 i = 0; i < per; i += 8u) {
    unsigned long long b[8];
#pragma unroll
    for (u32 k = 0; k < 8u; ++k) b[k] = w0 + i + k < nwords ? present[w0 + i + k] : 0ull;
#pragma unroll
    for (u32 k = 0; k < 8u; ++k) {
      if (w0 + i + k < nwords) table[w0 + i + k] = make_uint4((u32)b[k], (u32)(b[k] >> 32), run, 0u);
      run += (u32)__popcll(b[k]);
    }
  }
  if (threadIdx.x == 0) *distinct = total;
}

__device__ __forceinline__ u32 gram_code(const uint4* __restrict__ table, u32 v) {
  const uint4 e = table[v >> 6];
  const unsigned long long bits = (unsigned long long)e.x | ((unsigned long long)e.y << 32);
  return e.z + (u32)__popcll(bits & ((1ull << (v & 63u)) - 1ull));
}

// keys of G grams of g characters, b bits per gram code; slots in descending suffix order as in k_make_keys
//
// Long keys (lk.w != nullptr, see LongKey): the item also gets a second key word w = the NEXT lk.G2
// grams (characters G g .. (G + G2) g of the suffix), and the dense code of its predecessor character
// T[i-1] rides in spare bits of the key, so that the one long sort orders the suffixes by (G + G2) g
// characters and the ranking behind it can emit the transform's bytes without going back to T.
struct LongKey {
  u32* w; int G2;                // make: where the second words go, grams per word
  int first_r; u32 first_mask;   // a second word of fewer than eight bits: the sort's first digit is the bridge digit
                                 // (radix_sort_long), w under the key's lowest bits; first_r = w's bits, -1: the digit is w's
  int hi_shift;                  // key bits [hi_shift, hi_shift + 13): the upper bits of the suffix number
  int chr_shift; u32 chr_mask;   // key bits of the predecessor character's dense code
};
__global__ __launch_bounds__(256) void k_make_keys_gram(const u8* __restrict__ T, const u8* __restrict__ lut,
                                                        const uint4* __restrict__ gtable,
                                                        u64* __restrict__ keys, u32* __restrict__ idx, u32 n,
                                                        int g, int G, int b, u32 sigma, u32 top,
                                                        u8* __restrict__ plane, int idx_hi, LongKey lk) {
  constexpr u32 kTile = 1024;
  __shared__ u8 s_lut[256];
  __shared__ u8 s_code[kTile + 136];
  __shared__ u32 s_g[kTile + 128];
  __shared__ u64 s_key[kTile];
  __shared__ u32 s_w[kTile];
  s_lut[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const u32 J0 = blockIdx.x * kTile;
  if (J0 >= n) return;
  const u32 cntw = min(kTile, n - J0);
  const u32 i_base = n - J0 - cntw;
  const int Gall = G + (lk.w ? lk.G2 : 0);
  const u32 span = cntw + (u32)((Gall - 1) * g);     // gram starts this tile looks at (Gall g <= 96)
  const u32 pre = (lk.w && i_base > 0) ? 1u : 0u;    // long keys: one character before the tile's first suffix
  stage_codes(T, s_lut, s_code, i_base - pre, span + (u32)g + pre);
  __syncthreads();
  const u8* s_c = s_code + pre;                       // s_c[t] = code of T[i_base + t]
  {
    // span <= 1024 + 60: at most five gram starts per thread, strided, so that the five table
    // lookups (the kernel's only latency) are in flight together
    u32 v[5];
#pragma unroll
    for (int e = 0; e < 5; ++e) {
      const u32 t = threadIdx.x + 256u * e;
      v[e] = 0;
      if (t < span) for (int c = 0; c < g; ++c) v[e] = v[e] * sigma + s_c[t + c];
    }
    u32 cd[5];
#pragma unroll
    for (int e = 0; e < 5; ++e) cd[e] = threadIdx.x + 256u * e < span ? gram_code(gtable, v[e]) : 0u;
#pragma unroll
    for (int e = 0; e < 5; ++e) if (threadIdx.x + 256u * e < span) s_g[threadIdx.x + 256u * e] = cd[e];
  }
  __syncthreads();
  const u32 o = 4u * threadIdx.x;
  if (o < cntw) {
    const u32 cnt = min(4u, cntw - o);
    for (u32 s = 0; s < cnt; ++s) {
      u64 key = 0;
      for (int j = 0; j < G; ++j) key = (key << b) | (u64)s_g[o + s + (u32)(j * g)];
      if (lk.w) {
        u32 w = 0;
        for (int j = G; j < Gall; ++j) w = (w << b) | s_g[o + s + (u32)(j * g)];
        s_w[cntw - 1u - (o + s)] = w;
        const u32 i = i_base + o + s;                  // its predecessor's code (suffix 0 has none)
        key |= (u64)(i ? (u32)s_c[(int)(o + s) - 1] : 0u) << lk.chr_shift;
      }
      s_key[cntw - 1u - (o + s)] = key;
    }
  }
  __syncthreads();
  for (u32 j = threadIdx.x; j < cntw; j += 256u) {
    u64 key = s_key[j];
    if (idx_hi) key |= (u64)((n - 1u - (J0 + j)) >> 16) << (lk.w ? lk.hi_shift : 48);
    keys[J0 + j] = key;
    if (idx) idx[J0 + j] = n - 1u - (J0 + j);
    if (lk.w) {
      const u32 w = s_w[j];
      lk.w[J0 + j] = w;
      if (plane) plane[J0 + j] = lk.first_r < 0 ? (u8)w : (u8)((w | ((u32)key << lk.first_r)) & lk.first_mask);   // the long sort's first digit
    } else if (plane) plane[J0 + j] = (u8)s_key[j];
  }
}

// ---------------------------------------------------------------------------------------
// K2c initial keys from an order-preserving prefix code ("code keys", round 5).
//
//     Dense gram codes pack a fixed number of characters into the key: 20 of the 64-token generator's 40
//     symbols, but 12 of a real text's 170-210 -- and, worse, the same 12 of a string that occurs a
//     million times as of one that occurs twice, so that the frequent strings' groups stay huge.  Here
//     every character is written with a prefix code that depends on the character before it (an order-1
//     model of the block, taken from a sample of its character pairs), and a suffix's key is the first
//     kbits bits of its code string: frequent continuations cost few bits, so a frequent string gets MORE
//     of its characters into the key and the groups the sort leaves are levelled out (256 MiB of the
//     generator's text: 2.6 % of the suffixes tied after 72 key bits, against 21.5 % after 20 characters).
//
//     Order.  For every context the code is alphabetic (codewords in symbol order, prefix free: a
//     weight-balanced binary tree over the symbols, k_code_build), so two suffixes that first differ at
//     character j -- whose context, character j-1, they share -- have equal code strings up to that
//     character's codeword and differ there in symbol order: the code strings compare as the suffixes do,
//     and so do their first kbits bits (with ties).  A suffix's first character has no context inside the
//     suffix: it takes the order-0 code (row 256 of the table).  The terminator and the padding behind it
//     are symbol 0 (or share the smallest symbol's code, as in k_make_keys): the smallest continuation.
//
//     Depth.  Suffixes with equal keys share the key's COMPLETE codewords (the same bits decode the same
//     way), k characters, a number that differs from key to key.  The item carries it as a 3-bit level
//     (a lower bound: kCodeLevelDepth) in spare key bits; the ranking hands it on to the finisher, which
//     compares the text from there.
//
//     All codewords of the block's text are one bit string (a character's codeword does not depend on
//     which suffix it is read for), so a tile looks its characters' codewords up ONCE, scans their
//     lengths, writes the bits into LDS, and every suffix key is 72 bits cut out of that string at the
//     suffix's second character, behind its first character's order-0 codeword.
// ---------------------------------------------------------------------------------------
constexpr u32 kCodeRows = 257;                // contexts 0..255, and the order-0 row
constexpr u32 kCodeOrder0 = 256;
constexpr u32 kCodeMaxLen = 26;               // longest codeword k_code_build may make (27 code bits + 5 length bits per entry)
constexpr u32 kCodeLA = 72;                   // characters a tile looks ahead: a key has at most 72 bits, a codeword at least one
constexpr u32 kPairReplicas = 8;              // copies of the pair counters (hot pairs: one address per XCD-worth of workgroups)
constexpr u32 kPairTile = 4096;
// the level's depth: suffixes with equal keys of that level share at least so many characters
__host__ __device__ __forceinline__ u32 code_level_depth(u32 level) { return (u32)((0x30201814100C0801ull >> (8u * level)) & 255ull); }
__device__ __forceinline__ u32 code_level_of(u32 k) {
  return k >= 48u ? 7u : k >= 32u ? 6u : k >= 24u ? 5u : k >= 20u ? 4u : k >= 16u ? 3u : k >= 12u ? 2u : k >= 8u ? 1u : 0u;
}

// Character pairs of a sample of the text: workgroup b counts the pairs of tile floor(b * ntiles / gridDim).
// cnt[replica][a * 256 + b] over dense codes; pair (T[j-1], T[j]).
__global__ __launch_bounds__(256) void k_pair_counts(const u8* __restrict__ T, const u8* __restrict__ lut, u32 n,
                                                     u32* __restrict__ cnt) {
  __shared__ u8 s_lut[256];
  s_lut[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const u32 ntiles = (n + kPairTile - 1u) / kPairTile;
  const u32 tile = (u32)(((u64)blockIdx.x * ntiles) / gridDim.x);
  const u32 j0 = tile * kPairTile + 16u * threadIdx.x;
  if (j0 >= n) return;
  const uint4 v = *reinterpret_cast<const uint4*>(T + j0);          // T is padded by more than a tile
  const u32 w[4] = {v.x, v.y, v.z, v.w};
  u32 prev = j0 ? s_lut[T[j0 - 1u]] : 0u;
  u32* mine = cnt + (size_t)(blockIdx.x % kPairReplicas) * 65536u;
#pragma unroll
  for (u32 e = 0; e < 16u; ++e) {
    const u32 c = s_lut[(w[e >> 2] >> (8u * (e & 3u))) & 255u];
    if (j0 + e < n && j0 + e > 0u) atomicAdd(&mine[prev * 256u + c], 1u);
    prev = c;
  }
}

// One workgroup per context (row 256: all contexts together, the order-0 code): the alphabetic code of the
// sigma symbols for the weights count + 1 (every symbol can follow every context: unseen pairs get long
// codewords, not none).  Weight-balanced tree: an interval of symbols is cut where the two sides' weights
// differ least; every symbol walks down from the root on its own (the cuts it computes are the ones its
// neighbours compute).  codes[row * 256 + symbol] = length << 27 | bits.  Weights are scaled to sum below
// 2^16 + sigma: no codeword is longer than kCodeMaxLen (checked: *err is raised otherwise).
__global__ __launch_bounds__(256) void k_code_build(const u32* __restrict__ cnt, u32 sigma, u32* __restrict__ codes,
                                                    u32* __restrict__ err) {
  __shared__ u32 s_pre[257];
  __shared__ u32 scr[256 / kWave + 1];
  const u32 row = blockIdx.x, b = threadIdx.x;
  u32 c = 0;
  if (b < sigma) {
    if (row < 256u) {
      for (u32 r = 0; r < kPairReplicas; ++r) c += cnt[(size_t)r * 65536u + row * 256u + b];
    } else {
      for (u32 a = 0; a < sigma; ++a)
        for (u32 r = 0; r < kPairReplicas; ++r) c += cnt[(size_t)r * 65536u + a * 256u + b];
    }
  }
  u32 total;
  (void)block_scan_excl_add<256>(c, scr, &total);
  u32 sh = 0;
  while ((total >> sh) >= 65536u) ++sh;
  const u32 wt = b < sigma ? (c >> sh) + 1u : 0u;
  u32 wtot;
  const u32 ex = block_scan_excl_add<256>(wt, scr, &wtot);
  s_pre[b] = ex;
  if (b == 255u) s_pre[256] = wtot;
  __syncthreads();
  if (b >= sigma) { codes[row * 256u + b] = 0u; return; }
  u32 lo = 0, hi = sigma, bits = 0, len = 0;
  while (hi - lo > 1u && len <= 31u) {
    const u64 mid2 = (u64)s_pre[lo] + (u64)s_pre[hi];
    u32 a = lo + 1u, z = hi - 1u;                       // smallest cut in [lo+1, hi-1] with 2 pre[cut] >= mid2, else hi-1
    while (a < z) { const u32 mdl = (a + z) >> 1; if (2ull * s_pre[mdl] >= mid2) z = mdl; else a = mdl + 1u; }
    u32 cut = a;
    if (cut > lo + 1u) {
      const u64 da = 2ull * s_pre[cut] >= mid2 ? 2ull * s_pre[cut] - mid2 : mid2 - 2ull * s_pre[cut];
      const u64 db = mid2 - 2ull * s_pre[cut - 1u];       // 2 pre[cut-1] < mid2
      if (db <= da) cut = cut - 1u;
    }
    if (b < cut) { hi = cut; bits <<= 1; } else { lo = cut; bits = (bits << 1) | 1u; }
    ++len;
  }
  if (len == 0u) len = 1u;                              // (an alphabet of one symbol)
  if (len > kCodeMaxLen) atomicOr(err, 2u);
  codes[row * 256u + b] = (len << 27) | bits;
}

// keys of a tile of 1024 suffixes (slots in descending suffix order as in k_make_keys); kbits = key bits in all
// (8 .. 72: the low 32 go to the second word w, the rest to the key word)
struct CodeKey {
  const u32* codes;              // k_code_build's table
  int kbits;                     // key bits in all (33 .. 72)
  int hi_shift, chr_shift, lvl_shift;   // key word: the upper bits of the suffix number (split items), the predecessor's code, the level
};
__global__ __launch_bounds__(256) void k_make_keys_code(const u8* __restrict__ T, const u8* __restrict__ lut,
                                                        u64* __restrict__ keys, u32* __restrict__ wout, u32 n,
                                                        u8* __restrict__ plane, int idx_hi, CodeKey ck) {
  constexpr u32 kTile = 1024;
  constexpr u32 kPos = 1280;                            // codeword positions a tile can hold (five per thread)
  __shared__ u8 s_lut[256];
  __shared__ u8 s_code[kTile + kCodeLA + 24];
  __shared__ u32 s_c0[256];                             // the order-0 row
  __shared__ u32 s_bp[kPos + 8];                        // bit offset of every position's codeword
  __shared__ u32 s_bits[(kPos * kCodeMaxLen) / 32 + 8];
  __shared__ u32 s_ends[(kPos * kCodeMaxLen) / 32 + 8];   // a one at every codeword's last bit
  __shared__ u64 s_key[kTile];
  __shared__ u32 s_w[kTile];
  __shared__ u32 scr[256 / kWave + 1];
  s_lut[threadIdx.x] = lut[threadIdx.x];
  s_c0[threadIdx.x] = ck.codes[kCodeOrder0 * 256u + threadIdx.x];
  for (u32 i = threadIdx.x; i < (kPos * kCodeMaxLen) / 32 + 8; i += 256u) { s_bits[i] = 0u; s_ends[i] = 0u; }
  __syncthreads();
  const u32 J0 = blockIdx.x * kTile;
  if (J0 >= n) return;
  const u32 cntw = min(kTile, n - J0);
  const u32 i_base = n - J0 - cntw;
  const u32 pre = i_base > 0 ? 1u : 0u;
  const u32 span = cntw + kCodeLA + 2u;                 // positions 0 .. span-1 of the tile's stretch of T
  stage_codes(T, s_lut, s_code, i_base - pre, span + pre);
  __syncthreads();
  const u8* s_c = s_code + pre;                          // s_c[t] = code of T[i_base + t]
  // codewords of positions 1 .. span-1 (position t in the context of t-1), five consecutive ones per thread
  u32 cw[5];
  const u32 t0 = 1u + 5u * threadIdx.x;
#pragma unroll
  for (u32 e = 0; e < 5u; ++e) {
    const u32 t = t0 + e;
    cw[e] = t < span ? ck.codes[(u32)s_c[t - 1u] * 256u + (u32)s_c[t]] : 0u;
  }
  u32 lsum = 0;
#pragma unroll
  for (u32 e = 0; e < 5u; ++e) lsum += cw[e] >> 27;
  u32 total;
  u32 bp = block_scan_excl_add<256>(lsum, scr, &total);
#pragma unroll
  for (u32 e = 0; e < 5u; ++e) {
    const u32 t = t0 + e;
    const u32 len = cw[e] >> 27, code = cw[e] & 0x7FFFFFFu;
    if (t <= kPos) s_bp[t] = bp;
    if (len) {
      const u32 word = bp >> 5, off = bp & 31u;
      if (off + len <= 32u) atomicOr(&s_bits[word], code << (32u - off - len));
      else {
        const u32 r = off + len - 32u;                  // bits that spill into the next word
        atomicOr(&s_bits[word], code >> r);
        atomicOr(&s_bits[word + 1u], code << (32u - r));
      }
      const u32 last = bp + len - 1u;
      atomicOr(&s_ends[last >> 5], 0x80000000u >> (last & 31u));
    }
    bp += len;
  }
  if (threadIdx.x == 0) s_bp[0] = 0u;
  __syncthreads();
  const u32 o = 4u * threadIdx.x;
  if (o < cntw) {
    const u32 cnt = min(4u, cntw - o);
    for (u32 s = 0; s < cnt; ++s) {
      const u32 t = o + s;
      const u32 first = s_c0[s_c[t]];
      const u32 l0 = first >> 27;
      const u32 need = (u32)ck.kbits - l0;              // bits taken from the string, from position t+1 on
      const u32 start = s_bp[t + 1u];
      const u32 wi = start >> 5, off = start & 31u;
      const u64 a = ((u64)s_bits[wi] << 32) | (u64)s_bits[wi + 1u];
      const u64 bq = ((u64)s_bits[wi + 2u] << 32) | (u64)s_bits[wi + 3u];
      const u64 top = off ? (a << off) | (bq >> (64u - off)) : a;             // bits start .. start+63
      const u64 nxt = off ? (bq << off) : bq;                                // bits start+64 .. (at least 32 valid ones)
      // the first `need` bits (need <= 71) as a number
      u64 x_hi, x_lo;                                     // x = x_hi * 2^32 + (x_lo & 0xFFFFFFFF), x_lo = its low 32 bits
      if (need <= 64u) {
        const u64 x = need ? top >> (64u - need) : 0ull;
        x_hi = x >> 32; x_lo = x & 0xFFFFFFFFull;
      } else {
        const u32 extra = need - 64u;                     // 1 .. 7 bits of nxt
        const u64 lowpart = nxt >> (64u - extra);
        x_lo = ((top << extra) | lowpart) & 0xFFFFFFFFull;
        x_hi = (top >> (32u - extra));                    // top's upper bits: x >> 32
      }
      // key = first codeword * 2^need + x
      const u64 fc = (u64)(first & 0x7FFFFFFu);
      u64 k_hi, k_lo;                                     // key >> 32, key & 0xFFFFFFFF
      if (need >= 32u) { k_lo = x_lo; k_hi = x_hi | (fc << (need - 32u)); }
      else { const u64 full = (fc << need) | ((x_hi << 32) | x_lo); k_lo = full & 0xFFFFFFFFull; k_hi = full >> 32; }
      // complete codewords in the key: the first character, and the codewords that end inside bits [start, start + need)
      const u64 ea = ((u64)s_ends[wi] << 32) | (u64)s_ends[wi + 1u];
      const u64 eb = ((u64)s_ends[wi + 2u] << 32) | (u64)s_ends[wi + 3u];
      const u64 etop = off ? (ea << off) | (eb >> (64u - off)) : ea;
      u32 k = 1u;
      if (need <= 64u) k += need ? (u32)__popcll(etop >> (64u - need)) : 0u;
      else k += (u32)__popcll(etop) + (u32)__popcll((off ? (eb << off) : eb) >> (128u - need));
      const u32 i = i_base + t;
      u64 key = k_hi;
      key |= (u64)(i ? (u32)s_c[(int)t - 1] : 0u) << ck.chr_shift;
      key |= (u64)code_level_of(k) << ck.lvl_shift;
      s_key[cntw - 1u - t] = key;
      s_w[cntw - 1u - t] = (u32)k_lo;
    }
  }
  __syncthreads();
  for (u32 j = threadIdx.x; j < cntw; j += 256u) {
    u64 key = s_key[j];
    if (idx_hi) key |= (u64)((n - 1u - (J0 + j)) >> 16) << ck.hi_shift;
    keys[J0 + j] = key;
    const u32 w = s_w[j];
    wout[J0 + j] = w;
    if (plane) plane[J0 + j] = (u8)w;
  }
}

// ---------------------------------------------------------------------------------------
// K4  re-ranking of a sorted list (INIT: the whole SA after the initial sort; rounds: the
//     active list after sorting by (group, rank[s+h]), keys u64).
//
//     head[p]   = p starts a new group (key differs from p-1; INIT: or p-1 is no longer
//                 than the key, which makes it a finished singleton)
//     active[p] = p's group has >= 2 members
//     new rank of p = global SA slot of its group head
//     finished suffixes go to SA; active ones are compacted (order kept) into the next
//     active list with their global slot and a dense group number.
//     Three launches: tile reduce, one-workgroup scan of the tile aggregates, tile apply.
//
//     Every wave owns 512 consecutive list slots and walks them 64 at a time (coalesced).
//     Flags become 64-bit ballots, so every prefix inside the wave is a popcount of a
//     masked ballot and the running sums live in scalar registers; only the four wave
//     totals of a tile cross LDS.
// ---------------------------------------------------------------------------------------
constexpr int kRrTPB = 256;
constexpr int kRrWaves = kRrTPB / kWave;
constexpr int kRrE = 8;
constexpr int kRrChunk = kWave * kRrE;        // slots per wave
constexpr int kRrTile = kRrTPB * kRrE;        // slots per workgroup

__device__ __forceinline__ u64 shfl_up1(u64 v) {
  return (u64)__shfl_up((unsigned long long)v, 1, kWave);
}
__device__ __forceinline__ u32 shfl_up1(u32 v) { return __shfl_up(v, 1, kWave); }
__device__ __forceinline__ u64 shfl_down1(u64 v) {
  return (u64)__shfl_down((unsigned long long)v, 1, kWave);
}
__device__ __forceinline__ u32 shfl_down1(u32 v) { return __shfl_down(v, 1, kWave); }

template <typename K, bool INIT>
struct RrMasks {
  u64 head[kRrE], act[kRrE], valid[kRrE];
  u32 sfx[kRrE];       // idx[p] of this lane's slot in iteration e
  u32 chr[kRrE];       // bits 56..63 of its key (the carried character, when there is one)
};

__device__ __forceinline__ u64 lane_value(u64 v, int l) { return (u64)__shfl((unsigned long long)v, l, kWave); }
__device__ __forceinline__ u32 lane_value(u32 v, int l) { return __shfl(v, l, kWave); }

// Masks of one wave chunk starting at list slot wbase.  All of the chunk's keys and suffixes are
// loaded before anything is compared (one iteration at a time, each with its own neighbour
// loads, was eight memory latencies in a row per wave); the neighbours across an iteration's ends
// are the adjacent iterations' end lanes, and only the two slots next to the whole chunk are
// loaded on their own.
// LONG (initial ranking after the long-key sort, radix_sort_long): a suffix's order key is (key & kmask,
// w & wmask); the upper bits of its number sit at hi_shift, its predecessor character's code at chr_shift.
// wmask: the bits of w the sort ordered by.  How many characters the members of a group are known to share: code keys
// carry a 3-bit level at lvl_shift (code_level_depth); gram keys (lvl_shift < 0) all share `depth`.  It rides to the
// finisher in the upper byte of the list entry's 16-bit character field.
struct RrLong { const u32* w; u32 wmask; int hi_shift; int chr_shift; u32 chr_mask; int lvl_shift; u32 depth; };


template <typename K, bool INIT, bool SPLIT, bool LONG = false>
__device__ __forceinline__ void rr_masks(const K* __restrict__ key, const u32* __restrict__ idx,
                                         u32 m, u32 n, u32 short_len, K kmask, u32 wbase, u32 lane,
                                         RrMasks<K, INIT>& f, RrLong lg = RrLong()) {
  // SPLIT (initial ranking of 64-bit keys only): a suffix number's upper bits sit in bits 48.. of
  // its key and idx[] holds 16-bit lower halves (the sort moved 10 bytes per item instead of 12)
  auto suffix_of = [&](K kraw, u32 word, u32 p) -> u32 {     // the 16-bit half out of its aligned word (two lanes per word)
    if (sizeof(K) == 8 && SPLIT) {
      const u32 hi = LONG ? ((u32)((u64)kraw >> lg.hi_shift) & 0x1FFFu) : (u32)((u64)kraw >> 48);
      return (hi << 16) | ((p & 1u) ? word >> 16 : word & 0xFFFFu);
    }
    return word;
  };
  K kc[kRrE];
  u32 wc[LONG ? kRrE : 1] = {};
  {
    K kraw[kRrE];
    u32 word[kRrE];
#pragma unroll
    for (int e = 0; e < kRrE; ++e) {               // nothing but loads: sixteen in flight
      const u32 p = wbase + e * kWave + lane;
      const bool ok = p < m;
      kraw[e] = ok ? key[p] : (K)0;
      word[e] = ok ? idx[(sizeof(K) == 8 && SPLIT) ? p >> 1 : p] : 0u;
      if (LONG) wc[LONG ? e : 0] = ok ? lg.w[p] : 0u;
    }
#pragma unroll
    for (int e = 0; e < kRrE; ++e) {
      const u32 p = wbase + e * kWave + lane;
      f.sfx[e] = p < m ? suffix_of(kraw[e], word[e], p) : 0u;
      f.chr[e] = LONG ? ((u32)((u64)kraw[e] >> lg.chr_shift) & lg.chr_mask) : sizeof(K) == 8 ? (u32)((u64)kraw[e] >> 56) : 0u;
      if (LONG) {                                   // the depth its group is known to share, above the character's code
        f.chr[e] |= (lg.lvl_shift >= 0 ? code_level_depth((u32)((u64)kraw[e] >> lg.lvl_shift) & 7u) : lg.depth) << 8;
        wc[LONG ? e : 0] &= lg.wmask;
      }
      kc[e] = kraw[e] & kmask;
    }
  }
  K before = (K)0, after = (K)0;          // slots wbase - 1 and wbase + chunk (where they exist)
  u32 ibefore = 0u, wbefore = 0u, wafter = 0u;
  if (lane == 0 && wbase > 0 && wbase < m) {
    const K braw = key[wbase - 1];
    before = braw & kmask;
    if (LONG) wbefore = lg.w[wbase - 1] & lg.wmask;
    if (INIT) ibefore = suffix_of(braw, idx[(sizeof(K) == 8 && SPLIT) ? (wbase - 1) >> 1 : wbase - 1], wbase - 1);
  }
  if (lane == kWave - 1 && wbase + (u32)kRrChunk < m) {
    after = key[wbase + kRrChunk] & kmask;
    if (LONG) wafter = lg.w[wbase + kRrChunk] & lg.wmask;
  }
#pragma unroll
  for (int e = 0; e < kRrE; ++e) {
    const u32 p = wbase + e * kWave + lane;
    const bool ok = p < m;
    const u32 ic = f.sfx[e];
    K kp = shfl_up1(kc[e]);
    K kn = shfl_down1(kc[e]);
    u32 ip = shfl_up1(ic);
    const K kprev_end = e > 0 ? lane_value(kc[e > 0 ? e - 1 : 0], kWave - 1) : before;
    const u32 iprev_end = e > 0 ? lane_value(f.sfx[e > 0 ? e - 1 : 0], kWave - 1) : ibefore;
    const K knext_first = e + 1 < kRrE ? lane_value(kc[e + 1 < kRrE ? e + 1 : e], 0) : after;
    if (lane == 0) { kp = (ok && p > 0) ? kprev_end : (K)0; if (INIT) ip = (ok && p > 0) ? iprev_end : 0u; }
    if (lane == kWave - 1) kn = (p + 1 < m) ? knext_first : (K)0;
    bool wdiff_p = false, wdiff_n = false;           // LONG: the second key word differs from the neighbour's
    if (LONG) {
      const u32 wme = wc[LONG ? e : 0];
      u32 wp = shfl_up1(wme), wn = shfl_down1(wme);
      const u32 wprev_end = e > 0 ? lane_value(wc[LONG && e > 0 ? e - 1 : 0], kWave - 1) : wbefore;
      const u32 wnext_first = e + 1 < kRrE ? lane_value(wc[LONG && e + 1 < kRrE ? e + 1 : 0], 0) : wafter;
      if (lane == 0) wp = (ok && p > 0) ? wprev_end : 0u;
      if (lane == kWave - 1) wn = (p + 1 < m) ? wnext_first : 0u;
      wdiff_p = wme != wp;
      wdiff_n = wme != wn;
    }
    bool h = ok && (p == 0 || kc[e] != kp || wdiff_p);
    // (long items: a suffix is known to lie inside its key when it is no longer than the key's depth; equal keys
    // have equal depths, so the slot before counts with this one's)
    const u32 sl = LONG ? f.chr[e] >> 8 : short_len;
    if (INIT) h = h || (ok && p > 0 && (u64)ip + sl >= (u64)n);
    bool hn = (p + 1 >= m) || kn != kc[e] || wdiff_n;         // is p+1 a head (or past the end)?
    if (INIT) hn = hn || ((u64)ic + sl >= (u64)n);
    f.head[e] = __ballot(h);
    f.valid[e] = __ballot(ok);
    f.act[e] = __ballot(ok && (!h || !hn));
  }
}

__device__ __forceinline__ u32 top_bit(u64 v) { return 63u - (u32)__clzll((unsigned long long)v); }

template <typename K, bool INIT, bool SPLIT = false, bool LONG = false>
__global__ __launch_bounds__(kRrTPB) void k_rerank_reduce(const K* __restrict__ key,
                                                          const u32* __restrict__ idx, u32 m,
                                                          u32 n, u32 short_len, K kmask,
                                                          u32* __restrict__ aggA,
                                                          u32* __restrict__ aggB,
                                                          u32* __restrict__ aggC, RrLong lg = RrLong()) {
  __shared__ u32 s_tot[3][kRrWaves];
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x / kWave;
  const u32 wbase = blockIdx.x * kRrTile + wave * kRrChunk;
  RrMasks<K, INIT> f;
  rr_masks<K, INIT, SPLIT, LONG>(key, idx, m, n, short_len, kmask, wbase, lane, f, lg);
  u32 nact = 0, nha = 0, last = 0;
#pragma unroll
  for (int e = 0; e < kRrE; ++e) {
    nact += (u32)__popcll(f.act[e]);
    nha += (u32)__popcll(f.act[e] & f.head[e]);
    if (f.head[e]) last = wbase + e * kWave + top_bit(f.head[e]) + 1u;
  }
  if (lane == 0) { s_tot[0][wave] = nact; s_tot[1][wave] = nha; s_tot[2][wave] = last; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 a = 0, b = 0, c = 0;
    for (int w = 0; w < kRrWaves; ++w) { a += s_tot[0][w]; b += s_tot[1][w]; c = max(c, s_tot[2][w]); }
    aggA[blockIdx.x] = a; aggB[blockIdx.x] = b; aggC[blockIdx.x] = c;
  }
}

// Exclusive sum of aggA, aggB; exclusive max of aggC; totals -> counts[0..1].  A workgroup of 16
// waves owns 4096 consecutive entries, a wave 256 of them (four coalesced rows of 64, their twelve
// loads in flight together).  PHASE 0 leaves every workgroup's totals in part[]; PHASE 1 adds up the
// parts before it and scans its entries with that carried in.  (One workgroup walking all 131 072
// tile aggregates of a 256 MiB block took 107-168 us, three times per block, with the GPU idle beside it.)
constexpr u32 kRrScanChunk = 4096;
template <int PHASE>
__global__ __launch_bounds__(1024) void k_rerank_scan_tiles(u32* __restrict__ aggA,
                                                            u32* __restrict__ aggB,
                                                            u32* __restrict__ aggC, u32 ntiles,
                                                            u32* __restrict__ part,
                                                            u32* __restrict__ counts) {
  __shared__ u32 s_tot[3][16];
  __shared__ u32 s_base[3];
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x / kWave;
  const u32 b = blockIdx.x * kRrScanChunk + wave * 256u;
  const u32 e = min(b + 256u, ntiles);
  u32 va[4], vb[4], vc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const u32 i = b + (u32)r * kWave + lane;
    const bool ok = i < e;
    va[r] = ok ? aggA[i] : 0u; vb[r] = ok ? aggB[i] : 0u; vc[r] = ok ? aggC[i] : 0u;
  }
  u32 sa = 0, sb = 0, sc = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) { sa += va[r]; sb += vb[r]; sc = max(sc, vc[r]); }
  sa = wave_scan_add(sa, lane); sb = wave_scan_add(sb, lane); sc = wave_scan_max(sc, lane);
  if (lane == kWave - 1) { s_tot[0][wave] = sa; s_tot[1][wave] = sb; s_tot[2][wave] = sc; }
  if (PHASE == 1 && wave == 0) {
    // what the workgroups before this one hold (at most a few hundred parts), and the totals
    u32 pa = 0, pb = 0, pc = 0, ta = 0, tb = 0;
    for (u32 g = lane; g < gridDim.x; g += kWave) {
      const u32 x = part[3u * g], y = part[3u * g + 1u], z = part[3u * g + 2u];
      ta += x; tb += y;
      if (g < blockIdx.x) { pa += x; pb += y; pc = max(pc, z); }
    }
    pa = wave_scan_add(pa, lane); pb = wave_scan_add(pb, lane); pc = wave_scan_max(pc, lane);
    ta = wave_scan_add(ta, lane); tb = wave_scan_add(tb, lane);
    if (lane == kWave - 1) {
      s_base[0] = pa; s_base[1] = pb; s_base[2] = pc;
      if (blockIdx.x == 0) { counts[0] = ta; counts[1] = tb; }
    }
  }
  __syncthreads();
  if (PHASE == 0) {
    if (threadIdx.x == 0) {
      u32 ta = 0, tb = 0, tc = 0;
      for (u32 w = 0; w < 16; ++w) { ta += s_tot[0][w]; tb += s_tot[1][w]; tc = max(tc, s_tot[2][w]); }
      part[3u * blockIdx.x] = ta; part[3u * blockIdx.x + 1u] = tb; part[3u * blockIdx.x + 2u] = tc;
    }
    return;
  }
  u32 ca = s_base[0], cb = s_base[1], cc = s_base[2];
  for (u32 w = 0; w < wave; ++w) { ca += s_tot[0][w]; cb += s_tot[1][w]; cc = max(cc, s_tot[2][w]); }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const u32 i = b + (u32)r * kWave + lane;
    const u32 ia = wave_scan_add(va[r], lane), ib = wave_scan_add(vb[r], lane), ic = wave_scan_max(vc[r], lane);
    const u32 pc = __shfl_up(ic, 1, kWave);
    if (i < e) {
      aggA[i] = ca + ia - va[r];
      aggB[i] = cb + ib - vb[r];
      aggC[i] = max(cc, lane ? pc : 0u);
    }
    ca += __shfl(ia, kWave - 1, kWave);
    cb += __shfl(ib, kWave - 1, kWave);
    cc = max(cc, __shfl(ic, kWave - 1, kWave));
  }
}

// PAIRS: instead of scattering rank[s] = nr from here (one random 4-byte write per element,
// each costing a whole HBM sector), emit (s, nr) in list order; the caller partitions the
// pairs by the high bits of s and k_scatter_pairs then writes into one small window of
// rank[] at a time.
//
// MODE: how the new ranks reach rank[] and how the next round gets its list.
//   0  rank[s] = nr written from here (short lists); active list compacted (aidx/aglob/agrp)
//   1  (s, nr) pairs in list order for scatter_rank_pairs; active list compacted
//   2  "dense" (most of the list stays active): one record per slot for the window-ordered
//      route -- tri_key[p] = s << 32 | nr, tri_val[p] = new dense group number, or all ones for
//      a finished suffix -- and only aglob of the active list is compacted.  The records are
//      partitioned by the high bits of s; k_scatter_dense then updates rank[] one window at a
//      time and k_gather_dense builds the next round's sort keys from the same records, reading
//      rank[s+h] (and T[s-1]) inside one window instead of at random (bwt_engine.hip, host side).
//
// EMIT: the transform's bytes leave from here instead of from a gather over the finished
// suffix array.  1: the character rides in bits 56..63 of the round's key (k_gather_key2 /
// k_gather_dense put it there: a round's key has at most 56 bits, can_carry()); 2 (initial
// ranking: its keys have no room): it is read from T[s-1] here.  A suffix that becomes final
// writes out[slot] (slot out_n, which only exists in block mode, goes to *last_char; the slot
// of suffix 0 is the end-of-block row, *pidx); one that stays active hands its character on in
// achr_out (not in MODE 2: k_gather_dense reads it again, inside a window).  SA is only
// written when somebody wants it (SA != nullptr).
struct RrEmit {
  u8* out; u32 out_n; u32* last_char; u32* pidx; const u8* T; u8* achr_out;
  u8* rec_plane; int rec_shift;      // MODE 2: digit (s >> rec_shift) & 255 of every record, for the window partition's first pass
  const u8* inv_lut;                 // EMIT 3: dense code -> byte (the character rides in the long key as its code)
  // LF powers noted as suffixes become final (the routes that never complete rank[]):
  // LF[k] = slot of suffix n - k x, k = 1 .. lf_n - 1 (divsufsort.c:337-338,350,381,390); lf_inv = floor(2^32 / x)
  u32* lf; u32 lf_n, lf_x, lf_inv;
};

// Suffix s became final at `slot`: is it one of the sampled ones?
__device__ __forceinline__ void lf_note(const RrEmit& em, u32 n, u32 s, u32 slot) {
  if (em.lf_n <= 1u) return;
  const u32 t = n - s;                                  // 1 .. n
  u32 q = em.lf_x > 1u ? __umulhi(t, em.lf_inv) : t;    // floor(t / x) or one less (x = 1: a block of a few hundred bytes with 256 powers)
  u32 r = t - q * em.lf_x;
  if (r == em.lf_x) { r = 0u; ++q; }
  if (r == 0u && q >= 1u && q < em.lf_n) em.lf[q] = slot;
}

//   3  (finisher route, after the long-key sort) nothing goes to rank[]: finished suffixes leave
//      (out, SA, LF powers), the others are compacted into the list the finisher works on --
//      suffix, slot, slot of the group's head, character.
// EMIT 3: the character's dense code rides in the long key (RrLong::chr_shift); em.inv_lut turns it back.
template <typename K, bool INIT, int MODE, int EMIT, bool SPLIT = false, bool LONG = false>
__global__ __launch_bounds__(kRrTPB) void k_rerank_apply(
    const K* __restrict__ key, const u32* __restrict__ idx, const u32* __restrict__ aglob, u32 m,
    u32 n, u32 short_len, K kmask, const u32* __restrict__ aggA, const u32* __restrict__ aggB,
    const u32* __restrict__ aggC, u32* __restrict__ rank, u32* __restrict__ SA,
    u32* __restrict__ aidx_out, u32* __restrict__ aglob_out, u32* __restrict__ agrp_out,
    u32* __restrict__ pair_s, u32* __restrict__ pair_r, RrEmit em, RrLong lg = RrLong()) {
  __shared__ u32 s_tot[3][kRrWaves];
  __shared__ u8 s_inv[EMIT == 3 ? 256 : 4];
  if (EMIT == 3) s_inv[threadIdx.x] = em.inv_lut[threadIdx.x];     // kRrTPB == 256
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x / kWave;
  const u32 wbase = blockIdx.x * kRrTile + wave * kRrChunk;
  RrMasks<K, INIT> f;
  rr_masks<K, INIT, SPLIT, LONG>(key, idx, m, n, short_len, kmask, wbase, lane, f, lg);
  {
    u32 nact = 0, nha = 0, last = 0;
#pragma unroll
    for (int e = 0; e < kRrE; ++e) {
      nact += (u32)__popcll(f.act[e]);
      nha += (u32)__popcll(f.act[e] & f.head[e]);
      if (f.head[e]) last = wbase + e * kWave + top_bit(f.head[e]) + 1u;
    }
    if (lane == 0) { s_tot[0][wave] = nact; s_tot[1][wave] = nha; s_tot[2][wave] = last; }
  }
  __syncthreads();
  u32 q_run = aggA[blockIdx.x], g_run = aggB[blockIdx.x], hp_run = aggC[blockIdx.x];
  for (u32 w = 0; w < wave; ++w) {
    q_run += s_tot[0][w]; g_run += s_tot[1][w]; hp_run = max(hp_run, s_tot[2][w]);
  }
  const u64 lt = (1ull << lane) - 1ull;
  const u64 le = lt | (1ull << lane);
  // first every load of the chunk (global slots, new ranks, characters), then the stores
  u32 gs[kRrE], nrs[kRrE], grps[kRrE], qs[kRrE], cs[kRrE];
#pragma unroll
  for (int e = 0; e < kRrE; ++e) {
    const u32 base = wbase + e * kWave;
    const u32 p = base + lane;
    const u64 hd = f.head[e], ac = f.act[e];
    gs[e] = 0; nrs[e] = 0; grps[e] = 0; qs[e] = 0; cs[e] = 0;
    if ((f.valid[e] >> lane) & 1ull) {
      const bool a = (ac >> lane) & 1ull;
      const u64 hm = hd & le;
      const u32 hp1 = hm ? base + top_bit(hm) + 1u : hp_run;   // head slot + 1
      const u32 s = f.sfx[e];
      gs[e] = INIT ? p : aglob[p];
      nrs[e] = INIT ? hp1 - 1u : aglob[hp1 - 1u];
      grps[e] = g_run + (u32)__popcll(ac & hd & le) - 1u;      // dense number of p's group (if active)
      qs[e] = q_run + (u32)__popcll(ac & lt);
      if (EMIT == 1) cs[e] = f.chr[e];
      if (EMIT == 2 && (!a || MODE != 2)) cs[e] = s ? em.T[s - 1u] : 0u;
      if (EMIT == 3) cs[e] = s_inv[f.chr[e] & 0xFFu];
    }
    q_run += (u32)__popcll(ac);
    g_run += (u32)__popcll(ac & hd);
    if (hd) hp_run = base + top_bit(hd) + 1u;
  }
#pragma unroll
  for (int e = 0; e < kRrE; ++e) {
    const u32 p = wbase + e * kWave + lane;
    if ((f.valid[e] >> lane) & 1ull) {
      const bool a = (f.act[e] >> lane) & 1ull;
      const u32 s = f.sfx[e], g = gs[e], nr = nrs[e], grp = grps[e], c = cs[e];
      if (MODE == 2) {
        reinterpret_cast<u64*>(pair_s)[p] = ((u64)s << 32) | (u64)nr;
        pair_r[p] = a ? grp : 0xFFFFFFFFu;
        if (em.rec_plane) em.rec_plane[p] = (u8)(s >> em.rec_shift);
      } else if (MODE == 1) { pair_s[p] = s; pair_r[p] = nr; }
      else if (MODE == 0) { if (rank) rank[s] = nr; }
      if (a) {
        const u32 q = qs[e];
        aglob_out[q] = g;
        if (MODE != 2) {
          aidx_out[q] = s;
          agrp_out[q] = MODE == 3 ? nr : grp;
          if (MODE == 3) reinterpret_cast<unsigned short*>(em.achr_out)[q] = (unsigned short)(c | (f.chr[e] & 0xFF00u));   // character, known depth
          else if (EMIT) em.achr_out[q] = (u8)c;
        }
      } else {
        if (SA) SA[g] = s;
        if (EMIT) {
          if (s == 0u) *em.pidx = g;
          if (g < em.out_n) em.out[g] = (u8)c;
          else *em.last_char = c;
          lf_note(em, n, s, g);                    // nothing unless the caller asked for notes (lf_n)
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------
// Finisher (after the long-key sort): the suffixes that are still tied form groups -- contiguous in
// slot order -- whose members are known to share `depth` characters (a number every list entry
// carries: code keys cover more characters of a frequent string than of a rare one).  Prefix doubling
// would first have to complete rank[] for ALL suffixes (an ISA scatter of the whole block) before it
// could look at rank[s+h]; here a group is settled by comparing its members' next kFinChars characters
// directly.  The list entry of a suffix is (suffix, slot, slot of its group's head, predecessor
// character | depth << 8).
//
// A workgroup looks at a window of kFinWin consecutive entries and owns the groups that START in its
// first kFinWin - kFinMaxGroup entries and have at most kFinMaxGroup members (so they lie inside the
// window).  It stages the owned members' characters in LDS (the route's one random access: three
// aligned words of T per member) and every member counts the members of its group below it and equal
// to it.  (Round 5 also measured a bitonic network over the window instead of that loop -- 55 steps
// whatever the groups look like: 87 ps per entry on a real text whose groups average 34 members,
// against about 25 for the loop, whose cost is the group's size; the network was dropped.)
// A member without equals is final: out[slot], SA[slot], the LF powers.  The others form the next list
// in their new order (sub-group head = old head + members below), sixteen characters deeper, appended
// with one atomic per workgroup.
//
// Larger groups are nobody's: their entries go to the "hard" list, appended by the workgroup whose
// stride range they lie in, and take the text rounds / doubling rounds (bridge, below).
// "Proper prefix sorts first": characters past the end of T compare as zero bytes, and of two members
// that are equal under that rule and of which one reaches past the end, the shorter (larger suffix
// number) is the smaller -- they are never equal (the network's comparison knows).
// ---------------------------------------------------------------------------------------
#ifndef BWTC_FIN_TPB
#define BWTC_FIN_TPB 256
#endif
constexpr int kFinTPB = BWTC_FIN_TPB;             // threads of a finisher workgroup; a window is 1024 or 2048 entries
constexpr int kFinE1 = 1024 / kFinTPB, kFinE2 = 2048 / kFinTPB;   // entries per thread
constexpr u32 kFinNone = 0xFFFFFFFFu;

struct FinList { u32* S; u32* P; u32* H; unsigned short* C; };
// A list lives in up to kFinRegions regions of its arrays (a pass appends what it leaves with one atomic per workgroup:
// 300 000 workgroups on ONE counter were 3.4 ms of serialised atomics per pass over a real text, so workgroup b appends to
// one of 16 regions, each with a counter of its own: chunks of 64 consecutive workgroups take the regions in turn -- the
// workgroups on the chip at any moment spread over all of them, and a chunk leaves at most its 64 strides of entries plus
// one group, so the regions' sizes add up to the list's plus half a per cent).  Windows never straddle regions (groups
// do not: a workgroup's output goes to one region).
constexpr u32 kFinChunk = 64;             // (kFinRegions, FinRegions: bwt_engine.hpp)
struct FinOut { u32 base[kFinRegions]; };                 // where each region of the list a pass leaves starts

// sixteen characters T[pos .. pos+16) as two big-endian words; bytes at or past n read as zero
__device__ __forceinline__ void fin_chars(const u8* __restrict__ T, u32 pos, u32 n, u64* c0, u64* c1) {
  *c0 = 0; *c1 = 0;
  if (pos >= n) return;
  const u64* a = reinterpret_cast<const u64*>(T + (pos & ~7u));      // T is 256-byte aligned and padded
  const u32 sh = (pos & 7u) * 8u;
  const u64 w0 = a[0], w1 = a[1], w2 = a[2];
  u64 lo = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;               // bytes pos .. pos+7, first byte lowest
  u64 hi = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;
  const u32 left = n - pos;                                          // real bytes from pos on
  if (left < 16u) {
    if (left <= 8u) { hi = 0; if (left < 8u) lo &= (1ull << (8u * left)) - 1ull; }
    else hi &= (1ull << (8u * (left - 8u))) - 1ull;
  }
  *c0 = __builtin_bswap64(lo);
  *c1 = __builtin_bswap64(hi);
}

// 8 NW characters T[pos .. pos + 8 NW) as NW big-endian words; bytes at or past n read as zero
template <int NW>
__device__ __forceinline__ void fin_chars_n(const u8* __restrict__ T, u32 pos, u32 n, u64 (&c)[NW]) {
#pragma unroll
  for (int q = 0; q < NW; ++q) c[q] = 0;
  if (pos >= n) return;
  const u64* a = reinterpret_cast<const u64*>(T + (pos & ~7u));      // T is 256-byte aligned and padded
  const u32 sh = (pos & 7u) * 8u;
  u64 w[NW + 1];
#pragma unroll
  for (int q = 0; q <= NW; ++q) w[q] = a[q];
  const u32 left = n - pos;                                          // real bytes from pos on
#pragma unroll
  for (int q = 0; q < NW; ++q) {
    u64 v = sh ? (w[q] >> sh) | (w[q + 1] << (64u - sh)) : w[q];     // bytes pos + 8 q .. + 7, first byte lowest
    if (left < 8u * (u32)(q + 1)) v = left <= 8u * (u32)q ? 0ull : v & ((1ull << (8u * (left - 8u * (u32)q))) - 1ull);
    c[q] = __builtin_bswap64(v);
  }
}

// The finisher's comparison loop: how many members of the group at window positions [a, jend) are below the member at
// position lp, whose key is k (NW big-endian words)?  Four members per step, their LDS reads issued together (one member
// per step was a chain of LDS latencies); positions past the group's end are read (inside the arrays: they are padded)
// and not counted.  any_touch (the same for the whole workgroup): some member reaches past the end of T -- "proper
// prefix sorts first": characters past the end compare as zero bytes, and of two members equal under that rule of which
// one reaches past the end the shorter -- the larger suffix number -- is the smaller; they are never equal.
template <int NW>
__device__ __forceinline__ u32 fin_count_below(const u64* s_ch, const u32* s_S, u32 a, u32 jend, u32 lp, u32 sfx,
                                               const u64 (&k)[NW], bool any_touch) {
  u32 below = 0;
  if (!any_touch) {
    for (u32 j0 = a; j0 < jend; j0 += 4u) {
      u64 dd[4][NW];
#pragma unroll
      for (u32 u = 0; u < 4u; ++u) {
#pragma unroll
        for (int q = 0; q < NW; ++q) dd[u][q] = s_ch[(u32)NW * (j0 + u) + (u32)q];
      }
      // (no short cuts: with `||` and `&&` the compiler made the second word's read and comparison a branch per
      // member -- four dependent LDS round trips a step)
#pragma unroll
      for (u32 u = 0; u < 4u; ++u) {
        u32 lt = (u32)(dd[u][NW - 1] < k[NW - 1]);
#pragma unroll
        for (int q = NW - 2; q >= 0; --q) lt = (u32)(dd[u][q] < k[q]) | ((u32)(dd[u][q] == k[q]) & lt);
        below += lt & (u32)(j0 + u < jend);
      }
    }
  } else {
    const u32 me = s_S[lp];
    for (u32 j = a; j < jend; ++j) {
      const u32 sj = s_S[j];
      bool lt = false, eq = true;
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const u64 d = s_ch[(u32)NW * j + (u32)q];
        lt = lt || (eq && d < k[q]);
        eq = eq && d == k[q];
      }
      if (eq && ((sj | me) >> 31) && j != lp) lt = (sj & 0x7FFFFFFFu) > sfx;
      below += lt ? 1u : 0u;
    }
  }
  return below;
}

// hard_count[0]: entries of the hard list, [1]: the smallest depth among them.  Hard entries whose group shares fewer
// than shal.floor characters go to a list of their own (shal.count[0], [1]: entries, smallest depth): the rounds work
// at ONE depth for all their groups, the smallest, and a few groups of some exotic but repetitive string (bytes the
// code's sample never saw: a depth of one or two characters) would drag a whole text's hard groups down to theirs.
struct FinShallow { u32* S; u64* HP; u32* count; u32 floor; };
// RANK (FinRank rk; NW == 2): the members' keys are not characters but ranks -- rank[s + at] (and rank[s + at2]), the
// doubling step of Larsson and Sadakane done group by group in LDS instead of by a global sort of (group, rank): for the
// many small groups of deep repeats (copies of whole files).  rank[] must not change while a pass reads it, so the
// pass only NOTES every member's new rank (its slot when final, else its sub-group's head), by list position, and
// k_rank_updates writes them afterwards.
struct FinRank { const u32* rank; u32 at, at2; u32* us; u32* ur; };   // at2 == 0: one look-up
// NR (characters only; RANK: 1): the rounds of 8 NW characters a pass makes -- the members that a round leaves tied are
// compared again, sub-group by sub-group, without leaving the workgroup.  (A template parameter: with the rounds
// unrolled the kernel needs 30 registers less than with a loop over them, a workgroup more per CU.)
#ifdef BWTC_FIN_WAVES
#define BWTC_FIN_BOUNDS __launch_bounds__(kFinTPB, BWTC_FIN_WAVES)
#else
#define BWTC_FIN_BOUNDS __launch_bounds__(kFinTPB)
#endif
template <int kFinMaxGroup, int E_ = 4, int NW = 2, bool RANK = false, int NR = 1>
__global__ BWTC_FIN_BOUNDS void k_finish(FinList in, FinRegions rg, const u8* __restrict__ T, u32 n,
                                                    FinList next, FinOut ob, u32* __restrict__ next_count,
                                                    u32* __restrict__ hardS, u64* __restrict__ hardHP, unsigned short* __restrict__ hardC,
                                                    u32* __restrict__ hard_count, FinShallow shal, u32* __restrict__ SA, RrEmit em,
                                                    FinRank rk = FinRank()) {
  static_assert(!RANK || (NW == 2 && NR == 1), "rank keys are two words, one round");
  constexpr u32 nr = NR;
  constexpr int kFinE = E_;                                           // entries per thread: a window of 1024 or 2048
  constexpr int kFinWin = kFinTPB * kFinE;
  static_assert(kFinMaxGroup < kFinWin, "a group must fit the window");
  constexpr int kFinStride = kFinWin - kFinMaxGroup;                  // entries whose groups a workgroup owns
  constexpr u32 kChars = 8u * NW * NR;                                // characters compared by this pass
  __shared__ __attribute__((aligned(16))) u64 s_ch[(kFinWin + 4) * NW];   // the characters (+ padding for the comparison loop's reads); later the reorder staging
  __shared__ u32 s_S[kFinWin + 4];                                    // suffix | reaches past the end << 31
  __shared__ u32 s_H[kFinWin];
  __shared__ unsigned short s_g[kFinWin + 2];                         // group size by start position; [kFinWin]: the group cut by the window's start
  __shared__ u32 s_cnt2[NR > 1 ? kFinWin : 1];                        // the sub-group counters of the odd rounds (the even ones': s_H's bytes)
  __shared__ u32 scr[kFinTPB / kWave + 1];
  __shared__ u32 s_base;
  u32* st_S = reinterpret_cast<u32*>(s_ch);                           // staging, by new position
  u32* st_P = st_S + kFinWin;
  u32* st_H = st_P + kFinWin;                                         // | final << 31
  unsigned short* st_C = reinterpret_cast<unsigned short*>(st_H + kFinWin);
  const u32 tid = threadIdx.x;
  u32 reg = 0;
  while (reg + 1u < rg.nreg && blockIdx.x >= rg.wfirst[reg + 1u]) ++reg;
  const u32 w0 = rg.ebase[reg] + (blockIdx.x - rg.wfirst[reg]) * (u32)kFinStride;
  const u32 m = rg.ebase[reg] + rg.ecount[reg];                       // one past the region's last entry
  const u32 lane = lane_id();

  u32 S[kFinE], P[kFinE], H[kFinE], C[kFinE];
#pragma unroll
  for (int e = 0; e < kFinE; ++e) {
    const u32 lp = tid + (u32)e * kFinTPB, q = w0 + lp;
    const bool ok = q < m;
    S[e] = ok ? in.S[q] : 0u; P[e] = ok ? in.P[q] : 0u; H[e] = ok ? in.H[q] : kFinNone; C[e] = ok ? in.C[q] : 0u;
    s_H[lp] = H[e];
    s_g[lp] = 0xFFFFu;
  }
  if (tid < 2) s_g[kFinWin + tid] = 0xFFFFu;
  __syncthreads();
  // group sizes, from each group's last member
#pragma unroll
  for (int e = 0; e < kFinE; ++e) {
    const u32 lp = tid + (u32)e * kFinTPB, q = w0 + lp;
    if (q < m) {
      const bool last = (q + 1u == m) || (lp + 1u < (u32)kFinWin && s_H[lp + 1u] != H[e]);
      const u32 off = P[e] - H[e];
      if (last && off < (u32)kFinMaxGroup) {
        const int a = (int)lp - (int)off;
        s_g[a >= 0 ? a : kFinWin] = (unsigned short)(off + 1u);
      }
    }
  }
  __syncthreads();
  int A[kFinE];                   // group start (window position), or -1: not this workgroup's to settle
  u32 G[kFinE];
  u32 hard_mask = 0;
  u32* s_cnt = s_H;               // (the heads are in registers by now) members per new sub-group, by its first position
#pragma unroll
  for (int e = 0; e < kFinE; ++e) {
    const u32 lp = tid + (u32)e * kFinTPB, q = w0 + lp;
    A[e] = -1; G[e] = 0;
    s_cnt[lp] = 0u;
    if (q < m) {
      const int a = (int)lp - (int)(P[e] - H[e]);
      const u32 g = (P[e] - H[e] < (u32)kFinMaxGroup) ? s_g[a >= 0 ? a : kFinWin] : 0xFFFFu;
      if (g == 0xFFFFu) { if (lp < (u32)kFinStride) hard_mask |= 1u << e; }
      else if (a >= 0 && a < kFinStride) { A[e] = a; G[e] = g; }
    }
  }
  // Rounds.  A round compares 8 NW characters (RANK: the ranks, one round): every member counts the members of its
  // (sub-)group BELOW it (fin_count_below; broadcast LDS reads: the lanes of a wave walk a few neighbouring groups).
  // That number names its new sub-group -- equal members count the same -- and its place among its equals is the order
  // in which they arrive at the sub-group's counter (any order will do: they are a set until something tells them
  // apart).  The loop carries nothing else: a real text's first pass ran 1560 instructions per entry when it also
  // counted the equal members before and after.  The next round compares the sub-groups that are left, at their new
  // places, by the next characters -- read with the first round's: they come out of the cache line the first round paid
  // for, and a pass is bound by its random reads of the text, 149 bytes of HBM traffic per entry for the first sixteen
  // characters.
  const u32 kRound = 8u * NW;
  u64 all[kFinE][NW * NR];        // every round's characters, read in one go (RANK: the keys)
  u32 a_cur[kFinE], g_cur[kFinE], pos[kFinE];
  u32 slot0[kFinE];               // the slot of window position 0, were the group to start there (head slot - group start)
  int touches = 0;
#pragma unroll
  for (int e = 0; e < kFinE; ++e) {
    const u32 lp = tid + (u32)e * kFinTPB;
    a_cur[e] = A[e] >= 0 ? (u32)A[e] : 0u; g_cur[e] = A[e] >= 0 ? G[e] : 0u; pos[e] = lp;
    slot0[e] = H[e] - a_cur[e];
#pragma unroll
    for (int q = 0; q < NW * NR; ++q) all[e][q] = 0;
    if (A[e] >= 0) {
      if (RANK) {
        const u64 t1 = (u64)S[e] + rk.at, t2 = (u64)S[e] + rk.at2;
        all[e][0] = t1 < (u64)n ? (u64)rk.rank[t1] + 1ull : 0ull;            // (a suffix that ends before the look-up: the smallest key)
        all[e][NW - 1] = (rk.at2 && t2 < (u64)n) ? (u64)rk.rank[t2] + 1ull : 0ull;
      } else {
        // every round's characters in one go: the later rounds' come out of the line(s) the first round's are in.  (Loaded
        // a round later, while the round before counts, the line has left L2 again: 300 bytes of traffic per entry instead
        // of 150, and 1.2 ms more per pass over 233 M entries.)
        fin_chars_n<NW * NR>(T, S[e] + (C[e] >> 8), n, all[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < kFinE; ++e) {
    if (A[e] >= 0) {
      const u32 lp = tid + (u32)e * kFinTPB;
      const u32 touch = (!RANK && (u64)S[e] + (C[e] >> 8) + kRound > (u64)n) ? 0x80000000u : 0u;        // (in the first round)
#pragma unroll
      for (int q = 0; q < NW; ++q) s_ch[(u32)NW * lp + (u32)q] = all[e][q];
      s_S[lp] = S[e] | touch;
      touches |= (!RANK && (u64)S[e] + (C[e] >> 8) + kChars > (u64)n) ? 1 : 0;                           // (in any round)
    }
  }
  // (a member that reaches past the end of T: one workgroup in a pass, if any -- everybody else runs the lean loop)
  const bool any_touch = __syncthreads_or(touches) != 0;
#pragma unroll
  for (u32 r = 0; r < nr; ++r) {
    u32* cnt = (r & 1u) ? s_cnt2 : s_cnt;              // (the first round's was cleared above)
    if (r > 0) {
#pragma unroll
      for (int e = 0; e < kFinE; ++e) {
        cnt[tid + (u32)e * kFinTPB] = 0u;
        if (g_cur[e] > 1u) {
#pragma unroll
          for (int q = 0; q < NW; ++q) s_ch[(u32)NW * pos[e] + (u32)q] = all[e][r * NW + q];
          s_S[pos[e]] = S[e] | (((u64)S[e] + (C[e] >> 8) + (r + 1u) * kRound > (u64)n) ? 0x80000000u : 0u);
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < kFinE; ++e) {
      if (g_cur[e] > 1u) {
        u64 key[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) key[q] = all[e][r * NW + q];
        const u32 below = fin_count_below<NW>(s_ch, s_S, a_cur[e], a_cur[e] + g_cur[e], pos[e], S[e], key, any_touch);
        a_cur[e] += below;
        if (RANK) {               // the member's new rank, noted by list position (kFinNone: it keeps the one it has)
          const u32 q = w0 + pos[e];                                 // (a final member's slot is its sub-group's head)
          rk.us[q] = below ? S[e] : kFinNone;
          rk.ur[q] = slot0[e] + a_cur[e];
        }
        pos[e] = a_cur[e] + atomicAdd(&cnt[a_cur[e]], 1u);           // its place among its equals: as they arrive
      }
    }
    __syncthreads();               // every read of the round's characters is done, every arrival counted
#pragma unroll
    for (int e = 0; e < kFinE; ++e) {
      if (g_cur[e] > 1u) g_cur[e] = cnt[a_cur[e]];
    }
  }
  // (the characters' bytes become the staging)
#pragma unroll
  for (int e = 0; e < kFinE; ++e) st_P[tid + (u32)e * kFinTPB] = kFinNone;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < kFinE; ++e) {
    if (g_cur[e]) {                                                            // (a member this workgroup settles)
      const u32 np = pos[e];
      const u32 fin = g_cur[e] == 1u ? 0x80000000u : 0u;                       // nobody equals it
      st_S[np] = S[e];
      st_P[np] = slot0[e] + np;
      st_H[np] = (slot0[e] + a_cur[e]) | fin;
      st_C[np] = RANK ? (unsigned short)C[e] : (unsigned short)((C[e] & 0xFFu) | (min(255u, (C[e] >> 8) + kChars) << 8));
    }
  }
  // the hard entries of this workgroup's stride range, in any order; one atomic per workgroup and list (a wave's own
  // atomic per iteration was 600 000 on one address for a real text's 40 M hard entries)
  u32 hard_depth = 0xFFFFFFFFu, shal_depth = 0xFFFFFFFFu;
  {
    u32 hm = 0, sm = 0;                                   // this thread's hard entries: of the hard list, of the shallow one
#pragma unroll
    for (int e = 0; e < kFinE; ++e) {
      if ((hard_mask >> e) & 1u) { if ((C[e] >> 8) < shal.floor) sm |= 1u << e; else hm |= 1u << e; }
    }
    __shared__ u32 s_hbase[2];
    u32 htot, stot;
    u32 hat = block_scan_excl_add<kFinTPB>((u32)__popc(hm), scr, &htot);
    u32 sat = block_scan_excl_add<kFinTPB>((u32)__popc(sm), scr, &stot);
    if (tid == 0) { s_hbase[0] = htot ? atomicAdd(hard_count, htot) : 0u; s_hbase[1] = stot ? atomicAdd(shal.count, stot) : 0u; }
    __syncthreads();
    hat += s_hbase[0]; sat += s_hbase[1];
#pragma unroll
    for (int e = 0; e < kFinE; ++e) {
      if ((hm >> e) & 1u) {
        hardS[hat] = S[e]; hardHP[hat] = ((u64)H[e] << 32) | (u64)P[e]; hardC[hat] = (unsigned short)C[e]; ++hat;
        hard_depth = min(hard_depth, C[e] >> 8);
      } else if ((sm >> e) & 1u) {
        shal.S[sat] = S[e]; shal.HP[sat] = ((u64)H[e] << 32) | (u64)P[e]; ++sat;
        shal_depth = min(shal_depth, C[e] >> 8);
      }
    }
  }
  // (one atomic per wave that has something to say, and only when it lowers what is there)
  for (int o = kWave / 2; o > 0; o >>= 1) {
    hard_depth = min(hard_depth, (u32)__shfl_xor(hard_depth, o, kWave));
    shal_depth = min(shal_depth, (u32)__shfl_xor(shal_depth, o, kWave));
  }
  if (lane == 0 && hard_depth != 0xFFFFFFFFu && hard_depth < __hip_atomic_load(hard_count + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(hard_count + 1, hard_depth);
  if (lane == 0 && shal_depth != 0xFFFFFFFFu && shal_depth < __hip_atomic_load(shal.count + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(shal.count + 1, shal_depth);
  __syncthreads();
  // in position order: finals leave, the rest is compacted into the next list
  u32 oS[kFinE], oP[kFinE], oH[kFinE], oC[kFinE];
  u32 left = 0, heads = 0;
#pragma unroll
  for (int e = 0; e < kFinE; ++e) {
    const u32 lp = tid * (u32)kFinE + (u32)e;
    oP[e] = st_P[lp]; oS[e] = st_S[lp]; oH[e] = st_H[lp]; oC[e] = st_C[lp];
    if (oP[e] != kFinNone && !(oH[e] >> 31)) { ++left; if (oP[e] == oH[e]) ++heads; }
  }
  u32 total, total_heads;
  (void)block_scan_excl_add<kFinTPB>(heads, scr, &total_heads);
  u32 at = block_scan_excl_add<kFinTPB>(left, scr, &total);
  if (tid == 0) {
    const u32 oreg = (blockIdx.x / kFinChunk) % kFinRegions;
    s_base = ob.base[oreg] + (total ? atomicAdd(&next_count[oreg], total) : 0u);
    if (total_heads) atomicAdd(&next_count[kFinRegions], total_heads);      // groups of the list this pass leaves
  }
  __syncthreads();
  at += s_base;
#pragma unroll
  for (int e = 0; e < kFinE; ++e) {
    if (oP[e] == kFinNone) continue;
    if (oH[e] >> 31) {
      const u32 slot = oP[e], sfx = oS[e];
      SA[slot] = sfx;
      if (sfx == 0u) *em.pidx = slot;
      if (slot < em.out_n) em.out[slot] = (u8)oC[e];
      else *em.last_char = oC[e] & 0xFFu;
      lf_note(em, n, sfx, slot);
    } else {
      next.S[at] = oS[e]; next.P[at] = oP[e]; next.H[at] = oH[e]; next.C[at] = (unsigned short)oC[e];
      ++at;
    }
  }
}

// ---- bridge from the finisher to the doubling rounds: what is still tied (the hard list, with the
// last pass's leftovers appended) is sorted by slot and dressed up as a list the rounds understand
// (key = head slot | character << 56, value = suffix, positional slot array), and rank[] is completed
// for everybody: rank[SA[slot]] = slot for the finished, the group's head slot for the others.
// (entries of little depth -- a list no pass has looked at -- go to the shallow list instead: one atomic per entry, they are few;
// their places in the waiting list stay empty: suffix kFinNone, dropped when the list is dressed)
__global__ __launch_bounds__(256) void k_fin_to_hard(FinList in, u32 m, u32* __restrict__ hardS,
                                                     u64* __restrict__ hardHP, u32 at, u32* __restrict__ min_depth, FinShallow shal) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  u32 d = 0xFFFFFFFFu;
  if (i < m) {
    const u32 dep = in.C[i] >> 8;
    const u64 hp = ((u64)in.H[i] << 32) | (u64)in.P[i];
    if (dep < shal.floor) {
      const u32 j = atomicAdd(shal.count, 1u);
      shal.S[j] = in.S[i]; shal.HP[j] = hp;
      atomicMin(shal.count + 1, dep);
      hardS[at + i] = kFinNone; hardHP[at + i] = ~0ull;
    } else {
      hardS[at + i] = in.S[i];
      hardHP[at + i] = hp;
      d = dep;
    }
  }
  for (int o = kWave / 2; o > 0; o >>= 1) d = min(d, (u32)__shfl_xor(d, o, kWave));
  // the depth the rounds may start from (an atomic only where it lowers what is there: one per wave was 15 ms for 85 M entries)
  if (lane_id() == 0 && d != 0xFFFFFFFFu && d < __hip_atomic_load(min_depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(min_depth, d);
}
// rank[us[q]] = ur[q] for the list positions q of one region (k_finish<RANK>'s notes)
__global__ __launch_bounds__(256) void k_rank_updates(const u32* __restrict__ us, const u32* __restrict__ ur, u32 base, u32 count,
                                                      u32* __restrict__ rank) {
  const u32 i0 = blockIdx.x * 1024u + threadIdx.x;
  u32 s[4], r[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { const u32 i = i0 + (u32)e * 256u; s[e] = i < count ? us[base + i] : kFinNone; r[e] = i < count ? ur[base + i] : 0u; }
#pragma unroll
  for (int e = 0; e < 4; ++e) if (s[e] != kFinNone) rank[s[e]] = r[e];
}
// one region of a list, copied to the end of another list's arrays
__global__ __launch_bounds__(256) void k_list_copy(FinList in, u32 m, FinList out, u32 at, u32* __restrict__ min_depth) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  u32 d = 0xFFFFFFFFu;
  if (i < m) {
    const unsigned short c = in.C[i];
    out.S[at + i] = in.S[i]; out.P[at + i] = in.P[i]; out.H[at + i] = in.H[i]; out.C[at + i] = c;
    d = c >> 8;
  }
  for (int o = kWave / 2; o > 0; o >>= 1) d = min(d, (u32)__shfl_xor(d, o, kWave));
  if (lane_id() == 0 && d != 0xFFFFFFFFu && d < __hip_atomic_load(min_depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(min_depth, d);
}
// complete_ranks for a list that waits in finisher form: its suffixes rank as their groups' heads
__global__ __launch_bounds__(256) void k_bridge_pairs_fix_list(FinList in, u32 m, u32* __restrict__ pair_s, u32* __restrict__ pair_r) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= m) return;
  const u32 p = in.P[i];
  pair_s[p] = in.S[i];
  pair_r[p] = in.H[i];
}
__global__ __launch_bounds__(256) void k_raw_to_park(const u32* __restrict__ S, const u32* __restrict__ P, const u32* __restrict__ H, u32 m,
                                                     u32* __restrict__ parkS, u64* __restrict__ parkHP, u32 at) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= m) return;
  parkS[at + i] = S[i];
  parkHP[at + i] = ((u64)H[i] << 32) | (u64)P[i];
}
__global__ __launch_bounds__(256) void k_bridge_dress(const u64* __restrict__ hp, const u32* __restrict__ sfx, u32 m,
                                                      const u8* __restrict__ T, u64* __restrict__ key,
                                                      u32* __restrict__ aglob) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= m) return;
  const u64 x = hp[i];
  const u32 s = sfx[i];
  key[i] = (x >> 32) | ((u64)(s ? T[s - 1u] : 0u) << 56);
  aglob[i] = (u32)x;
}
__global__ __launch_bounds__(256) void k_bridge_pairs_all(const u32* __restrict__ SA, u32 n,
                                                          u32* __restrict__ pair_s, u32* __restrict__ pair_r) {
  const u32 p0 = blockIdx.x * 1024u + threadIdx.x;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const u32 p = p0 + (u32)e * 256u;
    if (p < n) { pair_s[p] = SA[p]; pair_r[p] = p; }      // unfinished slots hold anything: k_bridge_pairs_fix rewrites them
  }
}
__global__ __launch_bounds__(256) void k_bridge_pairs_fix(const u32* __restrict__ sfx, const u32* __restrict__ aglob, u32 m,
                                                          u32* __restrict__ pair_s, u32* __restrict__ pair_r) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= m) return;
  const u32 p = aglob[i];
  pair_s[p] = sfx[i];
  pair_r[p] = p;
}

// the same for a list that waits for its turn (its groups' head slots are its suffixes' ranks until then)
__global__ __launch_bounds__(256) void k_bridge_pairs_fix_parked(const u32* __restrict__ sfx, const u64* __restrict__ hp, u32 m,
                                                                 u32* __restrict__ pair_s, u32* __restrict__ pair_r) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= m) return;
  const u64 x = hp[i];
  pair_s[(u32)x] = sfx[i];
  pair_r[(u32)x] = (u32)(x >> 32);
}

constexpr int kSimpleE = 4;   // independent items per thread in the latency-bound kernels

// ---------------------------------------------------------------------------------------
// Window kernels.  The lists they sweep are partitioned by suffix window (the top bits of s),
// so a chunk of consecutive items touches one or two windows of rank[] / T -- but inside a
// window the order is arbitrary, and a wave's 64 accesses still hit 64 different cache lines:
// rocprofv3 showed these kernels stalled on instruction issue (SQ_WAIT_INST_ANY 0.4-0.5 of the
// wave cycles, the address pipe working through one line per lane), not on HBM.  Each workgroup
// therefore reorders its chunk in LDS by eight more bits of s first -- an unstable counting sort,
// one returning ds_add per item gives both the bin count and the item's place in its bin -- and
// performs the accesses in that order: neighbouring lanes then share cache lines.  The order in
// which a chunk's items are processed (and, for the gather, written back) does not matter.
// ---------------------------------------------------------------------------------------
constexpr int kWinTPB = 256;
constexpr int kWinE = 8;
constexpr int kWinTile = kWinTPB * kWinE;     // items per workgroup

// XCD-contiguous chunking (speed only): the lists are partitioned by destination window, so
// giving each XCD one contiguous eighth of the list keeps every window in ONE L2.
__device__ __forceinline__ bool win_chunk(u32 m, u32* blk_out) {
  const u32 nblk = (m + (u32)kWinTile - 1u) / (u32)kWinTile;
  const u32 per_xcd = (nblk + 7u) / 8u;
  const u32 blk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  *blk_out = blk;
  return blk < nblk;
}

// Places of the chunk's items in bin order.  bin[e] (< 256) of this thread's item e (ok bit e of
// okm) -> slot[e] in [0, count); s_cnt is 256 words of LDS, scr the block scan's scratch.
// Contains barriers; every thread of the workgroup must call it.
__device__ __forceinline__ void win_places(const u32 (&bin)[kWinE], u32 okm, u32 (&slot)[kWinE],
                                           u32* s_cnt, u32* scr) {
  s_cnt[threadIdx.x] = 0;                                 // kWinTPB == 256 bins
  __syncthreads();
#pragma unroll
  for (int e = 0; e < kWinE; ++e) slot[e] = ((okm >> e) & 1u) ? atomicAdd(&s_cnt[bin[e]], 1u) : 0u;
  __syncthreads();
  u32 total;
  const u32 base = block_scan_excl_add<kWinTPB>(s_cnt[threadIdx.x], scr, &total);
  s_cnt[threadIdx.x] = base;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < kWinE; ++e) slot[e] += s_cnt[bin[e]];
}

// rank[where[i]] = what[i] for pairs partitioned by the high bits of `where`.
__global__ __launch_bounds__(kWinTPB) void k_scatter_pairs(u32* __restrict__ dst,
                                                           const u32* __restrict__ where,
                                                           const u32* __restrict__ what, u32 m,
                                                           int bin_shift) {
  __shared__ u32 s_cnt[256], scr[kWinTPB / kWave + 1];
  __shared__ u32 s_w[kWinTile], s_v[kWinTile];
  u32 blk;
  if (!win_chunk(m, &blk)) return;
  const u32 p0 = blk * (u32)kWinTile + threadIdx.x;
  u32 w[kWinE], v[kWinE], bin[kWinE], slot[kWinE], okm = 0;
#pragma unroll
  for (int e = 0; e < kWinE; ++e) {
    const u32 p = p0 + e * kWinTPB;
    const bool ok = p < m;
    w[e] = ok ? where[p] : 0u;
    v[e] = ok ? what[p] : 0u;
    bin[e] = (w[e] >> bin_shift) & 255u;
    okm |= (ok ? 1u : 0u) << e;
  }
  win_places(bin, okm, slot, s_cnt, scr);
#pragma unroll
  for (int e = 0; e < kWinE; ++e) if ((okm >> e) & 1u) { s_w[slot[e]] = w[e]; s_v[slot[e]] = v[e]; }
  __syncthreads();
  const u32 count = min((u32)kWinTile, m - blk * (u32)kWinTile);
  for (u32 i = threadIdx.x; i < count; i += kWinTPB) dst[s_w[i]] = s_v[i];
}

// Dense route, step 2: rank[s] = nr for records (s << 32 | nr) partitioned by the high bits of s.
__global__ __launch_bounds__(kWinTPB) void k_scatter_dense(u32* __restrict__ rank,
                                                           const u64* __restrict__ rec, u32 m,
                                                           int bin_shift) {
  __shared__ u32 s_cnt[256], scr[kWinTPB / kWave + 1];
  __shared__ u64 s_r[kWinTile];
  u32 blk;
  if (!win_chunk(m, &blk)) return;
  const u32 p0 = blk * (u32)kWinTile + threadIdx.x;
  u64 r[kWinE];
  u32 bin[kWinE], slot[kWinE], okm = 0;
#pragma unroll
  for (int e = 0; e < kWinE; ++e) {
    const u32 p = p0 + e * kWinTPB;
    const bool ok = p < m;
    r[e] = ok ? rec[p] : 0ull;
    bin[e] = ((u32)(r[e] >> 32) >> bin_shift) & 255u;
    okm |= (ok ? 1u : 0u) << e;
  }
  win_places(bin, okm, slot, s_cnt, scr);
#pragma unroll
  for (int e = 0; e < kWinE; ++e) if ((okm >> e) & 1u) s_r[slot[e]] = r[e];
  __syncthreads();
  const u32 count = min((u32)kWinTile, m - blk * (u32)kWinTile);
  for (u32 i = threadIdx.x; i < count; i += kWinTPB) { const u64 x = s_r[i]; rank[(u32)(x >> 32)] = (u32)x; }
}

// Dense route, step 3 (after ALL of rank[] is updated): the next round's sort input, written
// over the records.  Record p of suffix s with dense group number grp (all ones: finished)
// becomes key = grp << b2 | rank[s+h]+1 (0 when s+h is past the end) | T[s-1] << 56 (emit),
// value = s; finished suffixes become the all-ones key, which the sort's first pass drops.
// The records are in window order of s and h is small against a window at the dense stage, so
// rank[s+h] and T[s-1] are read from the window being swept, not at random.  The chunk's items
// are written back in their LDS order (any order will do: the sort that follows does not care).
__global__ __launch_bounds__(kWinTPB) void k_gather_dense(u64* __restrict__ rec, u32* __restrict__ val,
                                                          const u32* __restrict__ rank,
                                                          const u8* __restrict__ T, u32 m, u32 n, u32 h,
                                                          int b2, int emit, int bin_shift) {
  __shared__ u32 s_cnt[256], scr[kWinTPB / kWave + 1];
  __shared__ u32 s_s[kWinTile], s_g[kWinTile];
  u32 blk;
  if (!win_chunk(m, &blk)) return;
  const u32 base = blk * (u32)kWinTile;
  const u32 p0 = base + threadIdx.x;
  u32 s[kWinE], g[kWinE], bin[kWinE], slot[kWinE], okm = 0;
#pragma unroll
  for (int e = 0; e < kWinE; ++e) {
    const u32 p = p0 + e * kWinTPB;
    const bool ok = p < m;
    s[e] = ok ? (u32)(rec[p] >> 32) : 0u;
    g[e] = ok ? val[p] : 0xFFFFFFFFu;
    bin[e] = (s[e] >> bin_shift) & 255u;
    okm |= (ok ? 1u : 0u) << e;
  }
  win_places(bin, okm, slot, s_cnt, scr);
#pragma unroll
  for (int e = 0; e < kWinE; ++e) if ((okm >> e) & 1u) { s_s[slot[e]] = s[e]; s_g[slot[e]] = g[e]; }
  __syncthreads();
  const u32 count = min((u32)kWinTile, m - base);
  for (u32 i = threadIdx.x; i < count; i += kWinTPB) {
    const u32 ss = s_s[i], gg = s_g[i];
    u64 key = ~0ull;
    if (gg != 0xFFFFFFFFu) {
      const u64 t = (u64)ss + (u64)h;
      const u32 r = t < (u64)n ? rank[t] + 1u : 0u;
      const u32 c = (emit && ss) ? T[ss - 1u] : 0u;
      key = ((u64)gg << b2) | (u64)r | ((u64)c << 56);
    }
    rec[base + i] = key;
    val[base + i] = ss;
  }
}

// ---------------------------------------------------------------------------------------
// K5  second sort key of a doubling round:  (group << b2) | key2,  key2 = rank[s+h] + 1,
//     or 0 for the one suffix whose h-successor is the empty suffix.
// ---------------------------------------------------------------------------------------
//     achr (may be null): the suffix's carried character, put into bits 56..63 of the key.
__global__ __launch_bounds__(256) void k_gather_key2(const u32* __restrict__ aidx,
                                                     const u32* __restrict__ agrp,
                                                     const u32* __restrict__ rank,
                                                     const u8* __restrict__ achr,
                                                     u64* __restrict__ key, u32 m, u32 n, u32 h,
                                                     int b2) {
  const u32 p0 = blockIdx.x * (256u * kSimpleE) + threadIdx.x;
  u64 t[kSimpleE];
  u32 r[kSimpleE], g[kSimpleE];
#pragma unroll
  for (int e = 0; e < kSimpleE; ++e) {
    const u32 p = p0 + e * 256u;
    t[e] = p < m ? (u64)aidx[p] + (u64)h : ~0ull;
    g[e] = p < m ? agrp[p] : 0u;
  }
#pragma unroll
  for (int e = 0; e < kSimpleE; ++e) r[e] = (t[e] < (u64)n) ? rank[t[e]] + 1u : 0u;
#pragma unroll
  for (int e = 0; e < kSimpleE; ++e) {
    const u32 p = p0 + e * 256u;
    if (p < m) key[p] = ((u64)g[e] << b2) | (u64)r[e] | (achr ? (u64)achr[p] << 56 : 0ull);
  }
}

// Text round (after the finisher, for the groups it could not take): the second sort key is the
// suffix's next c characters themselves, no rank[] needed:
//   key = group << (8c + 4) | characters << 4 | t,   t = c - (characters past the end of T), so that of
// suffixes equal on those characters the one that ends there comes first, shortest first.
__global__ __launch_bounds__(256) void k_gather_text(const u32* __restrict__ aidx, const u32* __restrict__ agrp,
                                                     const u8* __restrict__ T, const u8* __restrict__ achr,
                                                     u64* __restrict__ key, u32 m, u32 n, u32 h, u32 c) {
  const u32 p = blockIdx.x * 256u + threadIdx.x;
  if (p >= m) return;
  const u32 s = aidx[p];
  u64 c0, c1;
  fin_chars(T, s + h, n, &c0, &c1);                      // s + h < 2^31: h stays small in text rounds
  const u64 chars = c0 >> (64u - 8u * c);
  const u64 end = (u64)s + h + c;
  const u32 over = end > (u64)n ? (u32)min((u64)c, end - (u64)n) : 0u;
  key[p] = ((u64)agrp[p] << (8u * c + 4u)) | (chars << 4) | (u64)(c - over) | (achr ? (u64)achr[p] << 56 : 0ull);
}

// ---------------------------------------------------------------------------------------
// K9  BWT gather: out[j] = T[SA[j]-1]; the row of suffix 0 is the end-of-block row.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bwt_gather(const u32* __restrict__ SA,
                                                    const u8* __restrict__ T,
                                                    u8* __restrict__ out, u32 n,
                                                    u32* __restrict__ pidx) {
  const u32 w = blockIdx.x * 256u + threadIdx.x;
  const u32 j0 = w * 4u;
  if (j0 >= n) return;
  u32 word = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const u32 j = j0 + b;
    if (j < n) {
      const u32 s = SA[j];
      u32 c = 0;
      if (s) c = T[s - 1];
      else *pidx = j;
      word |= c << (8 * b);
    }
  }
  reinterpret_cast<u32*>(out)[w] = word;
}

// End-of-block patch + LF powers.  block mode: out[pidx] = out[n-1] (BWTransform.cpp:60);
// raw mode: out[pidx] = the input byte at pidx (divsufsort.c:507-511 leaves it untouched).
// LF[0] = pidx, LF[k] = ISA[n - k*(n/nLF)] (divsufsort.c:337-338,350,381,390).
// last_char (emitting rankers): the character of row n-1, which has no slot in a block's output.
// lf_noted: LF[1..] were noted as the suffixes became final (lf_note); rank[] is not complete then.
__global__ void k_finalize(u8* __restrict__ out, const u8* __restrict__ T,
                           const u32* __restrict__ rank, u32* __restrict__ lf, u32 n_lf, u32 n,
                           const u32* __restrict__ pidx, int raw, const u32* __restrict__ last_char, int lf_noted = 0) {
  const u32 p = *pidx;
  const u32 k = threadIdx.x;
  if (k == 0) {
    if (raw) out[p] = T[p];
    else if (p + 1u < n) out[p] = last_char ? (u8)*last_char : out[n - 1];
    lf[0] = p;
  } else if (k < n_lf && !lf_noted) {
    const u32 x = n / n_lf;
    lf[k] = rank[n - k * x];
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static u64 align_up(u64 v, u64 a) { return (v + a - 1) / a * a; }

struct ArenaPlan {
  u64 off_T, off_out, off_in, off_SA, off_rank, off_R1, off_R2, off_V0, off_V1, off_G0, off_G1,
      off_GRP, off_C0, off_C1, off_P0, off_P1, off_W0, off_W1, off_table, off_partial, off_aggA, off_aggB, off_aggC, off_agg_part, off_small, off_ent, off_comp, off_sweep, off_parkS, off_parkHP, off_hardS, off_hardHP, off_hardC, off_LP0, off_LH0, off_LH1, off_LC1, off_US, off_UR, off_pairs, off_codes, total;
};

static ArenaPlan plan_arena(u64 cap) {
  ArenaPlan a;
  u64 o = 0;
  auto take = [&](u64 bytes) { u64 r = o; o = align_up(o + bytes, 256); return r; };
  a.off_T = take(cap + kTextPad + kLoadTile);
  a.off_out = take(cap + 64);
  a.off_in = take(cap + 64);
  a.off_SA = take(cap * 4);
  a.off_rank = take(cap * 4);
  a.off_R1 = take(cap * 8);
  a.off_R2 = take(cap * 8);
  // (the finisher's lists live in regions sized for the most their workgroups can leave: up to 17 strides beyond the entries)
  const u64 lcap = cap + cap / 128 + 65536;
  a.off_V0 = take(lcap * 4);
  a.off_V1 = take(lcap * 4);
  a.off_G0 = take(lcap * 4);
  a.off_G1 = take(lcap * 4);
  a.off_GRP = take(lcap * 4);
  a.off_C0 = take(2 * lcap + 64);      // carried characters; the finisher's lists keep 16 bits per entry (character, depth)
  a.off_C1 = take(2 * lcap + 64);
  a.off_P0 = take(cap + 64);
  a.off_P1 = take(cap + 64);
  a.off_W0 = take(lcap * 4);           // (W0 and W1 are neighbours: together also one region of 8 * cap bytes)
  a.off_W1 = take(lcap * 4);
  a.off_table = take(radix_table_words(cap) * 4);
  a.off_partial = take(radix_partial_words(cap) * 4);
  const u64 rr_tiles = (cap + kRrTile - 1) / kRrTile + 1;
  a.off_aggA = take(rr_tiles * 4);
  a.off_aggB = take(rr_tiles * 4);
  a.off_aggC = take(rr_tiles * 4);
  a.off_agg_part = take((rr_tiles / 4096 + 2) * 12);
  a.off_small = take(1024 * 4);
  a.off_ent = take(2u << 20);
  a.off_comp = take(huffman_compress_bound(cap));
  a.off_sweep = take(SweepWs::small_words() * 4);
  a.off_parkS = take(cap * 4);          // finisher route: what waits for the rounds (the passes' leftovers, groups the code rounds gave up on)
  a.off_parkHP = take(cap * 8);
  a.off_hardS = take(lcap * 4);         // finisher route: the groups too large for a window, until they join the waiting list;
  a.off_hardHP = take(lcap * 8);        //   then, with the arrays below, the two homes of the local list and its rank notes (k_finish<RANK>)
  a.off_hardC = take(lcap * 2 + 64);
  a.off_LP0 = take(lcap * 4);
  a.off_LH0 = take(lcap * 4);
  a.off_LH1 = take(lcap * 4);
  a.off_LC1 = take(lcap * 2 + 64);
  a.off_US = take(lcap * 4);
  a.off_UR = take(lcap * 4);
  a.off_pairs = take((u64)kPairReplicas * 65536 * 4);
  a.off_codes = take((u64)kCodeRows * 256 * 4);
  a.total = o;
  return a;
}

// Contexts for tiny blocks still get room for the entropy front-ends' per-section tables
// (256 sections x 4096 run lengths).
static u64 arena_items(u32 max_block) { return std::max<u64>((u64)max_block + 1, 1u << 20); }

u64 BwtEngine::workspace_bytes(u32 max_block) { return plan_arena(arena_items(max_block)).total; }

int BwtEngine::init(int dev, u32 max_block_size) {
  device = dev;
  max_block = max_block_size;
  cap = arena_items(max_block_size);
  std::memset(&stats, 0, sizeof stats);
  BWTC_HIP_TRY(hipSetDevice(dev));
  BWTC_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  const ArenaPlan a = plan_arena(cap);
  arena_bytes = a.total;
  BWTC_HIP_TRY(hipMalloc(&arena, arena_bytes));
  u8* base = static_cast<u8*>(arena);
  d_T = base + a.off_T;
  d_out = base + a.off_out;
  d_in = base + a.off_in;
  d_SA = reinterpret_cast<u32*>(base + a.off_SA);
  d_rank = reinterpret_cast<u32*>(base + a.off_rank);
  d_R1 = base + a.off_R1;
  d_R2 = base + a.off_R2;
  d_V0 = reinterpret_cast<u32*>(base + a.off_V0);
  d_V1 = reinterpret_cast<u32*>(base + a.off_V1);
  d_G0 = reinterpret_cast<u32*>(base + a.off_G0);
  d_G1 = reinterpret_cast<u32*>(base + a.off_G1);
  d_GRP = reinterpret_cast<u32*>(base + a.off_GRP);
  d_C0 = base + a.off_C0;
  d_C1 = base + a.off_C1;
  d_P0 = base + a.off_P0;
  d_P1 = base + a.off_P1;
  d_W0 = reinterpret_cast<u32*>(base + a.off_W0);
  d_W1 = reinterpret_cast<u32*>(base + a.off_W1);
  d_table = reinterpret_cast<u32*>(base + a.off_table);
  d_partial = reinterpret_cast<u32*>(base + a.off_partial);
  d_aggA = reinterpret_cast<u32*>(base + a.off_aggA);
  d_aggB = reinterpret_cast<u32*>(base + a.off_aggB);
  d_aggC = reinterpret_cast<u32*>(base + a.off_aggC);
  d_agg_part = reinterpret_cast<u32*>(base + a.off_agg_part);
  d_small = reinterpret_cast<u32*>(base + a.off_small);
  d_ent = base + a.off_ent;
  d_comp = base + a.off_comp;
  d_sweep = reinterpret_cast<u32*>(base + a.off_sweep);
  d_parkS = reinterpret_cast<u32*>(base + a.off_parkS);
  d_parkHP = reinterpret_cast<u64*>(base + a.off_parkHP);
  d_hardS = reinterpret_cast<u32*>(base + a.off_hardS);
  d_hardHP = reinterpret_cast<u64*>(base + a.off_hardHP);
  d_hardC = reinterpret_cast<unsigned short*>(base + a.off_hardC);
  d_LP0 = reinterpret_cast<u32*>(base + a.off_LP0);
  d_LH0 = reinterpret_cast<u32*>(base + a.off_LH0);
  d_LH1 = reinterpret_cast<u32*>(base + a.off_LH1);
  d_LC1 = reinterpret_cast<unsigned short*>(base + a.off_LC1);
  d_US = reinterpret_cast<u32*>(base + a.off_US);
  d_UR = reinterpret_cast<u32*>(base + a.off_UR);
  d_pairs = reinterpret_cast<u32*>(base + a.off_pairs);
  d_codes = reinterpret_cast<u32*>(base + a.off_codes);
  {
    const char* e = std::getenv("BWTC_HIP_SORT");
    use_sweep = e && std::strcmp(e, "sweep") == 0;
    const char* w = std::getenv("BWTC_HIP_WAVELET");
    wavelet_on_host = w && std::strcmp(w, "host") == 0;
    const char* gmv = std::getenv("BWTC_HIP_MODELS");
    device_models = !(gmv && std::strcmp(gmv, "host") == 0);
    models_side_stream = !(std::getenv("BWTC_HIP_MODELS_STREAM") && std::getenv("BWTC_HIP_MODELS_STREAM")[0] == '0');
    const char* d = std::getenv("BWTC_HIP_WAVELET_DEPTH");
    if (d && std::atoi(d) > 0) max_inflight = (unsigned)std::atoi(d);
    dense_route = !(std::getenv("BWTC_HIP_DENSE") && std::getenv("BWTC_HIP_DENSE")[0] == '0');
    no_emit = std::getenv("BWTC_HIP_NO_EMIT") != nullptr;
    if (std::getenv("BWTC_HIP_WINDOW_BITS") && std::atoi(std::getenv("BWTC_HIP_WINDOW_BITS")) > 0)
      window_bits = std::min(24, std::atoi(std::getenv("BWTC_HIP_WINDOW_BITS")));
    digit_planes = !(std::getenv("BWTC_HIP_PLANES") && std::getenv("BWTC_HIP_PLANES")[0] == '0');
    split_index = !(std::getenv("BWTC_HIP_SPLIT_INDEX") && std::getenv("BWTC_HIP_SPLIT_INDEX")[0] == '0');
    if (const char* gr = std::getenv("BWTC_HIP_GRAMS")) {       // 0: base-sigma keys only; N > 0: N grams per key
      gram_keys = gr[0] != '0';
      gram_count_override = std::atoi(gr);
    }
    long_keys = !(std::getenv("BWTC_HIP_LONG") && std::getenv("BWTC_HIP_LONG")[0] == '0');
    if (std::getenv("BWTC_HIP_LONG_G2")) long_grams_override = std::atoi(std::getenv("BWTC_HIP_LONG_G2"));
    if (std::getenv("BWTC_HIP_FIN_GROUP")) fin_max_group = std::atoi(std::getenv("BWTC_HIP_FIN_GROUP"));
    if (std::getenv("BWTC_HIP_FIN_WINDOW")) fin_window = std::atoi(std::getenv("BWTC_HIP_FIN_WINDOW"));
    fin_shape_fixed = std::getenv("BWTC_HIP_FIN_GROUP") || std::getenv("BWTC_HIP_FIN_WINDOW");
    fin_wide_short = !(std::getenv("BWTC_HIP_FIN_WIDE") && std::getenv("BWTC_HIP_FIN_WIDE")[0] == '0');
    fin_window = fin_window >= 2048 ? 2048 : 1024;
    fin_max_group = fin_max_group >= 1024 ? 1024 : fin_max_group >= 512 ? 512 : 256;
    if (fin_max_group >= fin_window) fin_max_group = fin_window / 2;
    code_keys = !(std::getenv("BWTC_HIP_KEYS") && std::strcmp(std::getenv("BWTC_HIP_KEYS"), "grams") == 0);
    if (std::getenv("BWTC_HIP_CODE_BITS")) code_bits = std::min(72, std::max(40, std::atoi(std::getenv("BWTC_HIP_CODE_BITS"))));
    finisher = !(std::getenv("BWTC_HIP_FINISHER") && std::getenv("BWTC_HIP_FINISHER")[0] == '0');
    gm_partition_lines = std::getenv("BWTC_HIP_GM_PARTITION") && std::strcmp(std::getenv("BWTC_HIP_GM_PARTITION"), "lines") == 0;
    wt_segmented = !(std::getenv("BWTC_HIP_SEG_STEPS") && std::getenv("BWTC_HIP_SEG_STEPS")[0] == '0');
    gram_min_n = kGramMinN;
    if (std::getenv("BWTC_HIP_GRAM_MIN_N")) gram_min_n = (u32)std::max(64, std::atoi(std::getenv("BWTC_HIP_GRAM_MIN_N")));   // tests: small blocks through the gram / long-key routes
    if (std::getenv("BWTC_HIP_LONG_E")) long_items_per_thread = std::atoi(std::getenv("BWTC_HIP_LONG_E")) == 6 ? 6 : 8;
    if (std::getenv("BWTC_HIP_FIN_PASSES")) fin_max_passes = std::max(0, std::atoi(std::getenv("BWTC_HIP_FIN_PASSES")));
    if (std::getenv("BWTC_HIP_FIN_WORDS")) fin_words = std::min(4, std::max(2, std::atoi(std::getenv("BWTC_HIP_FIN_WORDS"))));
    if (std::getenv("BWTC_HIP_LONG_DIRECT")) long_direct = std::getenv("BWTC_HIP_LONG_DIRECT")[0] != '0';
    if (std::getenv("BWTC_HIP_FIN_ROUNDS")) fin_rounds = std::min(4, std::max(1, std::atoi(std::getenv("BWTC_HIP_FIN_ROUNDS"))));
    local_rounds = !(std::getenv("BWTC_HIP_LOCAL_ROUNDS") && std::getenv("BWTC_HIP_LOCAL_ROUNDS")[0] == '0');
    if (std::getenv("BWTC_HIP_FIN_FLOOR")) fin_floor = (u32)std::max(0, std::atoi(std::getenv("BWTC_HIP_FIN_FLOOR")));
    if (std::getenv("BWTC_HIP_TEXT_ROUNDS")) { text_rounds = std::max(0, std::atoi(std::getenv("BWTC_HIP_TEXT_ROUNDS"))); text_rounds_fixed = true; }
    const char* hg = std::getenv("BWTC_HIP_HUGE_MI");
    if (hg && std::atoi(hg) > 0) huge_group_elements = (u64)std::atoi(hg) << 20;
  }
  // Opt-in (BWTC_HIP_SCAN=chained): measured on MI355X the single-launch scan is SLOWER for the
  // sorter's tables (63 us against 25 + 11 + 6 us for 8.4 M words: 2048 tiles are too few to hide
  // the look-back hops), so the three-launch form stays the default.
  if (std::getenv("BWTC_HIP_SCAN") && std::strcmp(std::getenv("BWTC_HIP_SCAN"), "chained") == 0) {
    scan_chain.cap_tiles = (u32)(cap / kScanTile + 4096);
    BWTC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&scan_chain.status), ((u64)scan_chain.cap_tiles + 1) * 8));
    BWTC_HIP_TRY(hipMemset(scan_chain.status, 0, ((u64)scan_chain.cap_tiles + 1) * 8));
    scan_chain.ticket = scan_chain.status + scan_chain.cap_tiles;     // the last word
    scan_chain.err = d_small + 522;                                  // kSmallError: read back with every ranking step's counts
    scan_chain.issued = 0;
    scan_chain.epoch = 0;
  }
  BWTC_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_small), 1024 * 4, hipHostMallocDefault));
  BWTC_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_stage), cap + 32, hipHostMallocDefault));
  BWTC_HIP_TRY(hipEventCreateWithFlags(&ev_wait, hipEventBlockingSync | hipEventDisableTiming));
  {
    const char* m = std::getenv("BWTC_HIP_SYNC");
    wait_mode = m && std::strcmp(m, "spin") == 0 ? 1 : m && std::strcmp(m, "block") == 0 ? 2 : 0;
  }
  BWTC_HIP_TRY(hipEventCreate(&ev_begin));
  BWTC_HIP_TRY(hipEventCreate(&ev_end));
  for (int i = 0; i < kMaxSortEvents; ++i) BWTC_HIP_TRY(hipEventCreate(&ev_sort[i]));
  return 0;
}

hipError_t BwtEngine::ensure_d2h_stream() {
  if (d2h_stream) return hipSuccess;
  hipError_t rc = hipStreamCreateWithFlags(&d2h_stream, hipStreamNonBlocking);
  if (rc != hipSuccess) return rc;
  rc = hipEventCreateWithFlags(&ev_packed, hipEventDisableTiming);
  if (rc != hipSuccess) return rc;
  rc = hipEventCreateWithFlags(&ev_models, hipEventDisableTiming);
  if (rc != hipSuccess) return rc;
  return hipEventCreateWithFlags(&ev_codes, hipEventBlockingSync | hipEventDisableTiming);
}

hipError_t BwtEngine::codes_wait() {
  if (!codes_in_flight) return hipSuccess;
  codes_in_flight = false;
  return sleepy_wait(ev_codes, kWaitSlots - 1);
}

// A wait that gives the CPU away.  On this runtime hipEventSynchronize spins even on an event made with
// hipEventBlockingSync (scripts/r4/wait_probe.py: the waiting thread's CPU time equals the wall time), and
// with the 'B' coder the thread that feeds the GPU then takes a whole CPU of a quota that the range coders
// need (measured: 0.04 CPU-seconds per 256 MiB block, a twentieth of sixteen CPUs).  So: poll the event
// and sleep in between, in slices that are a quarter of what the wait is still expected to take (every
// wait of a block has a slot, and remembers how long it took for the previous blocks), at most 200 us and
// at least 20 us -- a wake-up comes at most a slice late.
hipError_t BwtEngine::sleepy_wait(hipEvent_t ev, int slot) {
  using clk = std::chrono::steady_clock;
  static thread_local bool slack_set = false;
  if (!slack_set) { (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL); slack_set = true; }   // default: 50 us of slack on every sleep
  const auto t0 = clk::now();
  float& ema = wait_us[slot < 0 ? 0 : slot >= kWaitSlots ? kWaitSlots - 1 : slot];
  double slice = 20.0;
  for (;;) {
    const hipError_t q = hipEventQuery(ev);
    const double el = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
    if (q != hipErrorNotReady) {
      ema = ema > 0.f ? 0.5f * ema + 0.5f * (float)el : (float)el;
      return q;
    }
    const double left = (double)ema - el;
    if (left > 80.0) slice = std::min(200.0, left * 0.25);
    else slice = std::min(200.0, std::max(20.0, slice * 1.25));
    timespec ts;
    ts.tv_sec = 0;
    ts.tv_nsec = (long)(slice * 1e3);
    nanosleep(&ts, nullptr);
  }
}

int BwtEngine::reserve_run_arrays() {
  if (d_run_start[0] && d_run_start[1] && d_run_sym[0] && d_run_sym[1]) return 0;
  // all four or none: a half-allocated set left behind by a failed call would let the next
  // block's scanner write through a null pointer
  bool ok = true;
  for (int b = 0; b < 2; ++b) {
    if (!d_run_start[b]) ok = ok && hipMalloc(reinterpret_cast<void**>(&d_run_start[b]), (cap + 2) * 4) == hipSuccess;
    if (!d_run_sym[b]) ok = ok && hipMalloc(reinterpret_cast<void**>(&d_run_sym[b]), cap + 64) == hipSuccess;
  }
  if (ok) return 0;
  (void)hipGetLastError();
  for (int b = 0; b < 2; ++b) {
    if (d_run_start[b]) (void)hipFree(d_run_start[b]);
    if (d_run_sym[b]) (void)hipFree(d_run_sym[b]);
    d_run_start[b] = nullptr; d_run_sym[b] = nullptr;
  }
  return -2;
}

hipError_t BwtEngine::wait() {
  const bool block = wait_mode == 2 || (wait_mode == 0 && pipeline != nullptr);
  if (!block || !ev_wait) return hipStreamSynchronize(stream);
  hipError_t rc = hipEventRecord(ev_wait, stream);
  if (rc != hipSuccess) return rc;
  if (wait_seq < kWaitSlots - 2) ++wait_seq;
  return sleepy_wait(ev_wait, wait_seq);
}

void BwtEngine::release() {
  wavelet_pipeline_release(*this);
  if (stream) (void)hipStreamSynchronize(stream);
  if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); copy_stream = nullptr; }
  if (ev_copy) { (void)hipEventDestroy(ev_copy); ev_copy = nullptr; }
  if (d2h_stream) { (void)hipStreamSynchronize(d2h_stream); (void)hipStreamDestroy(d2h_stream); d2h_stream = nullptr; }
  if (ev_packed) { (void)hipEventDestroy(ev_packed); ev_packed = nullptr; }
  if (ev_models) { (void)hipEventDestroy(ev_models); ev_models = nullptr; }
  if (scan_chain.status) { (void)hipFree(scan_chain.status); scan_chain = ScanChain(); }
  if (d_gm) { (void)hipFree(d_gm); d_gm = nullptr; gm_bytes = 0; }
  if (d_gm_w) { (void)hipFree(d_gm_w); d_gm_w = nullptr; gm_w_bytes = 0; }
  if (h_gm) { (void)hipHostFree(h_gm); h_gm = nullptr; h_gm_bytes = 0; }
  if (ev_gm_upload) { (void)hipEventDestroy(ev_gm_upload); ev_gm_upload = nullptr; gm_upload_pending = false; }
  if (gm_stream) { (void)hipStreamSynchronize(gm_stream); (void)hipStreamDestroy(gm_stream); gm_stream = nullptr; }
  if (ev_packed_ready) { (void)hipEventDestroy(ev_packed_ready); ev_packed_ready = nullptr; }
  if (ev_gm_done) { (void)hipEventDestroy(ev_gm_done); ev_gm_done = nullptr; gm_done_pending = false; }
  if (ev_codes) { (void)hipEventDestroy(ev_codes); ev_codes = nullptr; }
  if (arena) (void)hipFree(arena);
  if (h_small) (void)hipHostFree(h_small);
  if (h_stage) (void)hipHostFree(h_stage);
  for (int b = 0; b < 2; ++b) {
    if (d_run_start[b]) (void)hipFree(d_run_start[b]);
    if (d_run_sym[b]) (void)hipFree(d_run_sym[b]);
    d_run_start[b] = nullptr; d_run_sym[b] = nullptr;
  }
  if (d_wt) (void)hipFree(d_wt);
  if (h_wt) (void)hipHostFree(h_wt);
  if (h_stats) (void)hipHostFree(h_stats);
  h_stats = nullptr; h_stats_bytes = 0;
  d_wt = nullptr; h_wt = nullptr; wt_bytes = 0; h_wt_bytes = 0;
  if (ev_wait) (void)hipEventDestroy(ev_wait);
  ev_wait = nullptr;
  if (ev_begin) (void)hipEventDestroy(ev_begin);
  if (ev_end) (void)hipEventDestroy(ev_end);
  for (int i = 0; i < kMaxSortEvents; ++i) if (ev_sort[i]) (void)hipEventDestroy(ev_sort[i]);
  probe.destroy();
  if (stream) (void)hipStreamDestroy(stream);
  arena = nullptr; h_small = nullptr; h_stage = nullptr; stream = nullptr;
}

static constexpr int kSmallFreqs = 0, kSmallLf = 256, kSmallPidx = 512, kSmallCounts = 520,
                     kSmallLut = 600,   // 64 words = 256-byte alphabet remap table
                     kSmallInv = 664;   // 64 words: dense code -> byte

// Width of the initial sort key.  The alphabet of T is remapped to dense w-bit codes and as
// many characters as fit are packed into a 32- or 64-bit key: 16 characters of DNA or 4 raw
// bytes in 32 bits, 9 characters of text in 64 bits.  32 bits are used when the order-0
// entropy says the 32-bit key already separates almost all suffixes; the choice only
// changes speed, never the result.
struct KeyPlan { int k, bits; u32 sigma; bool wide; u8 lut[256]; };

// most characters per 8-bit radix pass for keys of at most `maxbits` bits; ties -> longer key
static void best_packing(u32 sigma, int maxbits, int* k_out, int* bits_out) {
  double best = -1.0;
  unsigned __int128 pow = 1;
  *k_out = 1; *bits_out = 8;
  for (int k = 1; k <= 32; ++k) {
    pow *= sigma;
    if (pow > ((unsigned __int128)1 << maxbits)) break;
    int bits = 0;
    for (unsigned __int128 v = pow - 1; v; v >>= 1) ++bits;
    if (bits == 0) bits = 1;
    const int passes = (bits + kRadixBits - 1) / kRadixBits;
    const double ratio = (double)k / passes;
    if (ratio >= best) { best = ratio; *k_out = k; *bits_out = bits; }
  }
}

// lone_sentinel: the only zero byte of T is the terminator planted at T[n-1].  It then
// shares code 0 with the smallest real symbol instead of costing a code of its own (DNA
// stays at 4 symbols): a suffix whose key window reaches the terminator has length <= k,
// and all of those are split off as singletons in true order by the "shorter first" rule
// of the initial ranking, so the merged code is never compared.
static KeyPlan plan_keys(const u32 hist[256], u32 n, bool lone_sentinel, bool force_wide = false) {
  KeyPlan p;
  int sigma = 0;
  double total = 0.0, h0 = 0.0;
  for (int c = 0; c < 256; ++c) {
    p.lut[c] = (u8)(sigma > 255 ? 255 : sigma);
    const bool present = hist[c] != 0 && !(c == 0 && lone_sentinel);
    if (present) { ++sigma; total += hist[c]; }
  }
  for (int c = lone_sentinel ? 1 : 0; c < 256; ++c)
    if (hist[c]) { const double q = hist[c] / total; h0 -= q * std::log2(q); }
  p.sigma = (u32)(sigma < 2 ? 2 : sigma);
  int k32, b32, k64, b64;
  best_packing(p.sigma, 32, &k32, &b32);
  best_packing(p.sigma, 64, &k64, &b64);
  const double need = std::log2((double)n + 1.0) + 2.0;
  p.wide = (k64 > k32) && (force_wide || k32 * h0 < need);
  p.k = p.wide ? k64 : k32;
  p.bits = p.wide ? b64 : b32;
  return p;
}

static constexpr int kSmallError = 522;   // sticky error word of the chained sort
static constexpr int kSmallLastChar = 524; // character of row n-1 (emitting rankers, block mode)

// Blocks for which a round's sort key (group bits + rank bits) leaves bits 56..63 free for the
// carried character in every round: groups <= n/2, ranks <= n.
static int rank_bits(u32 n) { return bit_width_u64((u64)n + 1); }   // rank + 1 <= n, and never all ones
static bool can_carry(u32 n) { return rank_bits(n) + bit_width_u64(n / 2 ? n / 2 - 1 : 0) <= 56; }

// Algorithmic bytes of P radix passes over `items` items of `item` bytes (key `key` bytes): each pass reads a
// digit plane (or, without planes, the keys) for its histograms, reads and writes the items, and all but the
// last leave the next pass's plane.
static u64 sort_bytes(u64 items, int passes, int item, int key, bool planes) {
  if (passes <= 0) return 0;
  return items * (u64)passes * (u64)(2 * item + (planes ? 1 : key)) + (planes ? items * (u64)(passes - 1) : 0);
}

template <typename K>
void BwtEngine::sort_pairs(K* k0, K* k1, u32* v0, u32* v1, u64 n, int nbits, K** ks, u32** vs,
                           bool probe_it, int bit_lo, u64 n_holes, bool plane_ready, bool values_descend) {
  if (use_sweep && n_holes == 0) {
    SweepWs ws;
    ws.hist_all = d_sweep;
    ws.bases = d_sweep + kSweepMaxPasses * kChains * kRadixBins;
    ws.tickets = d_sweep + 2 * kSweepMaxPasses * kChains * kRadixBins;
    ws.status = d_table;
    ws.error = d_small + kSmallError;
    radix_sort_pairs_sweep<K>(k0, k1, v0, v1, n, nbits, ws, stream, ks, vs, probe_it ? &probe : nullptr,
                              bit_lo);
  } else {
    radix_sort_pairs<K>(k0, k1, v0, v1, n, nbits, d_table, d_partial, stream, ks, vs,
                        probe_it ? &probe : nullptr, bit_lo, false, false, n_holes,
                        digit_planes ? d_P0 : nullptr, digit_planes ? d_P1 : nullptr, plane_ready && digit_planes,
                        values_descend);
  }
}

// rank[pairs_s[i]] = pairs_r[i] for i < m, made cache-friendly: the pairs (two arrays of
// `cap` words inside one 8*cap-byte region) are first partitioned by the top 16 bits of the
// destination with two stable radix passes (tmp = another 8*cap-byte region), after which
// consecutive pairs write into the same 4096-word window of rank[].
constexpr u32 kPairsMin = 1u << 21;

void BwtEngine::scatter_rank_pairs(u32* pairs, u32* tmp, u32 m, u32 n) {
  hipStream_t st = stream;
  const int bits = bit_width_u64(n ? n - 1 : 0);
  const int lo = bits > window_bits ? bits - window_bits : 0;
  u32 *ws = pairs, *wr = pairs + cap;
  if (m >= kPairsMin && bits > 12) {
    sort_pairs<u32>(pairs, tmp, pairs + cap, tmp + cap, m, bits, &ws, &wr, false, lo);
    stats.alg_bytes += sort_bytes(m, (bits - lo + kRadixBits - 1) / kRadixBits, 8, 4, digit_planes);
  }
  stats.alg_bytes += (u64)m * (8 + 4);                   // k_scatter_pairs
  hipLaunchKernelGGL(k_scatter_pairs, dim3((ceil_div(m, kWinTile) + 7u) / 8u * 8u), dim3(kWinTPB), 0, st,
                     d_rank, ws, wr, m, lo > 8 ? lo - 8 : 0);
}

// One ranking step over a sorted list (INIT: all suffixes by their initial key; rounds: the
// active list by (group, rank[s+h])): new ranks into rank[], finished suffixes out, and -- when
// anything stays active -- the next round's list, sorted.  Buffers: `rec_free` / `v_free` are
// an 8*cap-byte region and a 4*cap-byte array nobody uses; `rec_keys` / `v_keys` are the ones
// that hold ks / vs (free once the apply kernel has run).
template <typename K, bool INIT>
int BwtEngine::rank_step(const K* ks, const u32* vs, u32 m, u32 n, u32 short_len, K kmask,
                         RankBuffers& rb, RrEmit re, bool emit, u64 h_next, RankResult* res, u32 split,
                         const RrLong* lg, bool text, bool carry_in, bool raw_out) {
  hipStream_t st = stream;
  u32* counts = d_small + kSmallCounts;
  const u32 tiles = ceil_div(m, kRrTile);
  constexpr bool kCanSplit = INIT && sizeof(K) == 8;
  const bool lng = kCanSplit && lg && lg->w;              // items of the long-key sort (split: 16-bit values + upper bits in the key)
  const RrLong lgv = lng ? *lg : RrLong();
  if (lng && split) hipLaunchKernelGGL((k_rerank_reduce<K, INIT, kCanSplit, kCanSplit>), dim3(tiles), dim3(kRrTPB), 0, st, ks, vs, m, n,
                              short_len, kmask, d_aggA, d_aggB, d_aggC, lgv);
  else if (lng) hipLaunchKernelGGL((k_rerank_reduce<K, INIT, false, kCanSplit>), dim3(tiles), dim3(kRrTPB), 0, st, ks, vs, m, n,
                              short_len, kmask, d_aggA, d_aggB, d_aggC, lgv);
  else if (split) hipLaunchKernelGGL((k_rerank_reduce<K, INIT, kCanSplit>), dim3(tiles), dim3(kRrTPB), 0, st, ks, vs, m, n,
                                short_len, kmask, d_aggA, d_aggB, d_aggC, lgv);
  else hipLaunchKernelGGL((k_rerank_reduce<K, INIT, false>), dim3(tiles), dim3(kRrTPB), 0, st, ks, vs, m, n,
                          short_len, kmask, d_aggA, d_aggB, d_aggC, lgv);
  {
    const u32 parts = ceil_div(tiles, kRrScanChunk);
    hipLaunchKernelGGL(k_rerank_scan_tiles<0>, dim3(parts), dim3(1024), 0, st, d_aggA, d_aggB, d_aggC, tiles, d_agg_part, counts);
    hipLaunchKernelGGL(k_rerank_scan_tiles<1>, dim3(parts), dim3(1024), 0, st, d_aggA, d_aggB, d_aggC, tiles, d_agg_part, counts);
  }
  // how much stays active decides the route, so the counts are read before the apply kernel
  BWTC_HIP_TRY(hipMemcpyAsync(h_small + kSmallCounts, counts, 12, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(wait());
  if (h_small[kSmallError] & 1u) { std::fprintf(stderr, "bwtc_hip: chained radix sort timed out\n"); return -3; }
  if (h_small[kSmallError]) return -3;                  // (2: k_code_build -- suffix_sort retries the block with gram keys)
  const u32 m_next = h_small[kSmallCounts], groups = h_small[kSmallCounts + 1];
  {
    // reduce and apply read the list (key, suffix, slot; the second key word of long items); apply writes a
    // byte (and in the finisher route the suffix) per finished suffix and a list entry / record per active one
    const u64 in_bytes = sizeof(K) + (lng ? 2 + 4 : split ? 2 : 4) + (INIT ? 0 : 4);
    stats.alg_bytes += (u64)m * in_bytes * 2 + (u64)(m - m_next) * 5 + (u64)m_next * 13;
  }
  res->m = m_next;
  res->groups = groups;
  const int b2 = rank_bits(n);
  const int b1 = bit_width_u64(groups ? groups - 1 : 0);
  const int nbits = b1 + b2;
  // a key of more than 56 bits has no room for the carried character (blocks above 256 MiB with many groups): the
  // next list then goes without, and its ranking reads T[s-1] for what it finishes
  const bool carry_next = emit && nbits <= 56;
  res->carry = carry_next;
  const bool dense = !text && dense_route && m_next > 0 && (u64)m_next * 2 >= m && m >= kPairsMin && !use_sweep;
  u32* sa_out = (emit && !fin_active) ? nullptr : d_SA;   // finisher route: the bridge will want the finished suffixes' slots
  u32* rank_arg = (text && !ranks_live) ? nullptr : d_rank;   // text rounds before rank[] is completed: neither complete nor needed; after: kept exact
  // Long items of which at most half is still tied (a 256 MiB text block: 21 %, a 1 GiB one: 37 %): the finisher
  // settles the rest by direct comparison (suffix_sort); rank[] is not written at all.  Above that the list is
  // dominated by long repeats and the doubling rounds are the better tool.
  // (with the last gram's low bits left out of the sort -- long_drop -- the list also holds what those bits would have
  // told apart: the bound is three fifths then)
  res->finish = lng && emit && finisher;
  if (res->finish) {
    sa_out = d_SA;
    fin_active = true;
    if constexpr (kCanSplit) {
      if (split)
        hipLaunchKernelGGL((k_rerank_apply<K, INIT, 3, 3, true, true>), dim3(tiles), dim3(kRrTPB), 0, st, ks, vs,
                           (const u32*)rb.aglob, m, n, short_len, kmask, d_aggA, d_aggB, d_aggC, d_rank, sa_out,
                           rb.v_free, rb.aglob_next, d_GRP, (u32*)nullptr, (u32*)nullptr, re, lgv);
      else
        hipLaunchKernelGGL((k_rerank_apply<K, INIT, 3, 3, false, true>), dim3(tiles), dim3(kRrTPB), 0, st, ks, vs,
                           (const u32*)rb.aglob, m, n, short_len, kmask, d_aggA, d_aggB, d_aggC, d_rank, sa_out,
                           rb.v_free, rb.aglob_next, d_GRP, (u32*)nullptr, (u32*)nullptr, re, lgv);
    }
    res->carry = true;
    return 0;
  }
  u64* recA = static_cast<u64*>(rb.rec_free);
  u64* recB = static_cast<u64*>(rb.rec_keys);

#define BWTC_APPLY_S(MODE, EMIT, SP, LG, PS, PR, AIDX)                                                      \
  hipLaunchKernelGGL((k_rerank_apply<K, INIT, MODE, EMIT, SP, LG>), dim3(tiles), dim3(kRrTPB), 0, st, ks, vs, \
                     (const u32*)rb.aglob, m, n, short_len, kmask, d_aggA, d_aggB, d_aggC, rank_arg, sa_out, \
                     AIDX, rb.aglob_next, d_GRP, PS, PR, re, lgv)
  // Where a finished suffix's character comes from (see RrEmit): long items carry its code in the key (3); the
  // initial ranking of other items, and rounds whose keys had no room for it, read T[s-1] (2); rounds find it in
  // bits 56..63 of their key (1); 0: nothing is emitted.
#define BWTC_APPLY(MODE, PS, PR, AIDX)                                                                    \
  do { if (lng && split) { if (emit) BWTC_APPLY_S(MODE, 3, kCanSplit, kCanSplit, PS, PR, AIDX); else BWTC_APPLY_S(MODE, 0, kCanSplit, kCanSplit, PS, PR, AIDX); } \
       else if (lng) { if (emit) BWTC_APPLY_S(MODE, 3, false, kCanSplit, PS, PR, AIDX); else BWTC_APPLY_S(MODE, 0, false, kCanSplit, PS, PR, AIDX); } \
       else if (split) { if (emit) BWTC_APPLY_S(MODE, 2, kCanSplit, false, PS, PR, AIDX); else BWTC_APPLY_S(MODE, 0, kCanSplit, false, PS, PR, AIDX); } \
       else if (!emit) BWTC_APPLY_S(MODE, 0, false, false, PS, PR, AIDX);                                 \
       else if (INIT || !carry_in) BWTC_APPLY_S(MODE, 2, false, false, PS, PR, AIDX);                     \
       else BWTC_APPLY_S(MODE, 1, false, false, PS, PR, AIDX); } while (0)
  if (dense) {
    u32* tri_key = reinterpret_cast<u32*>(recA);
    // records partitioned by the top 16 bits of s (bits 32.. of the record), two stable passes
    const int bits = bit_width_u64(n ? n - 1 : 0);
    const int lo = bits > window_bits ? bits - window_bits : 0;
    re.rec_plane = digit_planes ? d_P0 : nullptr;
    re.rec_shift = lo;
    BWTC_APPLY(2, tri_key, rb.v_free, (u32*)nullptr);
    u64* ws = nullptr; u32* wv = nullptr;
    sort_pairs<u64>(recA, recB, rb.v_free, rb.v_keys, m, 32 + bits, &ws, &wv, false, 32 + lo, 0, true);
    u64* ws_other = ws == recA ? recB : recA;
    u32* wv_other = wv == rb.v_free ? rb.v_keys : rb.v_free;
    const u32 grid = (ceil_div(m, kWinTile) + 7u) / 8u * 8u;
    const int bin_shift = lo > 8 ? lo - 8 : 0;           // the eight bits of s below the window bits
    hipLaunchKernelGGL(k_scatter_dense, dim3(grid), dim3(kWinTPB), 0, st, d_rank, (const u64*)ws, m, bin_shift);
    hipLaunchKernelGGL(k_gather_dense, dim3(grid), dim3(kWinTPB), 0, st, ws, wv, (const u32*)d_rank,
                       (const u8*)d_T, m, n, (u32)(h_next > 0xFFFFFFFFull ? 0xFFFFFFFFu : h_next), b2,
                       carry_next ? 1 : 0, bin_shift);
    const bool timed = n_sort_events + 2 <= kMaxSortEvents;
    if (timed) BWTC_HIP_TRY(hipEventRecord(ev_sort[n_sort_events++], st));
    sort_pairs<u64>(ws, ws_other, wv, wv_other, m_next, nbits, &res->ks, &res->vs, true, 0, (u64)(m - m_next));
    if (timed) BWTC_HIP_TRY(hipEventRecord(ev_sort[n_sort_events++], st));
    stats.sort_pass_items += (u64)m + (u64)m_next * (u64)std::max(0, (nbits + kRadixBits - 1) / kRadixBits - 1);
    {
      const int P = (nbits + kRadixBits - 1) / kRadixBits;
      stats.alg_bytes += sort_bytes(m, 2, 12, 8, digit_planes)                         // window partition of the records
                         + (u64)m * (12 + 4) + (u64)m * (12 + 4 + 1 + 12)             // k_scatter_dense, k_gather_dense
                         + (u64)m * (8 + 12) + (u64)m_next * 12                        // hole-dropping first pass
                         + sort_bytes(m_next, std::max(0, P - 1), 12, 8, digit_planes);
    }
    res->rec_other = res->ks == recA ? recB : recA;
    res->v_other = res->vs == rb.v_free ? rb.v_keys : rb.v_free;
    return 0;
  }
  u32* pairs = reinterpret_cast<u32*>(recA);
  if (raw_out) {
    // the list goes back to its owner as it is now grouped: suffix, slot, head slot (rb.v_free, rb.aglob_next, d_GRP)
    BWTC_APPLY(3, (u32*)nullptr, (u32*)nullptr, rb.v_free);
    res->carry = emit;
    return 0;
  }
  if (text) {
    // text round: rank[] is neither complete nor needed
    res->text_chars = (u32)std::min(6, (56 - b1 - 4) / 8);
    BWTC_APPLY(0, (u32*)nullptr, (u32*)nullptr, rb.v_free);
    res->carry = emit;
    if (m_next == 0) return 0;
    const int tbits = b1 + 8 * (int)res->text_chars + 4;
    hipLaunchKernelGGL(k_gather_text, dim3(ceil_div(m_next, 256)), dim3(256), 0, st, (const u32*)rb.v_free, (const u32*)d_GRP,
                       (const u8*)d_T, emit ? (const u8*)re.achr_out : (const u8*)nullptr, recA, m_next, n, (u32)h_next, res->text_chars);
    sort_pairs<u64>(recA, recB, rb.v_free, rb.v_keys, m_next, tbits, &res->ks, &res->vs, false);
    stats.sort_pass_items += (u64)m_next * (u64)((tbits + kRadixBits - 1) / kRadixBits);
    stats.alg_bytes += (u64)m_next * (4 + 4 + 1 + 16 + 8) + sort_bytes(m_next, (tbits + kRadixBits - 1) / kRadixBits, 12, 8, digit_planes);
    stats.route |= 4u;
    res->rec_other = res->ks == recA ? recB : recA;
    res->v_other = res->vs == rb.v_free ? rb.v_keys : rb.v_free;
    return 0;
  }
  if (m >= kPairsMin || INIT) {
    BWTC_APPLY(1, pairs, pairs + cap, rb.v_free);
    scatter_rank_pairs(pairs, reinterpret_cast<u32*>(recB), m, n);
  } else {
    BWTC_APPLY(0, (u32*)nullptr, (u32*)nullptr, rb.v_free);
  }
#undef BWTC_APPLY
#undef BWTC_APPLY_S
  if (m_next == 0) return 0;
  hipLaunchKernelGGL(k_gather_key2, dim3(ceil_div(m_next, 256 * kSimpleE)), dim3(256), 0, st,
                     (const u32*)rb.v_free, (const u32*)d_GRP, (const u32*)d_rank,
                     carry_next ? (const u8*)re.achr_out : (const u8*)nullptr, recA, m_next, n,
                     (u32)(h_next > 0xFFFFFFFFull ? 0xFFFFFFFFu : h_next), b2);
  const bool timed = n_sort_events + 2 <= kMaxSortEvents;
  if (timed) BWTC_HIP_TRY(hipEventRecord(ev_sort[n_sort_events++], st));
  sort_pairs<u64>(recA, recB, rb.v_free, rb.v_keys, m_next, nbits, &res->ks, &res->vs, true);
  if (timed) BWTC_HIP_TRY(hipEventRecord(ev_sort[n_sort_events++], st));
  stats.sort_pass_items += (u64)m_next * (u64)((nbits + kRadixBits - 1) / kRadixBits);
  stats.alg_bytes += (u64)m_next * (4 + 4 + 4 + 1 + 8) + sort_bytes(m_next, (nbits + kRadixBits - 1) / kRadixBits, 12, 8, digit_planes);
  res->rec_other = res->ks == recA ? recB : recA;
  res->v_other = res->vs == rb.v_free ? rb.v_keys : rb.v_free;
  return 0;
}

// Decides whether the initial key is made of dense gram codes (K2g) and, if so, leaves the code
// table in d_rank (free until the first ranking step; the byte map lives in d_SA).  G = 0: keep
// the base-sigma key.  The gram length is the longest whose universe fits kGramMaxU; the key takes
// as few grams as cover the base-sigma key's characters, and is used when that saves a radix pass.
static constexpr int kSmallGram = 526;
static constexpr int kSmallFin = 528;     // finisher: counts and smallest depths of the hard and shallow lists (finish_list)
static constexpr int kSmallFinNext = 540; // finisher: entries of the next list's kFinRegions regions
int BwtEngine::plan_grams(const KeyPlan& plan, u32 n, const u8* d_lut, GramPlan* gp) {
  hipStream_t st = stream;
  gp->G = 0;
  if (!gram_keys || n < gram_min_n) return 0;
  int g = 1;
  u64 U = plan.sigma;
  while (U * plan.sigma <= kGramMaxU && U * plan.sigma <= (u64)cap / 2 && g < 32) { U *= plan.sigma; ++g; }
  if (g < 2) return 0;
  const u32 nwords = ceil_div((u32)U, 64u);                // <= 65536
  u8* present = reinterpret_cast<u8*>(d_SA);               // byte map, then (behind it) the bitmap
  unsigned long long* bits = reinterpret_cast<unsigned long long*>(present + 64ull * nwords);
  uint4* table = reinterpret_cast<uint4*>(d_rank);
  u32 top = 1;
  for (int t = 1; t < g; ++t) top *= plan.sigma;
  stats.alg_bytes += (u64)n + 64ull * nwords * 2;        // k_gram_mark reads T; the byte map is cleared and packed
  BWTC_HIP_TRY(hipMemsetAsync(present, 0, 64ull * nwords, st));
  hipLaunchKernelGGL(k_gram_mark, dim3(std::min<u32>(ceil_div(n, kGramTile), 4096u)), dim3(256), 0, st,
                     (const u8*)d_T, d_lut, n, g, plan.sigma, top, present);
  hipLaunchKernelGGL(k_gram_pack, dim3(ceil_div(nwords, 256u)), dim3(256), 0, st, (const u8*)present, nwords, bits);
  hipLaunchKernelGGL(k_gram_table, dim3(1), dim3(1024), 0, st, (const unsigned long long*)bits, nwords, table,
                     d_small + kSmallGram);
  BWTC_HIP_TRY(hipMemcpyAsync(h_small + kSmallGram, d_small + kSmallGram, 4, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(wait());
  const u32 distinct = h_small[kSmallGram];
  if (distinct == 0) return -3;
  const int b = std::max(1, bit_width_u64(distinct - 1));
  const int g_max = std::min(64 / b, 64 / g);                 // key bits, and the text's zero padding
  if (g_max < 1) return 0;
  int G = std::min(g_max, (plan.k + g - 1) / g);
  if (gram_count_override > 0) G = std::min(g_max, gram_count_override);
  const int passes = (b * G + kRadixBits - 1) / kRadixBits;
  const int base_passes = (plan.bits + kRadixBits - 1) / kRadixBits;
  // fewer passes for at least as many characters, or the same passes for more
  const bool better = (g * G >= plan.k && passes < base_passes) || (g * G > plan.k && passes <= base_passes);
  if (std::getenv("BWTC_HIP_DEBUG"))
    std::fprintf(stderr, "grams: sigma=%u g=%d universe=%llu present=%u bits=%d G=%d (%d chars, %d passes) vs base %d chars, %d passes -> %s\n",
                 plan.sigma, g, (unsigned long long)U, distinct, b, G, g * G, passes, plan.k, base_passes,
                 better || gram_count_override > 0 ? "grams" : "base");
  if (!better && gram_count_override <= 0) return 0;
  gp->g = g; gp->G = G; gp->b = b; gp->top = top;
  return 0;
}

// em != nullptr: the rankers emit the transform's bytes (see RrEmit) and the suffix array itself
// is not stored; em == nullptr: d_SA is filled and nothing is emitted.
int BwtEngine::suffix_sort(u32 n, const u32* hist, bool lone_sentinel, const EmitTarget* em) {
  hipStream_t st = stream;
  n_sort_events = 0;
  stats.rounds = 0;
  stats.active_sum = 0;
  stats.sort_pass_items = 0;
  stats.route = 0;
  stats.finisher_entries = 0;
  if (n == 0) return 0;

  const bool emit = em != nullptr;
  const KeyPlan plan = plan_keys(hist, n, lone_sentinel, gram_min_n < kGramMinN);   // (the tests' small blocks: 64-bit keys whatever the entropy says)
  std::memcpy(h_small + kSmallLut, plan.lut, 256);
  BWTC_HIP_TRY(hipMemcpyAsync(d_small + kSmallLut, h_small + kSmallLut, 256, hipMemcpyHostToDevice, st));
  const u8* d_lut = reinterpret_cast<const u8*>(d_small + kSmallLut);
  int key_bits = plan.bits;
  // suffixes of length <= k are finished by the initial ranking (a suffix of length exactly
  // k is a proper prefix of every other suffix with the same key)
  u32 short_len = (u32)plan.k;

  RrEmit re;
  re.out = emit ? em->out : nullptr;
  re.out_n = emit ? em->out_n : 0u;
  re.last_char = d_small + kSmallLastChar;
  re.pidx = d_small + kSmallPidx;
  re.T = d_T;
  re.achr_out = d_C0;
  re.rec_plane = nullptr;
  re.rec_shift = 0;
  re.inv_lut = reinterpret_cast<const u8*>(d_small + kSmallInv);
  re.lf = d_small + kSmallLf; re.lf_n = 0; re.lf_x = 0; re.lf_inv = 0;
  if (emit && em->n_lf > 1 && n / em->n_lf >= 1) {
    re.lf_n = em->n_lf; re.lf_x = n / em->n_lf; re.lf_inv = re.lf_x > 1 ? (u32)((1ull << 32) / re.lf_x) : 0u;
  }
  bridged = false;
  lf_noted = false;
  fin_active = false;
  u8* achr_other = d_C1;

  RankBuffers rb;
  rb.aglob = nullptr;
  rb.aglob_next = d_G0;
  RankResult res;
  res.finish = false;
  u64 h = (u64)plan.k;                    // the next round compares rank[s + h]

  // initial sort + ranking
  BWTC_HIP_TRY(hipEventRecord(ev_sort[n_sort_events++], st));
  int rc = 0;
  u8* key_plane = digit_planes && !use_sweep ? d_P0 : nullptr;     // the key makers leave the first pass's digits
  // slot j holds suffix n-1-j: the sort's first pass can make that up instead of reading it
  const bool implied_idx = !use_sweep && n > 1;
  u32* idx_out = implied_idx ? nullptr : d_V0;
  if (plan.wide) {
    u64* ka = static_cast<u64*>(d_R1);
    u64* kb = static_cast<u64*>(d_R2);
    GramPlan gp;
    const bool code_route = code_keys && finisher && !code_failed && long_keys && implied_idx && emit && key_plane && n >= gram_min_n;
    if (!code_route) {
      rc = plan_grams(plan, n, d_lut, &gp);
      if (rc) return rc;
    }
    if (gp.G > 0) {
      key_bits = gp.b * gp.G;
      short_len = (u32)(gp.g * gp.G);
      h = (u64)short_len;
    }
    // Keys of at most 48 bits leave room for the upper bits of the suffix number: the passes of
    // the initial sort then carry (u64 key, u16 low half) = 10 bytes per item instead of 12.
    const int split = (split_index && implied_idx && key_bits <= 48 && n <= (1u << 29)) ? 1 : 0;
    // Long keys: the sort also orders by a second key word and the items carry their predecessor character's code, so
    // that the ranking behind the ONE long sort can emit the transform's bytes and the finisher takes what is tied.
    // Two makers: the order-1 prefix code (k_make_keys_code, the default) and dense gram codes (round 4, BWTC_HIP_KEYS=grams).
    LongKey lk;
    lk.w = nullptr; lk.G2 = 0; lk.first_r = -1; lk.first_mask = 0; lk.hi_shift = 48; lk.chr_shift = 0; lk.chr_mask = 0;
    const bool long_ok = long_keys && implied_idx && emit && key_plane && n >= gram_min_n;
    const bool coded = long_ok && code_keys && finisher && !code_failed;
    auto upload_inverse_lut = [&]() -> int {
      u8 inv[256];
      std::memset(inv, 0, sizeof inv);
      for (int c = 255; c >= 0; --c) if (hist[c] && !(c == 0 && lone_sentinel)) inv[plan.lut[c]] = (u8)c;
      std::memcpy(h_small + kSmallInv, inv, 256);
      BWTC_HIP_TRY(hipMemcpyAsync(d_small + kSmallInv, h_small + kSmallInv, 256, hipMemcpyHostToDevice, st));
      return 0;
    };
    // (without the split -- blocks above 512 MiB, gram keys above 48 bits -- the items carry whole 32-bit suffix numbers:
    // 16 bytes instead of 14, and the key only needs room for the character)
    int w_bits = 0;
    RrLong lg;
    lg.w = nullptr; lg.wmask = ~0u; lg.hi_shift = 0; lg.chr_shift = 0; lg.chr_mask = 0; lg.lvl_shift = -1; lg.depth = 0;
    int split_now = split;
    if (coded) {
      // the block's order-1 statistics from a sample of its character pairs, the alphabetic codes, the keys
      split_now = (split_index && n <= (1u << 29)) ? 1 : 0;
      key_bits = code_bits - 32;                        // the key word's share; the second word takes 32
      w_bits = 32;
      const int hi_bits = split_now ? 13 : 0;
      CodeKey ck;
      ck.codes = d_codes; ck.kbits = code_bits; ck.hi_shift = key_bits; ck.chr_shift = key_bits + hi_bits; ck.lvl_shift = ck.chr_shift + 8;
      rc = upload_inverse_lut();
      if (rc) return rc;
      BWTC_HIP_TRY(hipMemsetAsync(d_pairs, 0, (size_t)kPairReplicas * 65536 * 4, st));
      const u32 sample_tiles = std::min<u32>(ceil_div(n, kPairTile), 512u);
      hipLaunchKernelGGL(k_pair_counts, dim3(sample_tiles), dim3(256), 0, st, (const u8*)d_T, d_lut, n, d_pairs);
      hipLaunchKernelGGL(k_code_build, dim3(kCodeRows), dim3(256), 0, st, (const u32*)d_pairs, plan.sigma, d_codes, d_small + kSmallError);
      hipLaunchKernelGGL(k_make_keys_code, dim3(ceil_div(n, 1024)), dim3(256), 0, st, (const u8*)d_T, d_lut, ka, d_W0, n, key_plane, split_now, ck);
      lg.hi_shift = ck.hi_shift; lg.chr_shift = ck.chr_shift; lg.chr_mask = 0xFFu; lg.lvl_shift = ck.lvl_shift;
      lk.w = d_W0;
      short_len = 0;                                    // every item has its own (rr_masks)
      h = 1;
      stats.alg_bytes += (u64)sample_tiles * kPairTile + (u64)kPairReplicas * 65536 * 8;
      if (std::getenv("BWTC_HIP_DEBUG"))
        std::fprintf(stderr, "code keys: %d + %d bits, sigma %u, %u sampled tiles, character code at bit %d, level at bit %d\n",
                     key_bits, w_bits, plan.sigma, sample_tiles, ck.chr_shift, ck.lvl_shift);
    } else if (long_ok && gp.G > 0) {
      const int cb = std::max(1, bit_width_u64(plan.sigma - 1));
      int G2 = std::min(32 / gp.b, (64 - gp.g * gp.G) / gp.g);
      G2 = std::min(G2, long_grams_override > 0 ? long_grams_override : 2);
      const int hi_bits = split ? 13 : 0;
      if (G2 >= 1 && key_bits + hi_bits + cb <= 64) {
        lk.w = d_W0; lk.G2 = G2; lk.hi_shift = key_bits; lk.chr_shift = key_bits + hi_bits; lk.chr_mask = (1u << cb) - 1u;
        rc = upload_inverse_lut();
        if (rc) return rc;
        w_bits = gp.b * lk.G2;
        if (w_bits < kRadixBits) { lk.first_r = w_bits; lk.first_mask = (1u << std::min(kRadixBits, w_bits + key_bits)) - 1u; }
        short_len = (u32)(gp.g * (gp.G + lk.G2));
        h = (u64)short_len;
        hipLaunchKernelGGL(k_make_keys_gram, dim3(ceil_div(n, 1024)), dim3(256), 0, st, d_T, d_lut,
                           (const uint4*)d_rank, ka, (u32*)nullptr, n, gp.g, gp.G, gp.b, plan.sigma, gp.top, key_plane, split, lk);
        lg.hi_shift = lk.hi_shift; lg.chr_shift = lk.chr_shift; lg.chr_mask = lk.chr_mask; lg.lvl_shift = -1; lg.depth = short_len;
        if (std::getenv("BWTC_HIP_DEBUG"))
          std::fprintf(stderr, "long keys: %d + %d bits, %d + %d grams of %d characters, character code at bit %d\n",
                       key_bits, w_bits, gp.G, lk.G2, gp.g, lk.chr_shift);
      }
    }
    if (lk.w) {
      u64* ks = nullptr; unsigned short* vs16 = nullptr; u32* ws = nullptr;
      u32* vs32 = nullptr;
      if (!split_now)
        radix_sort_long<u32, 6>(ka, kb, d_V0, d_V1, d_W0, d_W1, n, key_bits, w_bits, d_table, d_partial, stream, &ks, &vs32, &ws, &probe, d_P0, d_P1, 0, false, long_direct);
      else if (long_items_per_thread == 6)
        radix_sort_long<unsigned short, 6>(ka, kb, reinterpret_cast<unsigned short*>(d_V0), reinterpret_cast<unsigned short*>(d_V1),
                                           d_W0, d_W1, n, key_bits, w_bits, d_table, d_partial, stream, &ks, &vs16, &ws, &probe, d_P0, d_P1, 0, false, long_direct);
      else
        radix_sort_long<unsigned short, 8>(ka, kb, reinterpret_cast<unsigned short*>(d_V0), reinterpret_cast<unsigned short*>(d_V1),
                                           d_W0, d_W1, n, key_bits, w_bits, d_table, d_partial, stream, &ks, &vs16, &ws, &probe, d_P0, d_P1, 0, false, long_direct);
      BWTC_HIP_TRY(hipEventRecord(ev_sort[n_sort_events++], st));
      u32* vs = split_now ? reinterpret_cast<u32*>(vs16) : vs32;
      rb.rec_keys = ks; rb.rec_free = ks == ka ? kb : ka;
      rb.v_keys = vs; rb.v_free = vs == d_V0 ? d_V1 : d_V0;
      lg.w = ws; lg.wmask = w_bits >= 32 ? ~0u : (1u << w_bits) - 1u;
      rc = rank_step<u64, true>(ks, vs, n, n, short_len, (1ull << key_bits) - 1ull, rb, re, emit, h, &res, (u32)split_now, &lg);
      if (rc == -3 && coded && (h_small[kSmallError] & 2u)) {
        // a codeword came out longer than the table's fields hold (k_code_build): nothing has been emitted yet --
        // this block again, with gram keys
        if (std::getenv("BWTC_HIP_DEBUG")) std::fprintf(stderr, "code keys: a codeword above %u bits, the block takes gram keys\n", kCodeMaxLen);
        BWTC_HIP_TRY(hipMemsetAsync(d_small + kSmallError, 0, 4, st));
        code_failed = true;
        rc = suffix_sort(n, hist, lone_sentinel, em);
        code_failed = false;
        return rc;
      }
      if (rc) return rc;
      const int long_passes = (key_bits + w_bits + kRadixBits - 1) / kRadixBits;
      stats.sort_pass_items += (u64)n * (u64)long_passes;
      stats.alg_bytes += (u64)n * (1 + 12 + 1)               // the key maker: T read, key + second word + first plane written (the values are made up by the first pass)
                         + sort_bytes(n, long_passes, split_now ? 14 : 16, 8, true);
      stats.route |= coded ? 17u : 1u;
      key_bits = 0;                                      // counted
    } else if (gp.G > 0) {
      hipLaunchKernelGGL(k_make_keys_gram, dim3(ceil_div(n, 1024)), dim3(256), 0, st, d_T, d_lut,
                         (const uint4*)d_rank, ka, idx_out, n, gp.g, gp.G, gp.b, plan.sigma, gp.top, key_plane, split, lk);
    } else {
      u64 top = 1;
      for (int t = 1; t < plan.k; ++t) top *= plan.sigma;
      hipLaunchKernelGGL(k_make_keys<u64>, dim3(ceil_div(n, 1024)), dim3(256), 0, st, d_T, d_lut, ka,
                         idx_out, n, plan.k, plan.sigma, top, key_plane, split);
    }
    u64* ks = nullptr; u32* vs = nullptr;
    if (lk.w) {
      // sorted and ranked above
    } else if (split) {
      unsigned short* vs16 = nullptr;
      radix_sort_pairs<u64, unsigned short>(ka, kb, reinterpret_cast<unsigned short*>(d_V0), reinterpret_cast<unsigned short*>(d_V1),
                                            n, key_bits, d_table, d_partial, stream, &ks, &vs16, &probe, 0, false, false, 0,
                                            digit_planes ? d_P0 : nullptr, digit_planes ? d_P1 : nullptr,
                                            key_plane != nullptr, true);
      vs = reinterpret_cast<u32*>(vs16);
    } else {
      sort_pairs<u64>(ka, kb, d_V0, d_V1, n, key_bits, &ks, &vs, true, 0, 0, key_plane != nullptr, implied_idx);
    }
    if (!lk.w) {
      BWTC_HIP_TRY(hipEventRecord(ev_sort[n_sort_events++], st));
      rb.rec_keys = ks; rb.rec_free = ks == ka ? kb : ka;
      rb.v_keys = vs; rb.v_free = vs == d_V0 ? d_V1 : d_V0;
      rc = rank_step<u64, true>(ks, vs, n, n, short_len, split ? ((1ull << 48) - 1ull) : ~0ull, rb, re, emit, h, &res, (u32)split);
    }
  } else {
    u32* ka = static_cast<u32*>(d_R1);
    u32* kb = ka + cap;
    u32 top = 1;
    for (int t = 1; t < plan.k; ++t) top *= plan.sigma;
    hipLaunchKernelGGL(k_make_keys<u32>, dim3(ceil_div(n, 1024)), dim3(256), 0, st, d_T, d_lut, ka,
                       idx_out, n, plan.k, plan.sigma, top, key_plane, 0);
    u32* ks = nullptr; u32* vs = nullptr;
    sort_pairs<u32>(ka, kb, d_V0, d_V1, n, key_bits, &ks, &vs, true, 0, 0, key_plane != nullptr, implied_idx);
    BWTC_HIP_TRY(hipEventRecord(ev_sort[n_sort_events++], st));
    rb.rec_keys = d_R1; rb.rec_free = d_R2;             // both 32-bit key arrays live in R1
    rb.v_keys = vs; rb.v_free = vs == d_V0 ? d_V1 : d_V0;
    rc = rank_step<u32, true>(ks, vs, n, n, short_len, ~0u, rb, re, emit, h, &res);
  }
  if (rc) return rc;
  stats.sort_pass_items += (u64)n * (u64)((key_bits + kRadixBits - 1) / kRadixBits);
  if (key_bits) {
    const int item = plan.wide ? 12 : 8;                  // (u64, u32 or u16 + upper bits in the key) / (u32, u32)
    stats.alg_bytes += (u64)n * (1 + (u64)(plan.wide ? 8 : 4) + 1) + sort_bytes(n, (key_bits + kRadixBits - 1) / kRadixBits, item, plan.wide ? 8 : 4, digit_planes);
  }

  bool ranks_complete = true;
  // The rounds over the sorted list in res / rb, from depth h0 until nothing is tied.  text_rounds_now: rounds that compare
  // the text itself first (finisher route: rank[] is not complete, and completing it costs an ISA scatter of the whole block).
  // doubling == false: text rounds only; what they leave tied comes back raw in (rb.v_free, rb.aglob_next, d_GRP) --
  // suffix, slot, head slot -- *left entries at depth *h_left.
  u32 shallow_depth = 0xFFFFFFFFu;
  auto run_rounds = [&](u32 m, u64 h0, int text_rounds_now, bool keep_first, bool doubling, u32* left, u64* h_left) -> int {
    u64 h = h0;
    bool keep_h = keep_first;                // the list is sorted to depth h as it stands (no doubling before the next step)
    int text_left = text_rounds_now;
    int text_extra = text_rounds_fixed ? 0 : 12;        // further ones, one at a time, while the list is short (not when BWTC_HIP_TEXT_ROUNDS says how many)
    while (m > 0 || (doubling && local_m > 0)) {
      if (h >= (u64)n * 2 + 64) return -3;   // cannot happen: every group splits by then
      ++stats.rounds;
      stats.active_sum += m;
      const bool text = text_left > 0 && m > 0;
      const bool raw = !text && !doubling;               // out of text rounds and not allowed to double: hand the list back
      if (std::getenv("BWTC_HIP_DEBUG")) std::fprintf(stderr, "round %u%s: h=%llu m=%u groups=%u\n", stats.rounds, text ? " (text)" : "", (unsigned long long)h, m, res.groups);
      if (!keep_h) h *= 2;
      keep_h = false;
      if (!text && !raw && !ranks_complete) {
        // the doubling rounds start here: rank[] for everybody first (the finished from SA; the list's own
        // ranks come from the ranking step below, before anything reads them; a parked list's are its groups' heads)
        complete_ranks(n, res.vs, rb.aglob_next, m, res.rec_other == d_R2 ? (res.ks == (u64*)d_R1 ? (void*)d_W0 : d_R1) : d_R2, res.rec_other);
        ranks_complete = true;
        ranks_live = true;
        bridged = true;
        stats.route |= 8u;
        stats.alg_bytes += (u64)n * (4 + 8);                // k_bridge_pairs_all (the partition and scatter count themselves)
      }
      // what the local list's last pass noted reaches rank[] now: before this round's look-ups, after the last round's
      if (!text && !raw) { const int rcu = local_updates(); if (rcu) return rcu; }
      if (m == 0) {                                       // only the local list is left
        const int rcl = local_pass(n, 0, re);
        if (rcl) return rcl;
        continue;
      }
      // the list just sorted: positions are the active list's, aglob gives their global slots
      rb.aglob = rb.aglob_next;
      rb.aglob_next = rb.aglob == d_G0 ? d_G1 : d_G0;
      { u8* t = re.achr_out; re.achr_out = achr_other; achr_other = t; }
      rb.rec_keys = res.ks; rb.rec_free = res.rec_other;
      rb.v_keys = res.vs; rb.v_free = res.v_other;
      const u64* ks = res.ks;
      const u32* vs = res.vs;
      const bool carried = emit && res.carry;             // this list's keys hold the characters in bits 56..63
      const int rc2 = rank_step<u64, false>(ks, vs, m, n, 0u, carried ? ((1ull << 56) - 1ull) : ~0ull, rb, re, emit, h, &res, 0u, nullptr, text || raw, carried, raw);
      if (rc2) return rc2;
      m = res.m;
      if (raw) { *left = m; *h_left = h; return 0; }
      // the local list's doubling step beside the global list's (both read the rank[] this round's ranking step left)
      if (!text && doubling && local_m > 0) { const int rcl = local_pass(n, h, re); if (rcl) return rcl; }
      if (text) {
        h += res.text_chars; keep_h = true; --text_left;
        // Out of text rounds with a short list left: a few more cost tens of microseconds each, the doubling rounds
        // cost the completion of rank[] for the whole block first (5 ms per 256 MiB, 16 ms for the 1 GiB text -- the
        // block whose text rounds ended with 10 entries still tied paid it).  Long lists (deep repeats) go on to the
        // doubling rounds as before: their depth doubles there and only creeps here.
        if (text_left == 0 && m > 0 && (u64)m * 4096 < (u64)n && text_extra > 0) { text_left = 1; --text_extra; }
      }
    }
    if (left) { *left = 0; *h_left = h; }
    return 0;
  };
  ranks_live = false;
  parked = 0; park_holes = 0;
  local_m = 0; local_pending = false;
  if (res.finish) {
    if (re.lf_n == 0 && em->n_lf > 1) return -3;
    lf_noted = true;                       // every suffix that becomes final notes its LF power (lf_note), in every route
    ranks_complete = false;
    u32* cnt = d_small + kSmallFin;
    h_small[kSmallFin + 3] = 0xFFFFFFFFu; h_small[kSmallFin + 4] = 0; h_small[kSmallFin + 5] = 0xFFFFFFFFu; h_small[kSmallFin + 6] = 0xFFFFFFFFu;
    BWTC_HIP_TRY(hipMemcpyAsync(cnt + 3, h_small + kSmallFin + 3, 16, hipMemcpyHostToDevice, st));
    // the ranking's list through the finisher
    FinList la{rb.v_free, rb.aglob_next, d_GRP, reinterpret_cast<unsigned short*>(re.achr_out)};
    FinList lb{d_W0, d_W1, rb.aglob_next == d_G0 ? d_G1 : d_G0, reinterpret_cast<unsigned short*>(re.achr_out == d_C0 ? d_C1 : d_C0)};
    FinShallow shal{rb.v_keys, static_cast<u64*>(rb.rec_free), cnt + 4, fin_floor};
    FinOutcome fo;
    // Groups of hundreds of members on average (a period, one text many times over) are not the finisher's: its loop
    // costs a group's size per member.  The list then goes to the rounds as it is (shallow groups apart).
    const bool giant = res.groups > 0 && (u64)res.m > (u64)res.groups * 64;
    // The shape of the passes.  The comparison loop costs a group's size per member, which is why groups above 256 members
    // are not the finisher's on a real text (most of the block is on its list).  A SHORT list -- the long keys told
    // nearly everything apart: 2.6 % of the generator's text is left -- takes windows of 2048 entries and groups of up to
    // 1024 members instead: the few large groups (tokens repeated: 19 K entries of 7 M) are settled with the others and
    // no hard list is left for the text rounds, whose five rounds of tiny launches cost more than the whole pass
    // (256 MiB: 21.9 -> 21.3 ms).  Bounded: the wide shape costs at most 0.45 ns per entry more (measured on a real
    // text, where every group is large), n / 24 entries at most.
    fin_window_blk = fin_window; fin_group_blk = fin_max_group;
    // One pass only in that shape: what three rounds of sixteen characters leave tied in a short list is deep repeats
    // (copies of a long piece), the rounds' business -- further passes over groups of hundreds of members cost their size
    // per member and settle nothing (300 copies of a 10 KB piece in the generator's text: 32.6 ms with three passes, 30.0
    // in the narrow shape, where such groups are hard at once).
    bool wide = false;
    if (!fin_shape_fixed && fin_wide_short && !giant && (u64)res.m * 24 <= (u64)n) { fin_window_blk = 2048; fin_group_blk = 1024; wide = true; }
    rc = finisher_passes(n, res.m, la, lb, re, shal, &fo, local_rounds, giant ? 0 : wide ? std::min(1, fin_max_passes) : fin_max_passes);
    if (rc) return rc;
    const u32 shallow = h_small[kSmallFin + 4];
    const u64 h_sh = h_small[kSmallFin + 5];
    // text rounds -- a global sort per six characters -- are for what is left of a text; a long list is deep repeats,
    // which only doubling gets through
    auto text_for = [&](u32 m) { return ((u64)m * 16 < (u64)n || text_rounds_fixed) ? text_rounds : 0; };
    if (shallow) {
      // The few hard groups of little depth first, on their own: text rounds only (rank[] must not go live while
      // finisher passes are still to come); what they leave tied joins the waiting list, at the depth they reached.
      rc = dress_list(n, shallow, rb.v_keys, static_cast<u64*>(rb.rec_free), rb, &res);
      if (rc) return rc;
      u32 left = 0; u64 h_left = 0;
      rc = run_rounds(shallow, std::max<u64>(1, h_sh), std::max(text_rounds, 3), true, false, &left, &h_left);
      if (rc) return rc;
      if (left) {
        if ((u64)parked + left > cap) return -3;
        hipLaunchKernelGGL(k_raw_to_park, dim3(ceil_div(left, 256)), dim3(256), 0, st, (const u32*)rb.v_free, (const u32*)rb.aglob_next,
                           (const u32*)d_GRP, left, d_parkS, d_parkHP, parked);
        parked += left;
        shallow_depth = (u32)std::min<u64>(h_left, 0xFFFFFFFFull);
        if (std::getenv("BWTC_HIP_DEBUG")) std::fprintf(stderr, "shallow groups: %u entries still tied at depth %llu join the waiting list\n", left, (unsigned long long)h_left);
      }
    }
    const u32 hard = fo.hard, hard_depth = fo.hard_depth;
    if (parked || local_m) {
      BWTC_HIP_TRY(hipMemcpyAsync(h_small + kSmallFin + 3, cnt + 3, 4, hipMemcpyDeviceToHost, st));
      BWTC_HIP_TRY(wait());
      u64 h_pk = std::min<u64>(std::min<u64>(h_small[kSmallFin + 3], hard ? hard_depth : 0xFFFFFFFFu), shallow_depth);
      if (local_m) h_pk = std::min<u64>(h_pk, local_depth);       // every rank is at least as deep as the rounds' first look-up
      const u32 total = parked;
      if (std::getenv("BWTC_HIP_DEBUG"))
        std::fprintf(stderr, "finisher: %u entries (%u of groups too large) go on to the rounds at depth %llu; %u entries of small groups double beside them from depth %u\n",
                     total, hard, (unsigned long long)h_pk, local_m, local_m ? local_depth : 0u);
      if (total) {
        BWTC_HIP_TRY(hipMemcpyAsync(d_V0, d_parkS, (size_t)total * 4, hipMemcpyDeviceToDevice, st));
        BWTC_HIP_TRY(hipMemcpyAsync(d_R1, d_parkHP, (size_t)total * 8, hipMemcpyDeviceToDevice, st));
        parked = 0;                        // (the list is in the rounds' hands now: complete_ranks takes it from them)
        rc = dress_list(n, total, d_V0, static_cast<u64*>(d_R1), rb, &res, park_holes);
        if (rc) return rc;
      } else {
        // no global list: the rounds still want free regions to complete rank[] in
        res.m = 0; res.groups = 0;
        res.ks = reinterpret_cast<u64*>(d_W0); res.rec_other = d_R1;
        res.vs = d_V0; res.v_other = d_V1;
        rb.aglob_next = d_G0;
      }
      const u32 listed = total ? res.m : 0u;             // (the waiting list without its holes)
      rc = run_rounds(listed, std::max<u64>(1, h_pk), local_m ? 0 : text_for(listed), true, true, nullptr, nullptr);
      if (rc) return rc;
    }
  } else {
    rc = run_rounds(res.m, h, 0, false, true, nullptr, nullptr);
    if (rc) return rc;
  }
  BWTC_HIP_TRY(hipGetLastError());
  BWTC_HIP_TRY(take_sticky_error());
  return 0;
}

// The finisher passes over the list the long-key ranking left (k_finish).  Every entry knows the depth
// its group shares (the upper byte of its character field); a pass adds kFinChars to it.  Passes go on
// while the list shrinks: one that keeps more than three fifths of a list that is still longer than
// n / 64 has met deep repeats (copies of whole files, not of lines), which only doubling gets through.
// What is left for the rounds, as raw lists (suffix; head slot << 32 | slot), unsorted:
//   *shallow entries in (rb.v_keys, rb.rec_free): groups too large for a window that share fewer than
//    fin_floor characters; they share at least *h_shallow
//   `parked` entries in (d_parkS, d_parkHP): the other groups too large for a window, and what the last
//    pass left tied; at least *h_parked characters
int BwtEngine::finisher_passes(u32 n, u32 m, FinList a, FinList b, RrEmit& re, FinShallow shal, FinOutcome* fo, bool keep_local, int max_passes) {
  hipStream_t st = stream;
  u32* cnt = d_small + kSmallFin;      // [1] hard list, [2] its smallest depth; [3] smallest depth of what waits for the rounds; [4] shallow list, [5] its smallest depth
  u32* ncnt = d_small + kSmallFinNext; // entries of the next list, region by region
  h_small[kSmallFin] = 0; h_small[kSmallFin + 1] = 0; h_small[kSmallFin + 2] = 0xFFFFFFFFu;
  BWTC_HIP_TRY(hipMemcpyAsync(cnt, h_small + kSmallFin, 12, hipMemcpyHostToDevice, st));
  u32 hard = 0;
  passes_done = 0;
  const u32 park0 = parked;            // the groups too large for a window join the waiting list from here
  const int window = fin_window_blk, group = fin_group_blk;
  const u32 stride = (u32)(window - group);
  FinRegions rg;
  std::memset(&rg, 0, sizeof rg);
  rg.nreg = 1; rg.wfirst[0] = 0; rg.wfirst[1] = ceil_div(m, stride); rg.ebase[0] = 0; rg.ecount[0] = m;   // the list as given is one region
  u32 groups_left = 0;
  const u32 pass_chars = 8u * (u32)fin_words * (fin_words == 2 ? (u32)fin_rounds : 1u);
  const int depth_passes = (int)((255u - 48u) / pass_chars);      // the depth byte holds 255 (a level's depth is 48 at most)
  for (int it = 0; it < std::min(std::min(fin_max_passes, std::max(1, depth_passes)), max_passes) && m > 0; ++it) {
    ++stats.rounds;
    stats.active_sum += m;
    stats.finisher_entries += m;
    stats.route |= 2u;
    // a window of entries is read by the workgroups whose windows overlap there (4/3 on average), sixteen characters
    // (three aligned words) per owned entry, a byte and a suffix per finished one, an entry per member that stays
    stats.alg_bytes += (u64)m * (14 * 4 / 3 + (pass_chars + 8) + 5);
    BWTC_HIP_TRY(hipMemsetAsync(ncnt, 0, (kFinRegions + 1) * 4, st));
    const u32 grid = rg.wfirst[rg.nreg];
    // the regions of the list this pass leaves: chunk c of 64 workgroups appends to region c mod 16 and leaves at most
    // its workgroups' strides of entries and one group more
    FinOut ob;
    {
      u32 room[kFinRegions] = {};
      const u32 nchunks = ceil_div(grid, kFinChunk);
      for (u32 c = 0; c < nchunks; ++c) {
        const u32 wgs = std::min(kFinChunk, grid - c * kFinChunk);
        room[c % kFinRegions] += wgs * stride + (u32)group;
      }
      u32 at = 0;
      for (u32 r = 0; r < kFinRegions; ++r) { ob.base[r] = at; at += room[r]; }
      if ((u64)at > cap + cap / 128 + 65536) return -3;
    }
#define BWTC_FINISH_W(G, E, NW) hipLaunchKernelGGL((k_finish<G, E, NW>), dim3(grid), dim3(kFinTPB), 0, st, a, rg, (const u8*)d_T, n, \
                                             b, ob, ncnt, d_parkS + park0, d_parkHP + park0, d_hardC, cnt + 1, shal, d_SA, re)
#define BWTC_FINISH_R(G, E, NR) hipLaunchKernelGGL((k_finish<G, E, 2, false, NR>), dim3(grid), dim3(kFinTPB), 0, st, a, rg, (const u8*)d_T, n, \
                                             b, ob, ncnt, d_parkS + park0, d_parkHP + park0, d_hardC, cnt + 1, shal, d_SA, re)
#define BWTC_FINISH(G, E) do { if (fin_words >= 4) BWTC_FINISH_W(G, E, 4); else if (fin_words == 3) BWTC_FINISH_W(G, E, 3); \
                               else if (fin_rounds >= 4) BWTC_FINISH_R(G, E, 4); else if (fin_rounds == 3) BWTC_FINISH_R(G, E, 3); \
                               else if (fin_rounds == 2) BWTC_FINISH_R(G, E, 2); else BWTC_FINISH_W(G, E, 2); } while (0)
    if (window <= 1024) { if (group <= 256) BWTC_FINISH(256, kFinE1); else BWTC_FINISH(512, kFinE1); }
    else if (group <= 256) BWTC_FINISH(256, kFinE2);
    else if (group <= 512) BWTC_FINISH(512, kFinE2);
    else BWTC_FINISH(1024, kFinE2);
#undef BWTC_FINISH
#undef BWTC_FINISH_W
#undef BWTC_FINISH_R
    BWTC_HIP_TRY(hipMemcpyAsync(h_small + kSmallFin, cnt, 24, hipMemcpyDeviceToHost, st));
    BWTC_HIP_TRY(hipMemcpyAsync(h_small + kSmallFinNext, ncnt, (kFinRegions + 1) * 4, hipMemcpyDeviceToHost, st));
    BWTC_HIP_TRY(wait());
    const u32 m_was = m;
    groups_left = h_small[kSmallFinNext + kFinRegions];
    // what the pass left, region by region
    {
      FinRegions nx;
      std::memset(&nx, 0, sizeof nx);
      u32 total = 0;
      for (u32 r = 0; r < kFinRegions; ++r) {
        const u32 c = h_small[kSmallFinNext + r];
        if (!c) continue;
        nx.ebase[nx.nreg] = ob.base[r]; nx.ecount[nx.nreg] = c;
        nx.wfirst[nx.nreg + 1] = nx.wfirst[nx.nreg] + ceil_div(c, stride);
        ++nx.nreg;
        total += c;
      }
      if (nx.nreg == 0) nx.nreg = 1;
      rg = nx;
      m = total;
    }
    if (std::getenv("BWTC_HIP_DEBUG"))
      std::fprintf(stderr, "finisher pass %d: %u entries -> %u still tied; %u hard (depth from %u), %u shallow (from %u)\n", it, m_was, m,
                   h_small[kSmallFin + 1], h_small[kSmallFin + 2], h_small[kSmallFin + 4], h_small[kSmallFin + 5]);
    hard = h_small[kSmallFin + 1];
    std::swap(a, b);
    passes_done = it + 1;
    if (m < 4096u) break;                                                          // a handful: they join the waiting list (a pass of their own is a launch and a host wait for nothing)
    if (it >= 1 && (u64)m * 5 > (u64)m_was * 3 && (u64)m * 64 > (u64)n) break;     // deep repeats: on to the rounds
    if (it >= 2 && (u64)m * 4 > (u64)m_was * 3) break;                             // a list that hardly shrinks any more: the rounds double, this creeps
  }
  parked += hard;
  if ((u64)parked > cap) return -3;
  // (a pass has run: no group above the bound is left in the list; and its groups are small -- the comparison loop costs a
  // group's size per member: a block that is one text 256 times over has groups of 256 throughout and is the global rounds')
  if (m && keep_local && passes_done >= 1 && (u64)m * 64 >= (u64)n && (u64)groups_left * 24 >= (u64)m) {
    // deep repeats in small groups: the doubling rounds take them group by group (local_pass), not through the global sort
    FinList home{d_hardS, d_LP0, d_LH0, d_hardC};
    u32 at = 0;
    for (u32 r = 0; r < rg.nreg; ++r) {
      if (!rg.ecount[r]) continue;
      FinList part{a.S + rg.ebase[r], a.P + rg.ebase[r], a.H + rg.ebase[r], a.C + rg.ebase[r]};
      hipLaunchKernelGGL(k_list_copy, dim3(ceil_div(rg.ecount[r], 256)), dim3(256), 0, st, part, rg.ecount[r], home, at, cnt + 6);
      at += rg.ecount[r];
    }
    BWTC_HIP_TRY(hipMemcpyAsync(h_small + kSmallFin + 6, cnt + 6, 4, hipMemcpyDeviceToHost, st));
    BWTC_HIP_TRY(wait());
    local_depth = h_small[kSmallFin + 6];
    // its smallest depth: every member has taken the same passes, from a level's depth
    local_m = m;
    std::memset(&local_rg, 0, sizeof local_rg);
    local_rg.nreg = 1; local_rg.wfirst[1] = ceil_div(m, stride); local_rg.ecount[0] = m;
    local_home = 0;
    fo->left = 0;
    fo->hard = hard;
    fo->hard_depth = h_small[kSmallFin + 2];
    fo->local = m;
    return 0;
  }
  if (m) {
    // what the last pass left tied waits for the rounds
    if ((u64)parked + m > cap) return -3;
    for (u32 r = 0; r < rg.nreg; ++r) {
      if (!rg.ecount[r]) continue;
      FinList part{a.S + rg.ebase[r], a.P + rg.ebase[r], a.H + rg.ebase[r], a.C + rg.ebase[r]};
      hipLaunchKernelGGL(k_fin_to_hard, dim3(ceil_div(rg.ecount[r], 256)), dim3(256), 0, st, part, rg.ecount[r], d_parkS, d_parkHP, parked, cnt + 3,
                         passes_done ? FinShallow{shal.S, shal.HP, shal.count, 0u} : shal);
      parked += rg.ecount[r];
    }
    if (!passes_done) {
      // a list no pass has looked at: its shallow entries went to the shallow list and left holes behind
      const u32 before = h_small[kSmallFin + 4];
      BWTC_HIP_TRY(hipMemcpyAsync(h_small + kSmallFin + 4, cnt + 4, 8, hipMemcpyDeviceToHost, st));
      BWTC_HIP_TRY(wait());
      park_holes += h_small[kSmallFin + 4] - before;
    }
  }
  fo->hard = hard;
  fo->hard_depth = h_small[kSmallFin + 2];
  fo->left = m;
  return 0;
}

// The local list: what the finisher's passes left of a block with deep repeats -- small groups (none above the
// finisher's bound), many of them.  In the doubling rounds it is refined group by group in LDS (k_finish<RANK>)
// beside the global list, which keeps the groups too large for that.
FinList BwtEngine::local_list(int home) const {
  const u64 lcap = cap + cap / 128 + 65536;
  if (home == 0) return FinList{d_hardS, d_LP0, d_LH0, d_hardC};
  return FinList{reinterpret_cast<u32*>(d_hardHP), reinterpret_cast<u32*>(d_hardHP) + lcap, d_LH1, d_LC1};
}
int BwtEngine::local_updates() {
  if (!local_pending) return 0;
  hipStream_t st = stream;
  for (u32 r = 0; r < local_upd.nreg; ++r)
    if (local_upd.ecount[r])
      hipLaunchKernelGGL(k_rank_updates, dim3(ceil_div(local_upd.ecount[r], 1024)), dim3(256), 0, st, (const u32*)d_US, (const u32*)d_UR,
                         local_upd.ebase[r], local_upd.ecount[r], d_rank);
  local_pending = false;
  return 0;
}
// One doubling step of the local list.  h_global: the depth the global list's groups are sorted to (0: there is no
// global list); every rank[] entry is at least min(h_global, local_depth) deep, so a member looks up rank[s + local_depth]
// and, that far behind it, a second rank: the step adds twice that depth.  A local list far ahead of the global one
// (two of its steps and more) sits the round out.
int BwtEngine::local_pass(u32 n, u64 h_global, RrEmit& re) {
  hipStream_t st = stream;
  if (local_m == 0) return 0;
  if (h_global && (u64)local_depth >= 2 * h_global) return 0;
  const u64 delta = h_global ? std::min<u64>(h_global, local_depth) : local_depth;
  const u64 at = local_depth, at2 = (u64)local_depth + delta;
  if (at2 + delta > 0xFFFFFFF0ull) return -3;
  u32* cnt = d_small + kSmallFin;
  u32* ncnt = d_small + kSmallFinNext;
  const int window = fin_window_blk, group = fin_group_blk;   // the local list's groups are as large as the finisher's passes left them
  const u32 stride = (u32)(window - group);
  BWTC_HIP_TRY(hipMemsetAsync(ncnt, 0, kFinRegions * 4, st));
  BWTC_HIP_TRY(hipMemsetAsync(cnt + 1, 0, 4, st));
  const u32 grid = local_rg.wfirst[local_rg.nreg];
  FinOut ob;
  {
    u32 room[kFinRegions] = {};
    const u32 nchunks = ceil_div(grid, kFinChunk);
    for (u32 c = 0; c < nchunks; ++c) {
      const u32 wgs = std::min(kFinChunk, grid - c * kFinChunk);
      room[c % kFinRegions] += wgs * stride + (u32)group;
    }
    u32 a2 = 0;
    for (u32 r = 0; r < kFinRegions; ++r) { ob.base[r] = a2; a2 += room[r]; }
    if ((u64)a2 > cap + cap / 128 + 65536) return -3;
  }
  FinShallow none{d_parkS, d_parkHP, cnt + 4, 0u};
  FinRank rk{d_rank, (u32)at, (u32)at2, d_US, d_UR};
#define BWTC_LOCAL(G, E) hipLaunchKernelGGL((k_finish<G, E, 2, true>), dim3(grid), dim3(kFinTPB), 0, st, local_list(local_home), local_rg, (const u8*)d_T, n, \
                                            local_list(1 - local_home), ob, ncnt, d_parkS, d_parkHP, d_LC1 /* never written: no group is too large */, cnt + 1, none, d_SA, re, rk)
  if (window <= 1024) { if (group <= 256) BWTC_LOCAL(256, kFinE1); else BWTC_LOCAL(512, kFinE1); }
  else if (group <= 256) BWTC_LOCAL(256, kFinE2);
  else if (group <= 512) BWTC_LOCAL(512, kFinE2);
  else BWTC_LOCAL(1024, kFinE2);
#undef BWTC_LOCAL
  BWTC_HIP_TRY(hipMemcpyAsync(h_small + kSmallFin + 1, cnt + 1, 4, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(hipMemcpyAsync(h_small + kSmallFinNext, ncnt, kFinRegions * 4, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(wait());
  if (h_small[kSmallFin + 1]) return -3;                  // a group above the finisher's bound in a list that has none
  stats.finisher_entries += local_m;
  stats.active_sum += local_m;
  stats.route |= 64u;
  stats.alg_bytes += (u64)local_m * (14 * 4 / 3 + 8 + 8 + 5 + 8 + 4);
  local_upd = local_rg;
  local_pending = true;
  FinRegions nx;
  std::memset(&nx, 0, sizeof nx);
  u32 total = 0;
  for (u32 r = 0; r < kFinRegions; ++r) {
    const u32 c = h_small[kSmallFinNext + r];
    if (!c) continue;
    nx.ebase[nx.nreg] = ob.base[r]; nx.ecount[nx.nreg] = c;
    nx.wfirst[nx.nreg + 1] = nx.wfirst[nx.nreg] + ceil_div(c, stride);
    ++nx.nreg;
    total += c;
  }
  if (nx.nreg == 0) nx.nreg = 1;
  if (std::getenv("BWTC_HIP_DEBUG"))
    std::fprintf(stderr, "local pass: %u entries at depth %u (+ 2 x %llu) -> %u still tied\n", local_m, local_depth, (unsigned long long)delta, total);
  local_rg = nx;
  local_m = total;
  local_home = 1 - local_home;
  local_depth = (u32)std::min<u64>(at2 + delta, 0xFFFFFFF0ull);
  return 0;
}

// A raw list becomes a list the rounds understand: sorted by slot (groups contiguous, a positional slot
// array), key = head slot | character << 56, value = suffix.  S / HP must be one of (d_V0 | d_V1) and one
// of (d_R1 | d_R2); d_W0 / d_W1 and d_G0 receive the list.
int BwtEngine::dress_list(u32 n, u32 total, u32* S, u64* HP, RankBuffers& rb, RankResult* res, u32 holes) {
  hipStream_t st = stream;
  u64* hp_other = HP == static_cast<u64*>(d_R1) ? static_cast<u64*>(d_R2) : static_cast<u64*>(d_R1);
  u32* s_other = S == d_V0 ? d_V1 : d_V0;
  u64* hp_sorted = nullptr; u32* s_sorted = nullptr;
  sort_pairs<u64>(HP, hp_other, S, s_other, total, bit_width_u64(n ? n - 1 : 0), &hp_sorted, &s_sorted, false);
  u64* fkey = reinterpret_cast<u64*>(d_W0);              // d_W0 and d_W1 are neighbours in the arena: 8 * cap bytes
  total -= holes;                                        // (entries that left for the shallow list: all ones, sorted to the end)
  hipLaunchKernelGGL(k_bridge_dress, dim3(ceil_div(total, 256)), dim3(256), 0, st, (const u64*)hp_sorted, (const u32*)s_sorted,
                     total, (const u8*)d_T, fkey, d_G0);
  res->m = total; res->groups = 0;
  res->ks = fkey; res->rec_other = d_R1;
  res->vs = s_sorted; res->v_other = s_sorted == d_V0 ? d_V1 : d_V0;
  rb.aglob_next = d_G0;
  return 0;
}

// rank[] for everybody, late: rank[SA[slot]] = slot for the finished suffixes, any rank for the list's
// (the ranking step that follows writes theirs).  pairs / tmp: two free 8 * cap-byte regions.
void BwtEngine::complete_ranks(u32 n, const u32* list_sfx, const u32* list_slot, u32 m, void* pairs_region, void* tmp_region) {
  hipStream_t st = stream;
  u32* pairs = static_cast<u32*>(pairs_region);
  hipLaunchKernelGGL(k_bridge_pairs_all, dim3(ceil_div(n, 1024)), dim3(256), 0, st, (const u32*)d_SA, n, pairs, pairs + cap);
  if (m) hipLaunchKernelGGL(k_bridge_pairs_fix, dim3(ceil_div(m, 256)), dim3(256), 0, st, list_sfx, list_slot, m, pairs, pairs + cap);
  for (u32 r = 0; local_m && r < local_rg.nreg; ++r) {
    if (!local_rg.ecount[r]) continue;
    const FinList h = local_list(local_home);
    FinList part{h.S + local_rg.ebase[r], h.P + local_rg.ebase[r], h.H + local_rg.ebase[r], h.C + local_rg.ebase[r]};
    hipLaunchKernelGGL(k_bridge_pairs_fix_list, dim3(ceil_div(local_rg.ecount[r], 256)), dim3(256), 0, st, part, local_rg.ecount[r], pairs, pairs + cap);
  }
  if (parked) hipLaunchKernelGGL(k_bridge_pairs_fix_parked, dim3(ceil_div(parked, 256)), dim3(256), 0, st, (const u32*)d_parkS, (const u64*)d_parkHP, parked, pairs, pairs + cap);
  scatter_rank_pairs(pairs, static_cast<u32*>(tmp_region), n, n);
}

// Loads `ncopy` bytes of d_src into d_T (reversed or not), zero-fills up to n + padding and
// returns the histogram of T[0..n-1] (the implicit zeros included) in hist_T.
int BwtEngine::load_text(const u8* d_src, u32 ncopy, u32 n, bool reverse, u32* hist_T) {
  hipStream_t st = stream;
  BWTC_HIP_TRY(hipMemsetAsync(d_small, 0, 1024 * 4, st));
  stats.alg_bytes = 2ull * n;                            // k_load_hist: the block read, T written
  const u32 padded = (u32)((((u64)n + kTextPad + 15) / 16) * 16);
  const int aligned = (reinterpret_cast<uintptr_t>(d_src) & 15u) == 0;
  hipLaunchKernelGGL(k_load_hist, dim3(std::min<u32>(ceil_div(padded, kLoadTile), 2048u)), dim3(kLoadTPB), 0, st, d_src,
                     d_T, ncopy, padded, reverse ? 1 : 0, aligned, d_small + kSmallFreqs);
  BWTC_HIP_TRY(hipMemcpyAsync(h_small + kSmallFreqs, d_small + kSmallFreqs, 256 * 4,
                              hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(wait());
  for (int c = 0; c < 256; ++c) hist_T[c] = h_small[kSmallFreqs + c];
  hist_T[0] += n - ncopy;
  return 0;
}

int BwtEngine::transform(const u8* d_src, u8* d_dst, u32 size, bool raw, u32* lf, u32 n_lf,
                         u32* freqs) {
  BwtEngine::ScanScope scan_scope(*this);
  hipStream_t st = stream;
  // block mode: size source bytes, n = size + 1 suffixes.  raw mode: `size` is the length
  // of T including the caller's sentinel, so the source bytes that count are size - 1.
  const u32 n = raw ? size : size + 1u;
  const u32 nsrc = n - 1u;
  if (n_lf == 0 || n_lf > 256 || !lf) return -1;
  if ((u64)n > cap) return -1;
  BWTC_HIP_TRY(hipSetDevice(device));
  BWTC_HIP_TRY(hipEventRecord(ev_begin, st));
  wait_seq = 0;                                          // the block's waits come in the same order for every block of a stream
  u32 hist_T[256];
  int rc = load_text(d_src, nsrc, n, !raw, hist_T);
  if (rc) return rc;
  if (freqs) for (int c = 0; c < 256; ++c) freqs[c] += h_small[kSmallFreqs + c];
  // Blocks up to 256 MiB: the transform's bytes are written by the ranking kernels, straight
  // into the caller's buffer (d_src was consumed by load_text, so d_dst may alias it).
  // Larger blocks (a round's key then needs the character's bits): suffix array + gather.
  // The transform's bytes leave from the ranking kernels (a list whose keys have no room for the carried
  // character -- more than 56 bits, only possible above 256 MiB -- reads T[s-1] for what it finishes).
  const bool emit = !no_emit;
  if (emit) {
    EmitTarget em;
    em.out = d_dst;
    em.out_n = raw ? n : size;
    em.n_lf = n_lf;
    rc = suffix_sort(n, hist_T, h_small[kSmallFreqs] == 0, &em);
    if (rc) return rc;
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, st, d_dst, d_T, d_rank,
                       d_small + kSmallLf, n_lf, n, d_small + kSmallPidx, raw ? 1 : 0,
                       d_small + kSmallLastChar, lf_noted ? 1 : 0);
  } else {
    rc = suffix_sort(n, hist_T, h_small[kSmallFreqs] == 0, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(k_bwt_gather, dim3(ceil_div(ceil_div(n, 4), 256)), dim3(256), 0, st, d_SA,
                       d_T, d_out, n, d_small + kSmallPidx);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, st, d_out, d_T, d_rank,
                       d_small + kSmallLf, n_lf, n, d_small + kSmallPidx, raw ? 1 : 0,
                       (const u32*)nullptr);
    if (raw ? n : size)
      BWTC_HIP_TRY(hipMemcpyAsync(d_dst, d_out, raw ? n : size, hipMemcpyDeviceToDevice, st));
  }
  BWTC_HIP_TRY(hipMemcpyAsync(h_small + kSmallLf, d_small + kSmallLf, 264 * 4, hipMemcpyDeviceToHost, st));
  BWTC_HIP_TRY(hipEventRecord(ev_end, st));
  BWTC_HIP_TRY(wait());
  BWTC_HIP_TRY(hipGetLastError());
  BWTC_HIP_TRY(take_sticky_error());
  if (radix_fault_take()) { std::fprintf(stderr, "bwtc_hip: a radix pass's digit table and its items disagreed (checked build)\n"); return -3; }
  for (u32 k = 0; k < n_lf; ++k) lf[k] = h_small[kSmallLf + k];
  probe.harvest();
  stats.n = n;
  BWTC_HIP_TRY(hipEventElapsedTime(&stats.ms_total, ev_begin, ev_end));
  stats.ms_sort = 0.f;
  for (int i = 0; i + 1 < n_sort_events; i += 2) {
    float ms = 0.f;
    BWTC_HIP_TRY(hipEventElapsedTime(&ms, ev_sort[i], ev_sort[i + 1]));
    stats.ms_sort += ms;
  }
  return 0;
}

}  // namespace bwtc_hip
