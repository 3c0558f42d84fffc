// sa_kernels.hpp -- the non-sort kernels of the prefix-doubling loop.
//
// Reference loops each kernel stands in for (paths under /root/reference/include):
//   char_hist_kernel        alphabet.hpp:49-59        256-bin byte histogram
//   key_pairs_kernel        kmer.hpp:119-177 + shifting.hpp:33-122 (the (B1,B2) window of round 1, packed)
//   rebucket_first_kernel   suffix_array.hpp:1353-1396 (k-mer LCP) + bucketing.hpp:57-123, 21-53
//   isa_scatter_kernel      bulk_permute.hpp:14-73     ISA[SA[i]] = B[i]
//   compact_active_kernel   suffix_array.hpp:925-965   get_active
//   gather_keys_kernel      suffix_array.hpp:972-996   sparse_get_b2 (B2 = ISA[SA+h])
//   rebucket_refine_kernel  suffix_array.hpp:1092-1157 + :1444-1508 (new ids, LCP = h + range min)
//   pyramid_level_kernel    rmq.hpp:87-179 / par_rmq.hpp:199-332 (range-minimum structure)
//   isa_finalize_kernel     suffix_array.hpp:460-464   ISA -= 1
#pragma once
#ifndef RB_WAVES_ATTR
#define RB_WAVES_ATTR          // (tools/experiments: __attribute__((amdgpu_waves_per_eu(6, 6))) -- three workgroups of rebucket_first_kernel per CU)
#endif
#ifndef RB_ABLATE
#define RB_ABLATE 0          // (tools/experiments/rb_ablate.sh: output streams of rebucket_first_kernel left out to time them: 1 LCP, 2 SA, 4 the fused partition level, 8 its stores; results are wrong then)
#endif
#include <type_traits>
#include "dev_common.hpp"

namespace psacx {

constexpr int RADIX_P = 256;    // fan-out of one destination-partition pass

struct CodeTable { uint16_t c[256]; };

// OR / AND over all sort keys, accumulated by the kernels that produce the keys: bits where
// both agree are constant, so a radix digit made only of such bits needs no pass.
// summary[0] = OR(k1), [1] = AND(k1), [2] = OR(k2), [3] = AND(k2).
// Each workgroup stores its four partial words (no atomics: millions of waves updating four
// addresses serialise, and a cached copy of those words never refreshes); summary_reduce_kernel
// folds the partials afterwards.
template <typename T>
__device__ __forceinline__ void key_summary_add(unsigned long long* partials, T or1, T and1, T or2, T and2) {
    __shared__ unsigned long long red[4][16];
    or1 = wave_reduce<T>(or1, OpOr()); and1 = wave_reduce<T>(and1, OpAnd());
    or2 = wave_reduce<T>(or2, OpOr()); and2 = wave_reduce<T>(and2, OpAnd());
    const unsigned wave = threadIdx.x / WAVE, nw = (blockDim.x + WAVE - 1) / WAVE;
    const unsigned long long hi = ~(unsigned long long)(T)~(T)0;
    if (lane_id() == 0) {
        red[0][wave] = (unsigned long long)or1; red[1][wave] = (unsigned long long)and1 | hi;
        red[2][wave] = (unsigned long long)or2; red[3][wave] = (unsigned long long)and2 | hi;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        unsigned long long v = red[threadIdx.x][0];
        for (unsigned w = 1; w < nw; ++w) v = (threadIdx.x & 1) ? (v & red[threadIdx.x][w]) : (v | red[threadIdx.x][w]);
        partials[(size_t)blockIdx.x * 4 + threadIdx.x] = v;
    }
}

template <int TAG>
__global__ void summary_reduce_kernel(const unsigned long long* __restrict__ partials, unsigned nblocks,
                                      unsigned long long* __restrict__ summary) {
    unsigned long long o1 = 0, a1 = ~0ull, o2 = 0, a2 = ~0ull;
    for (unsigned b = threadIdx.x; b < nblocks; b += blockDim.x) {
        o1 |= partials[(size_t)b * 4]; a1 &= partials[(size_t)b * 4 + 1];
        o2 |= partials[(size_t)b * 4 + 2]; a2 &= partials[(size_t)b * 4 + 3];
    }
    __shared__ unsigned long long red[4][16];
    o1 = wave_reduce<uint64_t>(o1, OpOr()); a1 = wave_reduce<uint64_t>(a1, OpAnd());
    o2 = wave_reduce<uint64_t>(o2, OpOr()); a2 = wave_reduce<uint64_t>(a2, OpAnd());
    const unsigned wave = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
    if (lane_id() == 0) { red[0][wave] = o1; red[1][wave] = a1; red[2][wave] = o2; red[3][wave] = a2; }
    __syncthreads();
    if (threadIdx.x < 4) {
        unsigned long long v = red[threadIdx.x][0];
        for (unsigned w = 1; w < nw; ++w) v = (threadIdx.x & 1) ? (v & red[threadIdx.x][w]) : (v | red[threadIdx.x][w]);
        summary[threadIdx.x] = v;
    }
}

template <typename T>
__global__ void key_summary_kernel(const T* __restrict__ k1, const T* __restrict__ k2, uint64_t n,
                                   unsigned long long* __restrict__ summary) {
    T o1 = 0, a1 = ~(T)0, o2 = 0, a2 = ~(T)0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T x = k1[i], y = k2 ? k2[i] : (T)0;        // (two-word records have no second key word)
        o1 |= x; a1 &= x; o2 |= y; a2 &= y;
    }
    key_summary_add<T>(summary, o1, a1, o2, a2);
}

// ------------------------------------------------------------------ K1
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void char_hist_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                          unsigned long long* __restrict__ hist) {
    // 32 private histograms per workgroup, 257 words apart: the lanes of a wave that meet the same symbol (four
    // symbols for DNA) hit 32 different counters in 32 different banks instead of serialising on one
    constexpr int COPIES = 32, PITCH = 257;
    __shared__ unsigned lh[COPIES * PITCH];
    for (int i = threadIdx.x; i < COPIES * PITCH; i += BLOCK) lh[i] = 0;
    __syncthreads();
    const uint64_t nvec = ((uintptr_t)text % 16 == 0) ? n / 16 : 0;
    const uint4* tv = reinterpret_cast<const uint4*>(text);
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    unsigned* my = lh + (threadIdx.x & (COPIES - 1)) * PITCH;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < nvec; i += stride) {
        const uint4 v = tv[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            atomicAdd(&my[w[q] & 255u], 1u);
            atomicAdd(&my[(w[q] >> 8) & 255u], 1u);
            atomicAdd(&my[(w[q] >> 16) & 255u], 1u);
            atomicAdd(&my[w[q] >> 24], 1u);
        }
    }
    for (uint64_t i = nvec * 16 + (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride)
        atomicAdd(&my[text[i]], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += BLOCK) {
        unsigned c = 0;
#pragma unroll 8
        for (int k = 0; k < COPIES; ++k) c += lh[k * PITCH + i];
        if (c) atomicAdd(&hist[i], (unsigned long long)c);
    }
}

// ------------------------------------------------------------------ K2 (+K3 for round 1)
// First-round sort keys.  psac sorts by (B1,B2) = (k-mer at i, k-mer at i+k) with
// l = ceil(log2(sigma+1)) bits per character, code 0 being the end marker
// (kmer.hpp:119-177, shifting.hpp:33-122).  The same 2k-character window is packed
// here with lc = ceil(log2(sigma)) bits per character (codes 0..sigma-1, no code for
// the end marker): word 1 holds the first c1 characters, word 2 the remaining c2.
// That shortens the radix key (DNA, 32-bit words: 60 -> 40 bits).  The end marker is
// recovered from the record ORDER instead: the `spec` suffixes shorter than 2k
// characters come first in the input, shortest first, so the stable LSD sort leaves
// them in front of every suffix with the same packed key, in the order psac's
// zero-padded k-mers would give; rebucket_first_kernel caps their LCP by their length.
struct KeyShape {
    unsigned lc;        // bits per character in the packed key
    unsigned c1, c2;    // characters in word 1 / word 2, c1 + c2 = 2k
    uint64_t spec;      // number of suffixes shorter than 2k (= min(2k - 1, n))
};

// record j  <->  suffix start: the `spec` short suffixes first (shortest first), then 0, 1, 2, ...
__device__ __forceinline__ uint64_t record_suffix(uint64_t j, uint64_t spec, uint64_t n) {
    return j < spec ? n - 1 - j : j - spec;
}

//
// GSA (string sets, kmer.hpp:269-355): the codes are psac's 1..sigma with lc = l bits, 0 is the
// end marker, spec = 0, and every window is cut at the end of its string: slen[i] = characters
// from position i to the end of the string holding it.
//
// HIST: the tile shape equals the radix scatter tile of the sort that follows, and the digit histogram of
// word 1 at bit hist_shift is left in tile_hist[tile][256] -- the first pass then needs no
// radix_tile_hist_kernel (one read of the key word saved).
template <typename T, int BLOCK, int ITEMS, bool GSA = false, bool HIST = false>
__global__ __launch_bounds__(BLOCK) void key_pairs_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                          uint64_t n_text, CodeTable tab, KeyShape ks,
                                                          T* __restrict__ C1, T* __restrict__ C2,
                                                          unsigned long long* __restrict__ summary,
                                                          const T* __restrict__ slen = nullptr,
                                                          unsigned* __restrict__ tile_hist = nullptr, int hist_shift = 0) {
    constexpr int TILE = BLOCK * ITEMS;
    constexpr int HALO = 2 * 64 + 8;             // 2k <= 128 always
    // code i of the tile lives at SW(i): every 8 codes are followed by a 4-byte gap, so the lanes of a
    // wave, which walk windows that start 8 codes apart, read from different LDS banks
    // (stride 20 bytes instead of 16: 5 t mod 32 is a permutation of the banks)
#define SW(i) ((i) + (((i) >> 3) << 1))
    __shared__ uint16_t codes[SW(TILE + HALO) + 8];
    __shared__ uint16_t ctab[256];
    __shared__ unsigned dh[HIST ? 4 * RADIX : 1];
    for (int i = threadIdx.x; i < 256; i += BLOCK) ctab[i] = tab.c[i];
    if (HIST) for (int i = threadIdx.x; i < 4 * RADIX; i += BLOCK) dh[HIST ? i : 0] = 0;
    __syncthreads();
    const unsigned two_k = ks.c1 + ks.c2;
    const uint64_t base = (uint64_t)blockIdx.x * TILE;                 // first record of the tile
    const uint64_t i_lo = (base > ks.spec ? base : ks.spec) - ks.spec; // first regular suffix of the tile
    const unsigned need = TILE + two_k;
    // the text window of the tile, 16 bytes per lane where a whole aligned chunk lies inside the text
    // (the text pointer itself is 16-byte aligned: device allocations are), single bytes at the edges
    {
        const uint64_t a_lo = i_lo & ~15ull;                       // aligned start at or before the window
        const unsigned lead = (unsigned)(i_lo - a_lo);
        const bool aligned_ptr = (reinterpret_cast<uintptr_t>(text) & 15u) == 0;
        for (unsigned v = threadIdx.x * 16u; v < need + lead; v += BLOCK * 16u) {
            const uint64_t g0 = a_lo + v;
            if (aligned_ptr && g0 + 16 <= n_text) {
                const uint4 x = *reinterpret_cast<const uint4*>(text + g0);
                const unsigned wds[4] = {x.x, x.y, x.z, x.w};
                // (all sixteen table look-ups first, then the stores: written as one statement per byte, every look-up waited for
                //  the store before it -- both are LDS accesses the compiler cannot tell apart)
                uint16_t cd[16];
#pragma unroll
                for (int b = 0; b < 16; ++b) cd[b] = ctab[(wds[b >> 2] >> ((b & 3) * 8)) & 255u];
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const int i = (int)v + b - (int)lead;          // window index of this byte
                    if (i >= 0 && (unsigned)i < need) codes[SW(i)] = cd[b];
                }
            } else {
#pragma unroll 1
                for (int b = 0; b < 16; ++b) {
                    const int i = (int)v + b - (int)lead;
                    if (i >= 0 && (unsigned)i < need) codes[SW(i)] = (g0 + b) < n_text ? ctab[text[g0 + b]] : (uint16_t)0;
                }
            }
        }
    }
    __syncthreads();
    const unsigned lc = ks.lc;
    const T mask1 = (ks.c1 * lc >= sizeof(T) * 8) ? ~(T)0 : (T)(((T)1 << (ks.c1 * lc)) - 1);
    const T mask2 = (ks.c2 * lc >= sizeof(T) * 8) ? ~(T)0 : (T)(((T)1 << (ks.c2 * lc)) - 1);
    const uint64_t j0 = base + (uint64_t)threadIdx.x * ITEMS;
    T o1[ITEMS], o2[ITEMS];
    if (j0 >= ks.spec) {
        // ITEMS consecutive regular suffixes: rolling pack out of LDS
        const unsigned q = (unsigned)(j0 - ks.spec - i_lo);
        T w1 = 0, w2 = 0;
        for (unsigned t = 0; t + 1 < ks.c1; ++t) w1 = (T)(w1 << lc) | (T)codes[SW(q + t)];
        const bool want2 = ks.c2 != 0 && C2 != nullptr;          // word 2 is dropped when the first round sorts on word 1 only
        for (unsigned t = 0; want2 && t + 1 < ks.c2; ++t) w2 = (T)(w2 << lc) | (T)codes[SW(q + ks.c1 + t)];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            w1 = ((T)(w1 << lc) | (T)codes[SW(q + j + ks.c1 - 1)]) & mask1;
            if (want2) w2 = ((T)(w2 << lc) | (T)codes[SW(q + j + two_k - 1)]) & mask2;
            o1[j] = w1; o2[j] = w2;
        }
        if (GSA) {
            T len[ITEMS];
            load_run<T, ITEMS>(slen, j0, n, len, (T)1);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                // keep the first min(len, c1) characters of word 1 and the first min(len - c1, c2) of word 2
                const unsigned k1 = len[j] < (T)ks.c1 ? (unsigned)len[j] : ks.c1;
                const unsigned k2 = len[j] <= (T)ks.c1 ? 0u : (len[j] - (T)ks.c1 < (T)ks.c2 ? (unsigned)(len[j] - (T)ks.c1) : ks.c2);
                const unsigned d1 = (ks.c1 - k1) * lc, d2 = (ks.c2 - k2) * lc;
                o1[j] = d1 >= sizeof(T) * 8 ? (T)0 : (T)((T)(o1[j] >> d1) << d1);
                o2[j] = d2 >= sizeof(T) * 8 ? (T)0 : (T)((T)(o2[j] >> d2) << d2);
            }
        }
    } else {
        // the few threads whose records include short suffixes: pack each record from the text
#pragma unroll 1
        for (int j = 0; j < ITEMS; ++j) {
            const uint64_t rec = j0 + j;
            T w1 = 0, w2 = 0;
            if (rec < n) {
                const uint64_t i = record_suffix(rec, ks.spec, n);
                for (unsigned t = 0; t < ks.c1; ++t) w1 = (T)(w1 << lc) | (T)((i + t < n_text) ? tab.c[text[i + t]] : 0);
                for (unsigned t = 0; t < ks.c2; ++t) w2 = (T)(w2 << lc) | (T)((i + ks.c1 + t < n_text) ? tab.c[text[i + ks.c1 + t]] : 0);
            }
            o1[j] = w1; o2[j] = w2;
        }
    }
    store_run<T, ITEMS>(C1, j0, n, o1);
    if (C2) store_run<T, ITEMS>(C2, j0, n, o2);      // (not kept when the first round sorts on word 1 only)
    if (HIST) {
        unsigned* my = dh + ((threadIdx.x / WAVE) & 3) * RADIX;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) wave_hist_add(my, (unsigned)(o1[j] >> hist_shift) & (RADIX - 1), j0 + j < n);
        __syncthreads();
        for (int d = threadIdx.x; d < RADIX; d += BLOCK)
            tile_hist[(uint64_t)blockIdx.x * RADIX + d] = dh[d] + dh[RADIX + d] + dh[2 * RADIX + d] + dh[3 * RADIX + d];
    }
    T so1 = 0, sa1 = ~(T)0, so2 = 0, sa2 = ~(T)0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (j0 + j < n) { so1 |= o1[j]; sa1 &= o1[j]; so2 |= o2[j]; sa2 &= o2[j]; }
    key_summary_add<T>(summary, so1, sa1, so2, sa2);
#undef SW
}

// ------------------------------------------------------------------ keys straight into the one-word prefix sort
// (engine.hpp: prefix_sort_1w, fused front end).  key_pairs_kernel writes word 1 of every record and the pass on the top digit
// reads it back: 16 bytes per record that carry no information the text does not hold.  Here the tile histograms of the top
// digit come from the text itself (the top digit of word 1 is the first ceil(8 / lc) characters of the window) and the pass on
// the top digit computes word 1 of its tile in registers.
// top_digit_hist_kernel: tile_hist[tile][256] of the top eight bits of word 1 of the records of every tile of BLOCK * ITEMS records.
template <typename T, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void top_digit_hist_kernel(const uint8_t* __restrict__ text, uint64_t n, uint64_t n_text, CodeTable tab, KeyShape ks,
                                                               unsigned* __restrict__ tile_hist) {
    constexpr int TILE = BLOCK * ITEMS;
    __shared__ uint16_t ctab[256];
    __shared__ unsigned lh[4][RADIX];
    for (int i = threadIdx.x; i < 256; i += BLOCK) ctab[i] = tab.c[i];
    for (int i = threadIdx.x; i < 4 * RADIX; i += BLOCK) (&lh[0][0])[i] = 0;
    __syncthreads();
    const unsigned lc = ks.lc;
    const unsigned nch = (8 + lc - 1) / lc < ks.c1 ? (8 + lc - 1) / lc : ks.c1;     // characters that reach into the top digit
    const unsigned down = nch * lc > 8 ? nch * lc - 8 : 0, up = nch * lc < 8 ? 8 - nch * lc : 0;
    unsigned* my = lh[(threadIdx.x / WAVE) & 3];
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    static_assert(ITEMS == 8, "a thread takes eight consecutive records: their characters lie in three aligned 8-byte words");
    const uint64_t rec0 = base + (uint64_t)threadIdx.x * ITEMS;
    unsigned dig[ITEMS];
    const uint64_t i0 = rec0 - ks.spec;                          // (meaningful when rec0 >= spec)
    if (rec0 >= ks.spec && rec0 + ITEMS <= n && (i0 & ~7ull) + 24 <= n_text && (reinterpret_cast<uintptr_t>(text) & 7u) == 0) {
        const uint64_t* __restrict__ tw = reinterpret_cast<const uint64_t*>(text + (i0 & ~7ull));
        const uint64_t w0 = tw[0], w1 = tw[1], w2 = tw[2];
        const unsigned sh8 = (unsigned)(i0 & 7ull) * 8;
        // bytes i0 .. i0 + 15 in two words
        const uint64_t lo = sh8 ? (w0 >> sh8) | (w1 << (64 - sh8)) : w0, hi = sh8 ? (w1 >> sh8) | (w2 << (64 - sh8)) : w1;
        const unsigned tmask = nch * lc >= 32 ? ~0u : ((1u << (nch * lc)) - 1u);
        unsigned t = 0;
        for (unsigned q = 0; q + 1 < nch; ++q) t = (t << lc) | (unsigned)ctab[(lo >> (8 * q)) & 255u];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const unsigned k = (unsigned)j + nch - 1;                // the character that enters the window of record j
            const unsigned ch = (unsigned)((k < 8 ? lo >> (8 * k) : hi >> (8 * (k - 8))) & 255u);
            t = ((t << lc) | (unsigned)ctab[ch]) & tmask;
            dig[j] = ((t >> down) << up) & 255u;
        }
    } else {
#pragma unroll 1
        for (int j = 0; j < ITEMS; ++j) {
            const uint64_t rec = rec0 + j;
            unsigned t = 0;
            if (rec < n) {
                const uint64_t i = record_suffix(rec, ks.spec, n);
                for (unsigned q = 0; q < nch; ++q) t = (t << lc) | (unsigned)((i + q < n_text) ? ctab[text[i + q]] : (uint16_t)0);
            }
            dig[j] = ((t >> down) << up) & 255u;
        }
    }
    // (plain LDS atomics: the digits of a text that reaches this kernel vary -- a text whose prefixes repeat was turned away by the probe --
    //  and wave_hist_add's two ballots per record were a fifth of it: keys phase 24.5 -> 23.8 ms at 2^32)
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) if (rec0 + j < n) atomicAdd(&my[dig[j]], 1u);
    __syncthreads();
    for (int d = threadIdx.x; d < RADIX; d += BLOCK) tile_hist[(uint64_t)blockIdx.x * RADIX + d] = lh[0][d] + lh[1][d] + lh[2][d] + lh[3][d];
}

// cnt characters of the text from position q0 on as codes of lc bits each, the first on top (cnt * lc <= 64; zeros beyond the end of the
// text).  Away from the end the characters are read as 8-byte pieces, four asked for at a time, instead of one dependent byte load
// after the other (a text with interspersed repeats sends a quarter of its suffixes through gather_prefix_ties_kernel: 42 byte loads
// per suffix were 179 ms at 2^30 characters).
template <typename T>
__device__ __forceinline__ T packed_chars(const uint8_t* __restrict__ text, uint64_t n_text, const uint16_t* ctab, unsigned lc, uint64_t q0, unsigned cnt) {
    T w = 0;
    if (q0 + cnt + 32 <= n_text) {
        for (unsigned t0 = 0; t0 < cnt; t0 += 32) {
            uint64_t x[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { x[c] = 0; if (t0 + 8u * c < cnt) __builtin_memcpy(&x[c], text + q0 + t0 + 8u * c, 8); }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (t0 + 8u * c + i < cnt) w = (T)(w << lc) | (T)ctab[(unsigned)(x[c] >> (8 * i)) & 255u];
            }
        }
        return w;
    }
    for (unsigned t = 0; t < cnt; ++t) {
        const uint64_t q = q0 + t;
        w = (T)(w << lc) | (T)(q < n_text ? ctab[text[q]] : (uint16_t)0);
    }
    return w;
}
// prefix_dup_probe_kernel: does the text repeat itself massively?  Every `stride`-th suffix puts the sorted prefix of its word 1
// into an open-addressing table (zeroed, `slots` a power of two); dups counts the samples that met their own prefix there.
// Random text: none.  A tandem repeat: nearly all.  (The one-word prefix sort drops the bits of word 1 below the prefix; when
// most suffixes tie on the prefix they all read their windows from the text again, which costs more than it saves.)
template <typename T>
__global__ void prefix_dup_probe_kernel(const uint8_t* __restrict__ text, uint64_t n_text, CodeTable tab, KeyShape ks, unsigned lo1, uint64_t stride,
                                        uint64_t samples, unsigned long long* __restrict__ table, uint64_t slots, unsigned long long* __restrict__ dups) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= samples) return;
    const uint64_t i = s * stride;
    const T w1 = packed_chars<T>(text, n_text, tab.c, ks.lc, i, ks.c1);
    const unsigned long long key = ((unsigned long long)w1 >> lo1) + 1ull;
    uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> 20;
    for (int probe = 0; probe < 16; ++probe, ++h) {
        const unsigned long long old = atomicCAS(&table[h & (slots - 1)], 0ull, key);
        if (old == 0ull) return;
        if (old == key) { atomicAdd(dups, 1ull); return; }
    }
}

// key_scatter1w_kernel: the pass on the top digit with word 1 computed on the spot.  The text window of the tile is staged ONCE as a
// stream of packed codes (lc bits per character, first character on top, 32-bit words): a thread turns 16 bytes into 16 lc bits and
// stores them as lc half-words.  Word 1 of any record of the tile is then 64 bits of that stream at bit offset lc x (its place): three
// words read and two shifts -- every lane cuts out the records of its own places in the striped order radix_scatter_tile ranks in, so the
// words never pass through LDS again.  (Until round 5 the codes were staged as 16-bit entries, packed eight at a time into groups, put
// together per thread by a rolling window and handed to the ranking through the stage: 3.9 G of the kernel's 10.6 G vector
// instructions at 2^32 records, and the kernel is bound by vector issue: profiles/r4d_sq_counters_one_gpu_kernels.txt, tools/ubench_valu.hip.)
// The pass stays stable: the short suffixes at the head of the input keep their places in front.
template <int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK, 6) void key_scatter1w_kernel(const uint8_t* __restrict__ text, uint64_t n, uint64_t n_text, CodeTable tab, KeyShape ks,
                                                                 uint64_t* __restrict__ out, int shift, const unsigned long long* __restrict__ digit_base,
                                                                 const unsigned* __restrict__ tile_excl, const unsigned long long* __restrict__ slab_excl,
                                                                 unsigned* __restrict__ tile_counter, unsigned chunk, unsigned slab_tiles, unsigned lo1,
                                                                 uint64_t voff = 0, uint8_t* __restrict__ dnext = nullptr, int dnext_shift = 0, uint64_t out_pad = 0) {
    // lo1: bits of word 1 below the sorted prefix | width of the payload field << 16 (radix.hpp: ONEW_MAKE; 0 = 32)
    // voff: added to the suffix a record stands for (a rank's block of a distributed text)
    // dnext: the digit the first bucket pass sorts on, one byte per record (radix.hpp: radix_scatter_tile)
    typedef uint64_t T;
    constexpr int TILE = BLOCK * ITEMS;
    constexpr int NW = BLOCK / WAVE;
    constexpr int HALO = 2 * 64 + 8 + 16;
    __shared__ ScatterShared<T, TILE, NW> sh;
    __shared__ uint16_t ctab[256];
    static_assert(sizeof(uint32_t) * ((TILE + HALO) * 8 / 32 + 4) <= sizeof(sh.stage), "the packed codes of the tile live in the stage until the words are cut out");
    uint32_t* const stream = reinterpret_cast<uint32_t*>(sh.stage);
    uint16_t* const stream16 = reinterpret_cast<uint16_t*>(sh.stage);
    if (threadIdx.x == 0) sh.s_tile = tile_counter ? claim_tile(tile_counter, gridDim.x, chunk) : blockIdx.x;
    for (int i = threadIdx.x; i < NW * RADIX; i += BLOCK) sh.wcnt[i] = 0;
    for (int i = threadIdx.x; i < 256; i += BLOCK) ctab[i] = tab.c[i];
    __syncthreads();
    const unsigned tile = sh.s_tile;
    const uint64_t base = (uint64_t)tile * TILE;
    const unsigned lc = ks.lc;
    // the text from the first regular suffix of the tile on (aligned down to 16 bytes): character q of the stream = text[a_lo + q]
    const uint64_t i_lo = (base > ks.spec ? base : ks.spec) - ks.spec;
    const uint64_t a_lo = i_lo & ~15ull;
    const unsigned need = TILE + ks.c1 + 16;                   // characters the windows of the tile reach over (lead included)
    {
        const bool aligned_ptr = (reinterpret_cast<uintptr_t>(text) & 15u) == 0;
        for (unsigned v = threadIdx.x; v * 16u < need; v += BLOCK) {
            const uint64_t g0 = a_lo + (uint64_t)v * 16u;
            uint16_t cd[16];
            if (aligned_ptr && g0 + 16 <= n_text) {
                const uint4 x = *reinterpret_cast<const uint4*>(text + g0);
                const unsigned wds[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int b = 0; b < 16; ++b) cd[b] = ctab[(wds[b >> 2] >> ((b & 3) * 8)) & 255u];
            } else {
#pragma unroll
                for (int b = 0; b < 16; ++b) cd[b] = (g0 + b) < n_text ? ctab[text[g0 + b]] : (uint16_t)0;
            }
            // sixteen codes = lc half-words of the stream; half-word h of the stream is the upper half of word h / 2 when h is even
            uint32_t acc = 0;
            unsigned nb = 0, h = lc * v;
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                acc = (acc << lc) | cd[b];
                nb += lc;
                if (nb >= 16) { nb -= 16; stream16[(h++) ^ 1u] = (uint16_t)(acc >> nb); }
            }
        }
    }
    __syncthreads();
    // word 1 of the thread's records, in the order the ranking takes them: record (wave, i, lane) = base + wave * 64 * ITEMS + i * 64 + lane
    T kd[ITEMS];
    {
        const unsigned wbase = (threadIdx.x / WAVE) * (WAVE * ITEMS) + lane_id();
        const unsigned wbits = ks.c1 * lc;                     // 1 .. 64
        const long long rel = (long long)base - (long long)ks.spec - (long long)a_lo;      // stream character of record 0 of the tile (negative among the short suffixes)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const unsigned p = wbase + (unsigned)i * WAVE;
            const uint64_t rec = base + p;
            T w1 = 0;
            if (rec >= ks.spec) {
                const unsigned o = (unsigned)(rel + (long long)p) * lc;          // bit offset in the stream
                const unsigned m = o >> 5, sft = o & 31u;
                const uint64_t x = ((uint64_t)stream[m] << 32) | stream[m + 1];
                const uint64_t r = (x << sft) | (((uint64_t)stream[m + 2] << sft) >> 32);
                w1 = r >> (64u - wbits);
            } else if (rec < n) {
                // a suffix shorter than the window (the first records of the text): its characters one by one
                const uint64_t i0 = record_suffix(rec, ks.spec, n);
                for (unsigned t = 0; t < ks.c1; ++t) w1 = (T)(w1 << lc) | (T)((i0 + t < n_text) ? tab.c[text[i0 + t]] : 0);
            }
            kd[i] = rec < n ? w1 : (T)0;
        }
    }
    __syncthreads();                        // every thread has cut out its words: the stage takes the records now
    const uint64_t remain = n - base;
    if (remain >= (uint64_t)TILE)
        radix_scatter_tile<T, unsigned, BLOCK, ITEMS, true, false, false, true, 10>(sh, tile, (unsigned)TILE, nullptr, nullptr, nullptr, out, nullptr, nullptr, shift,
                                                                                   digit_base, nullptr, nullptr, nullptr, ks.spec, ks.spec ? n : (uint64_t)0, tile_excl, slab_excl, nullptr,
                                                                                   slab_tiles, voff, lo1, &kd, dnext, dnext_shift, out_pad);
    else
        radix_scatter_tile<T, unsigned, BLOCK, ITEMS, false, false, false, true, 10>(sh, tile, (unsigned)remain, nullptr, nullptr, nullptr, out, nullptr, nullptr, shift,
                                                                                    digit_base, nullptr, nullptr, nullptr, ks.spec, ks.spec ? n : (uint64_t)0, tile_excl, slab_excl, nullptr,
                                                                                    slab_tiles, voff, lo1, &kd, dnext, dnext_shift, out_pad);
}

// characters two packed windows share from the left (bitops.hpp:170-183 on the packed form)
template <typename T>
__device__ __forceinline__ unsigned window_lcp(T x1, T x2, T y1, T y2, const KeyShape& ks) {
    if (x1 != y1) return (clz_t<T>((T)(x1 ^ y1)) - (unsigned)(sizeof(T) * 8 - ks.c1 * ks.lc)) / ks.lc;
    if (x2 != y2) return ks.c1 + (clz_t<T>((T)(x2 ^ y2)) - (unsigned)(sizeof(T) * 8 - ks.c2 * ks.lc)) / ks.lc;
    return ks.c1 + ks.c2;
}

// Length of a suffix as far as the first round needs it (capped LCP, "shorter than 2k" test).
// Plain text: n - start.  String sets: the packed window carries end markers (code 0), so the
// length inside the window is 2k minus its trailing empty characters (suffix_array.hpp:1421,
// bucketing.hpp:138-143); 2k means "at least 2k".
template <bool GSA, typename T>
__device__ __forceinline__ uint64_t first_round_len(uint64_t ng, T sa, T w1, T w2, const KeyShape& ks) {
    if (!GSA) return ng - (uint64_t)sa;
    if (ks.c2 && w2 != 0) return ks.c1 + ks.c2 - ctz_t<T>(w2) / ks.lc;
    return w1 != 0 ? ks.c1 - ctz_t<T>(w1) / ks.lc : 0;
}

// What a rank needs to know about the records just outside its block of the globally sorted
// sequence (all zero on a single GPU).  psac gets the same with right_shift of the last tuple
// (bucketing.hpp:77,100) and exscan(max) of the bucket ids (bucketing.hpp:39).
template <typename T> struct Boundary {
    uint64_t off;            // global SA position of local record 0
    uint64_t base;           // id of the last bucket head on all lower ranks (0 = none)
    int has_prev, has_next;
    T prev1, prev2, prev3;   // last record (k1, k2, suffix) of the nearest non-empty lower rank
    T next1, next2, next3;   // first record of the nearest non-empty higher rank
};

// ------------------------------------------------------------------ the sorted records of a split round, read where they lie
// A refinement round whose records were split into heavy and light ones (heavy_keys.hpp) has its sorted order in three pieces: the light
// records sorted by (bucket number, rank h further) (SLK / SLV, suffixes as 32-bit entries), the heavy suffixes in their buckets' runs of an
// array in list order (HB), and per bucket where it starts in the list, its heavy rank, how many light records lie below it, the length of
// its heavy run and where its light records start.  Sorted record r of bucket b (o = r - bstart[b]) is light record lstart[b] + o for
// o < less[b], heavy suffix HB[r - less[b]] under the key (b << kb2 | value[b]) for less[b] <= o < less[b] + eq[b], light record
// lstart[b] + o - eq[b] behind that.  last_head_kernel and rebucket_refine_kernel read the records through this view instead of through two
// arrays a merge pass would have to write and they would have to read back (2^30 records: 7 + 5 ms of a round of 63).
template <typename T> struct HeavyView {
    const uint64_t* bstart = nullptr; const uint64_t* value = nullptr; const unsigned long long* less = nullptr;
    const unsigned long long* eq = nullptr; const uint64_t* lstart = nullptr;
    const T* SLK = nullptr; const uint32_t* SLV = nullptr; const uint32_t* HB = nullptr;
    uint32_t nb = 0; unsigned kb2 = 0;
    const uint64_t* rank = nullptr;            // per bucket: the rank (0-based) the members of its heavy run carry in ISA after the round; bit 63: they carry it already
    const ulonglong2* tile_b = nullptr;        // per scan tile whose records, the one before and the one after lie inside ONE heavy run: (their key, less of the bucket
                                               // | bit 62 when the run keeps its rank); else (0, ~0) (heavy_tiles_kernel)
};
constexpr uint64_t HEAVY_VIEW_KEEP = 1ull << 62, HEAVY_RANK_KEEP = 1ull << 63;
template <typename T>
__device__ __forceinline__ unsigned heavy_bucket_of(const HeavyView<T>& hv, uint64_t r) {
    unsigned lo = 0, hi = hv.nb;               // last bucket that starts at or before r
    while (hi - lo > 1) { const unsigned m = (lo + hi) >> 1; if (hv.bstart[m] <= r) lo = m; else hi = m; }
    return lo;
}
template <typename T>
__device__ __forceinline__ void heavy_record(const HeavyView<T>& hv, uint64_t r, T& key, T& val) {
    const unsigned b = heavy_bucket_of(hv, r);
    const uint64_t o = r - hv.bstart[b], less = hv.less[b], eq = hv.eq[b];
    if (o >= less && o < less + eq) { key = (T)(((uint64_t)b << hv.kb2) | hv.value[b]); val = (T)hv.HB[r - less]; }
    else { const uint64_t s = hv.lstart[b] + (o < less ? o : o - eq); key = hv.SLK[s]; val = (T)hv.SLV[s]; }
}
// ITEMS consecutive records from r0 on (keys and suffixes; zeros from n on): the bucket is looked up once and followed from there
// hrank[j]: the ISA rank of record j if it is a heavy one (bit j of *hmask), as the plan of the round has it
template <typename T, int ITEMS>
__device__ __forceinline__ void heavy_run(const HeavyView<T>& hv, uint64_t r0, uint64_t n, T (&key)[ITEMS], T (&val)[ITEMS], uint32_t (&hrank)[ITEMS], unsigned* hmask) {
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { key[j] = 0; val[j] = 0; hrank[j] = 0; }
    *hmask = 0;
    if (r0 >= n) return;
    unsigned b = heavy_bucket_of<T>(hv, r0);
    uint64_t s0 = hv.bstart[b], s1 = hv.bstart[b + 1], less = hv.less[b], eq = hv.eq[b], ls = hv.lstart[b];
    T hk = (T)(((uint64_t)b << hv.kb2) | hv.value[b]);
    uint32_t hr = (uint32_t)hv.rank[b];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint64_t r = r0 + j;
        if (r >= n) break;
        if (r >= s1) {          // (buckets have at least two records)
            ++b;
            s0 = s1; s1 = hv.bstart[b + 1]; less = hv.less[b]; eq = hv.eq[b]; ls = hv.lstart[b];
            hk = (T)(((uint64_t)b << hv.kb2) | hv.value[b]);
            hr = (uint32_t)hv.rank[b];
        }
        const uint64_t o = r - s0;
        if (o >= less && o < less + eq) { key[j] = hk; val[j] = (T)hv.HB[r - less]; hrank[j] = hr; *hmask |= 1u << j; }
        else { const uint64_t s = ls + (o < less ? o : o - eq); key[j] = hv.SLK[s]; val[j] = (T)hv.SLV[s]; }
    }
}
// do the records r0 .. r1 (both included) lie inside ONE heavy run?  Then they share *key and record r is the suffix HB[r - *less].
template <typename T>
__device__ __forceinline__ bool heavy_pure(const HeavyView<T>& hv, uint64_t r0, uint64_t r1, T* key, uint64_t* less) {
    const unsigned b = heavy_bucket_of(hv, r0);
    const uint64_t a = hv.bstart[b] + hv.less[b];
    if (r0 < a || r1 >= a + hv.eq[b]) return false;
    *key = (T)(((uint64_t)b << hv.kb2) | hv.value[b]); *less = hv.less[b];
    return true;
}

// tile_b[t] for the scan tiles of `tile` records each (a tile on its own would find its bucket by a chain of twelve dependent loads while
// the rest of its workgroup waits: 3 ms of a 13 ms rebucket pass over 2^30 records; here the chains of all tiles run side by side)
template <typename T>
__global__ void heavy_tiles_kernel(HeavyView<T> hv, uint64_t cnt, unsigned tile, uint64_t ntiles, ulonglong2* __restrict__ tile_b) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const uint64_t t0 = t * tile, t1 = t0 + tile < cnt ? t0 + tile : cnt - 1;
    ulonglong2 out; out.x = 0; out.y = ~0ull;
    if (t0 > 0) {
        T key; uint64_t less;
        if (heavy_pure<T>(hv, t0 - 1, t1, &key, &less))
            { out.x = (unsigned long long)key; out.y = less | ((hv.rank[(uint64_t)key >> hv.kb2] & HEAVY_RANK_KEEP) ? HEAVY_VIEW_KEEP : 0ull); }
    }
    tile_b[t] = out;
}

// ------------------------------------------------------------------ tile carries for the prefix-max
// Bucket ids are a prefix maximum over "head" positions (bucketing.hpp:21-53).  It is
// evaluated in three launches without any inter-workgroup waiting:
//   1. last_head_kernel: per tile, the id of its last head (found by a backward
//      search that normally stops within the last 64 records),
//   2. tile_scan_kernel: exclusive scan of those per-tile values,
//   3. rebucket_*_kernel: every tile recomputes its heads and fills from its carry.
// REFINE = false: heads of the first round, the 2k-character windows differ (packed pair
//                 differs, or one of the two suffixes is shorter than 2k).
// REFINE = true : heads inside old buckets, (K1,K2) differs or K2 == 0; id = pos + 1.
// HEAVY (REFINE, one-word keys): the records come through a HeavyView; a tile inside one heavy run has no head.
template <typename T, bool REFINE, bool GSA = false, bool HEAVY = false>
__global__ __launch_bounds__(256) void last_head_kernel(const T* __restrict__ A1, const T* __restrict__ A2,
                                 const T* __restrict__ pos, uint64_t cnt, unsigned tile_size,
                                 uint64_t ntiles, uint64_t* __restrict__ agg, const T* __restrict__ SA,
                                 KeyShape ks, uint64_t n_global, Boundary<T> bd, HeavyView<T> hv = HeavyView<T>()) {
    // Refinement rounds: one workgroup of four waves per tile, every wave searching its quarter of the tile backwards.  In
    // ordinary text each finds a head in its first window; in a text with long runs of equal records (tandem repeats) the
    // whole tile is walked, and one wave per tile was 96 dependent steps = 34 us of a 230 us refinement round on 2^20
    // suffixes.  The first round (once per construction, 2^20 tiles at 2^32 records) keeps one wave per tile.
    constexpr unsigned NW = 256 / WAVE;
    constexpr unsigned WPT = REFINE ? NW : 1;           // waves per tile
    constexpr unsigned TPB = NW / WPT;                  // tiles per workgroup
    __shared__ uint64_t wfound[NW];
    const unsigned lane = lane_id(), wv = threadIdx.x / WAVE;
    const uint64_t tile = (uint64_t)blockIdx.x * TPB + wv / WPT;
    const unsigned sub = wv % WPT;
    const uint64_t t_lo = tile < ntiles ? tile * tile_size : cnt;
    uint64_t t_hi = t_lo + tile_size;
    if (t_hi > cnt) t_hi = cnt;
    const unsigned quarter = ((tile_size + WPT - 1) / WPT + WAVE - 1) / WAVE * WAVE;
    const uint64_t lo = t_lo + (uint64_t)sub * quarter < t_hi ? t_lo + (uint64_t)sub * quarter : t_hi;
    const uint64_t hi = lo + quarter < t_hi ? lo + quarter : t_hi;
    uint64_t found = 0;
    bool quarter_pure = false;          // (HEAVY: this wave's records and the one before them lie in one heavy run: no head among them)
    if constexpr (HEAVY) {
        T k_; uint64_t l_;
        quarter_pure = tile < ntiles && hv.tile_b[tile].y != ~0ull;
        if (!quarter_pure) quarter_pure = hi > lo && lo > 0 && heavy_pure<T>(hv, lo - 1, hi - 1, &k_, &l_);
    }
    // walk backwards in windows of 64 records [w0, w0 + 64)
    for (uint64_t wend = hi; wend > lo && !quarter_pure; ) {
        const uint64_t w0 = wend >= lo + WAVE ? wend - WAVE : lo;
        const uint64_t e = w0 + lane;
        bool head = false;
        if (e < wend) {
            if (e == 0 && !bd.has_prev) head = true;
            else {
                // (A2 == nullptr: 64-bit words holding both keys of a refinement record, K1 << kb2 | K2, kb2 = ks.lc bits for K2)
                const uint64_t m2 = (~0ull) >> (64 - (REFINE ? ks.lc : 32u));
                T x1, y1;
                if constexpr (HEAVY) {
                    T v_;
                    heavy_record<T>(hv, e, x1, v_);
                    if (e) heavy_record<T>(hv, e - 1, y1, v_); else y1 = bd.prev1;
                } else { x1 = A1[e]; y1 = e ? A1[e - 1] : bd.prev1; }
                const T x2 = A2 ? A2[e] : (T)((uint64_t)x1 & m2);
                const T y2 = A2 ? (e ? A2[e - 1] : bd.prev2) : (T)((uint64_t)y1 & m2);
                head = (x1 != y1) || (x2 != y2) || (REFINE && x2 == 0);
                if (!REFINE && !head) {
                    // equal packed windows: still a boundary if either suffix is shorter than 2k
                    const uint64_t two_k = ks.c1 + ks.c2;
                    if (GSA) head = first_round_len<true, T>(n_global, (T)0, x1, x2, ks) < two_k;    // equal windows
                    else {
                        const T ysa = e ? SA[e - 1] : bd.prev3;
                        head = (n_global - (uint64_t)SA[e] < two_k) || (n_global - (uint64_t)ysa < two_k);
                    }
                }
            }
        }
        const uint64_t m = __ballot(head);
        if (m) {
            const uint64_t at = w0 + (63u - (unsigned)__builtin_clzll(m));
            found = REFINE ? (uint64_t)(pos ? pos[at] : (T)at) + 1 : bd.off + at + 1;
            break;
        }
        wend = w0;
    }
    if (lane == 0) wfound[wv] = found;
    __syncthreads();
    if (threadIdx.x < TPB) {
        const uint64_t t = (uint64_t)blockIdx.x * TPB + threadIdx.x;
        uint64_t f = 0;
        for (int w = (int)WPT - 1; w >= 0 && !f; --w) f = wfound[threadIdx.x * WPT + w];
        if (t < ntiles) agg[t] = f;
    }
}

// ------------------------------------------------------------------ one-word records read where they lie
// The prefix sort of the first round (engine.hpp: prefix_sort_1w) leaves the records of bucket b -- the top digit of the sorted prefix --
// at [off[b], off[b + 1]) as ONE word each: (rest of the prefix) << sfield | suffix.  The kernels that follow the sort (tie_resolve_1w_kernel,
// last_head_1w_kernel, rebucket_first_kernel<.., ONEW>) read them in that form -- the top digit is known from a record's place -- instead of
// having the last pass write word 1 and the suffix as two arrays (8 bytes per record less written by the pass and 8 less read by the tie
// stage; the suffix array itself is written by the rebucket kernel, which holds the suffixes anyway).
struct OneWordView {
    const unsigned long long* off;      // off[0 .. 256], off[256] = number of records
    unsigned low, sfield, lo1;          // prefix bits in the word, width of the suffix field, bits of word 1 below the sorted prefix
};
// the bucket of record e: the largest b with off[b] <= e (empty buckets are skipped)
__device__ __forceinline__ unsigned onew_bucket(const OneWordView& ow, uint64_t e) {
    unsigned b = 0;
#pragma unroll
    for (int s = 128; s; s >>= 1) if (ow.off[b + s] <= e) b += s;
    return b;
}
// the leading bits of word 1 (top digit | rest of the prefix), and word 1 with zeros below them
__device__ __forceinline__ uint64_t onew_lead(const OneWordView& ow, unsigned b, uint64_t rec) { return ((uint64_t)b << ow.low) | (rec >> ow.sfield); }
__device__ __forceinline__ uint64_t onew_word1(const OneWordView& ow, unsigned b, uint64_t rec) { return onew_lead(ow, b, rec) << ow.lo1; }
__device__ __forceinline__ uint64_t onew_suffix(const OneWordView& ow, uint64_t rec) { return rec & ((1ull << ow.sfield) - 1); }
// a thread that walks records in ascending order keeps its bucket and the end of that bucket in registers
struct OneWordCursor {
    unsigned b; uint64_t end;
    __device__ __forceinline__ void start(const OneWordView& ow, uint64_t e) { b = onew_bucket(ow, e); end = ow.off[b + 1]; }
    __device__ __forceinline__ unsigned at(const OneWordView& ow, uint64_t e) {
        while (b < 255u && e >= end) { ++b; end = ow.off[b + 1]; }
        return b;
    }
};

// last_head_kernel (first round) on one-word records: W1 / S2 hold both words of the suffixes that tie on the leading bits
// (tie_resolve_1w_kernel filled them), nothing elsewhere.  One wave per tile, four tiles per workgroup.
template <int TAG>
__global__ __launch_bounds__(256) void last_head_1w_kernel(const uint64_t* __restrict__ R, const uint64_t* __restrict__ W1, const uint64_t* __restrict__ S2,
                                                           uint64_t cnt, unsigned tile_size, uint64_t ntiles, uint64_t* __restrict__ agg, KeyShape ks,
                                                           OneWordView ow) {
    constexpr unsigned NW = 256 / WAVE;
    const unsigned lane = lane_id(), wv = threadIdx.x / WAVE;
    const uint64_t tile = (uint64_t)blockIdx.x * NW + wv;
    if (tile >= ntiles) return;
    const uint64_t lo = tile * tile_size;
    uint64_t hi = lo + tile_size;
    if (hi > cnt) hi = cnt;
    const uint64_t two_k = ks.c1 + ks.c2;
    uint64_t found = 0;
    for (uint64_t wend = hi; wend > lo; ) {
        const uint64_t w0 = wend >= lo + WAVE ? wend - WAVE : lo;
        const uint64_t e = w0 + lane;
        bool head = false;
        if (e < wend) {
            if (e == 0) head = true;
            else {
                const uint64_t x = R[e], y = R[e - 1];
                // (the buckets are looked up only when the rest of the prefix agrees: eight dependent loads each)
                head = (x >> ow.sfield) != (y >> ow.sfield);
                if (!head) head = onew_bucket(ow, e) != onew_bucket(ow, e - 1);
                if (!head) head = W1[e] != W1[e - 1] || S2[e] != S2[e - 1];
                if (!head) head = (cnt - onew_suffix(ow, x) < two_k) || (cnt - onew_suffix(ow, y) < two_k);
            }
        }
        const uint64_t m = __ballot(head);
        if (m) { found = w0 + (63u - (unsigned)__builtin_clzll(m)) + 1; break; }
        wend = w0;
    }
    if (lane == 0) agg[tile] = found;
}

// one-word records -> word 1 (zeros below the sorted prefix) and the suffixes as words: what the last pass of the prefix sort writes when the
// records are not read where they lie (the tie stage met a long group and takes its radix path)
template <int TAG>
__global__ __launch_bounds__(256) void onew_widen_kernel(const uint64_t* __restrict__ R, uint64_t n, OneWordView ow, uint64_t* __restrict__ S1, uint64_t* __restrict__ SA) {
    __shared__ unsigned long long s_off[257];
    for (int i = threadIdx.x; i < 257; i += 256) s_off[i] = ow.off[i];
    __syncthreads();
    ow.off = s_off;
    constexpr int ITEMS = 8;
    const uint64_t stride = (uint64_t)gridDim.x * 256 * ITEMS;
    for (uint64_t e0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * ITEMS; e0 < n; e0 += stride) {
        uint64_t x[ITEMS], w1[ITEMS], sa[ITEMS];
        load_run<uint64_t, ITEMS>(R, e0, n, x, 0ull);
        OneWordCursor cu;
        cu.start(ow, e0);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { w1[j] = e0 + j < n ? onew_word1(ow, cu.at(ow, e0 + j), x[j]) : 0ull; sa[j] = onew_suffix(ow, x[j]); }
        store_run<uint64_t, ITEMS>(S1, e0, n, w1);
        store_run<uint64_t, ITEMS>(SA, e0, n, sa);
    }
}

// Exclusive scan of `len` uint64 values by one workgroup (len is the number of tiles, at
// most a few tens of thousands).  total[0] receives the reduction of everything.
template <int BLOCK, typename Op>
__global__ __launch_bounds__(BLOCK) void tile_scan_kernel(uint64_t* __restrict__ a, uint64_t len, Op op,
                                                          uint64_t identity, uint64_t* __restrict__ total) {
    constexpr int PER = 4;             // (16 entries per thread and step were tried: 1.17 against 0.75 ms for 2^20 tiles)
    __shared__ uint64_t tmp[BLOCK / WAVE + 1];
    uint64_t carry = identity;
    for (uint64_t base = 0; base < len; base += (uint64_t)BLOCK * PER) {
        const uint64_t e0 = base + (uint64_t)threadIdx.x * PER;
        uint64_t v[PER];
        uint64_t run = identity;
#pragma unroll
        for (int j = 0; j < PER; ++j) { v[j] = (e0 + j < len) ? a[e0 + j] : identity; run = op(run, v[j]); }
        uint64_t tot;
        uint64_t ex = block_scan_exclusive<BLOCK, uint64_t>(run, op, identity, tmp, &tot);
        ex = op(carry, ex);
#pragma unroll
        for (int j = 0; j < PER; ++j) { if (e0 + j < len) a[e0 + j] = ex; ex = op(ex, v[j]); }
        carry = op(carry, tot);
    }
    if (threadIdx.x == 0 && total) total[0] = carry;
}

// Long arrays (2^20 tiles at 2^32 records: one workgroup took 0.75 ms): every workgroup scans a chunk of BLOCK * 4 entries in place and
// leaves the chunk's total (blockIdx.y picks the array: a0 / a1, totals at tot and tot + tot_stride), one workgroup scans the totals
// (tile_scan_kernel / tile_scan2_kernel on them), chunk_add_kernel puts every chunk's offset onto its entries.
template <int BLOCK, typename Op>
__global__ __launch_bounds__(BLOCK) void chunk_scan_kernel(uint64_t* __restrict__ a0, uint64_t* __restrict__ a1, uint64_t len, Op op, uint64_t identity,
                                                           uint64_t* __restrict__ tot, unsigned tot_stride) {
    constexpr int PER = 4;
    __shared__ uint64_t tmp[BLOCK / WAVE + 1];
    uint64_t* const a = blockIdx.y ? a1 : a0;
    const uint64_t e0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * PER;
    uint64_t v[PER];
    uint64_t run = identity;
#pragma unroll
    for (int j = 0; j < PER; ++j) { v[j] = (e0 + j < len) ? a[e0 + j] : identity; run = op(run, v[j]); }
    uint64_t total;
    uint64_t ex = block_scan_exclusive<BLOCK, uint64_t>(run, op, identity, tmp, &total);
#pragma unroll
    for (int j = 0; j < PER; ++j) { if (e0 + j < len) a[e0 + j] = ex; ex = op(ex, v[j]); }
    if (threadIdx.x == 0) tot[(size_t)blockIdx.y * tot_stride + blockIdx.x] = total;
}
template <int BLOCK, typename Op>
__global__ __launch_bounds__(BLOCK) void chunk_add_kernel(uint64_t* __restrict__ a0, uint64_t* __restrict__ a1, uint64_t len, Op op,
                                                          const uint64_t* __restrict__ off, unsigned off_stride) {
    constexpr int PER = 4;
    uint64_t* const a = blockIdx.y ? a1 : a0;
    const uint64_t o = off[(size_t)blockIdx.y * off_stride + blockIdx.x];
    const uint64_t e0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * PER;
#pragma unroll
    for (int j = 0; j < PER; ++j) if (e0 + j < len) a[e0 + j] = op(o, a[e0 + j]);
}

// The two sum scans that follow a rebucket step (active positions, buckets with more than one member) in one launch:
// block 0 scans a0, block 1 scans a1.  totals[b] receives block b's sum; host_totals (optional) is the same pair in
// host-pinned memory the device can write, so that the host reads the counters after a stream synchronisation without
// a copy operation in between.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void tile_scan2_kernel(uint64_t* __restrict__ a0, uint64_t* __restrict__ a1, uint64_t len,
                                                           uint64_t* __restrict__ totals, uint64_t* __restrict__ host_totals) {
    constexpr int PER = 4;
    __shared__ uint64_t tmp[BLOCK / WAVE + 1];
    uint64_t* const a = blockIdx.x ? a1 : a0;
    const OpSum op;
    uint64_t carry = 0;
    for (uint64_t base = 0; base < len; base += (uint64_t)BLOCK * PER) {
        const uint64_t e0 = base + (uint64_t)threadIdx.x * PER;
        uint64_t v[PER];
        uint64_t run = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { v[j] = (e0 + j < len) ? a[e0 + j] : 0; run += v[j]; }
        uint64_t tot;
        uint64_t ex = block_scan_exclusive<BLOCK, uint64_t>(run, op, (uint64_t)0, tmp, &tot);
        ex += carry;
#pragma unroll
        for (int j = 0; j < PER; ++j) { if (e0 + j < len) a[e0 + j] = ex; ex += v[j]; }
        carry += tot;
    }
    if (threadIdx.x == 0) {
        totals[blockIdx.x] = carry;
        if (host_totals) host_totals[blockIdx.x] = carry;
    }
}

// ------------------------------------------------------------------ K6 + K7, first round
// Sorted (S1,S2): writes the bucket id of every position, the LCP of the 2k-mers at
// every bucket boundary (suffix_array.hpp:1353-1396; sentinel n elsewhere) and, per
// tile, how many positions stay active (share their bucket) and how many buckets
// hold more than one suffix (bucketing.hpp:98-118).
// PCB > 0 (64-bit words, at most 2^32 positions, one GPU): the kernel also runs the first level of the SA -> ISA
// inversion on the records it holds anyway: the (position, rank) pairs (SA[e], id[e] - 1) leave as 32-bit pairs
// partitioned into 2^PCB destination classes by SA >> part_shift (see partition_pairs_kernel, whose first level this
// replaces: its 16 bytes per record of reads are saved).  part_key / part_val: the pair arrays, part_cursors: zeroed.
// PPK (with PCB): the pairs leave as ONE array of 64-bit entries (position | rank << 32, part_key viewed as uint64_t*,
// part_val unused): a class's run of a tile is one 64-byte piece instead of two of 32 bytes, and is staged through LDS once.
// ONEW (one GPU, 64-bit words): S1 holds the one-word records of the prefix sort (OneWordView), S1t word 1 of the suffixes that tie on the
// leading bits (S2 their word 2), SA is not read: the suffixes come out of the records and leave through sa_out.
template <typename T, int BLOCK, int ITEMS, bool WITH_LCP, bool GSA = false, int PCB = 0, bool PPK = false, bool ONEW = false>
__global__ __launch_bounds__(BLOCK) RB_WAVES_ATTR void rebucket_first_kernel(
    const T* __restrict__ S1, const T* __restrict__ S2, const T* __restrict__ SA, uint64_t n, KeyShape ks,
    T* __restrict__ Bsa, T* __restrict__ LCP, const uint64_t* __restrict__ carry_in,
    uint64_t* __restrict__ n_active, uint64_t* __restrict__ n_unf, uint64_t ng, Boundary<T> bd,
    T* __restrict__ pyr1 = nullptr, unsigned* __restrict__ sa_hist = nullptr, int sa_hist_shift = 0,
    uint32_t* __restrict__ part_key = nullptr, uint32_t* __restrict__ part_val = nullptr, unsigned part_shift = 0,
    unsigned* __restrict__ part_cursors = nullptr, T* __restrict__ sa_out = nullptr, int lazy_ids = 0,
    OneWordView ow = OneWordView(), const T* __restrict__ S1t = nullptr) {
    // lazy_ids: a tile without a single unresolved suffix does not write its bucket ids (they are e + 1 and nobody reads them
    // unless some OTHER tile has unresolved suffixes: the caller then fills them in, fill_resolved_ids_kernel)
    // sa_out (optional): the suffixes are written there as well (the multi-GPU engine's records end in scratch arrays; the
    // copy into the rank's SA block rides along instead of reading them again)
    // sa_hist (optional): per-tile histogram of the digit of SA at sa_hist_shift, for the first level of the
    // SA -> ISA inversion when it runs as radix passes over tiles of this size
    // n: records in this block; ng: length of the whole text (LCP sentinel, suffix lengths)
    // pyr1 (optional): level 1 of the min-pyramid over LCP, pyr1[g] = min(LCP[64 g .. 64 g + 63]), written
    // here so that nobody has to read the fresh LCP array again
    constexpr int TILE = BLOCK * ITEMS;
    __shared__ T scan_tmp[BLOCK / WAVE + 1];
    __shared__ unsigned red_tmp[BLOCK / WAVE + 1];
    const unsigned tile = blockIdx.x;
    const uint64_t e0 = (uint64_t)tile * TILE + (uint64_t)threadIdx.x * ITEMS;
    const unsigned two_k = ks.c1 + ks.c2;

    // Word 2 of a record only matters where word 1 equals a neighbour's (window_lcp looks at it then, and the prefix sort
    // of the two-stage first round fills it for exactly those records): it is fetched for those records only -- 6 % of
    // random DNA -- instead of streamed (w of the 5 w bytes per record this kernel moves).  String sets read the string
    // ends out of it, so they stream it.
    constexpr bool LAZY2 = !GSA;
    // (runs read and written through load_run_x / store_run_x: whole rows of a wave on the memory side instead of 16-byte pieces
    //  64 bytes apart -- 46.2 -> 39.3 ms at 2^32 64-bit records, both forms in one process; the region doubles as the stage of the
    //  pair partition at the end of the kernel)
    constexpr size_t XP_RUNS = sizeof(T) == 8 ? sizeof(T) * (BLOCK / WAVE) * XRUN_WORDS<ITEMS>::N : 16;
    constexpr size_t XP_STAGE = PCB > 0 ? (PPK ? sizeof(uint64_t) : sizeof(uint32_t)) * (size_t)TILE : 0;
    __shared__ __attribute__((aligned(16))) unsigned char xp_raw[XP_RUNS > XP_STAGE ? XP_RUNS : XP_STAGE];
    T* const xp = reinterpret_cast<T*>(xp_raw);
    T* const xw = xp + (threadIdx.x / WAVE) * XRUN_WORDS<ITEMS>::N;
    T a1[ITEMS], a2[ITEMS], sa[ITEMS];
    // (ONEW: the table of the bucket starts is asked for before the records, so that its way from memory is not waited for on its own)
    static_assert(!ONEW || BLOCK >= 257, "one thread per table entry");
    const unsigned long long my_off = (ONEW && threadIdx.x < 257) ? ow.off[threadIdx.x] : 0ull;
    load_run_x<T, ITEMS>(S1, e0, n, a1, (T)0, xw);
    if (!LAZY2) load_run_x<T, ITEMS>(S2, e0, n, a2, (T)0, xw);
    if constexpr (!ONEW) load_run_x<T, ITEMS>(SA, e0, n, sa, (T)0, xw);
    T p1 = 0, p2 = 0, psa = 0;
    bool have_p = false;
    const bool have_q = e0 + ITEMS <= n && (e0 + ITEMS < n || bd.has_next);
    const bool q_in = e0 + ITEMS < n;
    T q1v = 0, qsav = 0;
    __shared__ unsigned long long s_off[ONEW ? 257 : 1];
    if constexpr (ONEW) {
        // word 1 (zeros below the sorted prefix) and the suffix out of every record; the top digit from the record's place
        // (the table of the bucket starts in LDS: a thread's search is eight dependent reads)
        static_assert(!ONEW || (sizeof(T) == 8 && LAZY2), "one-word records: 64-bit words, no string sets");
        if (threadIdx.x < 257) s_off[threadIdx.x] = my_off;
        __syncthreads();
        ow.off = s_off;
        OneWordCursor cu;
        cu.start(ow, e0 ? (e0 - 1 < n ? e0 - 1 : (n ? n - 1 : 0)) : 0);
        if (e0 > 0 && e0 - 1 < n) { const uint64_t x = S1[e0 - 1]; p1 = (T)onew_word1(ow, cu.b, x); psa = (T)onew_suffix(ow, x); have_p = true; }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint64_t x = a1[j];
            sa[j] = (T)onew_suffix(ow, x);
            a1[j] = e0 + j < n ? (T)onew_word1(ow, cu.at(ow, e0 + j), x) : (T)0;
        }
        if (have_q && q_in) { const uint64_t x = S1[e0 + ITEMS]; q1v = (T)onew_word1(ow, cu.at(ow, e0 + ITEMS), x); qsav = (T)onew_suffix(ow, x); }
    } else {
        if (e0 > 0 && e0 - 1 < n) { p1 = S1[e0 - 1]; psa = SA[e0 - 1]; have_p = true; if (!LAZY2) p2 = S2[e0 - 1]; }
        else if (e0 == 0 && bd.has_prev) { p1 = bd.prev1; p2 = bd.prev2; psa = bd.prev3; }
        q1v = have_q ? (q_in ? S1[e0 + ITEMS] : bd.next1) : (T)0;
    }
    bool q_tied = false;            // (ONEW) the record after this run ties with the run's last one on the leading bits
    if (LAZY2) {
        const bool p_tied = have_p && p1 == a1[0] && e0 < n;
        if (ONEW) q_tied = have_q && q_in && q1v == a1[ITEMS - 1];
        unsigned tied = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const bool eq_prev = j ? a1[j] == a1[j - 1] : ((have_p || (e0 == 0 && bd.has_prev)) && a1[0] == p1);
            const bool eq_next = (e0 + j + 1 == n) ? (bd.has_next && a1[j] == bd.next1)          // last record of the block
                                                   : (j + 1 < ITEMS ? a1[j] == a1[j + 1] : (have_q && a1[j] == q1v));
            if ((eq_prev || eq_next) && e0 + j < n) tied |= 1u << j;
        }
        if (p_tied) { p2 = S2[e0 - 1]; if (ONEW) p1 = S1t[e0 - 1]; }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const bool t = (tied >> j) & 1u;
            a2[j] = t ? S2[e0 + j] : (T)0;
            if (ONEW && t) a1[j] = S1t[e0 + j];          // (the bits of word 1 below the sorted prefix matter between tied suffixes only)
        }
        if (ONEW && q_tied) q1v = S1t[e0 + ITEMS];
    }
    // head flag of the first record after this run (for the activity test)
    bool next_head = true;
    if (have_q) {
        const bool in = q_in;
        const T q1 = q1v;
        const T q2 = in ? ((ONEW ? q_tied : (!LAZY2 || q1 == a1[ITEMS - 1])) ? S2[e0 + ITEMS] : (T)0) : bd.next2;
        const T qsa = in ? (ONEW ? qsav : SA[e0 + ITEMS]) : bd.next3;
        uint64_t c = window_lcp<T>(a1[ITEMS - 1], a2[ITEMS - 1], q1, q2, ks);
        const uint64_t la = first_round_len<GSA, T>(ng, sa[ITEMS - 1], a1[ITEMS - 1], a2[ITEMS - 1], ks);
        const uint64_t lb = first_round_len<GSA, T>(ng, qsa, q1, q2, ks);
        c = c < la ? c : la; c = c < lb ? c : lb;
        next_head = c < two_k;
    }

    T id[ITEMS];
    T lc[ITEMS];
    unsigned heads = 0;
    T run = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint64_t e = e0 + j;
        // common characters of the two 2k windows, the end marker differing from everything
        uint64_t c = window_lcp<T>(p1, p2, a1[j], a2[j], ks);
        const uint64_t la = first_round_len<GSA, T>(ng, psa, p1, p2, ks), lb = first_round_len<GSA, T>(ng, sa[j], a1[j], a2[j], ks);
        c = c < la ? c : la; c = c < lb ? c : lb;
        const bool very_first = (e == 0) && !bd.has_prev;      // nothing sorts before this record
        const bool head = very_first || c < two_k;
        if (head || e >= n) heads |= 1u << j;
        id[j] = (e < n && head) ? (T)(bd.off + e + 1) : (T)0;
        if (WITH_LCP) lc[j] = head ? ((very_first && bd.off == 0) ? (T)0 : (T)c) : (T)ng;
        p1 = a1[j]; p2 = a2[j]; psa = sa[j];
        if (id[j] > run) run = id[j];
    }
    if (next_head) heads |= 1u << ITEMS;
    if (e0 < n && e0 + ITEMS > n) {
        // the block ends inside this run: the record after its last one lives on the next rank
        const unsigned lastj = (unsigned)(n - 1 - e0);
        bool nh = true;
        if (bd.has_next) {
            uint64_t c = window_lcp<T>(a1[lastj], a2[lastj], bd.next1, bd.next2, ks);
            const uint64_t la = first_round_len<GSA, T>(ng, sa[lastj], a1[lastj], a2[lastj], ks);
            const uint64_t lb = first_round_len<GSA, T>(ng, bd.next3, bd.next1, bd.next2, ks);
            c = c < la ? c : la; c = c < lb ? c : lb;
            nh = c < two_k;
        }
        heads = nh ? (heads | (1u << (lastj + 1))) : (heads & ~(1u << (lastj + 1)));
    }
    // activity: not a head, or a head followed by a non-head
    unsigned nact = 0, nub = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if (e0 + j < n) {
            const bool h = (heads >> j) & 1u, hn = (heads >> (j + 1)) & 1u;
            if (!h || !hn) ++nact;
            if (h && !hn) ++nub;
        }
    }
    T agg;
    T excl = block_scan_exclusive<BLOCK, T>(run, OpMax(), (T)0, scan_tmp, &agg);
    static_assert(TILE < (1 << 16), "both counters of a tile reduced in one word");
    const unsigned both_counts = block_reduce<BLOCK, unsigned>(nact | (nub << 16), OpSum(), red_tmp);
    const unsigned tact = both_counts & 0xFFFFu, tub = both_counts >> 16;
    if (threadIdx.x == 0) { n_active[tile] = tact; n_unf[tile] = tub; }
    T carry = (T)carry_in[tile];
    if ((T)bd.base > carry) carry = (T)bd.base;
    if (excl > carry) carry = excl;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if (id[j] == 0) id[j] = carry; else carry = id[j];
    }
    if (!(lazy_ids && tact == 0)) store_run_x<T, ITEMS>(Bsa, e0, n, id, xw);
    if (WITH_LCP && !(RB_ABLATE & 1)) store_run_x<T, ITEMS>(LCP, e0, n, lc, xw);
    if (sa_out && !(RB_ABLATE & 2)) store_run_x<T, ITEMS>(sa_out, e0, n, sa, xw);
    if (sa_hist) {
        __shared__ unsigned dh[4 * RADIX];
        for (int i = threadIdx.x; i < 4 * RADIX; i += BLOCK) dh[i] = 0;
        __syncthreads();
        unsigned* my = dh + ((threadIdx.x / WAVE) & 3) * RADIX;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) wave_hist_add(my, (unsigned)(sa[j] >> sa_hist_shift) & (RADIX - 1), e0 + j < n);
        __syncthreads();
        for (int d = threadIdx.x; d < RADIX; d += BLOCK)
            sa_hist[(uint64_t)tile * RADIX + d] = dh[d] + dh[RADIX + d] + dh[2 * RADIX + d] + dh[3 * RADIX + d];
    }
    if (WITH_LCP && pyr1) {
        static_assert(64 % ITEMS == 0 && (64 / ITEMS) <= WAVE, "a group of 64 entries must sit inside one wave");
        T m = ~(T)0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) if (e0 + j < n && lc[j] < m) m = lc[j];
#pragma unroll
        for (int d = 1; d < 64 / ITEMS; d <<= 1) { const T o = shfl_xor<T>(m, d); m = o < m ? o : m; }
        if ((threadIdx.x & (64 / ITEMS - 1)) == 0 && e0 < n) pyr1[e0 >> 6] = m;
    }
    if constexpr (PCB > 0 && !(RB_ABLATE & 4)) {
        constexpr int NCLS = 1 << PCB;
        static_assert(BLOCK >= NCLS, "one thread per class");
        typedef typename std::conditional<PPK, uint64_t, uint32_t>::type ST;
        ST* const stage = reinterpret_cast<ST*>(xp_raw);
        __shared__ unsigned pcnt[NCLS];
        __shared__ unsigned pstart[NCLS];
        __shared__ uint64_t pbase[NCLS];
        __shared__ unsigned pscan_tmp[BLOCK / WAVE + 1];
        const unsigned tid = threadIdx.x;
        const uint64_t tbase = (uint64_t)tile * TILE;
        const unsigned count = n - tbase < (uint64_t)TILE ? (unsigned)(n - tbase) : (unsigned)TILE;
        for (int i = tid; i < NCLS; i += BLOCK) pcnt[i] = 0;
        __syncthreads();
        unsigned slot[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const unsigned d = (unsigned)((uint64_t)sa[j] >> part_shift) & (NCLS - 1);
            slot[j] = e0 + j < n ? atomicAdd(&pcnt[d], 1u) : 0u;
        }
        __syncthreads();
        const unsigned tot = tid < NCLS ? pcnt[tid] : 0u;
        unsigned total;
        const unsigned bs = block_scan_exclusive<BLOCK, unsigned>(tot, OpSum(), 0u, pscan_tmp, &total);
        if (tid < NCLS) {
            pstart[tid] = bs;
            if (tot) {
                const unsigned at = atomicAdd(&part_cursors[tid], tot);
                pbase[tid] = ((uint64_t)tid << part_shift) + at - bs;
            }
        }
        __syncthreads();
        if constexpr (PPK) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const unsigned d = (unsigned)((uint64_t)sa[j] >> part_shift) & (NCLS - 1);
                if (e0 + j < n) stage[slot[j] + pstart[d]] = (uint64_t)(uint32_t)sa[j] | ((uint64_t)(uint32_t)(id[j] - 1) << 32);
            }
            __syncthreads();
            uint64_t* const out = reinterpret_cast<uint64_t*>(part_key);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const unsigned p = tid + j * BLOCK;
                if (p < count && !(RB_ABLATE & 8)) {
                    const uint64_t x = stage[p];
                    out[pbase[((uint32_t)x >> part_shift) & (NCLS - 1)] + p] = x;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const unsigned d = (unsigned)((uint64_t)sa[j] >> part_shift) & (NCLS - 1);
                slot[j] += pstart[d];
                if (e0 + j < n) stage[slot[j]] = (uint32_t)sa[j];
            }
            __syncthreads();
            uint64_t dest[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const unsigned p = tid + j * BLOCK;
                if (p < count) {
                    const uint32_t x = (uint32_t)stage[p];
                    dest[j] = pbase[(x >> part_shift) & (NCLS - 1)] + p;
                    part_key[dest[j]] = x;
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (e0 + j < n) stage[slot[j]] = (uint32_t)(id[j] - 1);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const unsigned p = tid + j * BLOCK;
                if (p < count) part_val[dest[j]] = (uint32_t)stage[p];
            }
        }
    }
}

// ------------------------------------------------------------------ K8
template <typename T>
__global__ void isa_scatter_kernel(const T* __restrict__ SA, const T* __restrict__ Bsa, uint64_t n,
                                   T* __restrict__ ISA, uint64_t koff) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        ISA[(uint64_t)SA[i] - koff] = Bsa[i] - 1;      // ISA holds 0-based ranks throughout (suffix_array.hpp:460-464)
}

// ------------------------------------------------------------------ K8, cache-friendly form
// ISA[SA[i]] = val[i] - 1 is the inverse of a permutation; written directly it is n
// random 4/8-byte stores (32 B of HBM traffic each).  Because SA is a permutation,
// the destination range [g << s, (g + 1) << s) receives EXACTLY its own size in
// records, so the pairs can be MSD-partitioned by destination with no histogram: each
// tile reserves room in bucket g with one atomicAdd (order inside a bucket is
// irrelevant), and after one or two such passes every bucket is a 4096-entry
// window that one workgroup scatters inside LDS and writes out as full lines.
constexpr int INV_WINDOW_BITS = 12;

// TI / TO: entry types of the pairs read and written.  A permutation of at most 2^32 positions held in 64-bit words is
// narrowed by its first level (SUB1: the value's -1 is applied there, so that bucket id n = 2^32 fits) and travels as
// 32-bit pairs from then on: 24 + 16 + 16 bytes per record over three levels instead of 3 x 32.
// CB: class bits of a level (2^CB destination classes per parent bucket; BLOCK >= 2^CB).
// (a register cap for a third workgroup per CU, as in the narrow scatter passes of radix.hpp, spills 13 registers here and
//  loses: 34.8-35.8 against 33.0-33.4 ms for the second inversion level + window scatter at 2^32)
template <typename TI, typename TO, int BLOCK, int ITEMS, bool SUB1 = false, int CB = 8>
__global__ __launch_bounds__(BLOCK) void partition_pairs_kernel(
    const TI* __restrict__ key_in, const TI* __restrict__ val_in, TO* __restrict__ key_out,
    TO* __restrict__ val_out, uint64_t n, unsigned shift, unsigned* __restrict__ cursors, uint64_t koff) {
    // koff is subtracted from every key on the way in (first level of a rank's block)
    constexpr int NCLS = 1 << CB;
    static_assert(BLOCK >= NCLS, "one thread per class");
    constexpr int TILE = BLOCK * ITEMS;
    __shared__ TO stage[TILE];
    __shared__ unsigned cnt[NCLS];
    __shared__ unsigned bstart[NCLS];
    __shared__ uint64_t gbase[NCLS];
    __shared__ unsigned scan_tmp[BLOCK / WAVE + 1];
    const unsigned tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    const uint64_t remain = n - base;
    const unsigned count = remain < (uint64_t)TILE ? (unsigned)remain : (unsigned)TILE;
    for (int i = tid; i < NCLS; i += BLOCK) cnt[i] = 0;
    __syncthreads();
    TO key[ITEMS], val[ITEMS];
    unsigned slot[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned loc = tid + i * BLOCK;
        if (loc < count) { key[i] = (TO)((uint64_t)key_in[base + loc] - koff); val[i] = (TO)((uint64_t)val_in[base + loc] - (SUB1 ? 1u : 0u)); }
        else { key[i] = 0; val[i] = 0; }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned loc = tid + i * BLOCK;
        const unsigned d = (unsigned)(key[i] >> shift) & (NCLS - 1);
        slot[i] = loc < count ? atomicAdd(&cnt[d], 1u) : 0u;
    }
    __syncthreads();
    // all keys of a tile share the bits above shift + CB (tiles never straddle a parent bucket)
    const unsigned tot = tid < NCLS ? cnt[tid] : 0u;
    unsigned total;
    const unsigned bs = block_scan_exclusive<BLOCK, unsigned>(tot, OpSum(), 0u, scan_tmp, &total);
    if (tid < NCLS) {
        bstart[tid] = bs;
        if (tot) {
            const uint64_t parent = ((uint64_t)key_in[base] - koff) >> shift >> CB;   // same for the whole tile
            const uint64_t g = (parent << CB) | tid;
            const unsigned at = atomicAdd(&cursors[g], tot);
            gbase[tid] = (g << shift) + at - bs;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned d = (unsigned)(key[i] >> shift) & (NCLS - 1);
        slot[i] += bstart[d];
        if (tid + i * BLOCK < count) stage[slot[i]] = key[i];
    }
    __syncthreads();
    // (32-bit pairs exist for at most 2^32 positions: their places fit 32 bits, sixteen registers less per thread)
    typedef typename std::conditional<sizeof(TO) == 4, uint32_t, uint64_t>::type DT;
    DT dest[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (p < count) {
            const TO x = stage[p];
            dest[j] = (DT)(gbase[(unsigned)(x >> shift) & (NCLS - 1)] + p);
            key_out[dest[j]] = x;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (tid + i * BLOCK < count) stage[slot[i]] = val[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (p < count) val_out[dest[j]] = stage[p];
    }
}

// one workgroup per window of 2^INV_WINDOW_BITS destinations (SUB1: the value's -1 is still to be applied)
// WB: window bits.  The window is held in LDS in the narrower of the two entry types (32-bit pairs widened into 64-bit
// results keep 4 bytes per entry there: 2^14 entries = 64 KiB).
template <typename TI, typename TO, int BLOCK, bool SUB1 = true, int WB = INV_WINDOW_BITS>
__global__ __launch_bounds__(BLOCK) void window_scatter_kernel(const TI* __restrict__ key, const TI* __restrict__ val,
                                                               uint64_t n, TO* __restrict__ out) {
    constexpr unsigned W = 1u << WB;
    __shared__ TI win[W];
    const uint64_t base = (uint64_t)blockIdx.x * W;
    const uint64_t remain = n - base;
    const unsigned count = remain < (uint64_t)W ? (unsigned)remain : W;
    for (unsigned p = threadIdx.x; p < count; p += BLOCK) win[(unsigned)(key[base + p]) & (W - 1)] = (TI)(val[base + p] - (SUB1 ? 1u : 0u));
    __syncthreads();
    for (unsigned p = threadIdx.x; p < count; p += BLOCK) out[base + p] = (TO)win[p];
}

// The same two kernels for pairs packed into one 64-bit entry (position in the low half, rank in the high half; at most
// 2^32 positions): one array, one staging round, runs of 64 to 128 bytes where the two-array form writes two of 32 to 64.
// FIRST = 1: the level reads (SA, bucket id) words and packs (position - koff, id - 1).
// FIRST = 2: the level makes up the rank REQUESTS of a refinement round (the B2 fetch of suffix_array.hpp:972-996 batched by address,
// as psac batches it by owner, bulk_rma.hpp:20-49): list entry j asks for the rank of text position q = SA[pos[j]] + h and carries the
// number of its bucket in the sort key, (q | number << 32); key_in = SA, val_in = pos, req_ord = the numbers.  A suffix with fewer than h
// characters left (q >= n_text: rank 0, suffix_array.hpp:1010-1016) travels under q - n_text with bit 63 set: those addresses lie in [0, h),
// the others in [h, n_text), so no two requests share one and a class of 2^k addresses never receives more than 2^k records.
template <typename TI, int BLOCK, int ITEMS, int FIRST, int CB>
__global__ __launch_bounds__(BLOCK) void partition_packed_kernel(const TI* __restrict__ key_in, const TI* __restrict__ val_in,
                                                                 const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t n,
                                                                 unsigned shift, unsigned* __restrict__ cursors, uint64_t koff,
                                                                 const unsigned* __restrict__ in_counts = nullptr, unsigned in_shift = 0,
                                                                 const uint32_t* __restrict__ req_ord = nullptr, uint64_t req_h = 0, uint64_t n_text = 0,
                                                                 const ulonglong2* __restrict__ skip_tiles = nullptr, unsigned skip_per_tile = 0) {
    // skip_tiles (the ISA entries of a split round, heavy_keys.hpp): this tile covers skip_per_tile scan tiles of the list; when all of them
    // lie inside heavy runs that keep their ranks (HeavyView::tile_b) there is nothing in it to store
    // in_counts (levels after the first, pairs of a SUBSET of the positions: the ISA update of a refinement round): the class regions of
    // the level before are filled to in_counts[class] only (class = index >> in_shift; a tile lies inside one region)
    constexpr int NCLS = 1 << CB;
    static_assert(BLOCK >= NCLS, "one thread per class");
    constexpr int TILE = BLOCK * ITEMS;
    __shared__ uint64_t stage[TILE];
    __shared__ unsigned cnt[NCLS];
    __shared__ unsigned bstart[NCLS];
    __shared__ uint64_t gbase[NCLS];
    __shared__ unsigned scan_tmp[BLOCK / WAVE + 1];
    const unsigned tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    uint64_t remain = n - base;
    if (in_counts) {
        const uint64_t filled = in_counts[base >> in_shift], into = base & ((1ull << in_shift) - 1);
        if (filled <= into) return;                  // (the same for every thread of the workgroup)
        remain = filled - into < remain ? filled - into : remain;
    }
    const unsigned count = remain < (uint64_t)TILE ? (unsigned)remain : (unsigned)TILE;
    if (skip_tiles) {
        bool all = true;
        for (unsigned q = 0; q < skip_per_tile; ++q) {
            const uint64_t st = (uint64_t)blockIdx.x * skip_per_tile + q;
            if (st * (TILE / skip_per_tile) >= n) break;
            const ulonglong2 tb = skip_tiles[st];
            all = all && tb.y != ~0ull && (tb.y & HEAVY_VIEW_KEEP) != 0;
        }
        if (all) return;                             // (the same for every thread of the workgroup)
    }
    for (int i = tid; i < NCLS; i += BLOCK) cnt[i] = 0;
    __syncthreads();
    uint64_t rec[ITEMS];
    unsigned slot[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned loc = tid + i * BLOCK;
        if (loc < count) {
            if (FIRST == 2) {
                const uint64_t sa = (uint64_t)key_in[(uint64_t)val_in[base + loc]], q = sa + req_h;
                const uint64_t num = (uint64_t)req_ord[base + loc] << 32;
                rec[i] = q < n_text ? (q | num) : ((q - n_text) | num | (1ull << 63));
            }
            else if (FIRST) rec[i] = (uint64_t)(uint32_t)((uint64_t)key_in[base + loc] - koff) | ((uint64_t)(uint32_t)((uint64_t)val_in[base + loc] - 1u) << 32);
            else rec[i] = in[base + loc];
        } else rec[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned loc = tid + i * BLOCK;
        const unsigned d = ((uint32_t)rec[i] >> shift) & (NCLS - 1);
        slot[i] = loc < count ? atomicAdd(&cnt[d], 1u) : 0u;
    }
    __syncthreads();
    const unsigned tot = tid < NCLS ? cnt[tid] : 0u;
    unsigned total;
    const unsigned bs = block_scan_exclusive<BLOCK, unsigned>(tot, OpSum(), 0u, scan_tmp, &total);
    if (tid < NCLS) {
        bstart[tid] = bs;
        if (tot) {
            // all keys of a tile share the bits above shift + CB (tiles never straddle a parent bucket)
            // (the requests of a round enter at the top level: no parent bucket)
            const uint64_t first_key = FIRST == 2 ? (uint64_t)0 : FIRST ? (uint64_t)key_in[base] - koff : (uint64_t)(uint32_t)in[base];
            const uint64_t g = ((first_key >> shift >> CB) << CB) | tid;
            const unsigned at = atomicAdd(&cursors[g], tot);
            gbase[tid] = (g << shift) + at - bs;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned d = ((uint32_t)rec[i] >> shift) & (NCLS - 1);
        if (tid + i * BLOCK < count) stage[slot[i] + bstart[d]] = rec[i];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (p < count) {
            const uint64_t x = stage[p];
            out[gbase[((uint32_t)x >> shift) & (NCLS - 1)] + p] = x;
        }
    }
}

template <typename TO, int BLOCK, int WB>
__global__ __launch_bounds__(BLOCK) void window_scatter_packed_kernel(const uint64_t* __restrict__ pairs, uint64_t n, TO* __restrict__ out) {
    constexpr unsigned W = 1u << WB;
    __shared__ uint32_t win[W];
    const uint64_t base = (uint64_t)blockIdx.x * W;
    const uint64_t remain = n - base;
    const unsigned count = remain < (uint64_t)W ? (unsigned)remain : W;
    for (unsigned p = threadIdx.x; p < count; p += BLOCK) { const uint64_t x = pairs[base + p]; win[(uint32_t)x & (W - 1)] = (uint32_t)(x >> 32); }
    __syncthreads();
    for (unsigned p = threadIdx.x; p < count; p += BLOCK) out[base + p] = (TO)win[p];
}

// The last step of the ISA update of a refinement round: window w holds counts[w] (position | value << 32) pairs at pairs[w << WB ...].
// The window of `out` is brought into LDS unless every entry of it has a pair, the pairs are applied there, and the window goes back in
// whole lines (values below 2^32: texts of at most 2^32 characters).  (Storing the pairs straight into `out` -- random 8-byte stores
// inside 128 KiB -- took 4 x as long.)
template <typename TO, int BLOCK, int WB>
__global__ __launch_bounds__(BLOCK) void window_store_sparse_kernel(const uint64_t* __restrict__ pairs, const unsigned* __restrict__ counts, uint64_t n,
                                                                    TO* __restrict__ out) {
    constexpr unsigned W = 1u << WB;
    __shared__ uint32_t win[W];
    const uint64_t base = (uint64_t)blockIdx.x << WB;
    const unsigned count = counts[blockIdx.x];
    if (count == 0) return;
    const uint64_t remain = n - base;
    const unsigned wn = remain < (uint64_t)W ? (unsigned)remain : W;
    if (count < wn) {
        for (unsigned p = threadIdx.x; p < wn; p += BLOCK) win[p] = (uint32_t)out[base + p];
        __syncthreads();
    }
    for (unsigned p = threadIdx.x; p < count; p += BLOCK) { const uint64_t x = pairs[base + p]; win[(uint32_t)x & (W - 1)] = (uint32_t)(x >> 32); }
    __syncthreads();
    for (unsigned p = threadIdx.x; p < wn; p += BLOCK) out[base + p] = (TO)win[p];
}

// The last step of the B2 fetch through partition levels (partition_packed_kernel<..., 2, ...>): window w holds counts[w] requests
// (q | number << 32) for text positions q of its 2^WB entries of ISA.  The window of ISA comes into LDS in whole lines (ranks below 2^32:
// texts of at most 2^32 characters) unless only a few of its entries are asked for, and every request leaves as a record of the round's
// sort: key (number << kb2 | rank + 1) and the suffix q - h as a 32-bit entry, windows back to back (offs = exclusive scan of counts).
// The records of a bucket are no longer neighbours -- the sort that follows orders by (number, rank) anyway -- and equal keys
// may come out in any order: they stay one bucket, whose inner order no result depends on.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void window_offsets_kernel(const unsigned* __restrict__ counts, uint64_t nwin, uint64_t* __restrict__ offs) {
    __shared__ uint64_t scan_tmp[BLOCK / WAVE + 1];
    const uint64_t per = (nwin + BLOCK - 1) / BLOCK, lo = (uint64_t)threadIdx.x * per, hi = lo + per < nwin ? lo + per : nwin;
    uint64_t mine = 0;
    for (uint64_t w = lo; w < hi; ++w) mine += counts[w];
    uint64_t total;
    uint64_t run = block_scan_exclusive<BLOCK, uint64_t>(mine, OpSum(), (uint64_t)0, scan_tmp, &total);
    for (uint64_t w = lo; w < hi; ++w) { offs[w] = run; run += counts[w]; }
    if (threadIdx.x == 0) offs[nwin] = total;
}

template <typename T, int BLOCK, int WB>
__global__ __launch_bounds__(BLOCK) void window_gather_kernel(const uint64_t* __restrict__ pairs, const unsigned* __restrict__ counts,
                                                              const uint64_t* __restrict__ offs, uint64_t n, uint64_t h, const T* __restrict__ ISA,
                                                              unsigned kb2, T* __restrict__ K1, uint32_t* __restrict__ V32,
                                                              unsigned long long* __restrict__ summary) {
    constexpr unsigned W = 1u << WB;
    __shared__ uint32_t win[W];
    const uint64_t base = (uint64_t)blockIdx.x << WB;
    const unsigned count = counts[blockIdx.x];
    const uint64_t remain = n - base;
    const unsigned wn = remain < (uint64_t)W ? (unsigned)remain : W;
    const bool staged = count * 4u >= wn;            // (the same for every thread of the workgroup)
    if (staged) {
        for (unsigned p = threadIdx.x; p < wn; p += BLOCK) win[p] = (uint32_t)ISA[base + p];
        __syncthreads();
    }
    const uint64_t o = offs[blockIdx.x];
    T o1 = 0, a1 = ~(T)0;
    for (unsigned p = threadIdx.x; p < count; p += BLOCK) {
        const uint64_t x = pairs[base + p];
        const uint32_t q = (uint32_t)x;
        const bool beyond = (x >> 63) != 0;
        const uint64_t num = (x >> 32) & 0x7FFFFFFFull;
        uint64_t b2 = 0;
        if (!beyond) b2 = (staged ? (uint64_t)win[q & (W - 1)] : (uint64_t)ISA[q]) + 1;
        const T kk = (T)((num << kb2) | b2);
        K1[o + p] = kk;
        V32[o + p] = (uint32_t)(beyond ? (uint64_t)q + n - h : (uint64_t)q - h);
        o1 |= kk; a1 &= kk;
    }
    key_summary_add<T>(summary, o1, a1, (T)0, ~(T)0);
}

// ------------------------------------------------------------------ K12
// ids: bucket ids of `cnt` consecutive list entries (SA order).  An entry is
// still active when it shares its id with a neighbour (suffix_array.hpp:925-965).
// pos_in == nullptr means list entry j sits at SA position j.  offset[tile] is the
// exclusive scan of the per-tile active counts the rebucket kernels produced.
// EMIT: also writes ids[e] and payload[e] of every active entry to out_id / out_payload (list order).
// Bucket ids of the tiles rebucket_first_kernel left out (lazy_ids): a tile without unresolved suffixes holds singleton buckets only,
// id = position + 1.  offset[]: exclusive scan of the per-tile counts of unresolved suffixes, *total: their sum.
template <typename T, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void fill_resolved_ids_kernel(T* __restrict__ ids, uint64_t n, const uint64_t* __restrict__ offset, const uint64_t* __restrict__ total) {
    constexpr int TILE = BLOCK * ITEMS;
    const unsigned tile = blockIdx.x;
    const uint64_t cnt = (tile + 1 < gridDim.x ? offset[tile + 1] : *total) - offset[tile];
    if (cnt != 0) return;
    const uint64_t base = (uint64_t)tile * TILE;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint64_t e = base + (uint64_t)i * BLOCK + threadIdx.x;
        if (e < n) ids[e] = (T)(e + 1);
    }
}

template <typename T, int BLOCK, int ITEMS, bool EMIT = false>
__global__ __launch_bounds__(BLOCK) void compact_active_kernel(
    const T* __restrict__ ids, const T* __restrict__ pos_in, uint64_t cnt, T* __restrict__ pos_out,
    const uint64_t* __restrict__ offset, uint64_t pos_off, T prev_id, T next_id, unsigned shift = 0,
    const T* __restrict__ payload = nullptr, T* __restrict__ out_id = nullptr, T* __restrict__ out_payload = nullptr,
    const uint64_t* __restrict__ unf_offset = nullptr, uint32_t* __restrict__ ord_out = nullptr, int payload32 = 0) {
    // payload32 (EMIT, 64-bit words): the payloads leave as 32-bit entries (suffixes of a text of at most 2^32 characters)
    // ord_out (with unf_offset = exclusive scan of the per-tile counts of buckets with more than one member): every list entry's bucket
    // counted from 0 in list order -- the bucket's number in the one-word sort keys of the next round (gather_keys_kernel)
    // shift: only the bits above `shift` of an id count (ties of a prefix sort by the leading bits)
    // prev_id / next_id: bucket id of the list entry just before / after this block (0 = none);
    // pos_off: SA position of entry 0 when pos_in is null
    constexpr int TILE = BLOCK * ITEMS;
    __shared__ unsigned scan_tmp[BLOCK / WAVE + 1];
    const unsigned tile = blockIdx.x;
    const uint64_t e0 = (uint64_t)tile * TILE + (uint64_t)threadIdx.x * ITEMS;

    __shared__ T xp[sizeof(T) == 8 ? (BLOCK / WAVE) * XRUN_WORDS<ITEMS>::N : 1];          // runs through whole rows of a wave (dev_common.hpp: load_run_x)
    T* const xw = xp + (threadIdx.x / WAVE) * XRUN_WORDS<ITEMS>::N;
    T v[ITEMS + 2];                    // ids[e0-1 .. e0+ITEMS]; ids are >= 1, so 0 never matches
    T raw[EMIT ? ITEMS : 1];
    {
        T mid[ITEMS];
        load_run_x<T, ITEMS>(ids, e0, cnt, mid, (T)0, xw);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { v[j + 1] = mid[j]; if (EMIT) raw[EMIT ? j : 0] = mid[j]; }
        v[0] = (e0 >= 1 && e0 - 1 < cnt) ? ids[e0 - 1] : (e0 == 0 ? prev_id : (T)0);
        v[ITEMS + 1] = (e0 + ITEMS < cnt) ? ids[e0 + ITEMS] : (e0 + ITEMS == cnt ? next_id : (T)0);
        if (e0 < cnt && e0 + ITEMS > cnt) v[(unsigned)(cnt - e0) + 1] = next_id;   // block ends inside this run
        if (shift) {
#pragma unroll
            for (int j = 0; j < ITEMS + 2; ++j) v[j] >>= shift;
        }
    }
    unsigned act = 0, nact = 0, uh = 0;          // uh: active entries that open their bucket
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const bool in = (e0 + j) < cnt;
        // (the very first entry opens its bucket whatever stands before it: ids of a prefix sort may be 0, like the "none" before them)
        if (in && (v[j + 1] == v[j] || v[j + 1] == v[j + 2])) { act |= 1u << j; ++nact; if (v[j + 1] != v[j] || (e0 + j == 0 && prev_id == 0)) uh |= 1u << j; }
    }
    unsigned agg;
    static_assert(TILE < (1 << 16), "active entries and bucket heads of a tile scanned in one word");
    const unsigned both_excl = block_scan_exclusive<BLOCK, unsigned>(nact | ((unsigned)__builtin_popcount(uh) << 16), OpSum(), 0u, scan_tmp, &agg);
    const unsigned excl = both_excl & 0xFFFFu;
    if ((agg & 0xFFFFu) == 0) return;
    uint64_t o = offset[tile] + excl;
    uint64_t ord = ord_out ? unf_offset[tile] + (both_excl >> 16) : 0;          // buckets opened before this thread's first entry
    T ps[ITEMS];
    if (pos_in) load_run_x<T, ITEMS>(pos_in, e0, cnt, ps, (T)0, xw);
    T pl[EMIT ? ITEMS : 1];
    if (EMIT && __ballot(nact != 0)) {          // (the whole wave or none of it: the rows are read by all its lanes)
        T tmp[ITEMS];
        load_run_x<T, ITEMS>(payload, e0, cnt, tmp, (T)0, xw);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) pl[EMIT ? j : 0] = tmp[j];
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if (act & (1u << j)) {
            if (EMIT) {
                out_id[o] = raw[EMIT ? j : 0];
                if (payload32) reinterpret_cast<uint32_t*>(out_payload)[o] = (uint32_t)pl[EMIT ? j : 0]; else out_payload[o] = pl[EMIT ? j : 0];
            }
            if (uh & (1u << j)) ++ord;
            if (ord_out) ord_out[o] = (uint32_t)(ord - 1);
            pos_out[o++] = pos_in ? ps[j] : (T)(pos_off + e0 + j);
        }
    }
}

// per-tile number of active entries (same predicate as compact_active_kernel), for callers that
// do not get the counts from a rebucket kernel
template <typename T, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void count_active_kernel(const T* __restrict__ ids, uint64_t cnt, T prev_id, T next_id,
                                                             uint64_t* __restrict__ n_active, unsigned shift = 0, uint64_t* __restrict__ n_unf = nullptr) {
    // n_unf (optional): per tile, the buckets with more than one member that start in it
    constexpr int TILE = BLOCK * ITEMS;
    __shared__ unsigned red_tmp[BLOCK / WAVE + 1];
    __shared__ T xp[sizeof(T) == 8 ? (BLOCK / WAVE) * XRUN_WORDS<ITEMS>::N : 1];
    T* const xw = xp + (threadIdx.x / WAVE) * XRUN_WORDS<ITEMS>::N;
    const uint64_t e0 = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * ITEMS;
    T v[ITEMS + 2];
    {
        T mid[ITEMS];
        load_run_x<T, ITEMS>(ids, e0, cnt, mid, (T)0, xw);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) v[j + 1] = mid[j];
        v[0] = (e0 >= 1 && e0 - 1 < cnt) ? ids[e0 - 1] : (e0 == 0 ? prev_id : (T)0);
        v[ITEMS + 1] = (e0 + ITEMS < cnt) ? ids[e0 + ITEMS] : (e0 + ITEMS == cnt ? next_id : (T)0);
        if (e0 < cnt && e0 + ITEMS > cnt) v[(unsigned)(cnt - e0) + 1] = next_id;
        if (shift) {
#pragma unroll
            for (int j = 0; j < ITEMS + 2; ++j) v[j] >>= shift;
        }
    }
    unsigned nact = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if ((e0 + j) < cnt && (v[j + 1] == v[j] || v[j + 1] == v[j + 2])) nact += 1u + ((v[j + 1] != v[j] || (e0 + j == 0 && prev_id == 0)) ? (1u << 16) : 0u);
    const unsigned t = block_reduce<BLOCK, unsigned>(nact, OpSum(), red_tmp);
    if (threadIdx.x == 0) { n_active[blockIdx.x] = t & 0xFFFFu; if (n_unf) n_unf[blockIdx.x] = t >> 16; }
}

// ------------------------------------------------------------------ first round in two stages
// When the first key word alone separates almost all suffixes (2^(c1 lc) >> n), the first sort
// moves only (word 1, suffix) and the few suffixes that still tie on word 1 are ordered by the
// full window afterwards.  K2rec holds word 2 in RECORD order (see record_suffix).
// word 2 of the packed window of suffix `sa`, straight from the text (same packing as key_pairs_kernel)
// word 1 of the packed window of suffix `sa`
template <typename T>
__device__ __forceinline__ T window_word1(const uint8_t* __restrict__ text, uint64_t n_text, const uint16_t* ctab,
                                          const KeyShape& ks, uint64_t sa) {
    return packed_chars<T>(text, n_text, ctab, ks.lc, sa, ks.c1);
}
// word 1 of suffix `sa` when its bits above lo1 are known (have): only the characters that reach below bit lo1 come from the text
template <typename T>
__device__ __forceinline__ T window_word1_low(const uint8_t* __restrict__ text, uint64_t n_text, const uint16_t* ctab,
                                              const KeyShape& ks, uint64_t sa, T have, unsigned lo1) {
    const unsigned nlow = (lo1 + ks.lc - 1) / ks.lc < ks.c1 ? (lo1 + ks.lc - 1) / ks.lc : ks.c1;      // characters with a bit below lo1
    const unsigned lowbits = nlow * ks.lc;
    const T w = packed_chars<T>(text, n_text, ctab, ks.lc, sa + (ks.c1 - nlow), nlow);
    if (lowbits >= sizeof(T) * 8) return w;
    return (T)(((have >> lowbits) << lowbits) | w);
}
template <typename T>
__device__ __forceinline__ T window_word2(const uint8_t* __restrict__ text, uint64_t n_text, const uint16_t* ctab,
                                          const KeyShape& ks, uint64_t sa) {
    return packed_chars<T>(text, n_text, ctab, ks.lc, sa + ks.c1, ks.c2);
}

// The characters of word 1 that reach below bit lo1 and all of word 2 -- one stretch of the text -- read together (at most 32
// characters: four 8-byte pieces in flight; longer stretches take the two words one after the other).  k1: `have` with its low
// characters filled in, k2: word 2.
template <typename T>
__device__ __forceinline__ void window_low_and_word2(const uint8_t* __restrict__ text, uint64_t n_text, const uint16_t* ctab, const KeyShape& ks,
                                                     uint64_t sa, T have, unsigned lo1, T& k1, T& k2) {
    const unsigned nlow = (lo1 + ks.lc - 1) / ks.lc < ks.c1 ? (lo1 + ks.lc - 1) / ks.lc : ks.c1;
    const unsigned lowbits = nlow * ks.lc;
    const uint64_t q0 = sa + ks.c1 - nlow;
    const unsigned L = nlow + ks.c2;
    if (L <= 32 && q0 + 32 <= n_text) {
        uint64_t x[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { x[c] = 0; if ((unsigned)(8 * c) < L) __builtin_memcpy(&x[c], text + q0 + 8 * c, 8); }
        T lo = 0, w2 = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned at = 8 * c + i;
                const T code = (T)ctab[(unsigned)(x[c] >> (8 * i)) & 255u];
                if (at < nlow) lo = (T)(lo << ks.lc) | code;
                else if (at < L) w2 = (T)(w2 << ks.lc) | code;
            }
        }
        k1 = lowbits >= sizeof(T) * 8 ? lo : (T)(((have >> lowbits) << lowbits) | lo);
        k2 = w2;
    } else {
        k2 = window_word2<T>(text, n_text, ctab, ks, sa);
        k1 = window_word1_low<T>(text, n_text, ctab, ks, sa, have, lo1);
    }
}

// Both words of the packed window of the suffixes q[j] (global positions) out of a rank's text block with its halo
// (text[0] = position off, text_len = block + 2k characters, zero beyond the end of the whole text): what the owner of a
// position answers when another rank asks for the window of a suffix that ties on the leading bits (multi.hpp).
template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void window_at_kernel(const uint8_t* __restrict__ text, uint64_t text_len, uint64_t off, const T* __restrict__ q,
                                                          uint64_t cnt, CodeTable tab, KeyShape ks, T* __restrict__ W1, T* __restrict__ W2) {
    __shared__ uint16_t ctab[256];
    for (int i = threadIdx.x; i < 256; i += BLOCK) ctab[i] = tab.c[i];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const uint64_t i = (uint64_t)q[j] - off;
        W1[j] = window_word1<T>(text, text_len, ctab, ks, i);
        W2[j] = window_word2<T>(text, text_len, ctab, ks, i);
    }
}

// Stage 2, common case: every group of suffixes that tie on the leading bits of word 1 is tiny.
// The thread that owns the first member of a group of at most G loads the group, fetches word 2 of
// each member from the text, orders the group by (word 1, word 2) with a stable odd-even
// transposition network and writes it back in place (S2 is filled for tied positions only).
// Groups longer than G are left alone and counted in big[0]; the caller then falls back to the
// compaction + radix path for all ties.
// FROM_ARRAY: word 2 of every record is already in S2 (records that carried both words through the prefix sort, the
// multi-GPU path); it is read from there instead of from the text and rewritten in the new order.
template <typename T, int BLOCK, int ITEMS, int G, bool FROM_ARRAY = false>
__global__ __launch_bounds__(BLOCK) void tie_resolve_kernel(T* __restrict__ S1, T* __restrict__ SA, T* __restrict__ S2,
                                                            uint64_t n, unsigned lo1, const uint8_t* __restrict__ text,
                                                            uint64_t n_text, CodeTable tab, KeyShape ks,
                                                            unsigned long long* __restrict__ big, bool packed = false) {
    constexpr int TILE = BLOCK * ITEMS;
    __shared__ uint16_t ctab[256];
    __shared__ unsigned leaders[TILE / 2 + 1];     // tile-relative position of the first member of every group
    __shared__ unsigned n_leaders;
    for (int i = threadIdx.x; i < 256; i += BLOCK) ctab[i] = tab.c[i];
    if (threadIdx.x == 0) n_leaders = 0;
    __syncthreads();
    // pass 1 (streaming): find the groups that start in this tile
    // (Tried: the tile in pieces of BLOCK x 8 records read through whole rows of a wave as in the rebucket kernels (load_run_x):
    //  13.6 against 10.7 ms at 2^32 -- the thirty-two-record runs keep sixteen 16-byte loads per thread in flight, the pieces do not.)
    const uint64_t t0 = (uint64_t)blockIdx.x * TILE;
    const uint64_t e0 = t0 + (uint64_t)threadIdx.x * ITEMS;
    if (e0 < n) {
        T v[ITEMS + 2];                 // leading bits of S1[e0-1 .. e0+ITEMS]
        T mid[ITEMS];
        load_run<T, ITEMS>(S1, e0, n, mid, (T)0);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) v[j + 1] = mid[j] >> lo1;
        v[0] = e0 ? (T)(S1[e0 - 1] >> lo1) : (T)0;
        v[ITEMS + 1] = e0 + ITEMS < n ? (T)(S1[e0 + ITEMS] >> lo1) : (T)0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint64_t e = e0 + j;
            const bool start = (e == 0) || v[j] != v[j + 1];
            if (e + 1 < n && start && v[j + 2] == v[j + 1]) leaders[atomicAdd(&n_leaders, 1u)] = (unsigned)(e - t0);
        }
    }
    __syncthreads();
    // pass 2: one thread per group, all loads of a step issued together
    // (a long group met by anybody sends ALL ties through the caller's radix path: what this kernel would still order is thrown away)
    const unsigned ng = __hip_atomic_load(big, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0u : n_leaders;
    for (unsigned g = threadIdx.x; g < ng; g += BLOCK) {
        const uint64_t e = t0 + leaders[g];
        T k1[G + 1], sa[G];
#pragma unroll
        for (int i = 0; i <= G; ++i) k1[i] = e + i < n ? S1[e + i] : (T)0;
#pragma unroll
        for (int i = 0; i < G; ++i) sa[i] = e + i < n ? SA[e + i] : (T)0;
        const T key = k1[0] >> lo1;
        unsigned len = 1;
#pragma unroll
        for (int i = 1; i <= G; ++i) if (len == (unsigned)i && e + i < n && (T)(k1[i] >> lo1) == key) len = i + 1;
        if (len > (unsigned)G) { atomicAdd(big, 1ull); continue; }
        T k2[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if ((unsigned)i < len) {
                k2[i] = FROM_ARRAY ? S2[e + i] : window_word2<T>(text, n_text, ctab, ks, (uint64_t)sa[i]);
                // packed payload (radix.hpp: VN 3 .. 6): the bits of word 1 below the sorted prefix went to the payload
                // (the bits above lo1 are all there: only the characters that reach below the prefix are read)
                if (!FROM_ARRAY && packed) k1[i] = window_word1_low<T>(text, n_text, ctab, ks, (uint64_t)sa[i], k1[i], lo1);
            } else { k1[i] = ~(T)0; k2[i] = ~(T)0; }
        }
        // adjacent exchanges of strictly descending neighbours only: stable
        bool moved = false;                 // a group that is already in order writes back word 2 only
#pragma unroll
        for (int r = 0; r < G; ++r) {
#pragma unroll
            for (int i = r & 1; i + 1 < G; i += 2) {
                const bool sw = k1[i] > k1[i + 1] || (k1[i] == k1[i + 1] && k2[i] > k2[i + 1]);
                moved |= sw;
                const T a1 = sw ? k1[i + 1] : k1[i], b1 = sw ? k1[i] : k1[i + 1];
                const T a2 = sw ? k2[i + 1] : k2[i], b2 = sw ? k2[i] : k2[i + 1];
                const T a3 = sw ? sa[i + 1] : sa[i], b3 = sw ? sa[i] : sa[i + 1];
                k1[i] = a1; k1[i + 1] = b1; k2[i] = a2; k2[i + 1] = b2; sa[i] = a3; sa[i + 1] = b3;
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if ((unsigned)i < len) {
                if ((lo1 && moved) || (!FROM_ARRAY && packed)) S1[e + i] = k1[i];
                if (!FROM_ARRAY || moved) S2[e + i] = k2[i];
                if (moved) SA[e + i] = sa[i];
            }
        }
    }
}

// tie_resolve_kernel on one-word records (OneWordView): the groups are found on (top digit, rest of the prefix); the members of a group share
// their prefix bits, so a group is put in order by rewriting the suffix fields of its records.  W1 / S2 receive both words of the window of
// every tied suffix (word 1 with the characters below the sorted prefix read from the text): rebucket_first_kernel<.., ONEW> reads them for
// exactly those records.
template <int BLOCK, int ITEMS, int G>
__global__ __launch_bounds__(BLOCK, 4) void tie_resolve_1w_kernel(uint64_t* __restrict__ R, uint64_t* __restrict__ W1, uint64_t* __restrict__ S2,
                                                               uint64_t n, OneWordView ow, const uint8_t* __restrict__ text, uint64_t n_text,
                                                               CodeTable tab, KeyShape ks, unsigned long long* __restrict__ big,
                                                               const uint8_t* __restrict__ tb = nullptr) {
    // tb (optional): the lowest byte of the prefix bits of the record at every place (written by the last pass of the sort): the groups are
    // looked for in these bytes -- neighbours that tie agree in them, others do once in 256 -- and only such neighbours' records are read
    typedef uint64_t T;
    constexpr int TILE = BLOCK * ITEMS;
    __shared__ uint16_t ctab[256];
    __shared__ unsigned leaders[TILE / 2 + 1];
    __shared__ unsigned n_leaders;
    __shared__ unsigned long long s_off[257];
    for (int i = threadIdx.x; i < 256; i += BLOCK) ctab[i] = tab.c[i];
    for (int i = threadIdx.x; i < 257; i += BLOCK) s_off[i] = ow.off[i];
    if (threadIdx.x == 0) n_leaders = 0;
    __syncthreads();
    ow.off = s_off;
    const uint64_t t0 = (uint64_t)blockIdx.x * TILE;
    const uint64_t e0 = t0 + (uint64_t)threadIdx.x * ITEMS;
    if (tb && e0 < n) {
        static_assert(ITEMS == 32, "a thread takes the 32 bytes of its places as two 16-byte pieces");
        auto starts_bucket = [&](uint64_t e) -> bool { return ow.off[onew_bucket(ow, e)] == e; };
        const uint4 q0 = *reinterpret_cast<const uint4*>(tb + e0), q1 = *reinterpret_cast<const uint4*>(tb + e0 + 16);
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        auto byte_at = [&](int j) -> unsigned { return (w[j >> 2] >> (8 * (j & 3))) & 255u; };
        const unsigned before = e0 ? (unsigned)tb[e0 - 1] : 256u, after = e0 + ITEMS < n ? (unsigned)tb[e0 + ITEMS] : 256u;
        // bit j + 1: the places e0 + j and e0 + j + 1 agree in their bytes (j = -1 .. 31)
        uint64_t cand = before == byte_at(0) ? 1ull : 0ull;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const unsigned nx = j + 1 < ITEMS ? byte_at(j + 1 < ITEMS ? j + 1 : 0) : after;
            if (byte_at(j) == nx && e0 + j + 1 < n) cand |= 1ull << (j + 1);
        }
        uint64_t tied = 0;                              // ... and their prefixes agree, inside one bucket
        while (cand) {
            const int jj = __builtin_ctzll(cand);
            cand &= cand - 1;
            const uint64_t e = e0 + (uint64_t)jj - 1;
            const T x = R[e], y = R[e + 1];
            if ((x >> ow.sfield) == (y >> ow.sfield) && !starts_bucket(e + 1)) tied |= 1ull << jj;
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (((tied >> (j + 1)) & 1ull) && !((tied >> j) & 1ull)) leaders[atomicAdd(&n_leaders, 1u)] = (unsigned)(e0 + j - t0);
    } else if (e0 < n) {
        // two neighbours tie when the rest of their prefixes agree and no bucket starts between them; the bucket table is asked only
        // where the rests agree (one record in 250 on random text), not walked along with every record
        T mid[ITEMS];
        load_run<T, ITEMS>(R, e0, n, mid, (T)0);
        const T before = e0 ? R[e0 - 1] : (T)0, after = e0 + ITEMS < n ? R[e0 + ITEMS] : (T)0;
        auto starts_bucket = [&](uint64_t e) -> bool { return ow.off[onew_bucket(ow, e)] == e; };
        bool eq_prev = e0 != 0 && (before >> ow.sfield) == (mid[0] >> ow.sfield) && !starts_bucket(e0);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint64_t e = e0 + j;
            const T x = j + 1 < ITEMS ? mid[j + 1 < ITEMS ? j + 1 : 0] : after;
            bool eq_next = e + 1 < n && (x >> ow.sfield) == (mid[j] >> ow.sfield);
            if (eq_next) eq_next = !starts_bucket(e + 1);
            if (e < n && !eq_prev && eq_next) leaders[atomicAdd(&n_leaders, 1u)] = (unsigned)(e - t0);
            eq_prev = eq_next;
        }
    }
    __syncthreads();
    // pass 2: eight lanes per group, one member each -- the members' windows come from the text side by side, a member's place in its group
    // is the number of members that sort before it (the lanes of a group ask each other), and every lane writes its own member there
    // (a long group met by anybody sends ALL ties through the caller's radix path: what this kernel would still order is thrown away)
    const unsigned ng = __hip_atomic_load(big, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0u : n_leaders;
    const T smask = (1ull << ow.sfield) - 1;
    const unsigned lane = lane_id(), m = lane & (G - 1), seg0 = lane & ~(unsigned)(G - 1);
    static_assert(G == 8, "eight lanes per group");
    for (unsigned g0 = 0; g0 < ng; g0 += BLOCK / G) {            // (the same trips for every lane of the workgroup: the lane moves below need whole waves)
        const unsigned g = g0 + threadIdx.x / G;
        const bool live = g < ng;
        const uint64_t e = t0 + (live ? leaders[g] : 0u);
        const unsigned b = onew_bucket(ow, e);
        const uint64_t bend = ow.off[b + 1];              // a group does not reach over the end of its bucket
        const T rec = (live && e + m < bend) ? R[e + m] : (T)0;
        const T beyond = (live && e + G < bend) ? R[e + G] : (T)0;
        const T key = shfl<T>(rec >> ow.sfield, (int)seg0);
        const bool same = live && e + m < bend && (rec >> ow.sfield) == key;
        const unsigned mine = (unsigned)(__ballot(same) >> seg0) & ((1u << G) - 1u);
        const unsigned len = (unsigned)__builtin_ctz(~mine);                // members: the leading run of equal prefixes
        const bool too_long = len == (unsigned)G && e + G < bend && (beyond >> ow.sfield) == key;
        const bool member = live && !too_long && m < len;
        if (live && too_long && m == 0) atomicAdd(big, 1ull);
        const T sa = rec & smask;
        T k1 = ~(T)0, k2 = ~(T)0;
        if (member) {
            window_low_and_word2<T>(text, n_text, ctab, ks, sa, (T)onew_word1(ow, b, rec), ow.lo1, k1, k2);
        }
        unsigned rank = 0;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const T o1 = shfl<T>(k1, (int)(seg0 + j)), o2 = shfl<T>(k2, (int)(seg0 + j));
            if (o1 < k1 || (o1 == k1 && (o2 < k2 || (o2 == k2 && (unsigned)j < m)))) ++rank;          // (equal windows keep their order: stable)
        }
        const bool moved = ((unsigned)(__ballot(member && rank != m) >> seg0) & ((1u << G) - 1u)) != 0;
        if (member) {
            W1[e + rank] = k1; S2[e + rank] = k2;
            if (moved) R[e + rank] = (key << ow.sfield) | sa;
        }
    }
}

// Stage 2, fallback: K1 / V are word 1 and suffix of the tied records (written by the compaction);
// fetches word 2 from the text.
template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void gather_prefix_ties_kernel(uint64_t cnt, T* __restrict__ K1, const T* __restrict__ V,
                                          const uint8_t* __restrict__ text, uint64_t n_text, CodeTable tab, KeyShape ks,
                                          T* __restrict__ K2, unsigned long long* __restrict__ summary, bool packed = false,
                                          const uint32_t* __restrict__ ord = nullptr, unsigned lo1 = 0, int v32 = 0) {
    // v32: V holds 32-bit entries
    // ord (with lo1): the tie groups counted from 0 in list order (compact_active_kernel).  The records are in the order of their
    // sorted prefixes already, so the sort that follows only needs the group's number above the bits of word 1 below the prefix:
    // K1 = ord << lo1 | low bits (fewer digits than the 64 bits of word 1; scatter_prefix_ties_kernel puts the low bits back under
    // the prefix that stands in S1).
    __shared__ uint16_t ctab[256];
    for (int i = threadIdx.x; i < 256; i += BLOCK) ctab[i] = tab.c[i];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const T lowmask = lo1 ? (T)(((T)1 << lo1) - 1) : (T)0;
    T o1 = 0, a1 = ~(T)0, o2 = 0, a2 = ~(T)0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        T k1 = K1[j], k2;
        const uint64_t sa = v32 ? (uint64_t)reinterpret_cast<const uint32_t*>(V)[j] : (uint64_t)V[j];
        if (packed && ord) window_low_and_word2<T>(text, n_text, ctab, ks, sa, k1, lo1, k1, k2);       // (its low bits are not in the record)
        else {
            k2 = window_word2<T>(text, n_text, ctab, ks, sa);
            if (packed) k1 = window_word1<T>(text, n_text, ctab, ks, sa);
        }
        if (ord) k1 = (T)(((T)ord[j] << lo1) | (k1 & lowmask));
        if (packed || ord) K1[j] = k1;
        K2[j] = k2;
        o1 |= k1; a1 &= k1; o2 |= k2; a2 &= k2;
    }
    key_summary_add<T>(summary, o1, a1, o2, a2);
}

// the tie groups are contiguous in SA order and pos[] ascends, so the sorted records go back in list order
// (S1 is rewritten as well when stage 1 left the low bits of word 1 unsorted)
template <typename T>
__global__ void scatter_prefix_ties_kernel(const T* __restrict__ pos, uint64_t cnt, const T* __restrict__ K1s,
                                           const T* __restrict__ K2s, const T* __restrict__ Vs, T* __restrict__ S1,
                                           T* __restrict__ S2, T* __restrict__ SA, unsigned merge_lo = 0) {
    // merge_lo: only the low merge_lo bits of K1s are word 1's (above them the sort key held the group's number): they go under the
    // prefix S1[p] holds -- a record comes back into its own group, whose prefix is the same at every place of it
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const T lowmask = merge_lo ? (T)(((T)1 << merge_lo) - 1) : ~(T)0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const uint64_t p = pos[j];
        if (K1s) S1[p] = merge_lo ? (T)((S1[p] & ~lowmask) | (K1s[j] & lowmask)) : K1s[j];
        S2[p] = K2s[j]; SA[p] = Vs[j];
    }
}

// ------------------------------------------------------------------ sparse B2 fetch
template <typename T>
__global__ void gather_keys_kernel(const T* __restrict__ pos, uint64_t cnt, const T* __restrict__ SA,
                                   const T* __restrict__ Bsa, const T* __restrict__ ISA, uint64_t n,
                                   uint64_t h, T* __restrict__ K1, T* __restrict__ K2, T* __restrict__ V,
                                   unsigned long long* __restrict__ summary, const T* __restrict__ slen = nullptr, unsigned kb2 = 32,
                                   const uint32_t* __restrict__ ord = nullptr) {
    // ord (with pos, K2 == nullptr): the bucket numbers of the list entries counted from 0 (compact_active_kernel) -- as many bits as the
    // buckets of the list need, so the sort runs fewer digit passes when few large buckets are left (a tandem repeat)
    // kb2 (K2 == nullptr): bits of the rank h further in the one-word key; above them the bucket's number.  With a list of positions
    // the number is DENSE: the list index of the bucket's head halved -- the members of a bucket are consecutive in SA and all of them
    // are in the list, so the head's index is j - (p - head position); buckets have at least two members, so halving keeps the numbers
    // apart and in order.  31 bits for 2^32 list entries: together with a 33-bit rank that is one 64-bit word at n = 2^32 too.
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    T o1 = 0, a1 = ~(T)0, o2 = 0, a2 = ~(T)0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const uint64_t p = pos ? (uint64_t)pos[j] : j;
        const T sa = SA[p];
        const uint64_t q = (uint64_t)sa + h;
        const T b1 = Bsa[p];
        // 1-based bucket id, 0 = past the end of the text, or of the own string (suffix_array.hpp:1010-1016)
        const bool inside = slen ? h < (uint64_t)slen[sa] : q < n;
        const T b2 = inside ? (T)(ISA[q] + 1) : (T)0;
        if (K2) { K1[j] = b1; K2[j] = b2; V[j] = sa; o1 |= b1; a1 &= b1; o2 |= b2; a2 &= b2; }
        else {
            // both keys in one 64-bit word, the suffix as a 32-bit entry (texts of at most 2^32 characters): two-word records
            const uint64_t num = ord ? (uint64_t)ord[j] : (pos ? ((j - (p - ((uint64_t)b1 - 1))) >> 1) : (uint64_t)b1);
            const T kk = (T)((num << kb2) | (uint64_t)b2);
            K1[j] = kk; reinterpret_cast<uint32_t*>(V)[j] = (uint32_t)sa; o1 |= kk; a1 &= kk;
        }
    }
    key_summary_add<T>(summary, o1, a1, o2, a2);
}

// How local is a walk through the text in SA order?  near[0] += the pairs of neighbouring SA entries (out of `samples`
// evenly spread ones) that lie within 64 text positions of each other.  Mostly near (one symbol repeated: SA is the text
// backwards): the fetch of the ranks h further and the ISA stores of a round are streams already, and sorting all n
// records in text order would only add work.  Mostly far (a tandem repeat: every bucket walks the text in strides of the
// period): the whole-round form below turns them into streams.
template <typename T>
__global__ void sa_locality_kernel(const T* __restrict__ SA, uint64_t n, uint64_t samples, unsigned long long* __restrict__ near) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned mine = 0;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < samples; s += stride) {
        // runs of 64 consecutive positions, the runs spread evenly over the array
        const uint64_t runs = (samples + 63) / 64, r = s / 64;
        const uint64_t p = (uint64_t)(((unsigned __int128)r * (n - 65)) / runs) + (s % 64);
        const uint64_t a = SA[p], b = SA[p + 1];
        mine += (a > b ? a - b : b - a) < 64 ? 1u : 0u;
    }
    if (mine) atomicAdd(near, (unsigned long long)mine);
}

// The same records for ALL suffixes in text order (what psac's doubling rounds sort, suffix_array.hpp:381-450 with
// shifting.hpp:33-122): K1[i] = id of suffix i, K2[i] = id of suffix i + h, V[i] = i -- two streaming reads of ISA
// instead of one random read per record.  Used for rounds in which almost every suffix is still unresolved.
template <typename T>
__global__ void shift_keys_kernel(const T* __restrict__ ISA, uint64_t n, uint64_t h, T* __restrict__ K1, T* __restrict__ K2,
                                  T* __restrict__ V, unsigned long long* __restrict__ summary, const T* __restrict__ slen = nullptr) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    T o1 = 0, a1 = ~(T)0, o2 = 0, a2 = ~(T)0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t q = i + h;
        const T b1 = (T)(ISA[i] + 1);
        const bool inside = slen ? h < (uint64_t)slen[i] : q < n;
        const T b2 = inside ? (T)(ISA[q] + 1) : (T)0;
        if (K2) { K1[i] = b1; K2[i] = b2; V[i] = (T)i; o1 |= b1; a1 &= b1; o2 |= b2; a2 &= b2; }
        else {
            const T kk = (T)(((uint64_t)b1 << 32) | (uint64_t)b2);
            K1[i] = kk; reinterpret_cast<uint32_t*>(V)[i] = (uint32_t)i; o1 |= kk; a1 &= kk;
        }
    }
    key_summary_add<T>(summary, o1, a1, o2, a2);
}

// ------------------------------------------------------------------ range minimum pyramid
// lvl[0] is the LCP array itself, lvl[L][i] = min(lvl[L-1][64 i .. 64 i + 63]).
constexpr int PYR_MAX = 8;
template <typename T> struct Pyramid {
    T* lvl[PYR_MAX] = {};
    uint64_t len[PYR_MAX] = {};
    int nlev = 0;
    // optional per-level helpers for range minima, rebuilt at the start of a refinement round:
    // pre[L][i] = min(lvl[L][64 (i / 64) .. i]), suf[L][i] = min(lvl[L][i .. 64 (i / 64) + 63])
    T* pre[PYR_MAX] = {};
    T* suf[PYR_MAX] = {};
};

// one wave per group of 64 entries: running minima from the left and from the right
template <typename T>
__global__ void pyramid_aux_kernel(const T* __restrict__ in, uint64_t len, T* __restrict__ pre, T* __restrict__ suf) {
    const uint64_t wave_id = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) / WAVE;
    const unsigned lane = lane_id();
    const uint64_t ngroups = (len + 63) >> 6;
    for (uint64_t c = wave_id; c < ngroups; c += nwaves) {
        const uint64_t i = c * 64 + lane;
        const T v = i < len ? in[i] : ~(T)0;
        const T p = wave_scan_inclusive<T>(v, OpMin());
        const T vr = shfl<T>(v, 63 - (int)lane);
        const T sr = wave_scan_inclusive<T>(vr, OpMin());
        const T sfx = shfl<T>(sr, 63 - (int)lane);
        if (i < len) { pre[i] = p; suf[i] = sfx; }
    }
}

// out[c] = min(in[64 c .. 64 c + 63]).  A lane reads 16 bytes, so a wave covers 1 KiB = 4 (32-bit) or 2 (64-bit) groups per load, and a group's
// minimum is folded over its 16 / 32 lanes (the form with one group per wave and one element per lane read a quarter of the bytes per
// instruction and folded over all 64 lanes: level 1 over 2^28 32-bit values 0.37 ms, a third of an ANSV pass).
template <typename T>
__global__ void pyramid_level_kernel(const T* __restrict__ in, uint64_t len_in, T* __restrict__ out,
                                     uint64_t len_out) {
    constexpr int PER = 16 / sizeof(T), LANES = 64 / PER, GPW = WAVE / LANES;       // elements per lane, lanes per group, groups per wave and step
    typedef T vec __attribute__((ext_vector_type(PER)));
    const uint64_t wave_id = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) / WAVE;
    const unsigned lane = lane_id();
    const bool vec_ok = (reinterpret_cast<uintptr_t>(in) & 15u) == 0;
    const uint64_t steps = (len_out + GPW - 1) / GPW;
    for (uint64_t s = wave_id; s < steps; s += nwaves) {
        const uint64_t c = s * GPW + lane / LANES;              // this lane's group
        const uint64_t i = c * 64 + (uint64_t)(lane % LANES) * PER;
        T v = ~(T)0;
        if (vec_ok && i + PER <= len_in) {
            const vec w = *reinterpret_cast<const vec*>(in + i);
#pragma unroll
            for (int d = 0; d < PER; ++d) v = w[d] < v ? w[d] : v;
        } else {
#pragma unroll
            for (int d = 0; d < PER; ++d) if (i + d < len_in) { const T x = in[i + d]; v = x < v ? x : v; }
        }
#pragma unroll
        for (int m = LANES / 2; m >= 1; m >>= 1) { const T o = shfl_xor<T>(v, m); v = o < v ? o : v; }
        if (lane % LANES == 0 && c < len_out) out[c] = v;
    }
}

// min over [l, r), l < r.  Each level contributes at most 63 leading and 63 trailing
// entries; they are fetched as predicated, fully unrolled batches so that the loads of a
// level are all in flight together instead of forming a dependent chain.
template <typename T>
__device__ __forceinline__ T pyramid_edge_min(const T* __restrict__ a, uint64_t lo, uint64_t hi, T m) {
    // hi - lo <= 128
#pragma unroll 1
    for (uint64_t b = lo; b < hi; b += 16) {
        T v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = (b + t < hi) ? a[b + t] : ~(T)0;
#pragma unroll
        for (int t = 0; t < 16; ++t) m = v[t] < m ? v[t] : m;
    }
    return m;
}

template <typename T>
__device__ __forceinline__ T pyramid_min(const Pyramid<T>& P, uint64_t l, uint64_t r) {
    T m = ~(T)0;
    // Every level below the top has its running minima (the rounds with many queries on one GPU in the normal layout): a level costs
    // at most two loads whose addresses follow from l and r alone, so the loads of ALL levels are asked for before the first is used --
    // one wait per query instead of one per level (a thread evaluates the queries of its records one after the other: 2^30 characters
    // of repeated reads spent two thirds of rebucket_refine_kernel in these waits).  Only a range inside one group that does not touch
    // either end of it, and what reaches the top level, is walked.
    bool all_tables = P.nlev >= 2;
#pragma unroll
    for (int L = 0; L < PYR_MAX - 1; ++L) if (L < P.nlev - 1 && P.pre[L] == nullptr) all_tables = false;
    if (all_tables) {
        T x[PYR_MAX - 1], y[PYR_MAX - 1];
        bool done = false;
        int wlev = -1; uint64_t wa = 0, wb = 0;
#pragma unroll
        for (int L = 0; L < PYR_MAX - 1; ++L) {
            x[L] = ~(T)0; y[L] = ~(T)0;
            if (!done && L < P.nlev - 1) {
                if ((l >> 6) == ((r - 1) >> 6)) {
                    if ((l & 63) == 0) x[L] = P.pre[L][r - 1];
                    else if ((r & 63) == 0) x[L] = P.suf[L][l];
                    else { wlev = L; wa = l; wb = r; }
                    done = true;
                } else {
                    if (l & 63) x[L] = P.suf[L][l];
                    if (r & 63) y[L] = P.pre[L][r - 1];
                    l = (l + 63) >> 6; r >>= 6;
                    if (l >= r) done = true;
                }
            }
        }
        if (!done) { wlev = P.nlev - 1; wa = l; wb = r; }
#pragma unroll
        for (int L = 0; L < PYR_MAX - 1; ++L) { m = x[L] < m ? x[L] : m; m = y[L] < m ? y[L] : m; }
        if (wlev >= 0) {
            const T* a = P.lvl[0];
#pragma unroll
            for (int L = 1; L < PYR_MAX; ++L) if (wlev == L) a = P.lvl[L];
            m = pyramid_edge_min<T>(a, wa, wb, m);
        }
        return m;
    }
    for (int L = 0; L < P.nlev; ++L) {
        const T* a = P.lvl[L];
        if (L == P.nlev - 1) return pyramid_edge_min<T>(a, l, r, m);
        const uint64_t lb = (l + 63) >> 6, rb = r >> 6;
        if (P.pre[L]) {
            // with the running minima of the groups a range that reaches into two groups or more costs two loads on this level however
            // short it is (a range of 100 entries used to be read entry by entry); only a range inside one group is walked
            if ((l >> 6) == ((r - 1) >> 6)) {
                if ((l & 63) == 0) { const T x = P.pre[L][r - 1]; return x < m ? x : m; }
                if ((r & 63) == 0) { const T x = P.suf[L][l]; return x < m ? x : m; }
                return pyramid_edge_min<T>(a, l, r, m);
            }
            // lb <= rb here: both partial groups are whole prefixes / suffixes of their groups
            if (l & 63) { const T x = P.suf[L][l]; m = x < m ? x : m; }
            if (r & 63) { const T x = P.pre[L][r - 1]; m = x < m ? x : m; }
        } else {
            if (r - l <= 128) return pyramid_edge_min<T>(a, l, r, m);
            m = pyramid_edge_min<T>(a, l, lb << 6, m);
            m = pyramid_edge_min<T>(a, rb << 6, r, m);
        }
        l = lb; r = rb;
        if (l >= r) return m;
    }
    return m;
}

template <typename T> __device__ __forceinline__ void atomic_min_t(T* p, T v);
template <> __device__ __forceinline__ void atomic_min_t<uint32_t>(uint32_t* p, uint32_t v) { atomicMin(p, v); }
template <> __device__ __forceinline__ void atomic_min_t<uint64_t>(uint64_t* p, uint64_t v) {
    atomicMin(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
}

template <typename T>
__device__ __forceinline__ void pyramid_set(const Pyramid<T>& P, uint64_t p, T v) {
    P.lvl[0][p] = v;
    uint64_t q = p;
    for (int L = 1; L < P.nlev; ++L) {
        q >>= 6;
        // entries only ever decrease, so a (possibly stale) value <= v proves the atomic is a no-op;
        // without this test every update of a round hammers the single top-level word
        if (P.lvl[L][q] <= v) break;
        atomic_min_t<T>(&P.lvl[L][q], v);
    }
}

// ------------------------------------------------------------------ K13 + K9-K11, later rounds
// Sorted active records (K1 = old bucket id, K2 = id of the suffix h further,
// V = suffix start) and their SA positions pos[] (ascending).  Writes back the
// refined order and ids, the LCP of every freshly split boundary, and the per-tile
// activity counts for the next round.
// DIST: this rank holds only a block of SA / Bsa / LCP (positions bd.off ...).  ISA is not
// written (the ids go to their owners afterwards), range minima are not evaluated here: every
// new boundary that needs one is appended to q_at / q_lo / q_hi (q_count = running length).
// HEAVY (one GPU, one-word keys): the sorted records come through a HeavyView (K1, K2, V unused).
template <typename T, int BLOCK, int ITEMS, bool WITH_LCP, bool DIST, bool HEAVY = false>
__global__ __launch_bounds__(BLOCK) void rebucket_refine_kernel(
    const T* __restrict__ K1, const T* __restrict__ K2, const T* __restrict__ V,
    const T* __restrict__ pos, uint64_t cnt, uint64_t n, uint64_t h, T* __restrict__ SA,
    T* __restrict__ Bsa, T* __restrict__ ISA, Pyramid<T> pyr, T* __restrict__ ids_out,
    const uint64_t* __restrict__ carry_in, uint64_t* __restrict__ n_active, uint64_t* __restrict__ n_unf,
    Boundary<T> bd, T* __restrict__ q_at, T* __restrict__ q_lo, T* __restrict__ q_hi,
    unsigned long long* __restrict__ q_count, unsigned kb2 = 32, uint64_t* __restrict__ pairs_out = nullptr,
    HeavyView<T> hv = HeavyView<T>()) {
    // pairs_out (one GPU, at most 2^32 characters): the ISA entries of the round leave as (suffix | new id - 1 << 32) pairs in list order
    // instead of one random store each; the caller takes them to their places through partition levels (construct.hpp: isa_update_by_levels)
    constexpr int TILE = BLOCK * ITEMS;
    __shared__ T scan_tmp[BLOCK / WAVE + 1];
    __shared__ unsigned red_tmp[BLOCK / WAVE + 1];
    const unsigned tile = blockIdx.x;
    const uint64_t e0 = (uint64_t)tile * TILE + (uint64_t)threadIdx.x * ITEMS;

    // K2 == nullptr: both keys of a record in one 64-bit word (K1 << kb2 | K2; texts of at most 2^32 characters, one GPU); only
    // equality of K1 matters here (a bucket's number), K2 is the rank of the suffix h further
    const bool both = HEAVY || K2 == nullptr;
    const uint64_t m2 = (~0ull) >> (64 - kb2);
    __shared__ T xp[sizeof(T) == 8 ? (BLOCK / WAVE) * XRUN_WORDS<ITEMS>::N : 1];          // runs through whole rows of a wave (dev_common.hpp: load_run_x)
    T* const xw = xp + (threadIdx.x / WAVE) * XRUN_WORDS<ITEMS>::N;
    T a1[ITEMS], a2[ITEMS], ps[ITEMS];
    // HEAVY: most tiles of a split round lie inside one heavy run and belong to rebucket_pure_kernel; the tiles here take their records
    // one by one through the view
    T sa_h[HEAVY ? ITEMS : 1];          // (HEAVY: the suffixes come with the keys)
    uint32_t hrank[HEAVY ? ITEMS : 1];  // (... and the ranks the heavy ones among them carry in ISA after the round)
    unsigned hmask = 0;
    if constexpr (HEAVY) {
        if (hv.tile_b[tile].y != ~0ull) return;        // (the same for the whole workgroup: one scalar load)
        heavy_run<T, ITEMS>(hv, e0, cnt, a1, sa_h, hrank, &hmask);
    } else
    load_run_x<T, ITEMS>(K1, e0, cnt, a1, (T)0, xw);
    if (!both) load_run_x<T, ITEMS>(K2, e0, cnt, a2, (T)0, xw);
    else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { a2[j] = (T)((uint64_t)a1[j] & m2); a1[j] = (T)((uint64_t)a1[j] >> kb2); }
    }
    if (pos) load_run_x<T, ITEMS>(pos, e0, cnt, ps, (T)0, xw);
    else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) ps[j] = (T)(e0 + j);
    }
    T p1 = 0, p2 = 0;
    if (e0 > 0 && e0 - 1 < cnt) {
        if constexpr (HEAVY) { T v_; heavy_record<T>(hv, e0 - 1, p1, v_); }
        else p1 = K1[e0 - 1];
        if (both) { p2 = (T)((uint64_t)p1 & m2); p1 = (T)((uint64_t)p1 >> kb2); } else p2 = K2[e0 - 1];
    }
    else if (e0 == 0 && bd.has_prev) { p1 = bd.prev1; p2 = bd.prev2; }
    bool next_head = true;
    if (e0 + ITEMS <= cnt && (e0 + ITEMS < cnt || bd.has_next)) {
        const bool in = e0 + ITEMS < cnt;
        T q1 = bd.next1;
        if (in) { if constexpr (HEAVY) { T v_; heavy_record<T>(hv, e0 + ITEMS, q1, v_); } else q1 = K1[e0 + ITEMS]; }
        T q2 = in ? (both ? (T)0 : K2[e0 + ITEMS]) : bd.next2;
        if (in && both) { q2 = (T)((uint64_t)q1 & m2); q1 = (T)((uint64_t)q1 >> kb2); }
        next_head = (q1 != a1[ITEMS - 1]) || (q2 != a2[ITEMS - 1]) || q2 == 0;
    }

    T id[ITEMS];
    unsigned heads = 0;
    T run = 0;
    // One GPU, at most 2^32 characters: the range minima of the new boundaries are not evaluated by the thread that finds them (a
    // thread's eight records one after the other, a few lanes of the wave busy each time, every query two dependent waits) but put
    // into a queue of the wave in LDS, four records of every lane at a time, and taken from there one per lane: all lanes busy, a
    // lane's queries a quarter as many (2^30 characters of repeated reads with mutations: rebucket_refine_kernel 246 -> 212 ms).
    constexpr int RQ_PER = 4, RQ_CAP = WAVE * RQ_PER;
    __shared__ uint32_t rq_buf[(WITH_LCP && !DIST) ? (BLOCK / WAVE) * RQ_CAP * 3 : 1];
    const bool rq_queue = WITH_LCP && !DIST && n <= (1ull << 32);
    const T p2_first = p2;
    unsigned rq = 0, rz = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint64_t e = e0 + j;
        const bool in = e < cnt;
        const bool same1 = (e > 0 || bd.has_prev) && a1[j] == p1;
        const bool head = !same1 || a2[j] != p2 || a2[j] == 0;
        if (head || !in) heads |= 1u << j;
        id[j] = (in && head) ? (T)(ps[j] + 1) : (T)0;
        if (WITH_LCP && in && same1 && head) {
            // boundary that appeared inside an old bucket (suffix_array.hpp:1457-1476)
            const uint64_t at = (uint64_t)ps[j] - bd.off;
            const T lo = p2 < a2[j] ? p2 : a2[j];
            const T hi = p2 < a2[j] ? a2[j] : p2;
            if (!DIST) {
                // (one GPU: every such boundary goes through the wave's queue below, also the ones beside a suffix that ends within h
                //  characters -- rank 0: LCP = h unless the entry is set already -- so that the walk over the pyramid and the store into
                //  it exist ONCE in the kernel: inlined per record they made 57 of its 68 KB of code)
                rq |= 1u << j;
                if (p2 == 0 || a2[j] == 0) rz |= 1u << j;
            } else if (p2 == 0 || a2[j] == 0) {
                if (pyr.lvl[0][at] == (T)n) pyramid_set<T>(pyr, at, (T)h);          // (DIST: a pyramid of level 0 only unless the caller keeps one)
            } else {
                const unsigned long long slot = atomicAdd(q_count, 1ull);
                q_at[slot] = ps[j]; q_lo[slot] = lo; q_hi[slot] = hi;
            }
        }
        p1 = a1[j]; p2 = a2[j];
        if (id[j] > run) run = id[j];
    }
    if constexpr (WITH_LCP && !DIST) {
        if (__ballot(rq != 0)) {
            // entries: place in LCP, then the two ranks - 1 as 32-bit words (texts of at most 2^32 characters); longer texts (no one-GPU
            // text is) keep their ranks as they are, in a queue of 64-bit words that takes a lane's records one at a time
            uint32_t* const qb = rq_buf + (threadIdx.x / WAVE) * (RQ_CAP * 3);
            const unsigned lane = lane_id();
            static_assert(ITEMS == 2 * RQ_PER, "the queue takes a lane's records in two parts");
            const int nparts = rq_queue ? 2 : ITEMS;
#pragma unroll 1
            for (int part = 0; part < nparts; ++part) {
                unsigned qn = 0;
#pragma unroll
                for (int jj = 0; jj < RQ_PER; ++jj) {
                    if (!rq_queue && jj) break;
                    // (a lane's record of this part, picked by selects: no register array is indexed by a loop variable)
                    const int j = rq_queue ? part * RQ_PER + jj : part;
                    const bool has = (rq >> j) & 1u;
                    const uint64_t mo = __ballot(has);
                    if (has) {
                        T cur = a2[0], pv = p2_first, pos_ = ps[0];
#pragma unroll
                        for (int t = 1; t < ITEMS; ++t) if (t == j) { cur = a2[t]; pv = a2[t - 1]; pos_ = ps[t]; }
                        const T lo = pv < cur ? pv : cur, hi = pv < cur ? cur : pv;
                        const bool zero = (rz >> j) & 1u;
                        const unsigned slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(mo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mo, 0u));
                        const uint64_t at = (uint64_t)pos_ - bd.off;
                        if (rq_queue) {
                            qb[slot * 3 + 0] = (uint32_t)at;
                            qb[slot * 3 + 1] = zero ? 0xFFFFFFFFu : (uint32_t)((uint64_t)lo - 1);          // (lo - 1 <= n - 2: never all ones)
                            qb[slot * 3 + 2] = (uint32_t)((uint64_t)hi - 1);
                        } else {
                            uint64_t* const qw = reinterpret_cast<uint64_t*>(qb);          // (64 entries of three words: half the queue's room)
                            qw[slot * 3 + 0] = at; qw[slot * 3 + 1] = zero ? ~0ull : (uint64_t)lo; qw[slot * 3 + 2] = (uint64_t)hi;
                        }
                    }
                    qn += (unsigned)__builtin_popcountll(mo);
                }
                xrun_order();
                for (unsigned i = lane; i < qn; i += WAVE) {
                    uint64_t at, lo, hi; bool zero;
                    if (rq_queue) { at = qb[i * 3 + 0]; zero = qb[i * 3 + 1] == 0xFFFFFFFFu; lo = (uint64_t)qb[i * 3 + 1] + 1; hi = (uint64_t)qb[i * 3 + 2] + 1; }
                    else { const uint64_t* const qw = reinterpret_cast<const uint64_t*>(qb); at = qw[i * 3 + 0]; zero = qw[i * 3 + 1] == ~0ull; lo = qw[i * 3 + 1]; hi = qw[i * 3 + 2]; }
                    // a boundary beside a suffix that ends within h characters: LCP = h unless set already; else h + the minimum between the ranks
                    const bool go = zero ? pyr.lvl[0][at] == (T)n : true;
                    const T m = zero ? (T)0 : pyramid_min<T>(pyr, lo, hi);
                    if (go) pyramid_set<T>(pyr, at, (T)(h + m));
                }
                xrun_order();
            }
        }
    }
    if (next_head) heads |= 1u << ITEMS;
    if (e0 < cnt && e0 + ITEMS > cnt) {
        const unsigned lastj = (unsigned)(cnt - 1 - e0);
        bool nh = true;
        if (bd.has_next) nh = (bd.next1 != a1[lastj]) || (bd.next2 != a2[lastj]) || bd.next2 == 0;
        heads = nh ? (heads | (1u << (lastj + 1))) : (heads & ~(1u << (lastj + 1)));
    }
    unsigned nact = 0, nub = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if (e0 + j < cnt) {
            const bool hd = (heads >> j) & 1u, hn = (heads >> (j + 1)) & 1u;
            if (!hd || !hn) ++nact;
            if (hd && !hn) ++nub;
        }
    }
    T agg;
    T excl = block_scan_exclusive<BLOCK, T>(run, OpMax(), (T)0, scan_tmp, &agg);
    static_assert(TILE < (1 << 16), "both counters of a tile reduced in one word");
    const unsigned both_counts = block_reduce<BLOCK, unsigned>(nact | (nub << 16), OpSum(), red_tmp);
    const unsigned tact = both_counts & 0xFFFFu, tub = both_counts >> 16;
    if (threadIdx.x == 0) { n_active[tile] = tact; n_unf[tile] = tub; }
    T carry = (T)carry_in[tile];
    if ((T)bd.base > carry) carry = (T)bd.base;
    if (excl > carry) carry = excl;
    T sa[ITEMS];
    if constexpr (HEAVY) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) sa[j] = sa_h[HEAVY ? j : 0];
    } else
    load_run_x<T, ITEMS>(V, e0, cnt, sa, (T)0, xw);
    // a wave whose 64 x ITEMS list entries are neighbours in SA too (a round in which nearly every suffix is unresolved: a tandem repeat, the
    // first rounds of repeated reads) writes SA and the ids through whole rows, like the arrays in list order, instead of entry by entry
    bool rows = false;
    uint64_t row0 = 0;
    if constexpr (sizeof(T) == 8) {
        const uint64_t wave_e0 = e0 - (uint64_t)lane_id() * ITEMS;
        const uint64_t p_first = shfl<uint64_t>((uint64_t)ps[0], 0), p_last = shfl<uint64_t>((uint64_t)ps[ITEMS - 1], WAVE - 1);
        // (pos ascends strictly -- every caller passes list positions in SA order --, so a span of 64 x ITEMS - 1 means consecutive positions;
        //  the check below costs eight compares and keeps a list that breaks the rule on the entry-by-entry path instead of writing wrong rows)
        bool consecutive = true;
#pragma unroll
        for (int j = 0; j + 1 < ITEMS; ++j) consecutive = consecutive && (uint64_t)ps[j + 1] == (uint64_t)ps[j] + 1;
        const uint64_t nxt = shfl<uint64_t>((uint64_t)ps[0], (int)((lane_id() + 1) & (WAVE - 1)));
        consecutive = consecutive && (lane_id() == WAVE - 1 || nxt == (uint64_t)ps[ITEMS - 1] + 1);
        rows = wave_e0 + (uint64_t)WAVE * ITEMS <= cnt && p_last - p_first == (uint64_t)WAVE * ITEMS - 1 && __ballot(!consecutive) == 0;
        row0 = p_first - bd.off;
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint64_t e = e0 + j;
        if (id[j] == 0) id[j] = carry; else carry = id[j];
        if (e < cnt) {
            if (!rows) {
                SA[(uint64_t)ps[j] - bd.off] = sa[j];
                Bsa[(uint64_t)ps[j] - bd.off] = id[j];
            }
            if (!DIST && ISA && !pairs_out) ISA[sa[j]] = id[j] - 1;
        }
    }
    if constexpr (sizeof(T) == 8) {
        if (rows) {
            const uint64_t rel = (uint64_t)lane_id() * ITEMS;          // (store_run_x addresses a thread's run by its first element: the wave's rows start at row0)
            store_run_x<T, ITEMS>(SA + row0, rel, (uint64_t)WAVE * ITEMS, sa, xw);
            store_run_x<T, ITEMS>(Bsa + row0, rel, (uint64_t)WAVE * ITEMS, id, xw);
        }
    }
    store_run_x<T, ITEMS>(ids_out, e0, cnt, id, xw);
    if constexpr (!DIST && sizeof(T) == 8) {
        if (pairs_out) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                uint32_t rk = (uint32_t)(id[j] - 1);
                // (split rounds: a heavy record takes the rank of its run -- the one it carries already, or that of the run's middle)
                if constexpr (HEAVY) { if (hmask & (1u << j)) rk = hrank[HEAVY ? j : 0]; }
                sa[j] = (T)((uint64_t)(uint32_t)sa[j] | ((uint64_t)rk << 32));
            }
            store_run_x<T, ITEMS>(reinterpret_cast<T*>(pairs_out), e0, cnt, sa, xw);
        }
    }
}

// rebucket_refine_kernel for the tiles of a split round that lie inside ONE heavy run (with the record before and the one after: HeavyView::tile_b) -- in a
// tandem repeat nearly all of them.  Such a tile has no bucket head, no new LCP entry and one id, the carry; what is left is moving its suffixes from the run
// to SA and writing the id three times: no ranking, no queue, a fifth of the registers of the general kernel (which runs one workgroup per CU at 190).
// Element e of every array is handled by one thread, rows of 64 x 8 bytes per wave.
template <typename T, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void rebucket_pure_kernel(HeavyView<T> hv, const T* __restrict__ pos, uint64_t cnt, T* __restrict__ SA, T* __restrict__ Bsa,
                                                              T* __restrict__ ISA, T* __restrict__ ids_out, const uint64_t* __restrict__ carry_in,
                                                              uint64_t* __restrict__ n_active, uint64_t* __restrict__ n_unf, uint64_t* __restrict__ pairs_out) {
    constexpr int TILE = BLOCK * ITEMS;
    const uint64_t tile = blockIdx.x;
    const ulonglong2 tb = hv.tile_b[tile];
    if (tb.y == ~0ull) return;
    const uint64_t base = tile * TILE, less = tb.y & (HEAVY_VIEW_KEEP - 1);
    const unsigned count = cnt - base < (uint64_t)TILE ? (unsigned)(cnt - base) : (unsigned)TILE;
    const T id = (T)carry_in[tile];
    const uint32_t rank = (uint32_t)hv.rank[tb.x >> hv.kb2];
    bool keep = (tb.y & HEAVY_VIEW_KEEP) != 0;
    if (keep) {          // (the levels leave out the entries of two scan tiles that both keep their ranks, IsaLevels::add)
        const uint64_t mate = tile ^ 1u;
        if (mate * TILE < cnt) { const ulonglong2 tm = hv.tile_b[mate]; keep = tm.y != ~0ull && (tm.y & HEAVY_VIEW_KEEP) != 0; }
    }
    T p[ITEMS]; uint32_t s[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned loc = threadIdx.x + (unsigned)i * BLOCK;
        p[i] = 0; s[i] = 0;
        if (loc < count) { p[i] = pos[base + loc]; s[i] = hv.HB[base + loc - less]; }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned loc = threadIdx.x + (unsigned)i * BLOCK;
        if (loc < count) {
            SA[(uint64_t)p[i]] = (T)s[i];
            Bsa[(uint64_t)p[i]] = id;
            ids_out[base + loc] = id;
            if (pairs_out) { if (!keep) pairs_out[base + loc] = (uint64_t)s[i] | ((uint64_t)rank << 32); }
            else if (ISA) ISA[s[i]] = id - 1;
        }
    }
    if (threadIdx.x == 0) { n_active[tile] = count; n_unf[tile] = 0; }
}

// ------------------------------------------------------------------ K14
template <typename T>
__global__ void isa_finalize_kernel(T* __restrict__ ISA, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) ISA[i] -= 1;
}

// slen[i] = off[t + 1] - i for the string t holding position i (off ascending, off[0] = 0, off[m] = n)
template <typename T>
__global__ void string_len_kernel(const uint64_t* __restrict__ off, uint64_t m, uint64_t n, T* __restrict__ slen) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t lo = 0, hi = m;                 // largest t with off[t] <= i
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
        slen[i] = (T)(off[lo + 1] - i);
    }
}

// bad[0] += offsets that break off[0] = 0, off[m] = n or strict ascent
template <int TAG>
__global__ void check_offsets_kernel(const uint64_t* __restrict__ off, uint64_t m, uint64_t n, unsigned long long* __restrict__ bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= m; t += stride) {
        const uint64_t o = off[t];
        const bool ok = (t == 0 ? o == 0 : o > off[t - 1]) && (t == m ? o == n : o < n);
        if (!ok) atomicAdd(bad, 1ull);
    }
}

// Left-branching characters (/root/reference/include/suffix_array.hpp:211-212, built there
// by :1365-1383 in the first round and carried through the range minima, par_rmq.hpp:334-481):
// Lc[i] = S[SA[i-1] + LCP[i]], the character of the left neighbour at the first mismatch, '\0'
// when that position is past the end (alphabet.hpp:168) and at i = 0.  The reference's own
// definition of the result is desa.hpp:262-264; with SA, LCP and the text resident in HBM the
// direct gather costs one pass (2w + 1 bytes streamed, one random byte read per suffix).
template <typename T>
__global__ void left_chars_kernel(const uint8_t* __restrict__ text, uint64_t n, const T* __restrict__ SA,
                                  const T* __restrict__ LCP, uint8_t* __restrict__ Lc) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint8_t ch = 0;
        if (i) {
            const uint64_t p = (uint64_t)SA[i - 1] + (uint64_t)LCP[i];
            if (p < n) ch = text[p];
        }
        Lc[i] = ch;
    }
}

// packed payload of an already sorted input (radix.hpp: VN 3 .. 6): put together again without a sort pass
template <typename T>
__global__ void unpack_payload_kernel(const T* __restrict__ k1, const void* __restrict__ hi, uint64_t n, unsigned pack, unsigned bytes, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const T mask = (T)(((uint64_t)1 << pack) - 1);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = bytes == 1 ? (uint64_t)static_cast<const uint8_t*>(hi)[i] : (uint64_t)static_cast<const uint16_t*>(hi)[i];
        out[i] = (T)((k1[i] & mask) | (T)(h << pack));
    }
}

template <typename T>
__global__ void widen32_kernel(const uint32_t* __restrict__ in, uint64_t n, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (T)in[i];
}

template <typename T>
__global__ void fill_kernel(T* __restrict__ a, uint64_t n, T v) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = v;
}

// a[g] = g, or the suffix start record g stands for in the first round (spec_n != 0)
template <typename T>
__global__ void iota_kernel(T* __restrict__ a, uint64_t n, uint64_t spec, uint64_t spec_n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        a[i] = (T)(spec_n ? record_suffix(i, spec, spec_n) : i);
}

} // namespace psacx
