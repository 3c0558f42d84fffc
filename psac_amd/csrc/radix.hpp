// radix.hpp -- rank-pair LSD radix sort for records (key1, key2, value).
//
// Stands in for idxsort_vectors() -> mxx::sort on (B1,B2,idx) tuples
// (/root/reference/include/idxsort.hpp:23-83).  The arrays stay struct-of-arrays
// in HBM; one sort is
//   * one histogram kernel over both key words producing all per-digit
//     histograms at once (LDS-staged 256-bin counters per digit), and
//   * one single-sweep scatter kernel per non-constant 8-bit digit: every tile
//     ranks its records with wave64 ballots, publishes its 256 digit counts
//     and resolves its global offsets by decoupled look-back, reorders the
//     tile through LDS and writes coalesced runs.
// Algorithmic HBM traffic: 2w bytes/record for the histogram and 6w
// bytes/record per scatter pass (w = sizeof(T)), SURVEY.md section 8(d).
#pragma once
#include "dev_common.hpp"

namespace psacx {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int MAX_PASSES = 2 * (64 / RADIX_BITS);   // two 64-bit key words

// pass p < passes_lo reads key2 (low word), the rest read key1
struct PassPlan {
    int n_pass;
    int word[MAX_PASSES];     // 0 = key1 (high), 1 = key2 (low)
    int shift[MAX_PASSES];
};

inline PassPlan make_plan(int key_bits) {
    PassPlan p;
    int per = (key_bits + RADIX_BITS - 1) / RADIX_BITS;
    p.n_pass = 0;
    for (int w = 1; w >= 0; --w)
        for (int i = 0; i < per; ++i) { p.word[p.n_pass] = w; p.shift[p.n_pass] = i * RADIX_BITS; p.n_pass++; }
    return p;
}

// --------------------------------------------------------------- histogram
struct HistArgs {
    int n_pass;
    int word[MAX_PASSES];
    int shift[MAX_PASSES];
};

template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void radix_hist_kernel(const T* __restrict__ k1,
                                                           const T* __restrict__ k2, uint64_t n,
                                                           HistArgs a,
                                                           unsigned long long* __restrict__ hist) {
    __shared__ unsigned lh[MAX_PASSES * RADIX];
    for (int i = threadIdx.x; i < a.n_pass * RADIX; i += BLOCK) lh[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const T a1 = k1[i];
        const T a2 = k2[i];
#pragma unroll 4
        for (int p = 0; p < a.n_pass; ++p) {
            const T w = a.word[p] ? a2 : a1;
            const unsigned d = (unsigned)(w >> a.shift[p]) & (RADIX - 1);
            atomicAdd(&lh[p * RADIX + d], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.n_pass * RADIX; i += BLOCK) {
        const unsigned c = lh[i];
        if (c) atomicAdd(&hist[i], (unsigned long long)c);
    }
}

// ------------------------------------------------------------ scatter pass
__device__ __forceinline__ uint64_t match_any8(unsigned d, bool valid) {
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

// T: record word type.  D: look-back descriptor word (uint32_t when n < 2^30).
// kd_*: the key word that carries this pass's digit; ko_*: the other key word;
// v_in may be null, in which case the payload is the record's global index.
template <typename T, typename D, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void radix_scatter_kernel(
    const T* __restrict__ kd_in, const T* __restrict__ ko_in, const T* __restrict__ v_in,
    T* __restrict__ kd_out, T* __restrict__ ko_out, T* __restrict__ v_out, uint64_t n, int shift,
    const unsigned long long* __restrict__ digit_base, D* __restrict__ desc,
    unsigned* __restrict__ tile_counter, unsigned* __restrict__ err) {
    constexpr int TILE = BLOCK * ITEMS;
    constexpr int NW = BLOCK / WAVE;
    static_assert(BLOCK >= RADIX, "one thread per digit needed");

    __shared__ T stage[TILE];
    __shared__ unsigned wcnt[NW * RADIX];   // per-wave digit counters -> per-wave exclusive bases
    __shared__ unsigned bstart[RADIX];      // tile-local start of each digit
    __shared__ T goff[RADIX];               // global offset of digit run minus bstart (wraps)
    __shared__ unsigned scan_tmp[NW + 1];
    __shared__ unsigned s_tile;

    const unsigned tid = threadIdx.x;
    const unsigned lane = lane_id();
    const unsigned wave = tid / WAVE;

    if (tid == 0) s_tile = atomicAdd(tile_counter, 1u);
    for (int i = tid; i < NW * RADIX; i += BLOCK) wcnt[i] = 0;
    __syncthreads();
    const unsigned tile = s_tile;
    const uint64_t base = (uint64_t)tile * TILE;
    const uint64_t remain = n - base;
    const unsigned count = remain < (uint64_t)TILE ? (unsigned)remain : (unsigned)TILE;

    // wave-striped load: record (wave, i, lane) = base + wave*64*ITEMS + i*64 + lane
    T kd[ITEMS], ko[ITEMS], vv[ITEMS];
    const unsigned wbase = wave * (WAVE * ITEMS) + lane;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned loc = wbase + i * WAVE;
        if (loc < count) {
            kd[i] = kd_in[base + loc];
            ko[i] = ko_in[base + loc];
            vv[i] = v_in ? v_in[base + loc] : (T)(base + loc);
        } else {
            kd[i] = 0; ko[i] = 0; vv[i] = 0;
        }
    }

    // rank inside the wave, round by round (keeps the sort stable)
    unsigned rank[ITEMS];
    unsigned* mycnt = wcnt + wave * RADIX;
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const bool valid = (wbase + i * WAVE) < count;
        const unsigned d = (unsigned)(kd[i] >> shift) & (RADIX - 1);
        const uint64_t m = match_any8(d, valid);
        unsigned prior = 0;
        const unsigned leader = m ? (unsigned)__builtin_ctzll(m) : 0u;
        if (valid && lane == leader) {
            prior = mycnt[d];
            mycnt[d] = prior + (unsigned)__builtin_popcountll(m);
        }
        prior = shfl<uint32_t>(prior, (int)leader);
        rank[i] = prior + (unsigned)__builtin_popcountll(m & lt);
    }
    __syncthreads();

    // per digit: exclusive bases over waves, tile total, look-back
    unsigned tot = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const unsigned c = wcnt[w * RADIX + tid];
            wcnt[w * RADIX + tid] = tot;
            tot += c;
        }
    }
    unsigned tile_total;
    unsigned bs = block_scan_exclusive<BLOCK, unsigned>(tid < RADIX ? tot : 0u, OpSum(), 0u, scan_tmp, &tile_total);
    if (tid < RADIX) {
        bstart[tid] = bs;
        D* my = desc + (uint64_t)tile * RADIX + tid;
        uint64_t excl = 0;
        if (tile == 0) {
            desc_store<D>(my, 2u, (D)tot);
        } else {
            desc_store<D>(my, 1u, (D)tot);
            long long t = (long long)tile - 1;
            while (t >= 0) {
                const D dsc = desc_wait<D>(desc + (uint64_t)t * RADIX + tid, err);
                excl += (uint64_t)(dsc & Desc<D>::MASK);
                if ((dsc >> Desc<D>::SHIFT) == 2u) break;
                --t;
            }
            desc_store<D>(my, 2u, (D)(excl + tot));
        }
        goff[tid] = (T)((uint64_t)digit_base[tid] + excl - (uint64_t)bs);
    }
    __syncthreads();

    // final tile-local position of every record
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned d = (unsigned)(kd[i] >> shift) & (RADIX - 1);
        rank[i] += bstart[d] + mycnt[d];
    }

    // move the three words through LDS one after the other
    T dest[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if ((wbase + i * WAVE) < count) stage[rank[i]] = kd[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (p < count) {
            const T x = stage[p];
            const unsigned d = (unsigned)(x >> shift) & (RADIX - 1);
            dest[j] = (T)(goff[d] + (T)p);
            kd_out[dest[j]] = x;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if ((wbase + i * WAVE) < count) stage[rank[i]] = ko[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (p < count) ko_out[dest[j]] = stage[p];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if ((wbase + i * WAVE) < count) stage[rank[i]] = vv[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (p < count) v_out[dest[j]] = stage[p];
    }
}

} // namespace psacx
