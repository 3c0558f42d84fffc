// bucket_sort.hpp -- the sort of a refinement round inside LDS, bucket by bucket.
// A refinement round (suffix_array.hpp:1092-1157: rebucket_bucket sorts every unfinished bucket by the ranks h further) sorts the list of
// unresolved suffixes by (bucket, rank h further).  The list is in SA order, so a bucket's records are neighbours before and after the
// sort: the sort only permutes inside buckets.  As long as no bucket is longer than what one workgroup holds in LDS, the global radix
// sort -- up to eight passes over all records, 32 bytes per record and pass -- is not needed: the list is cut at bucket boundaries into
// tasks of at most CAP records (window t = the buckets that start in list entries [t W, (t + 1) W)), one workgroup sorts a task by
// least-significant-digit passes that never leave LDS, and the records cross HBM once (12 bytes read, 16 written).  A bucket longer
// than CAP - W raises the flag; the caller then tries narrower windows (long tasks amortise the fixed cost of a pass -- barriers, the scan
// of the digit counters -- over more records: 2^30 characters of repeated reads with mutations 64 -> 51 ms for the four rounds it takes),
// or runs the global sort on the untouched input.
// Records: K = bucket number << kb2 | rank (gather_keys_kernel's one-word keys), V32 = the suffix as a 32-bit entry.  The order of
// equal keys is the order of the list, as the stable global sort leaves it.
#pragma once
#include "radix.hpp"

namespace psacx {

constexpr unsigned BSORT_W_WIDE = 8192, BSORT_W_NARROW = 2048;              // list entries per window
constexpr int BSORT_BLOCK = 1024, BSORT_ITEMS = 16;
constexpr unsigned BSORT_CAP = BSORT_BLOCK * BSORT_ITEMS;
constexpr unsigned BSORT_LI_BITS = 14;          // a record's place in its task rides below the key
static_assert((1u << BSORT_LI_BITS) == BSORT_CAP, "the place of a record in its task needs BSORT_LI_BITS bits");

// first list entry at or after x at which a bucket starts (cnt if none); *over is raised when none is found within CAP entries
__device__ __forceinline__ uint64_t bsort_next_head(const uint64_t* __restrict__ K, uint64_t cnt, unsigned kb2, uint64_t x, unsigned long long* over) {
    const unsigned lane = lane_id();
    for (uint64_t j0 = x; j0 < cnt; j0 += WAVE) {
        if (j0 >= x + BSORT_CAP + WAVE) { if (lane == 0) atomicAdd(over, 1ull); return j0; }
        const uint64_t j = j0 + lane;
        const bool head = j < cnt && (j == 0 || (K[j] >> kb2) != (K[j - 1] >> kb2));
        const uint64_t m = __ballot(head);
        if (m) return j0 + (unsigned)__builtin_ctzll(m);
    }
    return cnt;
}

// start[t] = first list entry of task t (t = 0 .. ntasks; start[ntasks] = cnt); *over counts the tasks longer than CAP.  One wave per task.
template <int TAG>
__global__ __launch_bounds__(256) void bucket_task_starts_kernel(const uint64_t* __restrict__ K, uint64_t cnt, unsigned kb2, uint64_t ntasks, unsigned W,
                                                                 uint64_t* __restrict__ start, unsigned long long* __restrict__ over) {
    const uint64_t t = (uint64_t)blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
    if (t > ntasks) return;
    const uint64_t a = t == ntasks ? cnt : bsort_next_head(K, cnt, kb2, t * W, over);
    if (lane_id() == 0) start[t] = a;
    if (t < ntasks) {
        const uint64_t b = t + 1 == ntasks ? cnt : bsort_next_head(K, cnt, kb2, (t + 1) * W, over);
        if (b - a > BSORT_CAP && lane_id() == 0) atomicAdd(over, 1ull);
    }
}

// One workgroup sorts task blockIdx.x.  The words live in registers between the passes, wave w holding the entries [w CH, (w + 1) CH) of
// the current order (CH = 64 x the passes' trip count, so a short task takes short passes); a pass ranks them by one digit (match_any8 and
// per-wave digit counters, as the scatter passes of radix.hpp do), puts them in their new places in LDS and takes them back in order.
template <int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void bucket_sort_lds_kernel(const uint64_t* __restrict__ K, const uint32_t* __restrict__ V32, const uint64_t* __restrict__ start,
                                                                unsigned kb2, uint64_t* __restrict__ Kout, uint64_t* __restrict__ Vout) {
    constexpr int NW = BLOCK / WAVE;
    constexpr unsigned CAP = BLOCK * ITEMS;
    static_assert(CAP == BSORT_CAP, "the tasks are cut for this capacity");
    static_assert(BLOCK >= RADIX, "one thread per digit");
    __shared__ uint64_t A[CAP];
    __shared__ unsigned wcnt[NW][RADIX];
    __shared__ unsigned dbase[RADIX];
    __shared__ unsigned scan_tmp[NW + 1];
    __shared__ unsigned long long red[2][NW];
    const uint64_t a = start[blockIdx.x], b = start[blockIdx.x + 1];
    if (b <= a || b - a > CAP) return;
    const unsigned cnt = (unsigned)(b - a);
    const unsigned tid = threadIdx.x, lane = lane_id(), w = tid / WAVE;
    const unsigned nit = (cnt + BLOCK - 1) / BLOCK;                // 1 .. ITEMS
    const unsigned CH = nit * WAVE;
    const uint64_t kmask = kb2 >= 64 ? ~0ull : ((1ull << kb2) - 1ull);
    const uint64_t first = K[a] >> kb2;                            // bucket number of the task's first record
    uint64_t word[ITEMS];
    uint64_t vor = 0, vand = ~0ull;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        word[i] = ~0ull;
        if ((unsigned)i < nit) {
            const unsigned e = w * CH + (unsigned)i * WAVE + lane;
            if (e < cnt) {
                const uint64_t k = K[a + e];
                word[i] = (((((k >> kb2) - first) << kb2) | (k & kmask)) << BSORT_LI_BITS) | e;
                vor |= word[i]; vand &= word[i];
            }
        }
    }
    // bits in which the task's words differ: a digit without any is not sorted on
    vor = wave_reduce<uint64_t>(vor, OpOr()); vand = wave_reduce<uint64_t>(vand, OpAnd());
    if (lane == 0) { red[0][w] = vor; red[1][w] = vand; }
    __syncthreads();
    vor = 0; vand = ~0ull;
#pragma unroll
    for (int i = 0; i < NW; ++i) { vor |= red[0][i]; vand &= red[1][i]; }
    const uint64_t varies = (vor ^ vand) >> BSORT_LI_BITS;
    for (unsigned shift = BSORT_LI_BITS; shift < 64 && (varies >> (shift - BSORT_LI_BITS)) != 0; shift += RADIX_BITS) {
        if (((varies >> (shift - BSORT_LI_BITS)) & (RADIX - 1)) == 0) continue;
#pragma unroll
        for (int i = 0; i < RADIX / WAVE; ++i) wcnt[w][i * WAVE + lane] = 0;
        xrun_order();
        unsigned short rk[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            rk[i] = 0;
            if ((unsigned)i < nit) {
                const unsigned d = (unsigned)(word[i] >> shift) & (RADIX - 1);
                const uint64_t peers = match_any8(d, true);
                const unsigned below = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
                const unsigned old = wcnt[w][d];
                xrun_order();
                if (below == 0) wcnt[w][d] = old + (unsigned)__popcll(peers);
                xrun_order();
                rk[i] = (unsigned short)(old + below);
            }
        }
        __syncthreads();
        unsigned tot = 0;
        if (tid < RADIX) {
            unsigned cw[NW];                    // (all reads first: they do not wait for one another)
#pragma unroll
            for (int v = 0; v < NW; ++v) cw[v] = wcnt[v][tid];
#pragma unroll
            for (int v = 0; v < NW; ++v) { wcnt[v][tid] = tot; tot += cw[v]; }
        }
        unsigned all;
        const unsigned ex = block_scan_exclusive<BLOCK, unsigned>(tot, OpSum(), 0u, scan_tmp, &all);
        if (tid < RADIX) dbase[tid] = ex;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if ((unsigned)i < nit) {
                const unsigned d = (unsigned)(word[i] >> shift) & (RADIX - 1);
                A[dbase[d] + wcnt[w][d] + rk[i]] = word[i];
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if ((unsigned)i < nit) word[i] = A[w * CH + (unsigned)i * WAVE + lane];
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if ((unsigned)i < nit) {
            const unsigned e = w * CH + (unsigned)i * WAVE + lane;
            if (e < cnt) {
                const uint64_t x = word[i] >> BSORT_LI_BITS;
                const unsigned li = (unsigned)word[i] & (BSORT_CAP - 1);
                Kout[a + e] = (((x >> kb2) + first) << kb2) | (x & kmask);
                Vout[a + e] = (uint64_t)V32[a + li];
            }
        }
}

} // namespace psacx
