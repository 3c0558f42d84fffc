// heavy_keys.hpp -- refinement rounds whose buckets are dominated by ONE rank h further.
//
// A refinement round (suffix_array.hpp:1092-1157) sorts every unfinished bucket by B2, the rank of the suffix h further.  In a repetitive
// text almost all members of a long bucket carry the SAME B2: suffix i + h is itself a member of one long unfinished bucket (a period-p
// tandem repeat: all suffixes i = r mod p with at least 2h characters left; of the n - h records of the round at h all but h), and only the
// few whose suffix h further was resolved meanwhile differ.  Larsson and Sadakane sort with a ternary split for exactly this reason; psac's
// sample sort and the radix passes of radix.hpp move every record through every digit.  Here the records of such a round are split where
// their keys are made (the window kernel of the B2 fetch through partition levels, construct.hpp: gather_by_levels):
//   * heavy: B2 equals the bucket's heavy value (the B2 of three probed members, majority) -- the suffix alone goes into the bucket's run
//            of an array in list order (per window and bucket one reservation; the order inside the run is free: equal keys stay one bucket);
//   * light: everything else -- a sort record as before, compacted; only these are radix-sorted, by (bucket number, B2);
// and the round's sorted order -- per bucket the light records below the heavy value, the heavy run, the light records above it -- is read
// where it lies by the kernels that follow (sa_kernels.hpp: HeavyView; last_head_kernel, rebucket_refine_kernel): a tile of the list inside one
// heavy run takes its suffixes straight out of the run.  A 4 GiB period-1024 tandem repeat: 26 rounds x 6 digit passes over 2^32 records become
// the sort of about one record in twenty.
// ISA.  The rank of an unresolved suffix only has to (a) be the same for all members of its bucket and (b) order the buckets -- ANY position
// inside the bucket's range of SA will do (psac writes the head's, bucketing.hpp:21-53; the final rank of a resolved suffix is its one position
// either way, and a range minimum over LCP between a position inside bucket A and one inside bucket B is the minimum between their heads: the
// entries inside unresolved buckets are still unset).  So a heavy run whose members carry a rank that still lies inside the run after the
// round keeps it -- no ISA store for any of them -- and a run that has to move takes the rank of its MIDDLE, which survives the most shrinking
// (a tandem repeat's runs lose h members at their front per round: a handful of stores per run and construction instead of one per round).
// Applies to 64-bit words, lists with at most HEAVY_MAXB buckets (their tables live in LDS) that take the partition levels (long buckets).
#pragma once
#include "sa_kernels.hpp"

namespace psacx {

constexpr unsigned HEAVY_MAXB = 4096;
constexpr unsigned HEAVY_PAD = 1;                 // stride of the per-bucket reservation counters in words: thread b of a workgroup adds to counter b, so a wave's 64 atomics go to
                                                  // eight lines and are combined per line (tools/ubench_atomic.hip: 43 - 170 G atomics/s side by side, 24 G/s on a line each)

struct HeavyTabs {
    uint64_t* bstart;          // [nb + 1] first list entry of every bucket
    uint64_t* value;           // [nb] the heavy B2 (>= 1)
    unsigned long long* eq;    // [nb * HEAVY_PAD] heavy records placed so far
    unsigned long long* less;  // [nb] light records below the heavy value
    unsigned long long* light; // [1] light records
    uint64_t* lstart;          // [nb + 1] where a bucket's light records start in the sorted light list
    uint64_t* rank;            // [nb] probe: the rank the bucket's members carry in ISA now; plan: the rank its heavy run carries after the round, bit 63
                               //      set when that is the rank it carries already (HEAVY_KEEP: no ISA entry of the run needs a store)
    static size_t words(unsigned nb) { return (size_t)(nb + 1) * 2 + (size_t)nb * (HEAVY_PAD + 3) + 8; }
};
constexpr uint64_t HEAVY_KEEP = 1ull << 63;

// first list entry whose bucket number is >= b (ord ascends along the list)
__device__ __forceinline__ uint64_t heavy_lower_bound(const uint32_t* __restrict__ ord, uint64_t cnt, uint32_t b) {
    uint64_t lo = 0, hi = cnt;
    while (lo < hi) { const uint64_t m = lo + ((hi - lo) >> 1); if (ord[m] < b) lo = m + 1; else hi = m; }
    return lo;
}

// One thread per bucket: where it starts in the list, and its heavy value -- the B2 of the members at 1/4, 1/2 and 3/4 of the bucket, the
// majority of those that have h characters left (a bucket has at most one member that has not: two such members would be equal strings).
template <typename T>
__global__ void heavy_probe_kernel(const uint32_t* __restrict__ ord, uint64_t cnt, uint32_t nb, const T* __restrict__ pos, const T* __restrict__ SA,
                                   const T* __restrict__ ISA, uint64_t n, uint64_t h, HeavyTabs ht) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    if (b == nb) { ht.bstart[nb] = cnt; ht.light[0] = 0; return; }
    const uint64_t lo = heavy_lower_bound(ord, cnt, b), hi = b + 1 == nb ? cnt : heavy_lower_bound(ord, cnt, b + 1);
    ht.bstart[b] = lo;
    const uint64_t len = hi - lo;
    uint64_t v[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const uint64_t j = lo + (len * (uint64_t)(s + 1)) / 4;
        const uint64_t q = (uint64_t)SA[(uint64_t)pos[j < hi ? j : hi - 1]] + h;
        v[s] = q < n ? (uint64_t)ISA[q] + 1 : 0;
    }
    uint64_t x = v[1];
    if (x == 0 || (v[0] == v[2] && v[0] != 0)) x = v[0] ? v[0] : v[2];
    ht.value[b] = x;                               // (0 only if the bucket is malformed; such a bucket then has no heavy records at all)
    ht.eq[(size_t)b * HEAVY_PAD] = 0; ht.less[b] = 0;
    ht.rank[b] = (uint64_t)ISA[(uint64_t)SA[(uint64_t)pos[lo]]];          // the rank every member of the bucket carries now
}

// The window kernel of gather_by_levels with the split: window w holds counts[w] requests (q | bucket number << 32), bit 63 = fewer than h
// characters left (construct.hpp).  Heavy requests leave as the suffix alone at HB[list position], light ones as sort records at LK / LV.
// LDS: the window of ISA (64 KiB) + 8 bytes per bucket (dynamic: 1024 buckets leave room for two workgroups per CU, whose waits on the
// reservations then overlap).
template <typename T, int BLOCK, int WB>
__global__ __launch_bounds__(BLOCK) void window_gather_heavy_kernel(const uint64_t* __restrict__ pairs, const unsigned* __restrict__ counts, uint64_t n, uint64_t h,
                                                                    const T* __restrict__ ISA, unsigned kb2, uint32_t nb, HeavyTabs ht, T* __restrict__ LK,
                                                                    uint32_t* __restrict__ LV, uint32_t* __restrict__ HB, unsigned long long* __restrict__ summary) {
    constexpr unsigned W = 1u << WB;
    constexpr int ITEMS = W / BLOCK;
    __shared__ uint32_t win[W];
    extern __shared__ uint32_t heavy_dyn[];
    uint32_t* const hval = heavy_dyn;            // [nb] heavy value - 1 (a rank); 0xFFFFFFFF also stands for "none" (value 0)
    uint32_t* const eqc = heavy_dyn + nb;        // [nb] heavy records of this window per bucket, then where their run starts in HB
    __shared__ unsigned nlight;
    __shared__ unsigned long long lbase;
    const uint64_t base = (uint64_t)blockIdx.x << WB;
    const unsigned count = counts[blockIdx.x];
    const uint64_t remain = n - base;
    const unsigned wn = remain < (uint64_t)W ? (unsigned)remain : W;
    const bool staged = count * 4u >= wn;
    T o1 = 0, a1 = ~(T)0;
    if (count) {            // (the same for every thread of the workgroup)
        for (unsigned b = threadIdx.x; b < nb; b += BLOCK) { hval[b] = (uint32_t)(ht.value[b] - 1); eqc[b] = 0; }
        if (threadIdx.x == 0) nlight = 0;
        if (staged) for (unsigned p = threadIdx.x; p < wn; p += BLOCK) win[p] = (uint32_t)ISA[base + p];
        __syncthreads();
        unsigned k[ITEMS];
        unsigned heavy_mask = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const unsigned p = threadIdx.x + (unsigned)i * BLOCK;
            k[i] = 0;
            bool hv = false, lt = false;
            uint32_t bb = 0;
            if (p < count) {
                const uint64_t x = pairs[base + p];
                const uint32_t q = (uint32_t)x, b = (uint32_t)(x >> 32) & 0x7FFFFFFFu;
                const bool beyond = (x >> 63) != 0;
                const uint32_t rank = beyond ? 0u : (staged ? win[q & (W - 1)] : (uint32_t)ISA[q]);
                hv = !beyond && rank == hval[b];
                if (hv && hval[b] == 0xFFFFFFFFu) hv = ht.value[b] != 0;          // (a bucket without a heavy value keeps 0 there)
                bb = b;
                lt = !hv;
            }
            if (hv) { k[i] = atomicAdd(&eqc[bb], 1u); heavy_mask |= 1u << i; }
            // (the light records of a wave take their places with one addition: late rounds of a tandem repeat are half light)
            const uint64_t lm = __ballot(lt);
            if (lm) {
                const int first = __builtin_ctzll(lm);
                unsigned at = 0;
                if ((int)lane_id() == first) at = atomicAdd(&nlight, (unsigned)__builtin_popcountll(lm));
                at = shfl<uint32_t>(at, first);
                if (lt) k[i] = at + __builtin_amdgcn_mbcnt_hi((unsigned)(lm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)lm, 0u));
            }
        }
        __syncthreads();
        for (unsigned b = threadIdx.x; b < nb; b += BLOCK) {
            const unsigned cb = eqc[b];
            if (cb) eqc[b] = (uint32_t)(ht.bstart[b] + atomicAdd(&ht.eq[(size_t)b * HEAVY_PAD], (unsigned long long)cb));
        }
        if (threadIdx.x == 0 && nlight) lbase = atomicAdd(ht.light, (unsigned long long)nlight);
        __syncthreads();
        const uint64_t lb = lbase;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const unsigned p = threadIdx.x + (unsigned)i * BLOCK;
            if (p < count) {
                const uint64_t x = pairs[base + p];
                const uint32_t q = (uint32_t)x, b = (uint32_t)(x >> 32) & 0x7FFFFFFFu;
                const bool beyond = (x >> 63) != 0;
                const uint32_t suffix = (uint32_t)(beyond ? (uint64_t)q + n - h : (uint64_t)q - h);
                if (heavy_mask & (1u << i)) HB[(uint64_t)eqc[b] + k[i]] = suffix;
                else {
                    const uint32_t rank = beyond ? 0u : (staged ? win[q & (W - 1)] : (uint32_t)ISA[q]);
                    const T kk = (T)(((uint64_t)b << kb2) | (beyond ? 0ull : (uint64_t)rank + 1));
                    LK[lb + k[i]] = kk; LV[lb + k[i]] = suffix;
                    o1 |= kk; a1 &= kk;
                }
            }
        }
    }
    key_summary_add<T>(summary, o1, a1, (T)0, ~(T)0);
}

// After the sort of the light records: one thread per bucket finds where its light records lie in the sorted list (SLK ascending by
// (bucket number, B2)) and how many of them come before the heavy value; err is raised when a bucket's records do not add up.
// ... and whether the bucket's heavy run keeps the rank its members carry (pos = the list: the bucket's first SA position is pos[bstart[b]], all its
// members are in the list, so it occupies the SA positions from there on)
template <typename T>
__global__ void heavy_plan_kernel(uint32_t nb, HeavyTabs ht, unsigned kb2, const T* __restrict__ SLK, uint64_t nlight, unsigned* __restrict__ err,
                                  const T* __restrict__ pos, int lazy) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    auto lower = [&](uint64_t key) { uint64_t lo = 0, hi = nlight; while (lo < hi) { const uint64_t m = lo + ((hi - lo) >> 1); if ((uint64_t)SLK[m] < key) lo = m + 1; else hi = m; } return lo; };
    if (b == nb) { ht.lstart[nb] = nlight; return; }
    const uint64_t s = lower((uint64_t)b << kb2), e = b + 1 == nb ? nlight : lower((uint64_t)(b + 1) << kb2);
    ht.lstart[b] = s;
    ht.less[b] = lower(((uint64_t)b << kb2) | ht.value[b]) - s;
    const uint64_t eq = ht.eq[(size_t)b * HEAVY_PAD], less = ht.less[b];
    if ((e - s) + eq != ht.bstart[b + 1] - ht.bstart[b]) atomicOr(err, 2u);
    const uint64_t first = (uint64_t)pos[ht.bstart[b]] + less, now = ht.rank[b];          // the run takes the SA positions first .. first + eq - 1
    ht.rank[b] = (lazy && eq && now >= first && now < first + eq) ? (now | HEAVY_KEEP) : first + (lazy ? eq / 2 : 0);
}

} // namespace psacx
