// dist_ops.hpp -- step-level ops of the block-distributed construction: one rank's share of every global
// step.  The exchanges between ranks are made by the host: multi.hpp (C++, RCCL; behind include/psacx.h) or, as
// the test harness, psac_amd/dist.py through the C ABI of include/psacx_ops.h (dist_ops.hip).
#pragma once
#include "../../include/psacx_ops.h"
#include "construct.hpp"
#include "nsv.hpp"
#include "multi_plan.hpp"      // BlkDist (mxx::blk_dist) and the host-side plans

namespace psacx {

// ---------------------------------------------------------------- small element-wise kernels
template <typename T>
__global__ void iota_from_kernel(T* __restrict__ out, uint64_t m, uint64_t start) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) out[i] = (T)(start + i);
}

template <typename T>
__global__ void owners_kernel(const T* __restrict__ g, uint64_t cnt, BlkDist d, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        uint64_t x = (uint64_t)g[i];
        if (x >= d.n) x = d.n - 1;
        out[i] = (T)d.rank_of(x);
    }
}
template <typename T>
__global__ void take_kernel(const T* __restrict__ block, const T* __restrict__ g, uint64_t cnt, uint64_t off, uint64_t n,
                            T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        uint64_t x = (uint64_t)g[i];
        if (x >= n) x = n - 1;
        out[i] = block[x - off];
    }
}
template <typename T>
__global__ void put_kernel(T* __restrict__ block, const T* __restrict__ g, uint64_t cnt, uint64_t off,
                           const T* __restrict__ vals, int64_t delta) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride)
        block[(uint64_t)g[i] - off] = (T)((int64_t)vals[i] + delta);
}
template <typename T>
__global__ void add_scalar_kernel(const T* __restrict__ in, uint64_t cnt, uint64_t s, uint64_t cap, T* __restrict__ out) {
    // saturating: in + s is formed in 64 bits and clamped to cap (= n, "past the end"), so SA + h cannot wrap
    // around a 32-bit index type and come back as a valid position
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        const uint64_t v = (uint64_t)in[i] + s;
        out[i] = (T)(v < cap ? v : cap);
    }
}
template <typename T>
__global__ void finish_b2_kernel(const T* __restrict__ ans, const T* __restrict__ q, uint64_t cnt, uint64_t n, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride)
        out[i] = (uint64_t)q[i] < n ? (T)(ans[i] + 1) : (T)0;
}
template <typename T>
__global__ void lcp_apply_kernel(T* __restrict__ block, const T* __restrict__ at, uint64_t cnt, uint64_t off,
                                 const T* __restrict__ mins, uint64_t h) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride)
        block[(uint64_t)at[i] - off] = (T)(h + (uint64_t)mins[i]);
}

// the same through a min-pyramid over the block that somebody keeps (multi.hpp: block_pyramid): the upper levels follow
template <typename T>
__global__ void lcp_apply_pyr_kernel(Pyramid<T> P, const T* __restrict__ at, uint64_t cnt, uint64_t off, const T* __restrict__ mins, uint64_t h) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride)
        pyramid_set<T>(P, (uint64_t)at[i] - off, (T)(h + (uint64_t)mins[i]));
}

// lower / upper bound of (q1,q2) in sorted pairs; use_second == 0 compares the first word only
template <typename T>
__global__ void pair_bounds_kernel(const T* __restrict__ s1, const T* __restrict__ s2, uint64_t n,
                                   const unsigned long long* __restrict__ q1, const unsigned long long* __restrict__ q2,
                                   unsigned nq, int use_second, unsigned long long* __restrict__ lb,
                                   unsigned long long* __restrict__ ub) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * nq) return;
    const unsigned qi = i >> 1;
    const bool upper = i & 1;
    const uint64_t a = q1[qi], b = q2[qi];
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const uint64_t x = (uint64_t)s1[mid], y = use_second ? (uint64_t)s2[mid] : 0;
        const uint64_t bb = use_second ? b : 0;
        const bool less = x < a || (x == a && y < bb);
        const bool leq = x < a || (x == a && y <= bb);
        if (upper ? leq : less) lo = mid + 1; else hi = mid;
    }
    (upper ? ub : lb)[qi] = lo;
}

template <typename T>
__global__ void rmq_split_kernel(const T* __restrict__ lo, const T* __restrict__ hi, uint64_t cnt, BlkDist d,
                                 T* own1, T* lo1, T* hi1, T* own2, T* lo2, T* hi2, T* ra, T* rb) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        const uint64_t l = lo[i], r = hi[i];
        const unsigned pl = d.rank_of(l), pr = d.rank_of(r - 1);
        own1[i] = (T)pl; lo1[i] = (T)l;
        own2[i] = (T)pr; hi2[i] = (T)r;
        if (pl == pr) { hi1[i] = (T)r; lo2[i] = (T)r; }
        else { hi1[i] = (T)(d.off(pl) + d.size(pl)); lo2[i] = (T)d.off(pr); }
        ra[i] = (T)(pl + 1); rb[i] = (T)pr;
    }
}
// one half of the same split at a time (half 0: the part inside the rank of lo; 1: the part inside the rank of hi - 1), so that a caller short of
// memory holds three arrays instead of eight -- the ranks in between are found again from lo / hi by rmq_combine_range_kernel
template <typename T>
__global__ void rmq_split_half_kernel(const T* __restrict__ lo, const T* __restrict__ hi, uint64_t cnt, BlkDist d, int half, T* own, T* l_out, T* h_out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        const uint64_t l = lo[i], r = hi[i];
        const unsigned pl = d.rank_of(l), pr = d.rank_of(r - 1);
        if (half == 0) { own[i] = (T)pl; l_out[i] = (T)l; h_out[i] = pl == pr ? (T)r : (T)(d.off(pl) + d.size(pl)); }
        else { own[i] = (T)pr; h_out[i] = (T)r; l_out[i] = pl == pr ? (T)r : (T)d.off(pr); }
    }
}
struct RankMins { unsigned long long v[64]; };
template <typename T>
__global__ void rmq_combine_range_kernel(const T* __restrict__ a1, const T* __restrict__ a2, const T* __restrict__ lo, const T* __restrict__ hi, uint64_t cnt, BlkDist d,
                                         RankMins rm, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        T m = a1[i] < a2[i] ? a1[i] : a2[i];
        const unsigned pl = d.rank_of((uint64_t)lo[i]), pr = d.rank_of((uint64_t)hi[i] - 1);
        for (unsigned r = pl + 1; r < pr; ++r) { const T x = (T)rm.v[r]; m = x < m ? x : m; }
        out[i] = m;
    }
}
template <typename T>
__global__ void rmq_combine_kernel(const T* __restrict__ a1, const T* __restrict__ a2, const T* __restrict__ ra,
                                   const T* __restrict__ rb, uint64_t cnt, RankMins rm, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        T m = a1[i] < a2[i] ? a1[i] : a2[i];
        for (unsigned r = (unsigned)ra[i]; r < (unsigned)rb[i]; ++r) { const T x = (T)rm.v[r]; m = x < m ? x : m; }
        out[i] = m;
    }
}

// min over block[lo-off .. hi-off) through a 64-ary pyramid of the block (empty range -> all ones)
template <typename T>
__global__ void range_min_kernel(Pyramid<T> P, const T* __restrict__ lo, const T* __restrict__ hi, uint64_t cnt,
                                 uint64_t off, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        const uint64_t l = (uint64_t)lo[i] - off, r = (uint64_t)hi[i] - off;
        out[i] = r > l ? pyramid_min<T>(P, l, r) : ~(T)0;
    }
}

template <typename T>
Boundary<T> to_boundary(const psacx_boundary* b) {
    Boundary<T> d;
    std::memset(&d, 0, sizeof(d));
    if (!b) return d;
    d.off = b->off; d.base = b->base; d.has_prev = b->has_prev; d.has_next = b->has_next;
    d.prev1 = (T)b->prev[0]; d.prev2 = (T)b->prev[1]; d.prev3 = (T)b->prev[2];
    d.next1 = (T)b->next[0]; d.next2 = (T)b->next[1]; d.next3 = (T)b->next[2];
    return d;
}

inline KeyShape shape_of(uint32_t l, uint32_t c1, uint32_t c2) { KeyShape ks; ks.lc = l; ks.c1 = c1; ks.c2 = c2; ks.spec = 0; return ks; }

#define OP_PROLOGUE(c) if (!(c)) return PSACX_EINVAL; PSACX_HIP(c, hipSetDevice((c)->device))

// tile scratch (carry / nact / nunf / totals) out of the ctx slab
struct TileScratch { uint64_t *carry, *nact, *nunf, *totals; };
inline int tile_scratch(psacx_ctx* c, uint64_t cnt, TileScratch& ts, size_t extra_bytes, char** extra) {
    const uint64_t nt = (cnt + SCAN_TILE_MIN - 1) / SCAN_TILE_MIN + 1;
    Arena dry(nullptr);
    auto lay = [&](Arena& a) { ts.carry = a.take<uint64_t>(nt); ts.nact = a.take<uint64_t>(nt); ts.nunf = a.take<uint64_t>(nt);
                               ts.totals = a.take<uint64_t>(8); if (extra) *extra = a.take<char>(extra_bytes); };
    lay(dry);
    PSACX_TRY(ensure_slab(c, dry.off + 4096));
    Arena ar(c->slab);
    lay(ar);
    return PSACX_OK;
}

template <typename T>
int op_make_keys(psacx_ctx* c, const uint8_t* text, uint64_t m, uint64_t text_len, const uint16_t* codes, uint32_t l,
                 uint32_t c1, uint32_t c2, T* k1, T* k2, const T* slen = nullptr) {
    // slen (string sets, construct_ss): characters from every position of the block to the end of its string; the codes are
    // then psac's 1 .. sigma with 0 = end and every window is cut at its string's end (key_pairs_kernel<..., GSA>)
    OP_PROLOGUE(c);
    if (m == 0) return PSACX_OK;
    CodeTable tab;
    for (int i = 0; i < 256; ++i) tab.c[i] = codes[i];
    constexpr int KB = 256, KI = 8;
    const uint64_t nb = (m + KB * KI - 1) / (KB * KI);
    Arena dry(nullptr); dry.take<unsigned long long>(nb * 4 + 16);
    PSACX_TRY(ensure_slab(c, dry.off + 4096));
    Arena ar(c->slab);
    unsigned long long* partials = ar.take<unsigned long long>(nb * 4 + 16);
    if (slen)
        hipLaunchKernelGGL((key_pairs_kernel<T, KB, KI, true>), dim3((unsigned)nb), dim3(KB), 0, c->stream, text, m, text_len, tab,
                           shape_of(l, c1, c2), k1, k2, partials, slen);
    else
        hipLaunchKernelGGL((key_pairs_kernel<T, KB, KI>), dim3((unsigned)nb), dim3(KB), 0, c->stream, text, m, text_len, tab,
                           shape_of(l, c1, c2), k1, k2, partials);
    PSACX_HIP(c, hipGetLastError());
    return PSACX_OK;
}

template <typename T>
int op_pair_sort(psacx_ctx* c, T* k1, T* k2, T* v, T* a1, T* a2, T* av, uint64_t n, uint32_t bits1, uint32_t bits2,
                 int32_t* where, uint32_t lo1 = 0, bool iota = false, uint64_t spec = 0, uint64_t spec_n = 0, bool v32_in = false) {
    // lo1: the low lo1 bits of word 1 are not sorted on (prefix sort by its leading bits; bits2 is then 0)
    // iota / spec / spec_n: the first pass makes up the payload (pair_sort); v32_in: v holds 32-bit entries (two-word records)
    OP_PROLOGUE(c);
    *where = 0;
    if (n < 2 && !iota && !v32_in) return PSACX_OK;      // (a made-up or 32-bit payload still has to be written out in full words)
    if (n == 0) return PSACX_OK;
    PSACX_TRY(ensure_pinned(c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 4096));
    SortScratch sc;
    auto layout = [&](Arena& a) {
        sc.d_hist = a.take<unsigned long long>((size_t)MAX_PASSES * RADIX);
        sc.d_base = a.take<unsigned long long>((size_t)MAX_PASSES * RADIX);
        sc.desc_bytes = sort_desc_bytes(n);
        sc.d_desc = a.take<char>(sc.desc_bytes);
        sc.d_err = a.take<unsigned>(64);
        sc.d_summary = a.take<unsigned long long>(8);
        sc.d_partials = a.take<unsigned long long>(((size_t)(n / 2048) + 8192) * 4);
    };
    { Arena dry(nullptr); layout(dry); PSACX_TRY(ensure_slab(c, dry.off + 4096)); }
    Arena ar(c->slab);
    layout(ar);
    sc.h_hist = reinterpret_cast<unsigned long long*>(c->pinned + 1024);
    sc.h_base = sc.h_hist + (size_t)MAX_PASSES * RADIX;
    sc.h_summary = reinterpret_cast<unsigned long long*>(c->pinned + 256);
    PSACX_HIP(c, hipMemsetAsync(sc.d_err, 0, 64 * sizeof(unsigned), c->stream));
    c->profile = c->profile_ops; c->ev_used = 0;
    SortBufs<T> in{k1, k2, v}, alt{a1, a2, av}, res;
    // a word with zero significant bits takes no pass
    PSACX_TRY(pair_sort<T>(c, sc, in, alt, n, iota, bits1, bits2, nullptr, &res, nullptr, spec, spec_n, false, lo1, -1, v32_in));
    *where = (res.k1 == k1) ? 0 : 1;
    if (res.v != (*where ? av : v))       // cannot happen without final_v, kept as a guard
        PSACX_HIP(c, hipMemcpyAsync(*where ? av : v, res.v, n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    PSACX_HIP(c, hipMemcpyAsync(c->pinned, sc.d_err, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    if (c->profile) { prof_accumulate(c); c->profile = false; }
    if (*reinterpret_cast<unsigned*>(c->pinned)) return PSACX_EDEVICE;
    return PSACX_OK;
}

template <typename T>
int op_split_by(psacx_ctx* c, const T* k1, const T* k2, const T* v, uint64_t n, const uint64_t* sk1, const uint64_t* sk2,
                const uint64_t* srank, const uint64_t* sidx, uint32_t nsplit, uint64_t my_rank, T* o1, T* o2, T* ov,
                uint64_t* class_start) {
    OP_PROLOGUE(c);
    if (nsplit > 63) return PSACX_EINVAL;
    for (uint32_t i = 0; i <= nsplit + 1; ++i) class_start[i] = 0;
    if (n == 0) return PSACX_OK;
    PSACX_TRY(ensure_pinned(c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 4096));
    SortScratch sc;
    T* cls = nullptr;
    auto layout = [&](Arena& a) {
        cls = a.take<T>(n);
        sc.d_base = a.take<unsigned long long>((size_t)MAX_PASSES * RADIX);
        sc.desc_bytes = sort_desc_bytes(n);
        sc.d_desc = a.take<char>(sc.desc_bytes);
    };
    { Arena dry(nullptr); layout(dry); PSACX_TRY(ensure_slab(c, dry.off + 4096)); }
    Arena ar(c->slab);
    layout(ar);
    Splitters sp;
    sp.n = nsplit;
    for (uint32_t i = 0; i < nsplit; ++i) { sp.k1[i] = sk1[i]; sp.k2[i] = sk2[i]; sp.rank[i] = srank[i]; sp.idx[i] = sidx[i]; }
    hipLaunchKernelGGL((classify_kernel<T>), dim3(grid_for(c, n, 256, 16)), dim3(256), 0, c->stream, k1, k2, n, sp,
                       (unsigned long long)my_rank, cls);
    PSACX_HIP(c, hipGetLastError());
    SortBufs<T> in{const_cast<T*>(k1), const_cast<T*>(k2), const_cast<T*>(v)}, out{o1, o2, ov};
    unsigned long long* starts = reinterpret_cast<unsigned long long*>(c->pinned + 1024);
    PSACX_TRY(class_partition<T>(c, sc, in, out, cls, n, starts));
    for (uint32_t i = 0; i <= nsplit; ++i) class_start[i] = starts[i];
    class_start[nsplit + 1] = n;
    return PSACX_OK;
}

template <typename T>
int op_put_perm(psacx_ctx* c, T* block, const T* gidx, uint64_t cnt, uint64_t off, const T* vals, T* s1, T* s2, T* s3, T* s4) {
    OP_PROLOGUE(c);
    if (cnt == 0) return PSACX_OK;
    const size_t ncur = (size_t)(cnt >> INV_WINDOW_BITS) + 2 + RADIX_P;
    PSACX_TRY(ensure_slab(c, ncur * sizeof(unsigned) + 8192));
    SortBufs<T> t1{s1, s2, nullptr}, t2{s3, s4, nullptr};
    return invert_permutation<T>(c, reinterpret_cast<unsigned*>(c->slab), gidx, vals, cnt, block, t1, t2, c->knobs, off);
}

template <typename T>
int op_pair_bounds(psacx_ctx* c, const T* s1, const T* s2, uint64_t n, const uint64_t* q1, const uint64_t* q2, uint32_t nq,
                   int use_second, uint64_t* lb, uint64_t* ub) {
    OP_PROLOGUE(c);
    if (nq == 0) return PSACX_OK;
    PSACX_TRY(ensure_pinned(c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 4096));
    Arena dry(nullptr); dry.take<unsigned long long>((size_t)nq * 4);
    PSACX_TRY(ensure_slab(c, dry.off + 4096));
    Arena ar(c->slab);
    unsigned long long* d = ar.take<unsigned long long>((size_t)nq * 4);
    PSACX_HIP(c, hipMemcpyAsync(d, q1, nq * 8, hipMemcpyHostToDevice, c->stream));
    PSACX_HIP(c, hipMemcpyAsync(d + nq, q2, nq * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL((pair_bounds_kernel<T>), dim3((2 * nq + 255) / 256), dim3(256), 0, c->stream, s1, s2, n, d, d + nq, nq,
                       use_second, d + 2 * nq, d + 3 * nq);
    PSACX_HIP(c, hipGetLastError());
    PSACX_HIP(c, hipMemcpyAsync(lb, d + 2 * nq, nq * 8, hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipMemcpyAsync(ub, d + 3 * nq, nq * 8, hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    return PSACX_OK;
}

template <typename T>
int run_last_head(psacx_ctx* c, int mode, const T* s1, const T* s2, const T* s3, uint64_t cnt, uint64_t n, KeyShape ks,
                  Boundary<T> bd, TileScratch& ts, bool gsa = false) {
    const uint64_t ntiles = (cnt + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
    if (mode == 0 && gsa)          // string sets: a suffix shorter than 2k shows in the end markers of its own window
        hipLaunchKernelGGL((last_head_kernel<T, false, true>), dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, c->stream, s1, s2,
                           (const T*)nullptr, cnt, (unsigned)ScanCfg<T>::TILE, ntiles, ts.carry, s3, ks, n, bd);
    else if (mode == 0)
        hipLaunchKernelGGL((last_head_kernel<T, false>), dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, c->stream, s1, s2,
                           (const T*)nullptr, cnt, (unsigned)ScanCfg<T>::TILE, ntiles, ts.carry, s3, ks, n, bd);
    else
        hipLaunchKernelGGL((last_head_kernel<T, true>), dim3((unsigned)ntiles), dim3(256), 0, c->stream, s1, s2, s3,
                           cnt, (unsigned)ScanCfg<T>::TILE, ntiles, ts.carry, (const T*)nullptr, ks, n, bd);
    PSACX_HIP(c, hipGetLastError());
    hipLaunchKernelGGL((tile_scan_kernel<1024, OpMax>), dim3(1), dim3(1024), 0, c->stream, ts.carry, ntiles, OpMax(), (uint64_t)0,
                       ts.totals);
    PSACX_HIP(c, hipGetLastError());
    return PSACX_OK;
}

template <typename T>
int op_last_head(psacx_ctx* c, int mode, const T* s1, const T* s2, const T* s3, uint64_t cnt, uint64_t n, uint32_t l,
                 uint32_t c1, uint32_t c2, const psacx_boundary* b, uint64_t* out, bool gsa = false) {
    OP_PROLOGUE(c);
    *out = 0;
    if (cnt == 0) return PSACX_OK;
    PSACX_TRY(ensure_pinned(c, 4096));
    TileScratch ts;
    PSACX_TRY(tile_scratch(c, cnt, ts, 0, nullptr));
    PSACX_TRY(run_last_head<T>(c, mode, s1, s2, s3, cnt, n, shape_of(l, c1, c2), to_boundary<T>(b), ts, gsa));
    PSACX_HIP(c, hipMemcpyAsync(c->pinned, ts.totals, 8, hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    *out = *reinterpret_cast<uint64_t*>(c->pinned);
    return PSACX_OK;
}

template <typename T>
int counts_back(psacx_ctx* c, TileScratch& ts, uint64_t ntiles, uint64_t* nact, uint64_t* nunf) {
    hipLaunchKernelGGL((tile_scan_kernel<1024, OpSum>), dim3(1), dim3(1024), 0, c->stream, ts.nact, ntiles, OpSum(), (uint64_t)0, ts.totals);
    hipLaunchKernelGGL((tile_scan_kernel<1024, OpSum>), dim3(1), dim3(1024), 0, c->stream, ts.nunf, ntiles, OpSum(), (uint64_t)0, ts.totals + 1);
    PSACX_HIP(c, hipGetLastError());
    PSACX_HIP(c, hipMemcpyAsync(c->pinned, ts.totals, 24, hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    *nact = reinterpret_cast<uint64_t*>(c->pinned)[0];
    *nunf = reinterpret_cast<uint64_t*>(c->pinned)[1];
    return PSACX_OK;
}

template <typename T>
int op_rebucket_first(psacx_ctx* c, const T* s1, const T* s2, const T* sa, uint64_t cnt, uint64_t n, uint32_t l, uint32_t c1,
                      uint32_t c2, const psacx_boundary* b, T* bsa, T* lcp, uint64_t* nact, uint64_t* nunf, bool gsa = false, T* sa_out = nullptr,
                      uint64_t* tile_nact_out = nullptr) {
    // tile_nact_out (optional, ceil(cnt / ScanCfg<T>::TILE) entries): the per-tile counts of unresolved positions the kernel
    // produces, kept for the compaction that follows (op_compact_counted) instead of counting them again
    OP_PROLOGUE(c);
    *nact = *nunf = 0;
    if (cnt == 0) return PSACX_OK;
    PSACX_TRY(ensure_pinned(c, 4096));
    TileScratch ts;
    PSACX_TRY(tile_scratch(c, cnt, ts, 0, nullptr));
    const KeyShape ks = shape_of(l, c1, c2);
    const Boundary<T> bd = to_boundary<T>(b);
    PSACX_TRY(run_last_head<T>(c, 0, s1, s2, sa, cnt, n, ks, bd, ts, gsa));
    const uint64_t ntiles = (cnt + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
    if (gsa && lcp)
        hipLaunchKernelGGL((rebucket_first_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, true, true>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0,
                           c->stream, s1, s2, sa, cnt, ks, bsa, lcp, ts.carry, ts.nact, ts.nunf, n, bd, (T*)nullptr, (unsigned*)nullptr, 0, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (unsigned*)nullptr, sa_out);
    else if (gsa)
        hipLaunchKernelGGL((rebucket_first_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, false, true>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0,
                           c->stream, s1, s2, sa, cnt, ks, bsa, (T*)nullptr, ts.carry, ts.nact, ts.nunf, n, bd, (T*)nullptr, (unsigned*)nullptr, 0, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (unsigned*)nullptr, sa_out);
    else if (lcp)
        hipLaunchKernelGGL((rebucket_first_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, true>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0,
                           c->stream, s1, s2, sa, cnt, ks, bsa, lcp, ts.carry, ts.nact, ts.nunf, n, bd, (T*)nullptr, (unsigned*)nullptr, 0, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (unsigned*)nullptr, sa_out);
    else
        hipLaunchKernelGGL((rebucket_first_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, false>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0,
                           c->stream, s1, s2, sa, cnt, ks, bsa, (T*)nullptr, ts.carry, ts.nact, ts.nunf, n, bd, (T*)nullptr, (unsigned*)nullptr, 0, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (unsigned*)nullptr, sa_out);
    PSACX_HIP(c, hipGetLastError());
    if (tile_nact_out) PSACX_HIP(c, hipMemcpyAsync(tile_nact_out, ts.nact, ntiles * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
    return counts_back<T>(c, ts, ntiles, nact, nunf);
}

template <typename T>
int op_rebucket_refine(psacx_ctx* c, const T* t1, const T* t2, const T* tv, const T* pos, uint64_t cnt, uint64_t n, uint64_t h,
                       const psacx_boundary* b, T* sa_block, T* bsa_block, T* lcp_block, T* ids_out, T* q_at, T* q_lo, T* q_hi,
                       uint64_t* nq, uint64_t* nact, uint64_t* nunf, const Pyramid<T>* kept = nullptr) {
    // kept: a min-pyramid over lcp_block that the caller keeps (level 0 = lcp_block): the entries this step lowers are lowered in its upper levels too
    OP_PROLOGUE(c);
    *nq = *nact = *nunf = 0;
    if (cnt == 0) return PSACX_OK;
    PSACX_TRY(ensure_pinned(c, 4096));
    TileScratch ts;
    PSACX_TRY(tile_scratch(c, cnt, ts, 0, nullptr));
    const Boundary<T> bd = to_boundary<T>(b);
    PSACX_TRY(run_last_head<T>(c, 1, t1, t2, pos, cnt, n, shape_of(1, 1, 0), bd, ts));
    const uint64_t ntiles = (cnt + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
    unsigned long long* qc = reinterpret_cast<unsigned long long*>(ts.totals + 2);
    PSACX_HIP(c, hipMemsetAsync(qc, 0, 8, c->stream));
    Pyramid<T> pyr;
    std::memset(&pyr, 0, sizeof(pyr));
    pyr.lvl[0] = lcp_block; pyr.nlev = lcp_block ? 1 : 0;
    if (kept && lcp_block && kept->nlev > 0 && kept->lvl[0] == lcp_block) pyr = *kept;
    if (lcp_block)
        hipLaunchKernelGGL((rebucket_refine_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, true, true>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0,
                           c->stream, t1, t2, tv, pos, cnt, n, h, sa_block, bsa_block, (T*)nullptr, pyr, ids_out, ts.carry, ts.nact,
                           ts.nunf, bd, q_at, q_lo, q_hi, qc);
    else
        hipLaunchKernelGGL((rebucket_refine_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, false, true>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0,
                           c->stream, t1, t2, tv, pos, cnt, n, h, sa_block, bsa_block, (T*)nullptr, pyr, ids_out, ts.carry, ts.nact,
                           ts.nunf, bd, q_at, q_lo, q_hi, qc);
    PSACX_HIP(c, hipGetLastError());
    PSACX_TRY(counts_back<T>(c, ts, ntiles, nact, nunf));
    *nq = reinterpret_cast<uint64_t*>(c->pinned)[2];
    return PSACX_OK;
}

template <typename T>
int op_compact(psacx_ctx* c, const T* ids, const T* pos, uint64_t cnt, uint64_t off, uint64_t prev_id, uint64_t next_id, T* pos_out,
               uint64_t* n_out) {
    OP_PROLOGUE(c);
    *n_out = 0;
    if (cnt == 0) return PSACX_OK;
    PSACX_TRY(ensure_pinned(c, 4096));
    TileScratch ts;
    PSACX_TRY(tile_scratch(c, cnt, ts, 0, nullptr));
    const uint64_t ntiles = (cnt + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
    hipLaunchKernelGGL((count_active_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0, c->stream, ids, cnt,
                       (T)prev_id, (T)next_id, ts.nact);
    hipLaunchKernelGGL((tile_scan_kernel<1024, OpSum>), dim3(1), dim3(1024), 0, c->stream, ts.nact, ntiles, OpSum(), (uint64_t)0, ts.totals);
    hipLaunchKernelGGL((compact_active_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0, c->stream, ids, pos,
                       cnt, pos_out, ts.nact, off, (T)prev_id, (T)next_id);
    PSACX_HIP(c, hipGetLastError());
    PSACX_HIP(c, hipMemcpyAsync(c->pinned, ts.totals, 8, hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    *n_out = *reinterpret_cast<uint64_t*>(c->pinned);
    return PSACX_OK;
}

// op_compact with the per-tile counts given (tile_nact: device array, turned into offsets in place): one pass over the ids
template <typename T>
int op_compact_counted(psacx_ctx* c, const T* ids, uint64_t cnt, uint64_t off, uint64_t prev_id, uint64_t next_id, uint64_t* tile_nact, T* pos_out) {
    OP_PROLOGUE(c);
    if (cnt == 0) return PSACX_OK;
    const uint64_t ntiles = (cnt + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
    hipLaunchKernelGGL((tile_scan_kernel<1024, OpSum>), dim3(1), dim3(1024), 0, c->stream, tile_nact, ntiles, OpSum(), (uint64_t)0, (uint64_t*)nullptr);
    hipLaunchKernelGGL((compact_active_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0, c->stream, ids, (const T*)nullptr,
                       cnt, pos_out, tile_nact, off, (T)prev_id, (T)next_id);
    PSACX_HIP(c, hipGetLastError());
    return PSACX_OK;
}

// Records of a prefix sort that still tie on the leading bits of word 1 (S1 >> lo1 equal to a neighbour's).  Phase 1
// (out arrays null): *n_out = their number.  Phase 2: pos_out = their positions, k1_out / v_out = their word 1 and payload,
// in order.  Tie groups never straddle a rank (the shuffle keeps equal prefixes together), so there is no neighbour id.
template <typename T>
int op_compact_ties(psacx_ctx* c, const T* s1, const T* v, uint64_t cnt, unsigned lo1, T* pos_out, T* k1_out, T* v_out, uint64_t* n_out, bool counted = false) {
    // counted (with pos_out): the call before this one on this context was the counting call for the same records (pos_out == nullptr), and
    // nothing has used the context's scratch since: the tiles' offsets are still there and the records are not read a third time.  The
    // counting call leaves a stamp (records, length, shift, generation of the scratch); a counted call whose stamp does not match counts again.
    OP_PROLOGUE(c);
    if (!pos_out) *n_out = 0;
    if (cnt == 0) return PSACX_OK;
    PSACX_TRY(ensure_pinned(c, 4096));
    const bool reuse = counted && pos_out && c->tie_stamp.s1 == (const void*)s1 && c->tie_stamp.cnt == cnt && c->tie_stamp.lo1 == lo1 &&
                       c->tie_stamp.gen == c->slab_gen;
    TileScratch ts;
    PSACX_TRY(tile_scratch(c, cnt, ts, 0, nullptr));
    const uint64_t ntiles = (cnt + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
    if (!reuse) {
        hipLaunchKernelGGL((count_active_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0, c->stream, s1, cnt,
                           (T)0, (T)0, ts.nact, lo1);
        hipLaunchKernelGGL((tile_scan_kernel<1024, OpSum>), dim3(1), dim3(1024), 0, c->stream, ts.nact, ntiles, OpSum(), (uint64_t)0, ts.totals);
    }
    PSACX_HIP(c, hipGetLastError());
    if (!pos_out) {
        PSACX_HIP(c, hipMemcpyAsync(c->pinned, ts.totals, 8, hipMemcpyDeviceToHost, c->stream));
        PSACX_HIP(c, hipStreamSynchronize(c->stream));
        *n_out = *reinterpret_cast<uint64_t*>(c->pinned);
        c->tie_stamp.s1 = s1; c->tie_stamp.cnt = cnt; c->tie_stamp.lo1 = lo1; c->tie_stamp.gen = c->slab_gen;
        return PSACX_OK;
    }
    c->tie_stamp.s1 = nullptr;
    hipLaunchKernelGGL((compact_active_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, true>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0, c->stream, s1,
                       (const T*)nullptr, cnt, pos_out, ts.nact, (uint64_t)0, (T)0, (T)0, lo1, v, k1_out, v_out);
    PSACX_HIP(c, hipGetLastError());
    return PSACX_OK;
}

// aux_queries > 0: also tabulates the running minima pyramid_min uses (see prepare_range_min in
// construct.hpp) when that many queries make it worth while
template <typename T>
int build_block_pyramid(psacx_ctx* c, const T* block, uint64_t m, Pyramid<T>& P, unsigned long long** scalar = nullptr,
                        uint64_t aux_queries = 0) {
    const bool aux_up = aux_queries >= (1u << 16);
    const bool aux0 = aux_up && aux_queries >= m / 32;
    T *pre[PYR_MAX] = {}, *suf[PYR_MAX] = {};
    auto layout = [&](Arena& a) {
        P = Pyramid<T>();
        P.lvl[0] = const_cast<T*>(block); P.len[0] = m; P.nlev = 1;
        uint64_t len = m;
        if (aux0) { pre[0] = a.take<T>(m); suf[0] = a.take<T>(m); }
        while (len > 128 && P.nlev < PYR_MAX) {
            len = (len + 63) / 64;
            P.lvl[P.nlev] = a.take<T>(len); P.len[P.nlev] = len;
            if (aux_up) { pre[P.nlev] = a.take<T>(len); suf[P.nlev] = a.take<T>(len); }
            P.nlev++;
        }
        unsigned long long* sc = a.take<unsigned long long>(8);
        if (scalar) *scalar = sc;
    };
    { Arena dry(nullptr); layout(dry); PSACX_TRY(ensure_slab(c, dry.off + 4096)); }
    Arena ar(c->slab);
    layout(ar);
    for (int L = 1; L < P.nlev; ++L) {
        hipLaunchKernelGGL((pyramid_level_kernel<T>), dim3(grid_for(c, P.len[L] * 64, 256, 8)), dim3(256), 0, c->stream, P.lvl[L - 1],
                           P.len[L - 1], P.lvl[L], P.len[L]);
        PSACX_HIP(c, hipGetLastError());
    }
    for (int L = 0; L + 1 < P.nlev; ++L) {
        if (!pre[L]) continue;
        hipLaunchKernelGGL((pyramid_aux_kernel<T>), dim3(grid_for(c, P.len[L], 256, 8)), dim3(256), 0, c->stream, P.lvl[L], P.len[L],
                           pre[L], suf[L]);
        PSACX_HIP(c, hipGetLastError());
        P.pre[L] = pre[L]; P.suf[L] = suf[L];
    }
    return PSACX_OK;
}

template <typename T>
__global__ void top_min_kernel(const T* __restrict__ a, uint64_t len, unsigned long long* __restrict__ out) {
    T m = ~(T)0;
    for (uint64_t i = threadIdx.x; i < len; i += blockDim.x) { const T x = a[i]; m = x < m ? x : m; }
    m = wave_reduce<T>(m, OpMin());
    __shared__ T red[16];
    if (lane_id() == 0) red[threadIdx.x / WAVE] = m;
    __syncthreads();
    if (threadIdx.x == 0) { T r = red[0]; for (unsigned w = 1; w < blockDim.x / WAVE; ++w) r = red[w] < r ? red[w] : r; out[0] = (unsigned long long)r; }
}

template <typename T>
int op_block_min(psacx_ctx* c, const T* block, uint64_t m, uint64_t* out) {
    OP_PROLOGUE(c);
    *out = (uint64_t)(T)~(T)0;
    if (m == 0) return PSACX_OK;
    PSACX_TRY(ensure_pinned(c, 4096));
    Pyramid<T> P;
    unsigned long long* d = nullptr;
    PSACX_TRY(build_block_pyramid<T>(c, block, m, P, &d));
    hipLaunchKernelGGL((top_min_kernel<T>), dim3(1), dim3(256), 0, c->stream, P.lvl[P.nlev - 1], P.len[P.nlev - 1], d);
    PSACX_HIP(c, hipGetLastError());
    PSACX_HIP(c, hipMemcpyAsync(c->pinned, d, 8, hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    *out = *reinterpret_cast<uint64_t*>(c->pinned);
    return PSACX_OK;
}

template <typename T>
int op_range_min(psacx_ctx* c, const T* block, uint64_t m, const T* lo, const T* hi, uint64_t cnt, uint64_t off, T* out) {
    OP_PROLOGUE(c);
    if (cnt == 0) return PSACX_OK;
    Pyramid<T> P;
    std::memset(&P, 0, sizeof(P));
    if (m) PSACX_TRY(build_block_pyramid<T>(c, block, m, P, nullptr, cnt));
    hipLaunchKernelGGL((range_min_kernel<T>), dim3(grid_for(c, cnt, 256, 16)), dim3(256), 0, c->stream, P, lo, hi, cnt, off, out);
    PSACX_HIP(c, hipGetLastError());
    return PSACX_OK;
}

// One step of the distributed ANSV (ansv.hpp:1304-1740 keeps per-rank stacks and merges them; here every
// open query is a search in the block's min-pyramid): nearest element of this block strictly beyond the
// global position start[j] (to the left or to the right) whose value is < thr[j] (strict) or <= thr[j].
// start may lie outside the block (or be -1 / n): the search then begins at the block's edge.
template <typename T>
__global__ void nsv_from_kernel(Pyramid<T> P, uint64_t m, uint64_t off, const long long* __restrict__ start,
                                const T* __restrict__ thr, uint64_t cnt, int strict, int left,
                                T* __restrict__ out_idx, T* __restrict__ out_val) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const long long s = start[j] - (long long)off;          // block-relative, may be < 0 or >= m
        const T v = thr[j];
        uint64_t r = NSV_NONE;
        if (m) {
            if (left) {
                if (s > 0) {
                    if ((uint64_t)s >= m) {                        // from beyond the right edge: element m - 1 counts
                        const T x = P.lvl[0][m - 1];
                        r = (strict ? x < v : x <= v) ? m - 1 : (m > 1 ? nsv_search<T, true>(P, m - 1, v, strict != 0) : NSV_NONE);
                    } else r = nsv_search<T, true>(P, (uint64_t)s, v, strict != 0);
                }
            } else {
                if (s < (long long)m - 1) {
                    if (s < 0) {                                   // from before the left edge: element 0 counts
                        const T x = P.lvl[0][0];
                        r = (strict ? x < v : x <= v) ? 0 : (m > 1 ? nsv_search<T, false>(P, 0, v, strict != 0) : NSV_NONE);
                    } else r = nsv_search<T, false>(P, (uint64_t)s, v, strict != 0);
                }
            }
        }
        out_idx[j] = r == NSV_NONE ? ~(T)0 : (T)(off + r);
        out_val[j] = r == NSV_NONE ? (T)0 : P.lvl[0][r];
    }
}

template <typename T>
int op_nsv_from(psacx_ctx* c, const T* block, uint64_t m, uint64_t off, const long long* start, const T* thr, uint64_t cnt,
                int strict, int left, T* out_idx, T* out_val) {
    OP_PROLOGUE(c);
    if (cnt == 0) return PSACX_OK;
    Pyramid<T> P;
    std::memset(&P, 0, sizeof(P));
    if (m) {
        { Arena dry(nullptr); nsv_pyramid_layout<T>(dry, block, m, P); PSACX_TRY(ensure_slab(c, dry.off + 4096)); }
        Arena ar(c->slab);
        nsv_pyramid_layout<T>(ar, block, m, P);
        for (int L = 1; L < P.nlev; ++L) {
            hipLaunchKernelGGL((pyramid_level_kernel<T>), dim3(grid_for(c, P.len[L] * 64, 256, 8)), dim3(256), 0, c->stream,
                               P.lvl[L - 1], P.len[L - 1], P.lvl[L], P.len[L]);
            PSACX_HIP(c, hipGetLastError());
        }
    }
    hipLaunchKernelGGL((nsv_from_kernel<T>), dim3(grid_for(c, cnt, 256, 16)), dim3(256), 0, c->stream, P, m, off, start, thr, cnt,
                       strict, left, out_idx, out_val);
    PSACX_HIP(c, hipGetLastError());
    return PSACX_OK;
}

#define SIMPLE_LAUNCH(c, kern, cnt, ...)                                                                     \
    do { if ((cnt) > 0) { hipLaunchKernelGGL(kern, dim3(grid_for(c, cnt, 256, 16)), dim3(256), 0, (c)->stream, __VA_ARGS__); \
                          PSACX_HIP(c, hipGetLastError()); } } while (0)

} // namespace psacx

