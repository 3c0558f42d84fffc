// dev_common.hpp -- shared device helpers for the psacx HIP engine (gfx950, wave64).
//
// Everything here is written for CDNA4 only: 64-lane wavefronts, __ballot()
// returning a 64-bit mask, per-XCD L2s that are not coherent with each other
// (inter-workgroup words go through relaxed agent-scope atomics, i.e. sc1
// accesses that bypass the per-CU L1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psacx {

constexpr int WAVE = 64;

__device__ __forceinline__ unsigned lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint64_t lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

template <typename T> __device__ __forceinline__ T shfl(T v, int src);
template <> __device__ __forceinline__ uint32_t shfl<uint32_t>(uint32_t v, int src) {
    return (uint32_t)__shfl((int)v, src, WAVE);
}
template <> __device__ __forceinline__ uint64_t shfl<uint64_t>(uint64_t v, int src) {
    uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, WAVE);
    uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, WAVE);
    return ((uint64_t)hi << 32) | lo;
}
template <typename T> __device__ __forceinline__ T shfl_up(T v, int d);
template <> __device__ __forceinline__ uint32_t shfl_up<uint32_t>(uint32_t v, int d) {
    return (uint32_t)__shfl_up((int)v, d, WAVE);
}
template <> __device__ __forceinline__ uint64_t shfl_up<uint64_t>(uint64_t v, int d) {
    uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, WAVE);
    uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, WAVE);
    return ((uint64_t)hi << 32) | lo;
}
template <> __device__ __forceinline__ unsigned long long shfl_up<unsigned long long>(unsigned long long v, int d) {
    return (unsigned long long)shfl_up<uint64_t>((uint64_t)v, d);
}
template <typename T> __device__ __forceinline__ T shfl_xor(T v, int m);
template <> __device__ __forceinline__ uint32_t shfl_xor<uint32_t>(uint32_t v, int m) {
    return (uint32_t)__shfl_xor((int)v, m, WAVE);
}
template <> __device__ __forceinline__ uint64_t shfl_xor<uint64_t>(uint64_t v, int m) {
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, WAVE);
    uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, WAVE);
    return ((uint64_t)hi << 32) | lo;
}

// XCD (chiplet) this wave runs on, 0..7.  Used for speed only (which L2 a tile's writes
// land in), never for correctness.
__device__ __forceinline__ unsigned xcc_id() {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    return (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;
}

// Hands out tile numbers so that `chunk` consecutive tiles are taken by workgroups of the
// SAME XCD right after one another: their output runs share cache lines, and only one L2
// can merge them into full-line write-backs.  Chunks are dealt round-robin to eight
// queues (queue q owns chunks q, q+8, ...); a workgroup draws from the queue of its own
// XCD and, once that is empty, from the others, so every tile is handed out exactly once
// whatever the placement.  qcnt: eight zeroed counters.  Call from one thread.
__device__ __forceinline__ unsigned claim_tile(unsigned* qcnt, unsigned ntiles, unsigned chunk) {
    if (chunk == 0) return atomicAdd(qcnt, 1u);
    const unsigned x = xcc_id();
    for (unsigned r = 0; r < 8; ++r) {
        const unsigned q = (x + r) & 7u;
        const unsigned k = atomicAdd(&qcnt[q], 1u);
        const unsigned long long tile = ((unsigned long long)(k / chunk) * 8u + q) * chunk + k % chunk;
        if (tile < ntiles) return (unsigned)tile;
    }
    return ntiles;   // unreachable when the grid has exactly ntiles workgroups
}

struct OpSum {
    template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; }
};
struct OpMax {
    template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; }
};
struct OpOr {
    template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a | b; }
};
struct OpAnd {
    template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a & b; }
};
struct OpMin {
    template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a < b ? a : b; }
};

template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce(T v, Op op) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = op(v, shfl_xor<T>(v, m));
    return v;
}

// inclusive scan across the 64 lanes of a wave (identity never needed)
template <typename T, typename Op>
__device__ __forceinline__ T wave_scan_inclusive(T v, Op op) {
    const unsigned lane = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        T o = shfl_up<T>(v, d);
        if (lane >= (unsigned)d) v = op(o, v);
    }
    return v;
}

// Block-wide exclusive scan of one value per thread.  `smem` needs BLOCK/64 + 1
// entries of T.  Returns the exclusive prefix; *total gets the block aggregate.
template <int BLOCK, typename T, typename Op>
__device__ __forceinline__ T block_scan_exclusive(T v, Op op, T identity, T* smem, T* total) {
    constexpr int NW = BLOCK / WAVE;
    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x / WAVE;
    T inc = wave_scan_inclusive<T>(v, op);
    if (lane == WAVE - 1) smem[wave] = inc;
    __syncthreads();
    T wprefix = identity;
    T tot = identity;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        T s = smem[w];
        if ((unsigned)w < wave) wprefix = op(wprefix, s);
        tot = op(tot, s);
    }
    T prev = shfl_up<T>(inc, 1);
    T excl = (lane == 0) ? wprefix : op(wprefix, prev);
    *total = tot;
    __syncthreads();   // smem reusable afterwards
    return excl;
}

template <int BLOCK, typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, Op op, T* smem) {
    constexpr int NW = BLOCK / WAVE;
    v = wave_reduce<T>(v, op);
    if (lane_id() == 0) smem[threadIdx.x / WAVE] = v;
    __syncthreads();
    T r = smem[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = op(r, smem[w]);
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------
// Decoupled look-back descriptors.  One naturally aligned word carries
// {status, value}; it is written by ONE relaxed agent-scope store and read by
// relaxed agent-scope loads, so no fence is needed (the data is the flag).
// Status 0 = not yet published (arrays are zeroed by hipMemsetAsync before
// every launch), 1 = tile aggregate, 2 = inclusive prefix.
// ---------------------------------------------------------------------------
template <typename D> struct Desc;
template <> struct Desc<uint64_t> {
    static constexpr int SHIFT = 62;
    static constexpr uint64_t MASK = (1ull << 62) - 1ull;
};
template <> struct Desc<uint32_t> {
    static constexpr int SHIFT = 30;
    static constexpr uint32_t MASK = (1u << 30) - 1u;
};

template <typename D> __device__ __forceinline__ void desc_store(D* p, unsigned status, D value) {
    __hip_atomic_store(p, (D)(((D)status << Desc<D>::SHIFT) | (value & Desc<D>::MASK)),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename D> __device__ __forceinline__ D desc_load(const D* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A spin that can never hang the GPU: after SPIN_LIMIT polls the kernel raises
// the error word and carries on with garbage; the host turns it into
// PSACX_EDEVICE.
constexpr unsigned SPIN_LIMIT = 1u << 24;

template <typename D>
__device__ __forceinline__ D desc_wait(const D* p, unsigned* err) {
    D d = desc_load<D>(p);
    unsigned spins = 0;
    while ((d >> Desc<D>::SHIFT) == 0) {
        __builtin_amdgcn_s_sleep(1);
        d = desc_load<D>(p);
        if (++spins > SPIN_LIMIT) { atomicOr(err, 1u); break; }
    }
    return d;
}

// Exclusive prefix of `aggregate` over all earlier tiles, computed by the 64
// lanes of ONE wave (call from wave 0 with a wave-uniform aggregate).  Publishes
// this tile's aggregate first and its inclusive prefix at the end.
template <typename Op>
__device__ __forceinline__ uint64_t lookback_wave(uint64_t* desc, unsigned tile, uint64_t aggregate,
                                                  Op op, uint64_t identity, unsigned* err) {
    const unsigned lane = lane_id();
    if (tile == 0) {
        if (lane == 0) desc_store<uint64_t>(desc, 2u, aggregate);
        return identity;
    }
    if (lane == 0) desc_store<uint64_t>(desc + tile, 1u, aggregate);
    uint64_t excl = identity;
    long long base = (long long)tile - 1;
    while (true) {
        long long t = base - (long long)lane;
        uint64_t d = (2ull << 62) | (identity & Desc<uint64_t>::MASK);   // before tile 0: inclusive identity
        if (t >= 0) d = desc_wait<uint64_t>(desc + t, err);
        const bool inc = (d >> 62) == 2u;
        const uint64_t incmask = __ballot(inc);
        const unsigned first = incmask ? (unsigned)__builtin_ctzll(incmask) : 64u;
        uint64_t v = (lane <= first) ? (d & Desc<uint64_t>::MASK) : identity;
        excl = op(excl, wave_reduce<uint64_t>(v, op));
        if (incmask) break;
        base -= WAVE;
    }
    if (lane == 0) desc_store<uint64_t>(desc + tile, 2u, op(excl, aggregate));
    return excl;
}

// ---------------------------------------------------------------------------
// ITEMS consecutive elements per thread, moved as 16-byte vectors when the run is
// complete and the address is 16-byte aligned (always true for workspace arrays
// and tile-aligned offsets), element-wise with bounds checks otherwise.
template <typename T, int ITEMS>
__device__ __forceinline__ void load_run(const T* __restrict__ p, uint64_t e0, uint64_t n, T (&out)[ITEMS], T fill) {
    constexpr int PER = 16 / sizeof(T);
    static_assert(ITEMS % PER == 0, "ITEMS must cover whole 16-byte vectors");
    const T* q = p + e0;
    if (e0 + ITEMS <= n && (reinterpret_cast<uintptr_t>(q) & 15u) == 0) {
        const uint4* v = reinterpret_cast<const uint4*>(q);
#pragma unroll
        for (int i = 0; i < ITEMS / PER; ++i) {
            const uint4 x = v[i];
            if constexpr (sizeof(T) == 4) {
                out[i * 4 + 0] = (T)x.x; out[i * 4 + 1] = (T)x.y; out[i * 4 + 2] = (T)x.z; out[i * 4 + 3] = (T)x.w;
            } else {
                out[i * 2 + 0] = (T)(((uint64_t)x.y << 32) | x.x);
                out[i * 2 + 1] = (T)(((uint64_t)x.w << 32) | x.z);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) out[i] = (e0 + i < n) ? q[i] : fill;
    }
}

template <typename T, int ITEMS>
__device__ __forceinline__ void store_run(T* __restrict__ p, uint64_t e0, uint64_t n, const T (&in)[ITEMS]) {
    constexpr int PER = 16 / sizeof(T);
    T* q = p + e0;
    if (e0 + ITEMS <= n && (reinterpret_cast<uintptr_t>(q) & 15u) == 0) {
        uint4* v = reinterpret_cast<uint4*>(q);
#pragma unroll
        for (int i = 0; i < ITEMS / PER; ++i) {
            uint4 x;
            if constexpr (sizeof(T) == 4) {
                x.x = (uint32_t)in[i * 4 + 0]; x.y = (uint32_t)in[i * 4 + 1];
                x.z = (uint32_t)in[i * 4 + 2]; x.w = (uint32_t)in[i * 4 + 3];
            } else {
                const uint64_t a = (uint64_t)in[i * 2 + 0], b = (uint64_t)in[i * 2 + 1];
                x.x = (uint32_t)a; x.y = (uint32_t)(a >> 32); x.z = (uint32_t)b; x.w = (uint32_t)(b >> 32);
            }
            v[i] = x;
        }
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) if (e0 + i < n) q[i] = in[i];
    }
}

// The same runs with the memory side coalesced: a wave reads (writes) its 64 * ITEMS consecutive elements as ITEMS rows of 64
// -- one 64 * sizeof(T)-byte piece per instruction instead of 64 pieces of 16 bytes, 64 bytes apart -- and turns rows into
// per-thread runs through its own padded LDS region xw (XRUN_WORDS<ITEMS> elements; e + e / 8: a thread's run starts nine
// slots after its neighbour's).  e0: first element of THIS thread, consecutive over the lanes of the wave.  LDS operations of a
// wave execute in order, so a region needs no barrier between its uses; the fences only keep the compiler from moving them.
// 64-bit words only: a thread's run of eight 32-bit entries is two 16-byte pieces 32 bytes apart, which the memory side takes
// as it is -- rebucket_first_kernel on 2^28 32-bit records 1.82 ms with plain runs, 2.74 ms through the rows (and 46.2 against
// 39.3 ms the other way round on 2^32 64-bit records) -- so for them load_run_x / store_run_x ARE load_run / store_run.
template <int ITEMS> struct XRUN_WORDS { static constexpr int N = WAVE * ITEMS + WAVE * ITEMS / 8; };
__device__ __forceinline__ void xrun_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <typename T, int ITEMS>
__device__ __forceinline__ void load_run_x(const T* __restrict__ p, uint64_t e0, uint64_t n, T (&out)[ITEMS], T fill, T* xw) {
    if constexpr (sizeof(T) < 8) { load_run<T, ITEMS>(p, e0, n, out, fill); return; }
    const unsigned lane = lane_id();
    const uint64_t wb = e0 - (uint64_t)lane * ITEMS;
    const T* __restrict__ q = p + wb + lane;
    T row[ITEMS];
    if (wb + (uint64_t)WAVE * ITEMS <= n) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) row[i] = q[i * WAVE];
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) row[i] = (wb + (uint64_t)i * WAVE + lane < n) ? q[i * WAVE] : fill;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) { const unsigned s = (unsigned)i * WAVE + lane; xw[s + (s >> 3)] = row[i]; }
    xrun_order();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { const unsigned s = lane * ITEMS + j; out[j] = xw[s + (s >> 3)]; }
    xrun_order();
}
template <typename T, int ITEMS>
__device__ __forceinline__ void store_run_x(T* __restrict__ p, uint64_t e0, uint64_t n, const T (&in)[ITEMS], T* xw) {
    if constexpr (sizeof(T) < 8) { store_run<T, ITEMS>(p, e0, n, in); return; }
    const unsigned lane = lane_id();
    const uint64_t wb = e0 - (uint64_t)lane * ITEMS;
    T* __restrict__ q = p + wb + lane;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { const unsigned s = lane * ITEMS + j; xw[s + (s >> 3)] = in[j]; }
    xrun_order();
    T row[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) { const unsigned s = (unsigned)i * WAVE + lane; row[i] = xw[s + (s >> 3)]; }
    xrun_order();
    if (wb + (uint64_t)WAVE * ITEMS <= n) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) q[i * WAVE] = row[i];
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) if (wb + (uint64_t)i * WAVE + lane < n) q[i * WAVE] = row[i];
    }
}

template <typename T> __device__ __forceinline__ unsigned clz_t(T x);
template <> __device__ __forceinline__ unsigned clz_t<uint32_t>(uint32_t x) { return x ? __clz((int)x) : 32u; }
template <> __device__ __forceinline__ unsigned clz_t<uint64_t>(uint64_t x) { return x ? __clzll((long long)x) : 64u; }
template <typename T> __device__ __forceinline__ unsigned ctz_t(T x);      // x != 0
template <> __device__ __forceinline__ unsigned ctz_t<uint32_t>(uint32_t x) { return (unsigned)__ffs((int)x) - 1u; }
template <> __device__ __forceinline__ unsigned ctz_t<uint64_t>(uint64_t x) { return (unsigned)__ffsll((long long)x) - 1u; }

} // namespace psacx
