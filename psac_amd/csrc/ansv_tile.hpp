// ansv_tile.hpp -- all nearest smaller values, tile form (the kernel behind psacx_ansv_* and the suffix-tree
// topology).  Semantics: /root/reference/include/ansv.hpp:48-65 (ansv_sequential), tie rules
// ansv_common.hpp:20-22 (nearest_sm / nearest_eq / furthest_eq), result contract ansv.hpp:2042-2051.
//
// The reference walks one monotone stack per rank.  Here a workgroup owns a tile of TB blocks of 64 consecutive
// elements, one element per lane, and every search is a binary descent over window minima -- O(log) steps for
// every lane at once, no data-dependent loops:
//   level 0  inside the own 64-block: the minima of the 1, 2, 4 .. 32 elements before (after) every lane are
//            built with 6 shuffles; 6 more (lane-indexed) find the nearest smaller element of every lane;
//   level 1  inside the tile: the same two steps over the TB block minima, which every wave holds one per lane,
//            then a 6-step binary search in the per-block suffix (prefix) minima kept in LDS;
//   beyond   only running minima of the tile are left.  Their answer depends on their VALUE alone (everything
//            between them and the tile edge is larger), so one wave-cooperative walk of the global 64-ary
//            min-pyramid per distinct value and side is shared through a small LDS table.
// furthest_eq adds pointer jumping over "same value, nothing smaller in between" links inside the tile (LDS,
// log2(tile) rounds) and one shared global query per value for runs that cross the tile edge.
// HBM traffic: the input once (plus 1/63 for the pyramid), both outputs once, coalesced.
#pragma once
#include "nsv.hpp"

namespace psacx {

constexpr int ANSV_WAVES = 4;
constexpr int ANSV_THREADS = ANSV_WAVES * WAVE;
constexpr unsigned ANSV_MEMO = 16;
constexpr uint64_t ANSV_NOCONT = ~0ull - 1;      // a run of equal values does not continue beyond the tile edge

template <typename T> struct AnsvTile { static constexpr int TB = sizeof(T) == 4 ? 64 : 32; };   // 64-blocks per tile

template <typename T> struct AnsvMemo {
    T val[ANSV_MEMO];
    unsigned long long res[ANSV_MEMO];
    unsigned kind[ANSV_MEMO];
    unsigned ready[ANSV_MEMO];
    unsigned cnt;
};

template <typename T, int TB> struct AnsvShared {
    T sm[TB * 64];              // sm[e] = min(v[e .. end of its block])
    T pm[TB * 64];              // pm[e] = min(v[start of its block .. e])
    T bm[64];                   // block minima (all ones beyond the tile)
    uint16_t link[2][TB * 64];  // equal-run links of furthest_eq (ping-pong)
    AnsvMemo<T> memo[2];        // shared answers of searches that leave the tile, per side
};

template <typename T> __device__ __forceinline__ T shfl_dn(T v, unsigned d) { return shfl<T>(v, (int)((lane_id() + d) & 63u)); }

// window minima before every lane: M[j][x] = min(v[x - 2^j .. x - 1]) clipped to the block (all ones if empty)
template <typename T> __device__ __forceinline__ void ansv_tables_left(T v, T (&M)[6]) {
    const unsigned lane = lane_id();
    const T up = shfl_up<T>(v, 1);
    M[0] = lane >= 1 ? up : ~(T)0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const T o = shfl_up<T>(M[j], 1 << j);
        const T c = lane >= (1u << j) ? o : ~(T)0;
        M[j + 1] = c < M[j] ? c : M[j];
    }
}
// window minima after every lane: R[j][x] = min(v[x + 1 .. x + 2^j]) clipped to the block
template <typename T> __device__ __forceinline__ void ansv_tables_right(T v, T (&R)[6]) {
    const unsigned lane = lane_id();
    const T dn = shfl_dn<T>(v, 1);
    R[0] = lane < 63 ? dn : ~(T)0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const T o = shfl_dn<T>(R[j], 1u << j);
        const T c = lane + (1u << j) <= 63 ? o : ~(T)0;
        R[j + 1] = c < R[j] ? c : R[j];
    }
}

// Nearest x < start (LEFT) / x > start (!LEFT) among the 64 entries the tables describe with entry < thr
// (strict) or <= thr.  start and thr are per lane.  Returns 64 when there is none.
template <typename T, bool LEFT>
__device__ __forceinline__ unsigned ansv_descend(const T (&W)[6], unsigned start, T thr, bool strict) {
    unsigned pos = start;
#pragma unroll
    for (int j = 5; j >= 0; --j) {
        const unsigned step = 1u << j;
        const T w = shfl<T>(W[j], (int)pos);
        const bool has = strict ? w < thr : w <= thr;
        if (!has) pos = LEFT ? (pos >= step ? pos - step : 0u) : (pos + step <= 63u ? pos + step : 63u);
    }
    if (LEFT) return pos > 0 ? pos - 1 : 64u;
    return pos < 63 ? pos + 1 : 64u;
}

template <typename T>
__device__ __forceinline__ bool ansv_memo_find(AnsvMemo<T>& m, T v, unsigned kind, uint64_t* res) {
    const unsigned lane = lane_id();
    unsigned c = __hip_atomic_load(&m.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (c > ANSV_MEMO) c = ANSV_MEMO;
    bool hit = false;
    if (lane < c && __hip_atomic_load(&m.ready[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP))
        hit = m.val[lane] == v && m.kind[lane] == kind;
    const uint64_t b = __ballot(hit);
    if (!b) return false;
    *res = m.res[__builtin_ctzll(b)];
    return true;
}
template <typename T>
__device__ __forceinline__ void ansv_memo_add(AnsvMemo<T>& m, T v, unsigned kind, uint64_t res) {
    if (lane_id() == 0) {
        const unsigned idx = atomicAdd(&m.cnt, 1u);
        if (idx < ANSV_MEMO) {
            m.val[idx] = v; m.kind[idx] = kind; m.res[idx] = res;
            __hip_atomic_store(&m.ready[idx], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// Answer of a search that leaves the tile (whole wave, wave-uniform arguments).  kind 0: the typed nearest
// smaller value beyond the tile edge for value v; kind 1 (furthest_eq): the far end of the run of values equal
// to v if the run continues beyond the edge, ANSV_NOCONT otherwise.
template <typename T, bool LEFT>
__device__ __forceinline__ uint64_t ansv_global(const Pyramid<T>& P, uint64_t n, uint64_t tile_base, uint64_t tile_end,
                                                T v, int type, unsigned kind, AnsvMemo<T>& memo) {
    uint64_t r;
    if (ansv_memo_find<T>(memo, v, kind, &r)) return r;
    const bool edge = LEFT ? tile_base == 0 : tile_end >= n;            // nothing beyond the edge
    const uint64_t start = LEFT ? tile_base : tile_end - 1;             // searches look strictly beyond `start`
    if (kind == 0) {
        r = edge ? NSV_NONE : nsv_typed_wave<T, LEFT>(P, n, start, v, type);
    } else {
        r = ANSV_NOCONT;
        if (!edge) {
            const uint64_t j = nsv_search_wave<T, LEFT>(P, start, v, false);
            if (j != NSV_NONE && P.lvl[0][j] == v) r = nsv_typed_wave<T, LEFT>(P, n, start, v, 2);
        }
    }
    ansv_memo_add<T>(memo, v, kind, r);
    return r;
}

// One side of the tile.  val[k]: the lane's element of block (wave * BPW + k).  bmv: block minimum of block
// `lane`.  out: result array of the side.
template <typename T, int TB, bool LEFT>
__device__ __forceinline__ void ansv_side(AnsvShared<T, TB>& sh, const Pyramid<T>& P, uint64_t n, uint64_t tile_base,
                                          const T (&val)[TB / ANSV_WAVES], T bmv, int type, uint64_t nonsv,
                                          uint64_t* __restrict__ out) {
    constexpr int BPW = TB / ANSV_WAVES;
    constexpr unsigned TILE = TB * 64;
    constexpr unsigned MASK = 0x7FFFu, EXT = 0x8000u, PEND = 0xFFFFu;
    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x / WAVE;
    const bool strict = type == 0;
    const uint64_t tile_end = tile_base + TILE < n ? tile_base + TILE : n;
    AnsvMemo<T>& memo = sh.memo[LEFT ? 0 : 1];

    T BW[6];                                   // window minima over the block minima of the tile
    if (LEFT) ansv_tables_left<T>(bmv, BW); else ansv_tables_right<T>(bmv, BW);

    unsigned pend = 0, pend_cont = 0;          // bit k: the search of block k's element leaves the tile
    unsigned code[BPW];                        // furthest_eq: tile position of the nearest <= element, or PEND
    T q[BPW];                                  // furthest_eq: value whose run has to be followed
#pragma unroll
    for (int k = 0; k < BPW; ++k) {
        const unsigned b = wave * BPW + k;
        const unsigned e = b * 64 + lane;
        const uint64_t g = tile_base + e;
        const T v = val[k];
        T W[6];
        if (LEFT) ansv_tables_left<T>(v, W); else ansv_tables_right<T>(v, W);
        unsigned c = ansv_descend<T, LEFT>(W, lane, v, strict);
        unsigned p = PEND;
        T u = 0;
        if (type == 2) { const T uu = shfl<T>(v, (int)(c & 63u)); if (c < 64) u = uu; }
        if (c < 64) p = b * 64 + c;
        // not inside the block: nearest block of the tile with a small enough minimum, then the nearest such
        // element inside it by binary search in its suffix (prefix) minima
        const unsigned bb = ansv_descend<T, LEFT>(BW, b, v, strict);
        const bool need = c >= 64 && bb < 64;
        {
            const unsigned base = (need ? bb : b) * 64;
            int lo = LEFT ? 0 : -1, hi = LEFT ? 64 : 63;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int mid = (lo + hi) >> 1;
                const T x = LEFT ? sh.sm[base + mid] : sh.pm[base + mid];
                const bool ok = strict ? x < v : x <= v;
                if (LEFT) { if (ok) lo = mid; else hi = mid; } else { if (ok) hi = mid; else lo = mid; }
            }
            if (need) {
                p = base + (unsigned)(LEFT ? lo : hi);
                if (type == 2) u = LEFT ? sh.sm[p] : sh.pm[p];
            }
        }
        if (tile_base + p >= n && p != PEND) p = PEND;     // padding past the end of the array is never an answer (right side)
        if (g >= n) { code[k] = PEND; q[k] = 0; continue; }
        if (type != 2) {
            if (p != PEND) out[g] = tile_base + p;
            else pend |= 1u << k;
            code[k] = 0; q[k] = v;
        } else {
            code[k] = p; q[k] = u;
            sh.link[0][e] = (uint16_t)((p != PEND && u == v) ? p : (e | (p == PEND ? EXT : 0u)));
        }
    }
    if (type == 2) {
        // elements past the end of the array must not carry stale links
#pragma unroll
        for (int k = 0; k < BPW; ++k) {
            const unsigned e = (wave * BPW + k) * 64 + lane;
            if (tile_base + e >= n) sh.link[0][e] = (uint16_t)e;
        }
        __syncthreads();
        // far end of every run of equal values inside the tile: link = link[link], log2(TILE) rounds
        int cur = 0;
#pragma unroll 1
        for (unsigned span = 1; span < TILE; span <<= 1) {
#pragma unroll
            for (int k = 0; k < BPW; ++k) {
                const unsigned e = (wave * BPW + k) * 64 + lane;
                sh.link[cur ^ 1][e] = sh.link[cur][sh.link[cur][e] & MASK];
            }
            cur ^= 1;
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < BPW; ++k) {
            const unsigned e = (wave * BPW + k) * 64 + lane;
            const uint64_t g = tile_base + e;
            if (g >= n) continue;
            if (code[k] == PEND) { pend |= 1u << k; q[k] = val[k]; continue; }       // nearest <= lies beyond the tile
            const unsigned hh = sh.link[cur][code[k]];
            out[g] = tile_base + (hh & MASK);
            if (hh & EXT) pend_cont |= 1u << k;            // the run may go on beyond the tile edge (q[k] = its value)
        }
        __syncthreads();                                   // the link buffers are reused by the other side
    }
    // searches that leave the tile: one shared walk of the global pyramid per distinct value
#pragma unroll
    for (int k = 0; k < BPW; ++k) {
        const uint64_t g = tile_base + (uint64_t)(wave * BPW + k) * 64 + lane;
        uint64_t m = __ballot((pend >> k) & 1u);
        while (m) {
            const int src = __builtin_ctzll(m);
            const T vq = shfl<T>(q[k], src);
            const uint64_t r = ansv_global<T, LEFT>(P, n, tile_base, tile_end, vq, type, 0u, memo);
            const bool mine = ((pend >> k) & 1u) && q[k] == vq;
            if (mine) out[g] = r == NSV_NONE ? nonsv : r;
            m &= ~__ballot(mine);
        }
        m = __ballot((pend_cont >> k) & 1u);
        while (m) {
            const int src = __builtin_ctzll(m);
            const T vq = shfl<T>(q[k], src);
            const uint64_t r = ansv_global<T, LEFT>(P, n, tile_base, tile_end, vq, 2, 1u, memo);
            const bool mine = ((pend_cont >> k) & 1u) && q[k] == vq;
            if (mine && r != ANSV_NOCONT) out[g] = r;
            m &= ~__ballot(mine);
        }
    }
    // elements without any answer on this side inside the tile and beyond were written as NSV_NONE -> nonsv above
}

template <typename T>
__global__ __launch_bounds__(ANSV_THREADS) void ansv_tile_kernel(Pyramid<T> P, uint64_t n, int left_type, int right_type,
                                                                 uint64_t nonsv, uint64_t* __restrict__ left,
                                                                 uint64_t* __restrict__ right) {
    constexpr int TB = AnsvTile<T>::TB;
    constexpr int BPW = TB / ANSV_WAVES;
    constexpr unsigned TILE = TB * 64;
    __shared__ AnsvShared<T, TB> sh;
    const T* __restrict__ in = P.lvl[0];
    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x / WAVE;
    const uint64_t tile_base = (uint64_t)blockIdx.x * TILE;
    if (threadIdx.x < 64) sh.bm[threadIdx.x] = ~(T)0;
    if (threadIdx.x < 2) sh.memo[threadIdx.x].cnt = 0;
    if (threadIdx.x < 2 * ANSV_MEMO) sh.memo[threadIdx.x / ANSV_MEMO].ready[threadIdx.x % ANSV_MEMO] = 0;
    __syncthreads();
    T val[BPW];
#pragma unroll
    for (int k = 0; k < BPW; ++k) {
        const uint64_t g = tile_base + (uint64_t)(wave * BPW + k) * 64 + lane;
        val[k] = g < n ? in[g] : ~(T)0;
    }
#pragma unroll
    for (int k = 0; k < BPW; ++k) {
        const unsigned b = wave * BPW + k;
        const unsigned e = b * 64 + lane;
        const T pre = wave_scan_inclusive<T>(val[k], OpMin());
        const T rev = shfl<T>(val[k], 63 - (int)lane);
        const T srv = wave_scan_inclusive<T>(rev, OpMin());
        const T suf = shfl<T>(srv, 63 - (int)lane);
        sh.pm[e] = pre; sh.sm[e] = suf;
        if (lane == 63) sh.bm[b] = pre;
    }
    __syncthreads();
    const T bmv = sh.bm[lane];
    ansv_side<T, TB, true>(sh, P, n, tile_base, val, bmv, left_type, nonsv, left);
    ansv_side<T, TB, false>(sh, P, n, tile_base, val, bmv, right_type, nonsv, right);
}

} // namespace psacx
