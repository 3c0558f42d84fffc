// nsv.hpp -- nearest-smaller-value search in a 64-ary min-pyramid (shared by ansv.hip and the
// distributed step ops).  Semantics: /root/reference/include/ansv.hpp:48-65, tie rules
// ansv_common.hpp:20-22.
#pragma once
#include "engine.hpp"

namespace psacx {

constexpr uint64_t NSV_NONE = ~0ull;

// nearest j < i (LEFT) or j > i (!LEFT) with in[j] < v (strict) or in[j] <= v
template <typename T, bool LEFT>
__device__ __forceinline__ uint64_t nsv_search(const Pyramid<T>& P, uint64_t i, T v, bool strict) {
    uint64_t pos = i, j = 0;
    int L = 0;
    bool found = false;
    while (!found) {
        const T* a = P.lvl[L];
        const uint64_t len = P.len[L];
        if (LEFT) {
            const uint64_t gstart = pos & ~63ull;
            for (uint64_t c = pos; c-- > gstart;) {
                const T x = a[c];
                if (strict ? x < v : x <= v) { j = c; found = true; break; }
            }
            if (!found && gstart == 0) return NSV_NONE;
        } else {
            uint64_t gend = (pos | 63ull) + 1;
            if (gend > len) gend = len;
            for (uint64_t c = pos + 1; c < gend; ++c) {
                const T x = a[c];
                if (strict ? x < v : x <= v) { j = c; found = true; break; }
            }
            if (!found && gend >= len) return NSV_NONE;
        }
        if (!found) { pos >>= 6; ++L; }     // the top level is a single group, so this never overruns
    }
    while (L > 0) {
        const T* a = P.lvl[L - 1];
        const uint64_t lo = j << 6;
        uint64_t hi = lo + 64;
        if (hi > P.len[L - 1]) hi = P.len[L - 1];
        if (LEFT) {
            for (uint64_t c = hi; c-- > lo;) { const T x = a[c]; if (strict ? x < v : x <= v) { j = c; break; } }
        } else {
            for (uint64_t c = lo; c < hi; ++c) { const T x = a[c]; if (strict ? x < v : x <= v) { j = c; break; } }
        }
        --L;
    }
    return j;
}


// ---- wave-cooperative forms: all 64 lanes of the calling wave take part, arguments are wave-uniform.
// One step looks at a whole 64-entry group with one coalesced load and one ballot, so a search costs a
// handful of memory round trips instead of one per entry walked.
// skip: the caller knows that the groups of `pos` on the first `skip` levels hold nothing on the searched side
// (pos is the edge element of an aligned tile), the walk starts above them
template <typename T, bool LEFT>
__device__ __forceinline__ uint64_t nsv_search_wave(const Pyramid<T>& P, uint64_t pos, T v, bool strict, int skip = 0) {
    const unsigned lane = lane_id();
    uint64_t p = pos, j = 0;
    int L = 0;
    while (L < skip && L + 1 < P.nlev) { p >>= 6; ++L; }
    for (;;) {
        const T* a = P.lvl[L];
        const uint64_t len = P.len[L];
        const uint64_t gstart = p & ~63ull;
        const uint64_t idx = gstart + lane;
        const bool in = idx < len && (LEFT ? idx < p : idx > p);
        const T x = in ? a[idx] : (T)0;
        const uint64_t m = __ballot(in && (strict ? x < v : x <= v));
        if (m) { j = gstart + (LEFT ? 63u - (unsigned)__builtin_clzll(m) : (unsigned)__builtin_ctzll(m)); break; }
        if (LEFT ? gstart == 0 : gstart + 64 >= len) return NSV_NONE;
        p >>= 6; ++L;
    }
    while (L > 0) {
        const T* a = P.lvl[L - 1];
        const uint64_t idx = (j << 6) + lane;
        const bool in = idx < P.len[L - 1];
        const T x = in ? a[idx] : (T)0;
        const uint64_t m = __ballot(in && (strict ? x < v : x <= v));      // never empty: the parent qualified
        j = (j << 6) + (LEFT ? 63u - (unsigned)__builtin_clzll(m) : (unsigned)__builtin_ctzll(m));
        --L;
    }
    return j;
}

// type 0 nearest_sm, 1 nearest_eq, 2 furthest_eq (ansv_common.hpp:20-22) for element i with value v
template <typename T, bool LEFT>
__device__ __forceinline__ uint64_t nsv_typed_wave(const Pyramid<T>& P, uint64_t n, uint64_t i, T v, int type) {
    if (type == 0) return nsv_search_wave<T, LEFT>(P, i, v, true);
    const uint64_t j = nsv_search_wave<T, LEFT>(P, i, v, false);
    if (type == 1 || j == NSV_NONE) return j;
    const T u = P.lvl[0][j];
    const uint64_t s = nsv_search_wave<T, LEFT>(P, j, u, true);      // first strictly smaller beyond j
    if (LEFT) {
        if (s == NSV_NONE) { if (P.lvl[0][0] <= u) return 0; return nsv_search_wave<T, false>(P, 0, u, false); }
        return nsv_search_wave<T, false>(P, s, u, false);
    } else {
        if (s == NSV_NONE) { if (P.lvl[0][n - 1] <= u) return n - 1; return nsv_search_wave<T, true>(P, n - 1, u, false); }
        return nsv_search_wave<T, true>(P, s, u, false);
    }
}

// levels of a search pyramid over `m` values: the top level is a single group of <= 64 entries
template <typename T>
inline void nsv_pyramid_layout(Arena& a, const T* values, uint64_t m, Pyramid<T>& P) {
    P.lvl[0] = const_cast<T*>(values); P.len[0] = m; P.nlev = 1;
    uint64_t len = m;
    while (len > 64 && P.nlev < PYR_MAX) {
        len = (len + 63) / 64;
        P.lvl[P.nlev] = a.take<T>(len); P.len[P.nlev] = len; P.nlev++;
    }
}

} // namespace psacx
