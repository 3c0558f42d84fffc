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
