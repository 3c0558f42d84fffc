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

// A tile is a chain of dependent LDS round trips per wave: 32-bit values take eight waves per tile (more, shorter chains;
// 4.7 against 5.4 ms at 2^28), 64-bit values four (their register count leaves no room for more waves anyway).
template <typename T> struct AnsvWaves { static constexpr int N = sizeof(T) == 4 ? 8 : 4; };
constexpr unsigned ANSV_MEMO = 16;
constexpr uint64_t ANSV_NOCONT = ~0ull - 1;      // a run of equal values does not continue beyond the tile edge

template <typename T> struct AnsvTile { static constexpr int TB = sizeof(T) == 4 ? 64 : 32; };   // 64-blocks per tile
// a search that starts at a tile edge finds nothing in the edge element's own 64-block, and nothing in the 64 blocks
// around it when a tile is exactly one level-1 group: the global walk may start that many levels up
template <typename T> struct ANSV_SKIP { static constexpr int LEVELS = AnsvTile<T>::TB == 64 ? 2 : 1; };

template <typename T> struct AnsvMemo {
    T val[ANSV_MEMO];
    unsigned long long res[ANSV_MEMO];      // the answer
    unsigned long long first[ANSV_MEMO];    // the nearest qualifying element the answer was derived from (NSV_NONE: none)
    unsigned kind[ANSV_MEMO];
    unsigned ready[ANSV_MEMO];
    unsigned cnt;
};

template <typename T, int TB, bool LF, bool RF> struct AnsvShared {
    T sm[TB * 64];              // sm[e] = min(v[e .. end of its block])
    T pm[TB * 64];              // pm[e] = min(v[start of its block .. e])
    T bm[64];                   // block minima (all ones beyond the tile)
    // furthest_eq only: tile position of the nearest <= element of every element (bit 15: it has the same value;
    // 0x7FFF: beyond the tile), and the equal-run links derived from it (ping-pong)
    uint16_t code_l[LF ? TB * 64 : 1];
    uint16_t code_r[RF ? TB * 64 : 1];
    // (the two link buffers of furthest_eq live in pm: the prefix minima are dead once the searches of a tile are done)
    __device__ __forceinline__ uint16_t* link(int which) { return reinterpret_cast<uint16_t*>(pm) + (size_t)which * TB * 64; }
    AnsvMemo<T> memo[2];        // shared answers of searches that leave the tile, per side
    int link_cur_left;          // link buffer with the final left-side links of the tile just finished (-1: none)
};

// The kernel is bound by its VALU work (PMC: 520 VALU instructions per 64 elements with the generic __shfl helpers,
// which recompute the lane id, the source lane and its byte address for every call), so the lane moves are issued
// directly: ds_bpermute with the byte address of the source lane (only address bits 7:2 count, so lane * 4 +- 4 d needs
// no wrap-around handling), DPP row shifts and row broadcasts for the two running-minimum scans of 32-bit values.
__device__ __forceinline__ uint32_t bperm(uint32_t v, int byte_addr) { return (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v); }
__device__ __forceinline__ uint64_t bperm(uint64_t v, int byte_addr) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// window minima before every lane: M[j][x] = min(v[x - 2^j .. x - 1]) clipped to the block (all ones if empty)
template <typename T> __device__ __forceinline__ void ansv_tables_left(T v, T (&M)[6]) {
    const int lane = (int)lane_id();
    const int a4 = lane << 2;
    const T up = bperm(v, a4 - 4);
    M[0] = lane >= 1 ? up : ~(T)0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const T o = bperm(M[j], a4 - (4 << j));
        const T c = lane >= (1 << j) ? o : ~(T)0;
        M[j + 1] = c < M[j] ? c : M[j];
    }
}
// window minima after every lane: R[j][x] = min(v[x + 1 .. x + 2^j]) clipped to the block
template <typename T> __device__ __forceinline__ void ansv_tables_right(T v, T (&R)[6]) {
    const int lane = (int)lane_id();
    const int a4 = lane << 2;
    const T dn = bperm(v, a4 + 4);
    R[0] = lane < 63 ? dn : ~(T)0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const T o = bperm(R[j], a4 + (4 << j));
        const T c = lane + (1 << j) <= 63 ? o : ~(T)0;
        R[j + 1] = c < R[j] ? c : R[j];
    }
}

// running minima of the 64 lanes from the left (pre) and from the right (suf)
template <typename T> __device__ __forceinline__ void ansv_min_scans(T v, T* pre, T* suf) {
    const unsigned lane = lane_id();
    *pre = wave_scan_inclusive<T>(v, OpMin());
    const T rev = bperm(v, (int)((63u - lane) << 2));
    const T srv = wave_scan_inclusive<T>(rev, OpMin());
    *suf = bperm(srv, (int)((63u - lane) << 2));
}
#define PSACX_DPP_MIN(x, ctrl, rowmask)                                                                  \
    { const uint32_t t__ = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)(x), ctrl, rowmask, 0xF, false); x = t__ < x ? t__ : x; }
template <> __device__ __forceinline__ void ansv_min_scans<uint32_t>(uint32_t v, uint32_t* pre, uint32_t* suf) {
    // inside the rows of 16 lanes: row_shr / row_shl by 1, 2, 4, 8 (a lane without a source keeps all ones)
    uint32_t p = v, s = v;
    PSACX_DPP_MIN(p, 0x111, 0xF) PSACX_DPP_MIN(p, 0x112, 0xF) PSACX_DPP_MIN(p, 0x114, 0xF) PSACX_DPP_MIN(p, 0x118, 0xF)
    PSACX_DPP_MIN(s, 0x101, 0xF) PSACX_DPP_MIN(s, 0x102, 0xF) PSACX_DPP_MIN(s, 0x104, 0xF) PSACX_DPP_MIN(s, 0x108, 0xF)
    // across rows, from the left: lane 15 of a row to the next row (rows 1 and 3), then lane 31 to rows 2 and 3
    PSACX_DPP_MIN(p, 0x142, 0xA) PSACX_DPP_MIN(p, 0x143, 0xC)
    // across rows, from the right: the first lane of every row holds the row's minimum
    const uint32_t r1 = (uint32_t)__builtin_amdgcn_readlane((int)s, 16), r2 = (uint32_t)__builtin_amdgcn_readlane((int)s, 32),
                   r3 = (uint32_t)__builtin_amdgcn_readlane((int)s, 48);
    const uint32_t q2 = r2 < r3 ? r2 : r3, q1 = r1 < q2 ? r1 : q2;
    const unsigned row = lane_id() >> 4;
    const uint32_t beyond = row == 0 ? q1 : row == 1 ? q2 : row == 2 ? r3 : 0xFFFFFFFFu;
    *pre = p;
    *suf = beyond < s ? beyond : s;
}
#undef PSACX_DPP_MIN

// Nearest x < start (LEFT) / x > start (!LEFT) among the 64 entries the tables describe with entry < thr
// (strict) or <= thr.  start and thr are per lane.  Returns 64 when there is none.
template <typename T, bool LEFT>
__device__ __forceinline__ unsigned ansv_descend(const T (&W)[6], unsigned start, T thr, bool strict) {
    unsigned pos = start;
#pragma unroll
    for (int j = 5; j >= 0; --j) {
        const unsigned step = 1u << j;
        const T w = bperm(W[j], (int)(pos << 2));
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
__device__ __forceinline__ void ansv_memo_add(AnsvMemo<T>& m, T v, unsigned kind, uint64_t res, uint64_t first) {
    if (lane_id() == 0) {
        const unsigned idx = atomicAdd(&m.cnt, 1u);
        if (idx < ANSV_MEMO) {
            m.val[idx] = v; m.kind[idx] = kind; m.res[idx] = res; m.first[idx] = first;
            __hip_atomic_store(&m.ready[idx], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// A workgroup walks its tiles in ascending order and carries the shared answers from tile to tile (wave 0, between
// two tiles, while the LDS arrays of the finished tile are still in place).
// Right side: an answer stays true while the element it was derived from lies beyond the right edge of the next tile.
template <typename T>
__device__ __forceinline__ void ansv_carry_right(AnsvMemo<T>& m, uint64_t next_end) {
    const unsigned lane = lane_id();
    const unsigned c = m.cnt < ANSV_MEMO ? m.cnt : ANSV_MEMO;
    if (lane < c && m.ready[lane] && !(m.first[lane] == NSV_NONE || m.first[lane] >= next_end)) m.ready[lane] = 0;
}
// Left side: when the finished tile holds an element that qualifies for an entry's value, the answer beyond the left edge
// of the NEXT tile is that element (the rightmost one: block minima, then the suffix minima of its block) -- for
// furthest_eq the far end of its run, which the finished tile's links give, continued by the finished tile's own
// "does the run of u go on" entry when the run reaches its left edge.  Otherwise the entry is still true as it stands.
// link_cur: which link buffer holds the finished tile's final links (-1: not available, such entries are dropped).
template <typename T, int TB, bool FUR, typename SH>
__device__ __forceinline__ void ansv_carry_left(SH& sh, int type, T bmv_prev, uint64_t prev_base, int link_cur) {
    AnsvMemo<T>& m = sh.memo[0];
    const unsigned lane = lane_id();
    const unsigned c = m.cnt < ANSV_MEMO ? m.cnt : ANSV_MEMO;
    bool live = lane < c && m.ready[lane] != 0;
    const T ev = live ? m.val[lane] : (T)0;
    const unsigned ek = live ? m.kind[lane] : 0u;
    const uint64_t er_old = live ? (uint64_t)m.res[lane] : 0ull;
    uint64_t er = er_old, ef = live ? (uint64_t)m.first[lane] : 0ull;
    const uint64_t live_mask = __ballot(live);
    for (unsigned idx = 0; idx < c; ++idx) {
        if (!((live_mask >> idx) & 1ull)) continue;
        const T v = shfl<T>(ev, (int)idx);
        const unsigned kind = (unsigned)__shfl((int)ek, (int)idx, WAVE);
        const bool strict = kind == 0 && type == 0;
        const uint64_t bal = __ballot(lane < (unsigned)TB && (strict ? bmv_prev < v : bmv_prev <= v));
        if (!bal) continue;                                   // nothing in the finished tile qualifies: still true
        const unsigned bb = 63u - (unsigned)__builtin_clzll(bal);
        int lo = 0, hi = 64;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int mid = (lo + hi) >> 1;
            const T x = sh.sm[bb * 64 + mid];
            if (strict ? x < v : x <= v) lo = mid; else hi = mid;
        }
        const unsigned p = bb * 64 + (unsigned)lo;
        uint64_t nres = prev_base + p;
        bool drop = false;
        if (FUR) {
            const T u = sh.sm[p];
            if (kind == 1 && u < v) nres = ANSV_NOCONT;       // something smaller comes first: the run of v ends here
            else if (link_cur < 0) drop = true;
            else {
                const unsigned hh = sh.link(link_cur)[p];
                nres = prev_base + (hh & 0x7FFFu);
                if (hh & 0x8000u) {
                    // the run reaches the left edge of the finished tile: its continuation is that tile's own entry for u
                    const uint64_t hit = __ballot(live && ek == 1u && ev == u);
                    if (!hit) drop = true;
                    else { const uint64_t r = shfl<uint64_t>(er_old, __builtin_ctzll(hit)); if (r != ANSV_NOCONT) nres = r; }
                }
            }
        }
        if (lane == idx) { er = nres; ef = prev_base + p; live = !drop; }
    }
    if (lane < c) { m.res[lane] = er; m.first[lane] = ef; m.ready[lane] = live ? 1u : 0u; }
}
// makes room when the table is nearly full of dropped entries (one lane; rare)
template <typename T>
__device__ __forceinline__ void ansv_memo_compact(AnsvMemo<T>& m) {
    if (m.cnt < ANSV_MEMO - 4) return;
    const unsigned c = m.cnt < ANSV_MEMO ? m.cnt : ANSV_MEMO;
    unsigned o = 0;
    for (unsigned i = 0; i < c; ++i) {
        if (!m.ready[i]) continue;
        if (o != i) { m.val[o] = m.val[i]; m.kind[o] = m.kind[i]; m.res[o] = m.res[i]; m.first[o] = m.first[i]; m.ready[o] = 1; }
        ++o;
    }
    for (unsigned i = o; i < ANSV_MEMO; ++i) m.ready[i] = 0;
    m.cnt = o;
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
    uint64_t j = NSV_NONE;
    r = kind == 0 ? NSV_NONE : ANSV_NOCONT;
    if (!edge) {
        j = nsv_search_wave<T, LEFT>(P, start, v, kind == 0 && type == 0, ANSV_SKIP<T>::LEVELS);
        if (kind == 0) {
            r = j;
            if (type == 2 && j != NSV_NONE) r = nsv_typed_wave<T, LEFT>(P, n, start, v, 2);
        } else if (j != NSV_NONE && P.lvl[0][j] == v) {
            r = nsv_typed_wave<T, LEFT>(P, n, start, v, 2);
        }
    }
    ansv_memo_add<T>(memo, v, kind, r, j);
    return r;
}

// Nearest element of the tile with a value < v (strict) or <= v on one side of element `lane` of block b.
// Returns its tile position or PEND (0x7FFF) when the search leaves the tile; *u receives its value when WANT_U.
constexpr unsigned ANSV_PEND = 0x7FFFu;
template <typename T, int TB, bool LEFT, bool WANT_U, typename SH>
__device__ __forceinline__ unsigned ansv_tile_search(SH& sh, const T (&BW)[6], unsigned b, T v, bool strict, uint64_t tile_base,
                                                     uint64_t n, T* u) {
    const unsigned lane = lane_id();
    T W[6];
    if (LEFT) ansv_tables_left<T>(v, W); else ansv_tables_right<T>(v, W);
    const unsigned c = ansv_descend<T, LEFT>(W, lane, v, strict);
    unsigned p = ANSV_PEND;
    T uu = 0;
    if (WANT_U) { const T x = bperm(v, (int)(c << 2)); if (c < 64) uu = x; }
    if (c < 64) p = b * 64 + c;
    // not inside the block: nearest block of the tile with a small enough minimum, then the nearest such element
    // inside it by binary search in its suffix (prefix) minima
    const unsigned bb = ansv_descend<T, LEFT>(BW, b, v, strict);
    const bool need = c >= 64 && bb < (unsigned)TB;
    if (__ballot(need)) {
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
            if (WANT_U) uu = LEFT ? sh.sm[p] : sh.pm[p];
        }
    }
    if (p != ANSV_PEND && tile_base + p >= n) p = ANSV_PEND;     // padding past the end of the array is never an answer
    if (WANT_U) *u = uu;
    return p;
}

// Both sides at once, step by step, so that the two chains of dependent lane moves and LDS reads overlap instead of
// following each other (the kernel is bound by these round trips, not by instruction issue).
template <typename T, int TB, bool WANT_UL, bool WANT_UR, typename SH>
__device__ __forceinline__ void ansv_tile_search2(SH& sh, const T (&BL)[6], const T (&BR)[6], unsigned b, T v, bool lstrict, bool rstrict,
                                                  uint64_t tile_base, uint64_t n, unsigned* pl_out, unsigned* pr_out, T* ul_out, T* ur_out, unsigned dbg = 0) {
    const int lane = (int)lane_id();
    const int a4 = lane << 2;
    // window minima inside the block, both directions
    T ML[6], MR[6];
    {
        const T up = bperm(v, a4 - 4), dn = bperm(v, a4 + 4);
        ML[0] = lane >= 1 ? up : ~(T)0;
        MR[0] = lane < 63 ? dn : ~(T)0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const T ol = bperm(ML[j], a4 - (4 << j)), orr = bperm(MR[j], a4 + (4 << j));
            const T cl = lane >= (1 << j) ? ol : ~(T)0, cr = lane + (1 << j) <= 63 ? orr : ~(T)0;
            ML[j + 1] = cl < ML[j] ? cl : ML[j];
            MR[j + 1] = cr < MR[j] ? cr : MR[j];
        }
    }
    // four descents in lockstep: in-block left / right, block minima of the tile left / right
    unsigned p0 = (unsigned)lane, p1 = (unsigned)lane, p2 = b, p3 = b;
#pragma unroll
    for (int j = 5; j >= 0; --j) {
        const unsigned step = 1u << j;
        const T w0 = bperm(ML[j], (int)(p0 << 2)), w1 = bperm(MR[j], (int)(p1 << 2));
        const T w2 = bperm(BL[j], (int)(p2 << 2)), w3 = bperm(BR[j], (int)(p3 << 2));
        if (!(lstrict ? w0 < v : w0 <= v)) p0 = p0 >= step ? p0 - step : 0u;
        if (!(rstrict ? w1 < v : w1 <= v)) p1 = p1 + step <= 63u ? p1 + step : 63u;
        if (!(lstrict ? w2 < v : w2 <= v)) p2 = p2 >= step ? p2 - step : 0u;
        if (!(rstrict ? w3 < v : w3 <= v)) p3 = p3 + step <= 63u ? p3 + step : 63u;
    }
    const unsigned cl = p0 > 0 ? p0 - 1 : 64u, cr = p1 < 63 ? p1 + 1 : 64u;
    const unsigned bl = p2 > 0 ? p2 - 1 : 64u, br = p3 < 63 ? p3 + 1 : 64u;
    unsigned pl = ANSV_PEND, pr = ANSV_PEND;
    T ul = 0, ur = 0;
    if (WANT_UL) { const T x = bperm(v, (int)(cl << 2)); if (cl < 64) ul = x; }
    if (WANT_UR) { const T x = bperm(v, (int)(cr << 2)); if (cr < 64) ur = x; }
    if (cl < 64) pl = b * 64 + cl;
    if (cr < 64) pr = b * 64 + cr;
    const bool needl = cl >= 64 && bl < (unsigned)TB, needr = cr >= 64 && br < (unsigned)TB;
    if (!(dbg & 2u) && __ballot(needl || needr)) {
        // nearest qualifying element inside the block found on the tile level: binary search in its suffix / prefix minima
        const unsigned basel = (needl ? bl : b) * 64, baser = (needr ? br : b) * 64;
        int lol = 0, hil = 64, lor = -1, hir = 63;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int ml = (lol + hil) >> 1, mr = (lor + hir) >> 1;
            const T xl = sh.sm[basel + ml], xr = sh.pm[baser + mr];
            if (lstrict ? xl < v : xl <= v) lol = ml; else hil = ml;
            if (rstrict ? xr < v : xr <= v) hir = mr; else lor = mr;
        }
        if (needl) { pl = basel + (unsigned)lol; if (WANT_UL) ul = sh.sm[pl]; }
        if (needr) { pr = baser + (unsigned)hir; if (WANT_UR) ur = sh.pm[pr]; }
    }
    if (pl != ANSV_PEND && tile_base + pl >= n) pl = ANSV_PEND;
    if (pr != ANSV_PEND && tile_base + pr >= n) pr = ANSV_PEND;     // padding past the end of the array is never an answer
    *pl_out = pl; *pr_out = pr;
    if (WANT_UL) *ul_out = ul;
    if (WANT_UR) *ur_out = ur;
}

// The lanes flagged in `pend` ask for the answer beyond the tile edge for their value myq: one shared walk per
// distinct value (whole wave).  kind 0: out = answer (nonsv if none); kind 1: out = far end of the run if it continues.
template <typename T, bool LEFT>
__device__ __forceinline__ void ansv_resolve_pending(const Pyramid<T>& P, uint64_t n, uint64_t tile_base, uint64_t tile_end,
                                                     bool pend, T myq, int type, unsigned kind, AnsvMemo<T>& memo, uint64_t nonsv,
                                                     uint64_t* __restrict__ out, uint64_t g) {
    uint64_t m = __ballot(pend);
    while (m) {
        const int src = __builtin_ctzll(m);
        const T vq = shfl<T>(myq, src);
        const uint64_t r = ansv_global<T, LEFT>(P, n, tile_base, tile_end, vq, type, kind, memo);
        const bool mine = pend && myq == vq;
        if (mine) {
            if (kind == 0) out[g] = r == NSV_NONE ? nonsv : r;
            else if (r != ANSV_NOCONT) out[g] = r;
        }
        m &= ~__ballot(mine);
    }
}

// furthest_eq, after the searches of all blocks: follow the runs of equal values inside the tile (pointer
// jumping over the links, log2(tile) rounds), then the runs that reach the tile edge beyond it.
template <typename T, int TB, bool LEFT, typename SH>
__device__ __forceinline__ void ansv_finish_furthest(SH& sh, const Pyramid<T>& P, uint64_t n, uint64_t tile_base, uint64_t nonsv,
                                                     uint64_t* __restrict__ out) {
    constexpr int BPW = TB / AnsvWaves<T>::N;
    constexpr unsigned TILE = TB * 64;
    constexpr unsigned MASK = 0x7FFFu, EXT = 0x8000u;
    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x / WAVE;
    const uint16_t* code = LEFT ? sh.code_l : sh.code_r;
    const T* __restrict__ in = P.lvl[0];
    const uint64_t tile_end = tile_base + TILE < n ? tile_base + TILE : n;
    AnsvMemo<T>& memo = sh.memo[LEFT ? 0 : 1];
#pragma unroll 1
    for (int k = 0; k < BPW; ++k) {
        const unsigned e = (wave * BPW + k) * 64 + lane;
        const unsigned cd = code[e];
        const unsigned p = cd & MASK;
        // same value as the nearest <= element: part of its run; otherwise the element heads its own run, which
        // may go on beyond the tile edge when nothing <= was found inside
        sh.link(0)[e] = (uint16_t)((p != ANSV_PEND && (cd & EXT)) ? p : (e | (p == ANSV_PEND ? EXT : 0u)));
    }
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (unsigned span = 1; span < TILE; span <<= 1) {
        // (runs are short on real data: stop as soon as a round moves nothing)
        bool moved = false;
#pragma unroll 4
        for (int k = 0; k < BPW; ++k) {
            const unsigned e = (wave * BPW + k) * 64 + lane;
            const uint16_t a = sh.link(cur)[e];
            const uint16_t b = sh.link(cur)[a & MASK];
            moved |= a != b;
            sh.link(cur ^ 1)[e] = b;
        }
        cur ^= 1;
        if (!__syncthreads_or(moved)) break;
    }
#pragma unroll 1
    for (int k = 0; k < BPW; ++k) {
        const unsigned e = (wave * BPW + k) * 64 + lane;
        const uint64_t g = tile_base + e;
        const bool in_range = g < n;
        const unsigned p = code[e] & MASK;
        const bool direct = in_range && p == ANSV_PEND;              // the nearest <= element lies beyond the tile
        bool cont = false;
        T q = 0;
        if (in_range && !direct) {
            const unsigned hh = sh.link(cur)[p];
            out[g] = tile_base + (hh & MASK);
            cont = (hh & EXT) != 0;                                     // the run may go on beyond the tile edge
            if (cont) q = in[tile_base + p];
        }
        if (direct) q = in[g];
        ansv_resolve_pending<T, LEFT>(P, n, tile_base, tile_end, direct, q, 2, 0u, memo, nonsv, out, g);
        ansv_resolve_pending<T, LEFT>(P, n, tile_base, tile_end, cont, q, 2, 1u, memo, nonsv, out, g);
    }
    if (LEFT && threadIdx.x == 0) sh.link_cur_left = cur;
    __syncthreads();                                                    // the link buffers are reused by the other side
}

template <typename T, bool LF, bool RF>
__global__ __launch_bounds__(AnsvWaves<T>::N * WAVE, (sizeof(T) == 4 && !LF && !RF) ? 8 : 1) void ansv_tile_kernel(Pyramid<T> P, uint64_t n, int left_type, int right_type,
                                                                 uint64_t nonsv, uint64_t* __restrict__ left,
                                                                 uint64_t* __restrict__ right, uint64_t ntiles, unsigned dbg) {
    constexpr int TB = AnsvTile<T>::TB;
    constexpr int BPW = TB / AnsvWaves<T>::N;
    constexpr unsigned TILE = TB * 64;
    typedef AnsvShared<T, TB, LF, RF> SH;
    __shared__ SH sh;
    const T* __restrict__ in = P.lvl[0];
    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x / WAVE;
    const int lt = LF ? 2 : left_type, rt = RF ? 2 : right_type;
    const bool lstrict = lt == 0, rstrict = rt == 0;
    // a contiguous range of tiles per workgroup, so that answers found beyond a tile edge carry over to the next tile
    const uint64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    const uint64_t t_lo = (uint64_t)blockIdx.x * per;
    const uint64_t t_hi = t_lo + per < ntiles ? t_lo + per : ntiles;
    if (threadIdx.x < 2) sh.memo[threadIdx.x].cnt = 0;
    if (threadIdx.x < 2 * ANSV_MEMO) sh.memo[threadIdx.x / ANSV_MEMO].ready[threadIdx.x % ANSV_MEMO] = 0;
    if (threadIdx.x < 64) sh.bm[threadIdx.x] = ~(T)0;       // (entries beyond the tile's blocks stay all ones)
    if (threadIdx.x == 0) sh.link_cur_left = -1;
    T bmv_prev = ~(T)0;
    for (uint64_t t = t_lo; t < t_hi; ++t) {
        const uint64_t tile_base = t * TILE;
        const uint64_t tile_end = tile_base + TILE < n ? tile_base + TILE : n;
        __syncthreads();                       // every wave is done with the previous tile (LDS arrays, shared answers)
        if (t > t_lo && wave == 0) {
            ansv_carry_left<T, TB, LF>(sh, lt, bmv_prev, tile_base - TILE, (LF && !RF) ? sh.link_cur_left : -1);
            ansv_carry_right<T>(sh.memo[1], tile_end);
            if (lane == 0) { ansv_memo_compact<T>(sh.memo[0]); ansv_memo_compact<T>(sh.memo[1]); }
        }
        __syncthreads();
#pragma unroll 2
        for (int k = 0; k < BPW; ++k) {
            const unsigned b = wave * BPW + k;
            const unsigned e = b * 64 + lane;
            const uint64_t g = tile_base + e;
            const T v = g < n ? in[g] : ~(T)0;
            T pre, suf;
            ansv_min_scans<T>(v, &pre, &suf);
            sh.pm[e] = pre; sh.sm[e] = suf;
            if (lane == 63) sh.bm[b] = pre;
        }
        __syncthreads();
        const T bmv = sh.bm[lane];
        bmv_prev = bmv;
        T BL[6], BR[6];                        // window minima over the block minima of the tile
        ansv_tables_left<T>(bmv, BL);
        ansv_tables_right<T>(bmv, BR);
        // (second read of the tile, out of L2; the next block's element is fetched while this one is searched)
        T vnext;
        { const uint64_t g0 = tile_base + (uint64_t)(wave * BPW) * 64 + lane; vnext = g0 < n ? in[g0] : ~(T)0; }
#pragma unroll 1
        for (int k = 0; k < BPW; ++k) {
            const unsigned b = wave * BPW + k;
            const unsigned e = b * 64 + lane;
            const uint64_t g = tile_base + e;
            const bool in_range = g < n;
            const T v = vnext;
            if (k + 1 < BPW) { const uint64_t g1 = g + 64; vnext = g1 < n ? in[g1] : ~(T)0; }
            T ul = 0, ur = 0;
            unsigned pl, pr;
            ansv_tile_search2<T, TB, LF, RF>(sh, BL, BR, b, v, lstrict, rstrict, tile_base, n, &pl, &pr, &ul, &ur, dbg);
            if (LF) sh.code_l[LF ? e : 0] = (uint16_t)(pl | ((pl != ANSV_PEND && ul == v) ? 0x8000u : 0u));
            else {
                if (in_range && pl != ANSV_PEND && !(dbg & 4u)) left[g] = tile_base + pl;
                if (!(dbg & 1u)) ansv_resolve_pending<T, true>(P, n, tile_base, tile_end, in_range && pl == ANSV_PEND, v, lt, 0u, sh.memo[0], nonsv, left, g);
            }
            if (RF) sh.code_r[RF ? e : 0] = (uint16_t)(pr | ((pr != ANSV_PEND && ur == v) ? 0x8000u : 0u));
            else {
                if (in_range && pr != ANSV_PEND && !(dbg & 4u)) right[g] = tile_base + pr;
                if (!(dbg & 1u)) ansv_resolve_pending<T, false>(P, n, tile_base, tile_end, in_range && pr == ANSV_PEND, v, rt, 0u, sh.memo[1], nonsv, right, g);
            }
        }
        if (LF || RF) __syncthreads();
        if (LF) ansv_finish_furthest<T, TB, true>(sh, P, n, tile_base, nonsv, left);
        if (RF) ansv_finish_furthest<T, TB, false>(sh, P, n, tile_base, nonsv, right);
    }
}

template <typename T>
void launch_ansv_seq(psacx_ctx* c, const Pyramid<T>& P, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* d_l, uint64_t* d_r);

// grid: a few workgroups per CU, each with a contiguous share of the tiles
template <typename T>
inline void launch_ansv_tiles(psacx_ctx* c, const Pyramid<T>& P, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* d_l, uint64_t* d_r) {
    constexpr uint64_t TILE = (uint64_t)AnsvTile<T>::TB * 64;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
    const unsigned dbg = 0u;
    // exactly as many workgroups as fit on the chip at once: every workgroup then walks an equal, contiguous share of the tiles
#define PSACX_ANSV(LF, RF)                                                                                                       \
    do {                                                                                                                         \
        int occ = 0;                                                                                                             \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ansv_tile_kernel<T, LF, RF>, AnsvWaves<T>::N * WAVE, 0) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 2; } \
        const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, (uint64_t)c->n_cu * occ);                                     \
        hipLaunchKernelGGL((ansv_tile_kernel<T, LF, RF>), dim3(grid), dim3(AnsvWaves<T>::N * WAVE), 0, c->stream, P, n, lt, rt, nonsv, d_l, d_r, ntiles, dbg); \
    } while (0)
    if (lt != 2 && rt != 2) { launch_ansv_seq<T>(c, P, n, lt, rt, nonsv, d_l, d_r); return; }      // (ansv_seq.hpp: stack walk inside the blocks)
    if (lt == 2 && rt == 2) PSACX_ANSV(true, true);
    else if (lt == 2) PSACX_ANSV(true, false);
    else if (rt == 2) PSACX_ANSV(false, true);
    else PSACX_ANSV(false, false);
#undef PSACX_ANSV
}

} // namespace psacx
