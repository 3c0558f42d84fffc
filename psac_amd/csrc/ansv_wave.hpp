// ansv_wave.hpp -- all nearest smaller values, third tile form: every WAVE owns its tiles (64 runs of 16 elements: a run per lane) and
// works through them on its own -- no workgroup barrier anywhere, so the 16 waves of a CU sit in different phases and their global loads,
// register work and LDS traffic overlap instead of meeting at seven barriers per tile (the round-4 kernel: 71 % of the wave cycles waiting;
// here 49 %).
// Semantics: /root/reference/include/ansv.hpp:48-65 (ansv_sequential), tie rules ansv_common.hpp:20-22, contract ansv.hpp:2042-2051.
//
// One side of one tile is a pass:
//   1. a lane loads its run (64 contiguous bytes), keeps it in registers and compares all pairs: every element gets the position of its
//      answer inside the run, or stays open; the run goes to the wave's LDS area, its minimum stays in the lane;
//   2. the open elements (a third on an LCP array) are compacted into a queue (popcounts + one scan over the lanes);
//   3. the queue, 64 entries a step: binary descent over window minima of the 64 run minima (held one per lane: ansv_tile.hpp), the run
//      found is read from LDS and searched in registers; what finds nothing in the tile asks the table of answers beyond the tile edge
//      (one wave-cooperative walk of the global min-pyramid per distinct value);
//   4. furthest_eq (the nearest <= element, then on through the values equal to IT while nothing smaller lies between): the nearest <=
//      answers of the tile are links where the values agree; pointer doubling over the links gives every chain its far end -- a few
//      rounds over a uint16 array in LDS instead of a second search per element;
//   5. all answers of the tile leave as coalesced 512-byte stores;
//   6. the table of answers beyond the edge is brought up to date against this tile, for the next one on the same side.
// A wave walks its range of tiles from both ends at once: the left side upwards, the right side downwards, so that on either side the
// tile just finished is the one the carried answers have to be checked against (in the round-4 kernel the right side lost its answers at
// every tile and walked the pyramid again: 272 against 147 time units in its queue loop).  The input is read twice (8 of 24 bytes per
// element); the all-pairs work is the same (one side per visit).
#pragma once
#include "ansv_tile.hpp"

#ifndef AW_FINAL_W
#define AW_FINAL_W 4
#endif
#ifndef AW_ABLATE
#define AW_ABLATE 0          // (tools/experiments/ansv_ablate.sh: parts of a pass left out, to time them; results are wrong then.  furthest_eq: 1 pointer
                             //  doubling, 2 links, 4 the answers out, 8 the carried table; 16 no pyramid walks (ansv_tile.hpp); nearest types: 32 the answers out;
                             //  64 the second queue loop, 128 both queue loops, 256 the carried table)
#endif

namespace psacx {

constexpr unsigned AW_PEND = 0xFFFFu;       // the answer lies beyond the edge of the tile
constexpr unsigned AW_DONE = 0xFFFEu;       // written by the search beyond the edge: the store pass leaves it alone
constexpr unsigned AW_SLOT0 = 0xFF00u;      // furthest_eq: AW_SLOT0 + s = the answer lies beyond the edge and has been looked up already: entry s of slot0 / slot1
// slots per tile and side (a tile with more such elements -- one falling run -- leaves the others at AW_PEND: they ask one by one).  64-bit
// values: 16, which keeps a workgroup at 53312 bytes of LDS -- three of them on a CU; with 512 bytes more the occupancy query still says
// three, two are resident, and the grid sized for three runs in two rounds (1.57 against 1.24 ms at 2^26)
template <typename T> struct AwSlots { static constexpr unsigned N = sizeof(T) == 4 ? 32 : 16; };

template <typename T> struct AnsvWaveShared {
    static constexpr int RUN = 16, TILE = 64 * RUN;
    __attribute__((aligned(16))) T v[TILE];                   // the tile
    __attribute__((aligned(16))) uint16_t ans[TILE];          // tile position of every element's answer (nearest types; nearest <= for furthest_eq)
    __attribute__((aligned(16))) uint16_t q[TILE];            // queue of open elements; afterwards the far ends of the chains (furthest_eq)
    AnsvMemo<T> memo[2];                                      // answers beyond the edge, left / right side
    unsigned long long slot0[AwSlots<T>::N];                       // furthest_eq, per element without a <= element in the tile: its answer beyond the edge,
    unsigned long long slot1[AwSlots<T>::N];                       // ... and the far end of the run of its value beyond the edge (ANSV_NOCONT: the run ends with it)
#ifdef AW_PAD_BYTES
    char pad[AW_PAD_BYTES];                                   // (tools/experiments: fewer workgroups per CU)
#endif
};

// (four waves a workgroup: four workgroups of 32-bit values, three of 64-bit values on the 160 KB of a CU)
static_assert(sizeof(AnsvWaveShared<uint32_t>) * 4 <= 40960 && sizeof(AnsvWaveShared<uint64_t>) * 4 <= 53312, "LDS of the ANSV kernel: a workgroup fewer per CU");

// rightmost (LEFT) / leftmost index of the 16 values that qualifies (-1: none)
template <typename T, bool LEFT>
__device__ __forceinline__ int answ_in_run(const T (&a)[16], T x, bool strict) {
    int r = -1;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int i = LEFT ? s : 15 - s;
        const bool ok = strict ? a[i] < x : a[i] <= x;
        r = ok ? i : r;
    }
    return r;
}
template <typename T>
__device__ __forceinline__ void answ_load_run(const T* __restrict__ p, T (&a)[16]) {       // p: 16-byte aligned, LDS or global
    constexpr int PER = 16 / sizeof(T);
    typedef T vec __attribute__((ext_vector_type(PER)));
    const vec* __restrict__ q = reinterpret_cast<const vec*>(p);
#pragma unroll
    for (int c = 0; c < 16 / PER; ++c) {
        const vec w = q[c];
#pragma unroll
        for (int d = 0; d < PER; ++d) a[c * PER + d] = w[d];
    }
}

// type: 0 nearest_sm, 1 nearest_eq, 2 furthest_eq (FUR)

// the run of this lane in tile t (padding past the end of the array: all ones)
template <typename T>
__device__ __forceinline__ void answ_fetch(const T* __restrict__ in, uint64_t n, uint64_t t, T (&a)[16]) {
    const uint64_t g0 = t * AnsvWaveShared<T>::TILE + (uint64_t)lane_id() * 16;
    if ((reinterpret_cast<uintptr_t>(in) & 15u) == 0 && g0 + 16 <= n) answ_load_run<T>(in + g0, a);
    else {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = g0 + i < n ? in[g0 + i] : ~(T)0;
    }
}

// (asking for the next pass's run as soon as the registers of this one are free was measured: 16 registers held through the pass, a few
//  spills, nearest_sm pair 2.69 against 2.60 ms -- the pass is bound by its instruction count, not by the load at its head)
// TYPE: 0 nearest_sm, 1 nearest_eq, 2 furthest_eq -- a template parameter since round 6: the compare of every pair is one instruction
// instead of two compares and a select, and a kernel holds the code of its two passes only (45 - 90 KB of code per kernel before, against
// 64 KB of instruction cache shared by two CUs)
template <typename T, bool LEFT, int TYPE>
__device__ __forceinline__ void ansv_wave_pass(AnsvWaveShared<T>& sh, const Pyramid<T>& P, uint64_t n, uint64_t t, uint64_t nonsv,
                                               uint64_t* __restrict__ out) {
    typedef AnsvWaveShared<T> SH;
    constexpr unsigned TILE = SH::TILE, RUN = SH::RUN;
    constexpr int SKIP = 1;                       // the tile edges are multiples of 64: nothing beyond them on level 0 of the pyramid
    const T* __restrict__ in = P.lvl[0];
    const unsigned lane = lane_id();
    constexpr bool FUR = TYPE == 2;
    constexpr int type = TYPE;
    constexpr bool strict = TYPE == 0;
    AnsvMemo<T>& memo = sh.memo[LEFT ? 0 : 1];
    const uint64_t tile_base = t * TILE;
    const uint64_t tile_end = tile_base + TILE < n ? tile_base + TILE : n;
    const unsigned n_rel = (unsigned)(tile_end - tile_base);          // elements of the tile that exist (the rest is padding)
    // ---- 1. the run in registers and in LDS, all pairs
    T mn;
    unsigned total;
    unsigned nslots = 0;                          // (furthest_eq) elements whose answers beyond the edge have been looked up, wave-uniform
    MemoRegs<T> mr;
    {
        T a[16];
        answ_fetch<T>(in, n, t, a);
        {
            constexpr int PER = 16 / sizeof(T);
            typedef T vec __attribute__((ext_vector_type(PER)));
            vec* __restrict__ q = reinterpret_cast<vec*>(sh.v + lane * RUN);
            mn = a[0];
#pragma unroll
            for (int c = 0; c < 16 / PER; ++c) {
                vec w;
#pragma unroll
                for (int d = 0; d < PER; ++d) { w[d] = a[c * PER + d]; mn = a[c * PER + d] < mn ? a[c * PER + d] : mn; }
                q[c] = w;
            }
        }
        unsigned open = 0;                        // bit j: element j finds nothing inside its run
        uint32_t w16[8];                          // sixteen tile positions, two per word
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int al = -1;
            if (LEFT) {
#pragma unroll
                for (int i = 0; i < j; ++i) al = (strict ? a[i] < a[j] : a[i] <= a[j]) ? i : al;
            } else {
#pragma unroll
                for (int i = 15; i > j; --i) al = (strict ? a[i] < a[j] : a[i] <= a[j]) ? i : al;
            }
            // (an answer in the padding past the end of the array is none: the element goes on as open and ends beyond the edge)
            const bool none = al < 0 || lane * RUN + (unsigned)(al < 0 ? 0 : al) >= n_rel;
            if (none && lane * RUN + (unsigned)j < n_rel) open |= 1u << j;
            const uint32_t code = none ? AW_PEND : lane * RUN + (unsigned)al;
            if (j & 1) w16[j >> 1] |= code << 16; else w16[j >> 1] = code;
        }
        typedef uint32_t vec4 __attribute__((ext_vector_type(4)));
        vec4* __restrict__ qa = reinterpret_cast<vec4*>(sh.ans + lane * RUN);
        vec4 x0, x1;
#pragma unroll
        for (int d = 0; d < 4; ++d) { x0[d] = w16[d]; x1[d] = w16[4 + d]; }
        qa[0] = x0; qa[1] = x1;
        // ---- 2. the open elements into the queue
        const unsigned mine = (unsigned)__builtin_popcount(open);
        const unsigned incl = wave_scan_inclusive<uint32_t>(mine, OpSum());
        total = shfl<uint32_t>(incl, 63);
        unsigned o = incl - mine;
#pragma unroll
        for (int j = 0; j < 16; ++j) if (open & (1u << j)) sh.q[o++] = (uint16_t)(lane * RUN + (unsigned)j);
    }
    xrun_order();
    // ---- 3. the queue.  First the run next to the own one (five of six open elements of an LCP array end there): one run read and
    //      searched, no descent; what is still open is compacted to the front of the queue.  Then the nearest run whose minimum
    //      qualifies (descent over window minima of the 64 run minima) and the search inside it.
    {
        unsigned total2 = 0;
        nslots = 0;
        mr = ansv_memo_load<T>(memo);          // (the table in registers from here to the end of the pass: ansv_tile.hpp)
#pragma unroll 1
        for (unsigned i0 = 0; i0 < ((AW_ABLATE & 128) ? 0u : total); i0 += 64) {
            const unsigned i = i0 + lane;
            const bool valid = i < total;
            const unsigned e = valid ? sh.q[i] : 0u;
            const T x = sh.v[e];
            const unsigned r = e >> 4;
            const bool has_nb = LEFT ? r > 0 : r < 63;
            const unsigned nb = has_nb ? (LEFT ? r - 1 : r + 1) : r;
            T b[16];
            answ_load_run<T>(sh.v + nb * RUN, b);
            const int j = answ_in_run<T, LEFT>(b, x, strict);
            const unsigned pos = nb * RUN + (unsigned)(j < 0 ? 0 : j);
            const bool hit = valid && has_nb && j >= 0 && pos < n_rel;
            if (hit) sh.ans[e] = (uint16_t)pos;
            const bool rest = valid && !hit;
            const uint64_t m = __ballot(rest);
            if (rest) sh.q[total2 + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (uint16_t)e;
            total2 += (unsigned)__builtin_popcountll(m);
        }
        xrun_order();
        T W[6];
        if (!(AW_ABLATE & 512)) { if (LEFT) ansv_tables_left<T>(mn, W); else ansv_tables_right<T>(mn, W); }
#pragma unroll 1
        for (unsigned i0 = 0; i0 < ((AW_ABLATE & (64 | 128)) ? 0u : total2); i0 += 64) {
            const unsigned i = i0 + lane;
            const bool valid = i < total2;
            const unsigned e = valid ? sh.q[i] : 0u;
            const T x = sh.v[e];
            const unsigned tr = (AW_ABLATE & 512) ? (LEFT ? ((e >> 4) >= 2 ? (e >> 4) - 2 : 64u) : ((e >> 4) + 2 < 64 ? (e >> 4) + 2 : 64u))      // (512: no descent, a made-up run)
                                                  : ansv_descend<T, LEFT>(W, e >> 4, x, strict);
            const bool found = valid && tr < 64;
            unsigned pos = AW_PEND;
            if (__ballot(found)) {
                T b[16];
                answ_load_run<T>(sh.v + (found ? tr : 0u) * RUN, b);
                const int j = answ_in_run<T, LEFT>(b, x, strict);
                if (found) { pos = tr * RUN + (unsigned)(j < 0 ? 0 : j); if (pos >= n_rel) pos = AW_PEND; }
            }
            const bool pend = valid && pos == AW_PEND;
            if (!FUR) {
                if (!(AW_ABLATE & 1024)) ansv_resolve_pending<T, LEFT>(P, n, tile_base, tile_end, pend, x, type, 0u, memo, mr, nonsv, out, tile_base + e, SKIP);
                if (pend) pos = AW_DONE;
            } else {
                // furthest_eq: both answers beyond the edge are looked up here, once per distinct value of a queue step, and kept per
                // element: step 4 then reads them lane by lane (a walk per group of 64 elements there cost a third of the pass)
                uint64_t m = __ballot(pend);
                if (m) {
                    const unsigned slot = nslots + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    nslots += (unsigned)__builtin_popcountll(m);
                    const bool fits = pend && slot < AwSlots<T>::N;
                    while (m) {
                        const int src = __builtin_ctzll(m);
                        const T vq = shfl<T>(x, src);
                        bool goes_on;
                        const uint64_t r0 = ansv_global_fur<T, LEFT>(P, n, tile_base, tile_end, vq, memo, mr, SKIP, &goes_on);
                        const bool mine = pend && x == vq;
                        if (mine && fits) { sh.slot0[slot] = r0 == NSV_NONE ? nonsv : r0; sh.slot1[slot] = goes_on ? r0 : ANSV_NOCONT; }
                        m &= ~__ballot(mine);
                    }
                    if (fits) pos = AW_SLOT0 + slot;
                }
            }
            if (valid) sh.ans[e] = (uint16_t)pos;
        }
    }
    xrun_order();
    if (!FUR && (AW_ABLATE & 32)) {
    } else if (!FUR) {
        // ---- 5. the answers out, lane = element
        uint64_t* __restrict__ o = out + tile_base + lane;
#pragma unroll 4
        for (unsigned k = 0; k < RUN; ++k) {
            const unsigned e = k * 64 + lane;
            const unsigned a = sh.ans[e];
            if (e < n_rel && a != AW_DONE) o[k * 64] = tile_base + a;
        }
    } else {
        // ---- 4. furthest_eq: links to the nearest <= element where it is EQUAL, pointer doubling to the far end of every chain
#pragma unroll
        for (unsigned h = 0; h < ((AW_ABLATE & 2) ? 0 : 2); ++h) {            // eight elements at a time: their loads travel together
            unsigned ne[8]; T xe[8], xn[8];
#pragma unroll
            for (unsigned k = 0; k < 8; ++k) { ne[k] = sh.ans[(h * 8 + k) * 64 + lane]; xe[k] = sh.v[(h * 8 + k) * 64 + lane]; }
#pragma unroll
            for (unsigned k = 0; k < 8; ++k) xn[k] = sh.v[ne[k] < AW_SLOT0 ? ne[k] : 0u];
#pragma unroll
            for (unsigned k = 0; k < 8; ++k) sh.q[(h * 8 + k) * 64 + lane] = (uint16_t)((ne[k] < AW_SLOT0 && xn[k] == xe[k]) ? ne[k] : (h * 8 + k) * 64 + lane);
        }
        xrun_order();
        for (; !(AW_ABLATE & 1);) {
            bool changed = false;
#pragma unroll
            for (unsigned h = 0; h < 2; ++h) {            // eight elements at a time: their loads travel together
                unsigned f[8], ff[8];
#pragma unroll
                for (unsigned k = 0; k < 8; ++k) f[k] = sh.q[(h * 8 + k) * 64 + lane];
#pragma unroll
                for (unsigned k = 0; k < 8; ++k) ff[k] = sh.q[f[k]];
#pragma unroll
                for (unsigned k = 0; k < 8; ++k) if (ff[k] != f[k]) { sh.q[(h * 8 + k) * 64 + lane] = (uint16_t)ff[k]; changed = true; }
            }
            xrun_order();
            if (!__ballot(changed)) break;
        }
        // the answer of e: the far end of the chain of its nearest <= element; a chain whose far end inside the tile has ITS nearest <=
        // element beyond the edge may go on there; an element without a <= element in the tile has its whole answer there (both looked
        // up by the queue loop: slot0 / slot1)
        const bool overflow = nslots > AwSlots<T>::N;
#pragma unroll 1
        for (unsigned h = 0; h < ((AW_ABLATE & 4) ? 0 : 16 / AW_FINAL_W); ++h) {            // AW_FINAL_W elements at a time
            unsigned ne[AW_FINAL_W], r[AW_FINAL_W], ar[AW_FINAL_W];
#pragma unroll
            for (unsigned k = 0; k < AW_FINAL_W; ++k) ne[k] = sh.ans[(h * AW_FINAL_W + k) * 64 + lane];
#pragma unroll
            for (unsigned k = 0; k < AW_FINAL_W; ++k) r[k] = ne[k] < AW_SLOT0 ? sh.q[ne[k]] : 0u;
#pragma unroll
            for (unsigned k = 0; k < AW_FINAL_W; ++k) ar[k] = sh.ans[r[k]];
#pragma unroll
            for (unsigned k = 0; k < AW_FINAL_W; ++k) {
                const unsigned e = (h * AW_FINAL_W + k) * 64 + lane;
                const uint64_t g = tile_base + e;
                const bool in_range = e < n_rel;
                const bool beyond = ne[k] >= AW_SLOT0;                       // no <= element in the tile
                const unsigned code = beyond ? ne[k] : ar[k];               // (the element itself / the far end of its chain)
                const bool looked_up = code >= AW_SLOT0 && code < AW_SLOT0 + AwSlots<T>::N;
                uint64_t res = tile_base + r[k];
                if (looked_up) {
                    const uint64_t far = beyond ? sh.slot0[code - AW_SLOT0] : sh.slot1[code - AW_SLOT0];
                    if (beyond || far != ANSV_NOCONT) res = far;
                }
                if (in_range && (!beyond || looked_up)) out[g] = res;
            }
        }
        if (overflow) {
            // more elements without a <= element in the tile than slots (one falling run): those without a slot ask one by one, and so do
            // the chains that end at one of them.  (Its own loop, not unrolled: the walks it holds were inlined eight times into the
            // loop above -- 30 of the pass's 44 KB of code.)
#pragma unroll 1
            for (unsigned k = 0; k < RUN; ++k) {
                const unsigned e = k * 64 + lane;
                const uint64_t g = tile_base + e;
                const bool in_range = e < n_rel;
                const unsigned ne = sh.ans[e];
                const bool beyond = ne >= AW_SLOT0;
                const unsigned r = beyond ? 0u : sh.q[ne];
                const unsigned ar = sh.ans[r];
                const bool pend = in_range && ne == AW_PEND;
                const bool cont = in_range && !beyond && ar == AW_PEND;
                const T x = sh.v[e], u = sh.v[r];
                ansv_resolve_pending<T, LEFT>(P, n, tile_base, tile_end, pend, x, 2, 0u, memo, mr, nonsv, out, g, SKIP);
                ansv_resolve_pending<T, LEFT>(P, n, tile_base, tile_end, cont, u, 2, 1u, memo, mr, nonsv, out, g, SKIP);
            }
        }
    }
    // ---- 6. the answers beyond the edge, for the next tile on this side.  An entry stays true unless this tile holds a qualifying element:
    //      that element, the nearest one to the next tile, is then the new nearest answer.  furthest_eq: from that element j the chain
    //      runs to its far end r inside the tile (sh.q, step 4); when r's own nearest <= element lies beyond this tile's edge the chain may
    //      go on there, which the table may know (value u = in[j], kind 1): the entries as the previous tile left them.
    if (!((FUR && (AW_ABLATE & 8)) || (AW_ABLATE & 256))) {
        // (the entries in registers, one per lane, since the queue loops: mr)
        const T mval = mr.val;
        const unsigned mkind = mr.kind;
        const bool mlive = mr.live;
        {
            // an entry per lane, all at once (round 5 took an entry at a time: a sixth of a furthest_eq pass, 7 % of a nearest one); every
            // entry reads the table as the previous tile left it
            if (__ballot(mlive)) {
                const uint64_t myres = mr.res;
                // the run nearest to the next tile whose minimum qualifies, entry by entry: a compare and a ballot over the run minima the
                // lanes hold (no table of window minima, no descent: twelve dependent lane moves)
                unsigned rr = 64u;
                uint64_t todo = __ballot(mlive);
                while (todo) {
                    const unsigned i = (unsigned)__builtin_ctzll(todo);
                    todo &= todo - 1;
                    const T x = answ_readlane<T>(mval, i);
                    const uint64_t bal = __ballot(strict ? mn < x : mn <= x);
                    const unsigned ri = bal ? (LEFT ? 63u - (unsigned)__builtin_clzll(bal) : (unsigned)__builtin_ctzll(bal)) : 64u;
                    if (lane == i) rr = ri;
                }
                T b[16];
                answ_load_run<T>(sh.v + (mlive && rr < 64 ? rr : 0u) * RUN, b);
                const int jj = answ_in_run<T, LEFT>(b, mval, strict);
                const bool any = mlive && rr < 64 && jj >= 0;
                const unsigned j = any ? rr * RUN + (unsigned)jj : 0u;
                const uint64_t gp = tile_base + j;
                if (!FUR) {
                    if (any) { if (j >= n_rel) memo.ready[lane] = 0; else memo.res[lane] = gp; }
                } else {
                    const T u = sh.v[j];
                    const unsigned r = sh.q[j];                 // the far end inside the tile of the chain of value u from j (step 4)
                    const bool at_edge = sh.ans[r] >= AW_SLOT0; // ... whose own nearest <= element lies beyond this tile: the chain may go on there
                    // what the table knew about the value u (u < x; u == x: the entry itself)
                    bool known = false, far_on = false; uint64_t far = 0;
                    if (__ballot(any && at_edge && u < mval)) {
                        uint64_t live = __ballot(mlive);
                        while (live) {
                            const unsigned i = (unsigned)__builtin_ctzll(live);
                            live &= live - 1;
                            const T vi = answ_readlane<T>(mval, i);
                            const uint64_t ri = answ_readlane<uint64_t>(myres, i);
                            const unsigned ki = (unsigned)__builtin_amdgcn_readlane((int)mkind, (int)i);
                            if (!known && vi == u) { known = true; far = ri; far_on = ki != 0; }
                        }
                    }
                    if (any) {
                        if (j >= n_rel) memo.ready[lane] = 0;                                   // (padding past the end of the array)
                        else {
                            // the chain of value u from j ends at r inside the tile, or goes on beyond it where the table says so
                            uint64_t res = tile_base + r;
                            bool keep = true;
                            if (at_edge) {
                                if (u == mval) { if (mkind != 0) res = myres; }
                                else if (known) { if (far_on) res = far; }
                                else keep = false;
                            }
                            if (keep) { memo.res[lane] = res; memo.kind[lane] = u == mval ? 1u : 0u; } else memo.ready[lane] = 0;
                        }
                    }
                }
            }
        }
    }
    xrun_order();
}

template <typename T, int LT, int RT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, sizeof(T) == 4 ? 4 : 3) void ansv_wave_kernel(Pyramid<T> P, uint64_t n, int left_type, int right_type, uint64_t nonsv,
                                                               uint64_t* __restrict__ left, uint64_t* __restrict__ right, uint64_t ntiles) {
    // left / right may be null: that side is not computed (the distributed search asks for one side at a time)
    __shared__ AnsvWaveShared<T> shw[WAVES];
    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x / WAVE;
    AnsvWaveShared<T>& sh = shw[wave];
    const uint64_t gw = (uint64_t)blockIdx.x * WAVES + wave, nw = (uint64_t)gridDim.x * WAVES;
    const uint64_t per = (ntiles + nw - 1) / nw;
    const uint64_t t_lo = gw * per;
    const uint64_t t_hi = t_lo + per < ntiles ? t_lo + per : ntiles;
    if (lane < 2) sh.memo[lane].cnt = 0;
    if (lane < 2 * ANSV_MEMO) sh.memo[lane / ANSV_MEMO].ready[lane % ANSV_MEMO] = 0;
    xrun_order();
    for (uint64_t k = 0; t_lo + k < t_hi; ++k) {
        if (left) ansv_wave_pass<T, true, LT>(sh, P, n, t_lo + k, nonsv, left);
        if (right) ansv_wave_pass<T, false, RT>(sh, P, n, t_hi - 1 - k, nonsv, right);
    }
}

template <typename T>
void launch_ansv_wave(psacx_ctx* c, const Pyramid<T>& P, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* d_l, uint64_t* d_r) {
    constexpr int WAVES = 4;
    constexpr uint64_t TILE = AnsvWaveShared<T>::TILE;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
#define PSACX_ANSW(LT, RT)                                                                                                       \
    do {                                                                                                                         \
        static int occ = 0;          /* (asked once per form: the kernel and its LDS are fixed) */                             \
        if (occ < 1 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ansv_wave_kernel<T, LT, RT, WAVES>, 64 * WAVES, 0) != hipSuccess || occ < 1)) { (void)hipGetLastError(); occ = 1; } \
        const unsigned grid = (unsigned)std::min<uint64_t>((ntiles + WAVES - 1) / WAVES, (uint64_t)c->n_cu * occ);               \
        hipLaunchKernelGGL((ansv_wave_kernel<T, LT, RT, WAVES>), dim3(grid), dim3(64 * WAVES), 0, c->stream, P, n, lt, rt, nonsv, d_l, d_r, ntiles); \
    } while (0)
    switch (lt * 3 + rt) {
    case 0: PSACX_ANSW(0, 0); break; case 1: PSACX_ANSW(0, 1); break; case 2: PSACX_ANSW(0, 2); break;
    case 3: PSACX_ANSW(1, 0); break; case 4: PSACX_ANSW(1, 1); break; case 5: PSACX_ANSW(1, 2); break;
    case 6: PSACX_ANSW(2, 0); break; case 7: PSACX_ANSW(2, 1); break; default: PSACX_ANSW(2, 2); break;
    }
#undef PSACX_ANSW
}

} // namespace psacx
