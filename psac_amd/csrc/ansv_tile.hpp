// ansv_tile.hpp -- pieces the ANSV kernel (ansv_wave.hpp) is built from: window-minima tables over 64 lane-held values and the binary
// descent over them, and the answers of searches that leave a tile (one wave-cooperative walk of the global 64-ary min-pyramid per
// distinct value, shared through a small LDS table that is carried from tile to tile).
// Semantics: /root/reference/include/ansv.hpp:48-65 (ansv_sequential), tie rules ansv_common.hpp:20-22 (nearest_sm / nearest_eq /
// furthest_eq), result contract ansv.hpp:2042-2051.  (Rounds 2-3 had a kernel of their own here: one element per lane, binary
// descents inside every 64-block; round 4 a run per lane in workgroup tiles of 4096 elements with seven barriers a tile; round 5 the
// wave-owned tiles of ansv_wave.hpp.)
#pragma once
#include "nsv.hpp"

namespace psacx {

#ifndef AW_ABLATE
#define AW_ABLATE 0          // (tools/experiments/ansv_ablate.sh: parts left out to time them; bit 16: no pyramid walks, the answers beyond a tile edge are wrong then)
#endif
#ifndef AW_MEMO_ENTRIES
#define AW_MEMO_ENTRIES 16
#endif
constexpr unsigned ANSV_MEMO = AW_MEMO_ENTRIES;
constexpr uint64_t ANSV_NOCONT = ~0ull - 1;      // a run of equal values does not continue beyond the tile edge


template <typename T> struct AnsvMemo {
    T val[ANSV_MEMO];
    unsigned long long res[ANSV_MEMO];      // the answer
    unsigned kind[ANSV_MEMO];
    unsigned ready[ANSV_MEMO];
    unsigned cnt;
};

// The lane moves are issued directly (the generic __shfl helpers recompute the lane id, the source lane and its byte address for every
// call): ds_bpermute with the byte address of the source lane (only address bits 7:2 count, so lane * 4 +- 4 d needs
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

template <typename T> __device__ __forceinline__ T answ_readlane(T v, unsigned src);      // src: wave-uniform
template <> __device__ __forceinline__ uint32_t answ_readlane<uint32_t>(uint32_t v, unsigned src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src); }
template <> __device__ __forceinline__ uint64_t answ_readlane<uint64_t>(uint64_t v, unsigned src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)src);
    return ((uint64_t)hi << 32) | lo;
}

// The table of one side in registers for the length of a pass: lane i holds entry i.  Every question a pass asks of the table is then a
// compare and a ballot; through LDS (count, flags, values, kinds, then the answer: five dependent reads per distinct value, a handful of
// values per tile and side) the questions were a fifth of the kernel (left out: nearest pair 2.64 -> 2.10 ms per call, of which the walks
// are 0.07 and bringing the table up to date 0.2).  New entries go to both copies; the pass's last step writes the table back.
template <typename T> struct MemoRegs { T val; unsigned long long res; unsigned kind; bool live; unsigned cnt; };
template <typename T>
__device__ __forceinline__ MemoRegs<T> ansv_memo_load(const AnsvMemo<T>& m) {
    MemoRegs<T> r;
    const unsigned lane = lane_id();
    unsigned c = m.cnt;
    if (c > ANSV_MEMO) c = ANSV_MEMO;
    const bool in = lane < c;
    r.cnt = c;
    r.val = in ? m.val[lane] : (T)0; r.res = in ? m.res[lane] : 0ull; r.kind = in ? m.kind[lane] : 0u; r.live = in && m.ready[lane] != 0;
    return r;
}
template <typename T>
__device__ __forceinline__ bool ansv_memo_find(const MemoRegs<T>& r, T v, unsigned kind, uint64_t* res) {
    const uint64_t b = __ballot(r.live && r.val == v && r.kind == kind);
    if (!b) return false;
    *res = answ_readlane<uint64_t>((uint64_t)r.res, (unsigned)__builtin_ctzll(b));
    return true;
}
// (whole wave) a new entry: at the end of the table, else in the place of a dropped one, else in the place of the LARGEST value held if v is
// smaller -- the values asked for beyond a tile edge are prefix minima, and it is the small ones that are asked for again tile after tile.
// (Round 5 dropped what came after the table had filled: on an LCP array with more than 16 distinct values at tile edges the later ones
// walked the pyramid every time: 0.1 of 2.7 ms.)
template <typename T>
__device__ __forceinline__ void ansv_memo_add(AnsvMemo<T>& m, MemoRegs<T>& r, T v, unsigned kind, uint64_t res) {
    const unsigned lane = lane_id();
    unsigned idx = r.cnt;
    if (r.cnt >= ANSV_MEMO) {
        const bool in = lane < ANSV_MEMO;
        const uint64_t d = __ballot(in && !r.live);
        if (d) idx = (unsigned)__builtin_ctzll(d);
        else {
            const T key = in ? r.val : (T)0;
            const T mx = shfl<T>(wave_scan_inclusive<T>(key, OpMax()), 63);
            if (!(v < mx)) return;
            idx = (unsigned)__builtin_ctzll(__ballot(in && key == mx));
        }
    } else {
        r.cnt++;
        if (lane == 0) m.cnt = r.cnt;
    }
    if (lane == idx) {
        r.val = v; r.res = res; r.kind = kind; r.live = true;
        m.val[idx] = v; m.kind[idx] = kind; m.res[idx] = res; m.ready[idx] = 1u;
    }
}

// furthest_eq keeps ONE entry per value v: res = the answer beyond the edge for an element of value v that has no <= element in its tile (the
// far end of the chain that starts at the nearest <= element there; NSV_NONE: nothing there), kind = 1 when that nearest element has
// the value v itself -- a run of v's inside the tile then goes on to the same far end, otherwise it ends inside the tile.  (Two entries per
// value, one per question, halved the values the table holds and walked the pyramid twice per value: 0.5 of 4.3 ms of the psac -t pair.)
template <typename T>
__device__ __forceinline__ bool ansv_memo_find_value(const MemoRegs<T>& r, T v, uint64_t* res, unsigned* kind) {
    const uint64_t b = __ballot(r.live && r.val == v);
    if (!b) return false;
    const unsigned i = (unsigned)__builtin_ctzll(b);
    *res = answ_readlane<uint64_t>((uint64_t)r.res, i);
    *kind = (unsigned)__builtin_amdgcn_readlane((int)r.kind, (int)i);
    return true;
}
template <typename T, bool LEFT>
__device__ __forceinline__ uint64_t ansv_global_fur(const Pyramid<T>& P, uint64_t n, uint64_t tile_base, uint64_t tile_end, T v,
                                                    AnsvMemo<T>& memo, MemoRegs<T>& mr, int skip, bool* run_goes_on) {
    uint64_t r; unsigned k;
    if (ansv_memo_find_value<T>(mr, v, &r, &k)) { *run_goes_on = k != 0; return r; }
    const bool edge = LEFT ? tile_base == 0 : tile_end >= n;            // nothing beyond the edge
    const uint64_t start = LEFT ? tile_base : tile_end - 1;             // searches look strictly beyond `start`
    uint64_t j = NSV_NONE;
    r = NSV_NONE; k = 0;
    if (!edge && !(AW_ABLATE & 16)) {
        j = nsv_search_wave<T, LEFT>(P, start, v, false, skip);
        if (j != NSV_NONE) { r = nsv_typed_wave<T, LEFT>(P, n, start, v, 2); k = P.lvl[0][j] == v ? 1u : 0u; }
    }
    ansv_memo_add<T>(memo, mr, v, k, r);
    *run_goes_on = k != 0;
    return r;
}

// Answer of a search that leaves the tile (whole wave, wave-uniform arguments).  kind 0: the typed nearest
// smaller value beyond the tile edge for value v; kind 1 (furthest_eq): the far end of the run of values equal
// to v if the run continues beyond the edge, ANSV_NOCONT otherwise.
// skip: levels of the pyramid on which the walk from the tile edge can find nothing (1 for a tile whose edges are multiples of 64)
template <typename T, bool LEFT>
__device__ __forceinline__ uint64_t ansv_global(const Pyramid<T>& P, uint64_t n, uint64_t tile_base, uint64_t tile_end,
                                                T v, int type, unsigned kind, AnsvMemo<T>& memo, MemoRegs<T>& mr, int skip) {
    uint64_t r;
    if (type == 2) {
        bool goes_on;
        r = ansv_global_fur<T, LEFT>(P, n, tile_base, tile_end, v, memo, mr, skip, &goes_on);
        return kind == 0 ? r : goes_on ? r : ANSV_NOCONT;
    }
    if (ansv_memo_find<T>(mr, v, 0u, &r)) return r;
    const bool edge = LEFT ? tile_base == 0 : tile_end >= n;            // nothing beyond the edge
    const uint64_t start = LEFT ? tile_base : tile_end - 1;             // searches look strictly beyond `start`
    r = (edge || (AW_ABLATE & 16)) ? NSV_NONE : nsv_search_wave<T, LEFT>(P, start, v, type == 0, skip);
    ansv_memo_add<T>(memo, mr, v, 0u, r);
    return r;
}

// The lanes flagged in `pend` ask for the answer beyond the tile edge for their value myq: one shared walk per
// distinct value (whole wave).  kind 0: out = answer (nonsv if none); kind 1: out = far end of the run if it continues.
template <typename T, bool LEFT>
__device__ __forceinline__ void ansv_resolve_pending(const Pyramid<T>& P, uint64_t n, uint64_t tile_base, uint64_t tile_end,
                                                     bool pend, T myq, int type, unsigned kind, AnsvMemo<T>& memo, MemoRegs<T>& mr, uint64_t nonsv,
                                                     uint64_t* __restrict__ out, uint64_t g, int skip) {
    uint64_t m = __ballot(pend);
    while (m) {
        const unsigned src = (unsigned)__builtin_ctzll(m);
        const T vq = answ_readlane<T>(myq, src);
        const uint64_t r = ansv_global<T, LEFT>(P, n, tile_base, tile_end, vq, type, kind, memo, mr, skip);
        const bool mine = pend && myq == vq;
        if (mine) {
            if (kind == 0) out[g] = r == NSV_NONE ? nonsv : r;
            else if (r != ANSV_NOCONT) out[g] = r;
        }
        m &= ~__ballot(mine);
    }
}

template <typename T>
void launch_ansv_wave(psacx_ctx* c, const Pyramid<T>& P, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* d_l, uint64_t* d_r);

// left_type / right_type: 0 nearest_sm, 1 nearest_eq, 2 furthest_eq; P: the 64-ary min-pyramid over the input (level 0 = the input);
// d_l / d_r may be null: that side is not computed
template <typename T>
inline void launch_ansv_tiles(psacx_ctx* c, const Pyramid<T>& P, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* d_l, uint64_t* d_r) {
    launch_ansv_wave<T>(c, P, n, lt, rt, nonsv, d_l, d_r);
}

} // namespace psacx
