// ansv_seq.hpp -- all nearest smaller values, second tile form: every LANE owns a run of 16 consecutive elements, finds the
// neighbours inside its run in registers, and only the elements a run leaves open go on to the levels above it.
// Semantics: /root/reference/include/ansv.hpp:48-65 (ansv_sequential), tie rules ansv_common.hpp:20-22.
//
// ansv_tile.hpp gives every element a lane and finds its neighbour inside a 64-block with binary descents over window minima:
// 24 dependent lane moves per block and side, paid by all 64 lanes although most elements have their answer a few positions away.
// Here a workgroup of 512 lanes takes a tile of 4096 elements (64 blocks of 4 runs):
//   1. a lane loads its run (64 contiguous bytes), keeps it in registers and compares all pairs -- 120 compare / select pairs per side,
//      no LDS round trip, no divergence -- which gives every element the position of its answer inside the run or marks it open;
//      the run also goes to LDS, with its minimum;
//   2. the codes are turned into coalesced stores, lane = element; the open elements (a third on an LCP array, whose small values are
//      the ones that stay open) are compacted into a queue;
//   3. the queue is worked off 64 entries per wave and step: the other runs of the own block (their four minima are one 16-byte
//      read), else the nearest block of the tile whose minimum qualifies (one binary descent over the 64 block minima, held one per
//      lane) and its nearest run; the run found is read with four 16-byte loads and searched in registers.  What leaves the tile asks
//      the shared table of answers beyond the tile edge (one wave-cooperative walk of the global min-pyramid per distinct value,
//      ansv_tile.hpp: ansv_global).
// nearest_sm compares with <, nearest_eq with <=.  furthest_eq (the nearest <= element, then on through the values equal to IT while
// nothing smaller lies between): inside a run every element carries the far end of its own chain of equal values and whether something
// smaller precedes it, so the answer of element j is the chain of its nearest <= element, picked up in the same pass over the pairs; a
// chain that reaches the start of its run goes on in O(1) steps over the run and block minima (the first run / block whose minimum is
// smaller ends it; the furthest one before that whose minimum EQUALS the value holds its far end: first occurrence of a run's minimum).
#pragma once
#include "ansv_tile.hpp"

namespace psacx {

constexpr unsigned ANSQ_OPEN = 0x80u;       // nothing qualifies inside the run
constexpr unsigned ANSQ_CONT = 0x40u;       // furthest_eq: the chain of equal values reaches the edge of the run and may go on beyond it
constexpr unsigned ANSQ_NOPOS = 0xFFFFu;

template <typename T> struct AnsvSeqShared {
    static constexpr int TB = 64, TILE = TB * 64, RUN = 16, NRUN = TILE / RUN;
    __attribute__((aligned(16))) T v[TILE];                   // the tile
    __attribute__((aligned(16))) T rm[NRUN];                  // run minima (four per block: one 16-byte read for 32-bit values)
    T bm[64];                                                 // block minima
    uint8_t frm[2][NRUN];                                     // first / last occurrence of a run's minimum inside the run
    __attribute__((aligned(16))) uint8_t code[2][TILE];       // per side: position of the answer inside the own run, or ANSQ_OPEN
    uint16_t queue[TILE];                                     // tile positions of the open elements of the side being worked on
    unsigned qcnt[2];
    AnsvMemo<T> memo[2];
};

// rightmost (LEFT) / leftmost index of the 16 values that qualifies (-1: none)
template <typename T, bool LEFT>
__device__ __forceinline__ int ansq_in_run(const T (&a)[16], T x, bool strict) {
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
__device__ __forceinline__ void ansq_load_run(const T* __restrict__ p, T (&a)[16]) {       // p: 16-byte aligned, LDS or global
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

// Nearest run of the tile beyond run rr of block b (towards the left when LEFT) whose minimum qualifies for x: the other runs of the
// own block first, then the nearest block whose minimum qualifies (binary descent over the block minima) and its nearest run.  -1: none.
template <typename T, bool LEFT, typename SH>
__device__ __forceinline__ int ansq_find_run(SH& sh, const T (&BW)[6], unsigned b, unsigned rr, T x, bool strict) {
    int run = -1;
    {
        const T* q = sh.rm + b * 4;
        const T mm[4] = {q[0], q[1], q[2], q[3]};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = LEFT ? s : 3 - s;               // (the last qualifying one in this order is the nearest)
            const bool beyond = LEFT ? (unsigned)c < rr : (unsigned)c > rr;
            if (beyond && (strict ? mm[c] < x : mm[c] <= x)) run = (int)(b * 4) + c;
        }
    }
    const unsigned bb = ansv_descend<T, LEFT>(BW, b, x, strict);
    if (run < 0 && bb < 64) {
        const T* q = sh.rm + bb * 4;
        const T mm[4] = {q[0], q[1], q[2], q[3]};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = LEFT ? s : 3 - s;
            if (strict ? mm[c] < x : mm[c] <= x) run = (int)(bb * 4) + c;
        }
    }
    return run;
}

// furthest_eq: a chain of elements equal to u reaches the edge of run r0 (its start when LEFT) with its far end so far at `far` (tile
// position).  Follows it through the tile: *far = the far end inside the tile; returns true when the chain reaches the tile edge too.
template <typename T, bool LEFT, typename SH>
__device__ __forceinline__ bool ansq_chain_on(SH& sh, const T (&BS)[6], const T (&BO)[6], bool active, unsigned r0, T u, unsigned* far) {
    // BS: window minima of the block minima on the searched side, BO: on the other side (for the way back)
    const unsigned b0 = r0 >> 2, rr0 = r0 & 3u;
    unsigned f = *far;
    bool closed = !active;
    int scan_run = -1;                                 // the run that holds the first smaller element: searched at the end
    {
        // the runs of the own block beyond r0, nearest first
        const T* q = sh.rm + b0 * 4;
        const T mm[4] = {q[0], q[1], q[2], q[3]};
#pragma unroll
        for (int s = 1; s < 4; ++s) {
            const int c = LEFT ? (int)rr0 - s : (int)rr0 + s;
            if (c < 0 || c > 3 || closed) continue;
            const unsigned rc = b0 * 4 + (unsigned)c;
            if (mm[c] == u) f = rc * 16 + sh.frm[LEFT ? 0 : 1][rc];
            else if (mm[c] < u) { scan_run = (int)rc; closed = true; }
        }
    }
    // blocks: the nearest one with a smaller minimum ends the chain; before it, the furthest block whose minimum equals u holds the far end
    const unsigned bs = ansv_descend<T, LEFT>(BS, b0, u, true);
    // the furthest block strictly between bs and b0 with a minimum <= u (then == u): the nearest one seen from bs, or from the tile edge
    // when no block ends the chain (the edge block itself counts then).  (One descent for all lanes: the lane moves inside it must not
    // sit in divergent branches.)
    const unsigned eb = LEFT ? 0u : 63u;
    const unsigned from = bs >= 64 ? eb : bs;
    unsigned be = LEFT ? ansv_descend<T, false>(BO, from, u, false) : ansv_descend<T, true>(BO, from, u, false);
    if (bs >= 64 && sh.bm[eb] <= u) be = eb;
    bool edge = false;
    if (!closed) {
        const bool be_ok = be < 64 && (LEFT ? be < b0 : be > b0);
        if (be_ok) {
            const T* q = sh.rm + be * 4;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int c = LEFT ? 3 - s : s;          // (the last one in this order is the furthest)
                if (q[c] == u) f = (be * 4 + (unsigned)c) * 16 + sh.frm[LEFT ? 0 : 1][be * 4 + c];
            }
        }
        if (bs < 64) {
            // inside the block that ends the chain: its nearest run with a smaller minimum, the runs before it whose minimum equals u
            const T* q = sh.rm + bs * 4;
            bool hit = false;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int c = LEFT ? 3 - s : s;          // nearest first
                if (hit) continue;
                if (q[c] < u) { scan_run = (int)(bs * 4) + c; hit = true; }
                else if (q[c] == u) f = (bs * 4 + (unsigned)c) * 16 + sh.frm[LEFT ? 0 : 1][bs * 4 + c];
            }
            closed = true;
        } else edge = true;
    }
    // the run with the first smaller element: elements equal to u before it (seen from the chain) still belong to the chain
    if (__ballot(scan_run >= 0)) {
        T a[16];
        ansq_load_run<T>(sh.v + (scan_run >= 0 ? scan_run : 0) * 16, a);
        bool stop = false;
        int ff = -1;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int i = LEFT ? 15 - s : s;
            stop = stop || a[i] < u;
            ff = (!stop && a[i] == u) ? i : ff;
        }
        if (scan_run >= 0 && ff >= 0) f = (unsigned)scan_run * 16 + (unsigned)ff;
    }
    *far = f;
    return edge;
}

template <typename T, bool LF, bool RF>
__global__ __launch_bounds__(512, (sizeof(T) == 4 && !LF && !RF) ? 8 : (sizeof(T) == 4 ? 6 : 4)) void ansv_seq_kernel(Pyramid<T> P, uint64_t n, int left_type, int right_type, uint64_t nonsv,
                                                        uint64_t* __restrict__ left, uint64_t* __restrict__ right, uint64_t ntiles) {
    typedef AnsvSeqShared<T> SH;
    constexpr int TB = SH::TB, RUN = SH::RUN, NW = 8, BPW = TB / NW;
    constexpr unsigned TILE = SH::TILE;
    __shared__ SH sh;
    const T* __restrict__ in = P.lvl[0];
    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x / WAVE;
    const int lt = LF ? 2 : left_type, rt = RF ? 2 : right_type;
    const bool lstrict = lt == 0, rstrict = rt == 0;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(in) & 15u) == 0;
    const uint64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    const uint64_t t_lo = (uint64_t)blockIdx.x * per;
    const uint64_t t_hi = t_lo + per < ntiles ? t_lo + per : ntiles;
    if (threadIdx.x < 2) sh.memo[threadIdx.x].cnt = 0;
    if (threadIdx.x < 2 * ANSV_MEMO) sh.memo[threadIdx.x / ANSV_MEMO].ready[threadIdx.x % ANSV_MEMO] = 0;
    T bmv_prev = ~(T)0;
    for (uint64_t t = t_lo; t < t_hi; ++t) {
        const uint64_t tile_base = t * TILE;
        const uint64_t tile_end = tile_base + TILE < n ? tile_base + TILE : n;
        __syncthreads();                       // every wave is done with the previous tile
        if (t > t_lo && wave == 0) {
            // answers beyond the tile edge carried to the next tile (ansv_tile.hpp: ansv_carry_*).  Left side: an entry whose value
            // finds a qualifying element in the finished tile now answers with the rightmost such element (nearest types); a
            // furthest_eq entry is dropped, unless it asks whether a chain goes on and the element found is smaller: it does not.
            AnsvMemo<T>& m = sh.memo[0];
            const unsigned c = m.cnt < ANSV_MEMO ? m.cnt : ANSV_MEMO;
            for (unsigned idx = 0; idx < c; ++idx) {
                if (!m.ready[idx]) continue;
                const T x = m.val[idx];
                const uint64_t bal = __ballot(lstrict ? bmv_prev < x : bmv_prev <= x);
                if (!bal) continue;
                const unsigned bb = 63u - (unsigned)__builtin_clzll(bal);
                const T y = sh.v[bb * 64 + lane];
                const uint64_t in_b = __ballot(lstrict ? y < x : y <= x);
                const unsigned pl = 63u - (unsigned)__builtin_clzll(in_b);
                const unsigned p = bb * 64 + pl;
                if (LF) {
                    const T u = shfl<T>(y, (int)pl);
                    if (lane == 0) { if (m.kind[idx] == 1 && u < x) { m.res[idx] = ANSV_NOCONT; m.first[idx] = tile_base - TILE + p; } else m.ready[idx] = 0; }
                } else if (lane == 0) { m.res[idx] = tile_base - TILE + p; m.first[idx] = tile_base - TILE + p; }
            }
            ansv_carry_right<T>(sh.memo[1], tile_end);
            if (lane == 0) { ansv_memo_compact<T>(sh.memo[0]); ansv_memo_compact<T>(sh.memo[1]); }
        }
        if (threadIdx.x < 2) sh.qcnt[threadIdx.x] = 0;
        __syncthreads();
        // ---- 1. lane = run: the run in registers, all pairs.  The lower half of the workgroup does the left side of the 256 runs (and puts
        //      the runs into LDS), the upper half the right side.
        {
            const bool do_left = threadIdx.x < (unsigned)SH::NRUN;
            const unsigned r = do_left ? threadIdx.x : threadIdx.x - (unsigned)SH::NRUN;
            {
                const uint64_t g0 = tile_base + (uint64_t)r * RUN;
                T a[16];
                if (vec_ok && g0 + RUN <= n) ansq_load_run<T>(in + g0, a);
                else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) a[i] = g0 + i < n ? in[g0 + i] : ~(T)0;
                }
                if (do_left) {   // the run into LDS, its minimum and where it first / last occurs
                    constexpr int PER = 16 / sizeof(T);
                    typedef T vec __attribute__((ext_vector_type(PER)));
                    vec* __restrict__ q = reinterpret_cast<vec*>(sh.v + r * RUN);
                    T mn = a[0];
#pragma unroll
                    for (int c = 0; c < 16 / PER; ++c) {
                        vec w;
#pragma unroll
                        for (int d = 0; d < PER; ++d) { w[d] = a[c * PER + d]; mn = a[c * PER + d] < mn ? a[c * PER + d] : mn; }
                        q[c] = w;
                    }
                    sh.rm[r] = mn;
                    if (LF || RF) {
                        int f0 = 15, f1 = 0;
#pragma unroll
                        for (int i = 15; i >= 0; --i) f0 = a[i] == mn ? i : f0;
#pragma unroll
                        for (int i = 0; i < 16; ++i) f1 = a[i] == mn ? i : f1;
                        sh.frm[0][r] = (uint8_t)f0; sh.frm[1][r] = (uint8_t)f1;
                    }
                }
                uint32_t wl[4] = {0, 0, 0, 0}, wr[4] = {0, 0, 0, 0};       // sixteen one-byte codes per side
                if (!do_left) {
                } else if (!LF) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        int al = -1;
#pragma unroll
                        for (int i = 0; i < j; ++i) al = (lstrict ? a[i] < a[j] : a[i] <= a[j]) ? i : al;
                        wl[j >> 2] |= (uint32_t)(al >= 0 ? (unsigned)al : ANSQ_OPEN) << (8 * (j & 3));
                    }
                } else {
                    // fc[i]: far end of the chain of equal values element i belongs to (low 4 bits), bit 4: something smaller precedes it
                    unsigned fc[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        unsigned sel = 0xFFu; T u = 0; bool smj = false;
#pragma unroll
                        for (int i = 0; i < j; ++i) {
                            const bool ok = a[i] <= a[j];
                            sel = ok ? fc[i] : sel; u = ok ? a[i] : u;
                            smj = smj || a[i] < a[j];
                        }
                        const unsigned cd = sel == 0xFFu ? ANSQ_OPEN : ((sel & 15u) | ((sel & 16u) ? 0u : ANSQ_CONT));
                        wl[j >> 2] |= cd << (8 * (j & 3));
                        fc[j] = (sel != 0xFFu && u == a[j]) ? sel : ((unsigned)j | (smj ? 16u : 0u));
                    }
                }
                if (do_left) {
                } else if (!RF) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        int ar = -1;
#pragma unroll
                        for (int i = 15; i > j; --i) ar = (rstrict ? a[i] < a[j] : a[i] <= a[j]) ? i : ar;
                        wr[j >> 2] |= (uint32_t)(ar >= 0 ? (unsigned)ar : ANSQ_OPEN) << (8 * (j & 3));
                    }
                } else {
                    unsigned fc[16];
#pragma unroll
                    for (int j = 15; j >= 0; --j) {
                        unsigned sel = 0xFFu; T u = 0; bool smj = false;
#pragma unroll
                        for (int i = 15; i > j; --i) {
                            const bool ok = a[i] <= a[j];
                            sel = ok ? fc[i] : sel; u = ok ? a[i] : u;
                            smj = smj || a[i] < a[j];
                        }
                        const unsigned cd = sel == 0xFFu ? ANSQ_OPEN : ((sel & 15u) | ((sel & 16u) ? 0u : ANSQ_CONT));
                        wr[j >> 2] |= cd << (8 * (j & 3));
                        fc[j] = (sel != 0xFFu && u == a[j]) ? sel : ((unsigned)j | (smj ? 16u : 0u));
                    }
                }
                typedef uint32_t vec4 __attribute__((ext_vector_type(4)));
                vec4 qq;
#pragma unroll
                for (int d = 0; d < 4; ++d) qq[d] = do_left ? wl[d] : wr[d];
                *reinterpret_cast<vec4*>(sh.code[do_left ? 0 : 1] + r * RUN) = qq;
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const T* q = sh.rm + threadIdx.x * 4;
            T m = q[0]; m = q[1] < m ? q[1] : m; m = q[2] < m ? q[2] : m; m = q[3] < m ? q[3] : m;
            sh.bm[threadIdx.x] = m;
        }
        __syncthreads();
        // ---- 2. answers inside the own run out (coalesced), the open elements into the queue; 3. the open elements: the other runs of
        //      the block, the nearest block with a qualifying minimum, else beyond the tile.  One side after the other (one queue).
        const T bmv = sh.bm[lane];
        bmv_prev = bmv;
        T BL[6], BR[6];
        ansv_tables_left<T>(bmv, BL);
        ansv_tables_right<T>(bmv, BR);
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            uint64_t* __restrict__ out = side == 0 ? left : right;
            const bool strict = side == 0 ? lstrict : rstrict;
            const bool fur = side == 0 ? LF : RF;
#pragma unroll 2
            for (int k = 0; k < BPW; ++k) {
                const unsigned e = (wave * BPW + k) * 64 + lane;
                const uint64_t g = tile_base + e;
                const bool in_range = g < n;
                const unsigned c8 = sh.code[side][e];
                const unsigned cd = (e & ~15u) + (c8 & 15u);
                // (an answer in the padding past the end of the array is none: the element goes on as open and ends beyond the edge)
                const bool open = in_range && ((c8 & (ANSQ_OPEN | ANSQ_CONT)) || tile_base + cd >= n);
                if (in_range && !open) out[g] = tile_base + cd;
                const uint64_t mo = __ballot(open);
                if (mo) {
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(&sh.qcnt[side], (unsigned)__builtin_popcountll(mo));
                    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                    if (open) sh.queue[base + __builtin_amdgcn_mbcnt_hi((unsigned)(mo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mo, 0u))] = (uint16_t)e;
                }
            }
            __syncthreads();
            const unsigned cnt = sh.qcnt[side];
#pragma unroll 1
            for (unsigned i0 = wave * 64; i0 < cnt; i0 += NW * 64) {
                const unsigned i = i0 + lane;
                const bool valid = i < cnt;
                const unsigned e = valid ? sh.queue[i] : 0u;
                const unsigned b = e >> 6, rr = (e >> 4) & 3u;           // block, run inside the block
                const T x = valid ? sh.v[e] : (T)0;
                const uint64_t g = tile_base + e;
                const unsigned c8 = fur ? sh.code[side][e] : ANSQ_OPEN;
                const bool is_cont = fur && valid && (c8 & ANSQ_CONT) && tile_base + (e & ~15u) + (c8 & 15u) < n;
                // the nearest run beyond the own one that holds a qualifying element, and that element
                int run;
                if (side == 0) run = ansq_find_run<T, true>(sh, BL, b, rr, x, strict); else run = ansq_find_run<T, false>(sh, BR, b, rr, x, strict);
                if (is_cont) run = -1;
                bool pend = valid && !is_cont && run < 0;
                unsigned pos = ANSQ_NOPOS;
                if (__ballot(valid && run >= 0)) {
                    T a[16];
                    ansq_load_run<T>(sh.v + (run >= 0 ? run : 0) * RUN, a);
                    const int j = side == 0 ? ansq_in_run<T, true>(a, x, strict) : ansq_in_run<T, false>(a, x, strict);
                    if (valid && run >= 0) {
                        pos = (unsigned)run * RUN + (unsigned)(j < 0 ? 0 : j);
                        if (tile_base + pos >= n) { pos = ANSQ_NOPOS; pend = true; }       // (padding past the end of the array is never an answer)
                    }
                }
                if (!fur) {
                    if (pos != ANSQ_NOPOS) out[g] = tile_base + pos;
                    if (side == 0) ansv_resolve_pending<T, true>(P, n, tile_base, tile_end, pend, x, lt, 0u, sh.memo[0], nonsv, out, g);
                    else ansv_resolve_pending<T, false>(P, n, tile_base, tile_end, pend, x, rt, 0u, sh.memo[1], nonsv, out, g);
                    continue;
                }
                // furthest_eq: the chain of the element found (or the element's own chain inside its run), followed on while it is open
                T u = x; unsigned far = ANSQ_NOPOS, r0 = e >> 4; bool open = false;
                if (is_cont) { far = (e & ~15u) + (c8 & 15u); u = sh.v[far]; open = true; }
                else if (pos != ANSQ_NOPOS) {
                    u = sh.v[pos];
                    const unsigned cp = sh.code[side][pos];
                    r0 = pos >> 4;
                    if (cp & ANSQ_OPEN) { far = pos; open = true; }
                    else {
                        const unsigned f = (pos & ~15u) + (cp & 15u);
                        if (sh.v[f] == u) { far = f; open = (cp & ANSQ_CONT) != 0; } else far = pos;
                    }
                }
                bool edge;
                if (side == 0) edge = ansq_chain_on<T, true>(sh, BL, BR, open, r0, u, &far); else edge = ansq_chain_on<T, false>(sh, BR, BL, open, r0, u, &far);
                if (valid && far != ANSQ_NOPOS) out[g] = tile_base + far;
                const bool cont = valid && open && edge;
                if (side == 0) {
                    ansv_resolve_pending<T, true>(P, n, tile_base, tile_end, pend, x, 2, 0u, sh.memo[0], nonsv, out, g);
                    ansv_resolve_pending<T, true>(P, n, tile_base, tile_end, cont, u, 2, 1u, sh.memo[0], nonsv, out, g);
                } else {
                    ansv_resolve_pending<T, false>(P, n, tile_base, tile_end, pend, x, 2, 0u, sh.memo[1], nonsv, out, g);
                    ansv_resolve_pending<T, false>(P, n, tile_base, tile_end, cont, u, 2, 1u, sh.memo[1], nonsv, out, g);
                }
            }
            __syncthreads();                   // the queue is reused by the other side
        }
    }
}

template <typename T>
void launch_ansv_seq(psacx_ctx* c, const Pyramid<T>& P, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* d_l, uint64_t* d_r) {
    constexpr uint64_t TILE = AnsvSeqShared<T>::TILE;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
#define PSACX_ANSQ(LF, RF)                                                                                                       \
    do {                                                                                                                         \
        int occ = 0;                                                                                                             \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ansv_seq_kernel<T, LF, RF>, 512, 0) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 1; } \
        const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, (uint64_t)c->n_cu * occ);                                     \
        hipLaunchKernelGGL((ansv_seq_kernel<T, LF, RF>), dim3(grid), dim3(512), 0, c->stream, P, n, lt, rt, nonsv, d_l, d_r, ntiles); \
    } while (0)
    if (lt == 2 && rt == 2) PSACX_ANSQ(true, true);
    else if (lt == 2) PSACX_ANSQ(true, false);
    else if (rt == 2) PSACX_ANSQ(false, true);
    else PSACX_ANSQ(false, false);
#undef PSACX_ANSQ
}

} // namespace psacx
