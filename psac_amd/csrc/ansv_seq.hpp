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
// nearest_sm compares with <, nearest_eq with <=.  furthest_eq sides stay with ansv_tile.hpp.
#pragma once
#include "ansv_tile.hpp"

namespace psacx {

constexpr unsigned ANSQ_OPEN = 0x80u;

template <typename T> struct AnsvSeqShared {
    static constexpr int TB = 64, TILE = TB * 64, RUN = 16, NRUN = TILE / RUN;
    __attribute__((aligned(16))) T v[TILE];                   // the tile
    __attribute__((aligned(16))) T rm[NRUN];                  // run minima (four per block: one 16-byte read for 32-bit values)
    T bm[64];                                                 // block minima
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

template <typename T>
__global__ __launch_bounds__(512, sizeof(T) == 4 ? 8 : 4) void ansv_seq_kernel(Pyramid<T> P, uint64_t n, int lt, int rt, uint64_t nonsv,
                                                        uint64_t* __restrict__ left, uint64_t* __restrict__ right, uint64_t ntiles) {
    typedef AnsvSeqShared<T> SH;
    constexpr int TB = SH::TB, RUN = SH::RUN, NW = 8, BPW = TB / NW;
    constexpr unsigned TILE = SH::TILE;
    __shared__ SH sh;
    const T* __restrict__ in = P.lvl[0];
    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x / WAVE;
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
            // finds a qualifying element in the finished tile now answers with the rightmost such element
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
                const unsigned p = bb * 64 + (63u - (unsigned)__builtin_clzll(in_b));
                if (lane == 0) { m.res[idx] = tile_base - TILE + p; m.first[idx] = tile_base - TILE + p; }
            }
            ansv_carry_right<T>(sh.memo[1], tile_end);
            if (lane == 0) { ansv_memo_compact<T>(sh.memo[0]); ansv_memo_compact<T>(sh.memo[1]); }
        }
        if (threadIdx.x < 2) sh.qcnt[threadIdx.x] = 0;
        __syncthreads();
        // ---- 1. lane = run: the run in registers, all pairs
        {
            const unsigned r = threadIdx.x;                     // 512 lanes: runs 0 .. 255 of the tile twice?  no: 256 runs, two lanes share none
            if (r < (unsigned)SH::NRUN) {
                const uint64_t g0 = tile_base + (uint64_t)r * RUN;
                T a[16];
                if (vec_ok && g0 + RUN <= n) ansq_load_run<T>(in + g0, a);
                else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) a[i] = g0 + i < n ? in[g0 + i] : ~(T)0;
                }
                {   // the run into LDS, its minimum
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
                }
                uint32_t wl[4] = {0, 0, 0, 0}, wr[4] = {0, 0, 0, 0};       // sixteen one-byte codes per side
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    int al = -1, ar = -1;
#pragma unroll
                    for (int i = 0; i < j; ++i) al = (lstrict ? a[i] < a[j] : a[i] <= a[j]) ? i : al;
#pragma unroll
                    for (int i = 15; i > j; --i) ar = (rstrict ? a[i] < a[j] : a[i] <= a[j]) ? i : ar;
                    wl[j >> 2] |= (uint32_t)(al >= 0 ? (unsigned)al : ANSQ_OPEN) << (8 * (j & 3));
                    wr[j >> 2] |= (uint32_t)(ar >= 0 ? (unsigned)ar : ANSQ_OPEN) << (8 * (j & 3));
                }
                typedef uint32_t vec4 __attribute__((ext_vector_type(4)));
                vec4 ql, qr;
#pragma unroll
                for (int d = 0; d < 4; ++d) { ql[d] = wl[d]; qr[d] = wr[d]; }
                *reinterpret_cast<vec4*>(sh.code[0] + r * RUN) = ql;
                *reinterpret_cast<vec4*>(sh.code[1] + r * RUN) = qr;
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
#pragma unroll 2
            for (int k = 0; k < BPW; ++k) {
                const unsigned e = (wave * BPW + k) * 64 + lane;
                const uint64_t g = tile_base + e;
                const bool in_range = g < n;
                const unsigned c8 = sh.code[side][e];
                const unsigned cd = (e & ~15u) + (c8 & 15u);
                // (an answer in the padding past the end of the array is none: the element goes on as open and ends beyond the edge)
                const bool open = in_range && ((c8 & ANSQ_OPEN) || tile_base + cd >= n);
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
                // the other runs of the own block on the searched side, nearest first
                int run = -1;
                {
                    const T* q = sh.rm + b * 4;
                    const T m0 = q[0], m1 = q[1], m2 = q[2], m3 = q[3];
                    const T mm[4] = {m0, m1, m2, m3};
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int c = side == 0 ? s : 3 - s;               // (the last qualifying one in this order is the nearest)
                        const bool beyond = side == 0 ? (unsigned)c < rr : (unsigned)c > rr;
                        if (beyond && (strict ? mm[c] < x : mm[c] <= x)) run = (int)(b * 4) + c;
                    }
                }
                unsigned bb;
                if (side == 0) bb = ansv_descend<T, true>(BL, b, x, strict); else bb = ansv_descend<T, false>(BR, b, x, strict);
                if (run < 0 && bb < 64) {
                    const T* q = sh.rm + bb * 4;
                    const T mm[4] = {q[0], q[1], q[2], q[3]};
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int c = side == 0 ? s : 3 - s;
                        if (strict ? mm[c] < x : mm[c] <= x) run = (int)(bb * 4) + c;
                    }
                }
                bool pend = valid && run < 0;
                if (__ballot(valid && run >= 0)) {
                    T a[16];
                    ansq_load_run<T>(sh.v + (run >= 0 ? run : 0) * RUN, a);
                    const int j = side == 0 ? ansq_in_run<T, true>(a, x, strict) : ansq_in_run<T, false>(a, x, strict);
                    if (valid && run >= 0) {
                        const uint64_t ans = tile_base + (uint64_t)run * RUN + (unsigned)(j < 0 ? 0 : j);
                        if (ans < n) out[g] = ans; else pend = true;         // (padding past the end of the array is never an answer)
                    }
                }
                if (side == 0) ansv_resolve_pending<T, true>(P, n, tile_base, tile_end, pend, x, lt, 0u, sh.memo[0], nonsv, out, g);
                else ansv_resolve_pending<T, false>(P, n, tile_base, tile_end, pend, x, rt, 0u, sh.memo[1], nonsv, out, g);
            }
            __syncthreads();                   // the queue is reused by the other side
        }
    }
}

template <typename T>
void launch_ansv_seq(psacx_ctx* c, const Pyramid<T>& P, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* d_l, uint64_t* d_r) {
    constexpr uint64_t TILE = AnsvSeqShared<T>::TILE;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ansv_seq_kernel<T>, 512, 0) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 1; }
    const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, (uint64_t)c->n_cu * occ);
    hipLaunchKernelGGL((ansv_seq_kernel<T>), dim3(grid), dim3(512), 0, c->stream, P, n, lt, rt, nonsv, d_l, d_r, ntiles);
}

} // namespace psacx
