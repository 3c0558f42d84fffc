// radix.hpp -- rank-pair LSD radix sort for records (key1, key2, value).
//
// Stands in for idxsort_vectors() -> mxx::sort on (B1,B2,idx) tuples
// (/root/reference/include/idxsort.hpp:23-83).  The arrays stay struct-of-arrays
// in HBM; one sort is
//   * one histogram kernel over both key words producing all per-digit
//     histograms at once (LDS-staged 256-bin counters per digit), and
//   * one single-sweep scatter kernel per non-constant 8-bit digit: every tile
//     ranks its records with wave64 ballots, publishes its 256 digit counts
//     and resolves its global offsets by decoupled look-back, reorders the
//     tile through LDS and writes coalesced runs.
// Algorithmic HBM traffic: 2w bytes/record for the histogram and 6w
// bytes/record per scatter pass (w = sizeof(T)), SURVEY.md section 8(d).
#pragma once
#include <type_traits>
#include "dev_common.hpp"

namespace psacx {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int MAX_PASSES = 2 * (64 / RADIX_BITS);   // two 64-bit key words
constexpr int SLAB_TILES = 64;                       // tiles per slab of the three-kernel offset scan (smallest slab)
// Very long inputs use longer slabs: the scan over the slab totals is one workgroup walking them one after the other
// (1.7 ms per pass at 2^20 tiles with 64-tile slabs, 2 % of a 4 GiB construction), the scan inside a slab runs in
// parallel over slabs.
// (a slab is one workgroup walking its tiles one after the other, the slab totals one workgroup walking the slabs: measured with
//  both sizes in one process, 2^29 64-bit records (2^17 tiles): 4.3 ms of histogram + scans per sort with slabs of 1024 tiles,
//  3.2 with 256 or 128; 2^32 records (2^20 tiles): 29.1 with 1024, 28.1 with 512, 31.3 with 128)
inline unsigned slab_tiles_for(uint64_t ntiles) { return ntiles > (1u << 19) ? 512u : ntiles > (1u << 16) ? 256u : (unsigned)SLAB_TILES; }

// pass p < passes_lo reads key2 (low word), the rest read key1
struct PassPlan {
    int n_pass;
    int word[MAX_PASSES];     // 0 = key1 (high), 1 = key2 (low)
    int shift[MAX_PASSES];
};

// bits1 / bits2: significant low bits of key word 1 (major) and key word 2 (minor)
// lo1: the low lo1 bits of word 1 are not sorted on (prefix sort by the leading bits only)
inline PassPlan make_plan(int bits1, int bits2, int lo1 = 0) {
    PassPlan p;
    p.n_pass = 0;
    for (int w = 1; w >= 0; --w) {
        const int lo = w ? 0 : lo1;
        const int per = ((w ? bits2 : bits1) - lo + RADIX_BITS - 1) / RADIX_BITS;
        for (int i = 0; i < per; ++i) { p.word[p.n_pass] = w; p.shift[p.n_pass] = lo + i * RADIX_BITS; p.n_pass++; }
    }
    return p;
}

// --------------------------------------------------------------- histogram
// One LDS counter update per lane, except when the whole wave agrees on the digit (sorted
// bucket ids give long runs of equal high digits; 64 same-address LDS atomics serialise).
__device__ __forceinline__ void wave_hist_add(unsigned* hist, unsigned d, bool valid) {
    const uint64_t vm = __ballot(valid);
    if (vm == 0) return;                                        // wave-uniform
    const int first = __builtin_ctzll(vm);
    const unsigned dv = (unsigned)__builtin_amdgcn_readlane((int)d, first);
    if (__ballot(valid && d != dv) == 0) {                      // every valid lane holds digit dv
        if ((int)lane_id() == first) atomicAdd(&hist[dv], (unsigned)__builtin_popcountll(vm));
    } else if (valid) {
        atomicAdd(&hist[d], 1u);
    }
}

struct HistArgs {
    int n_pass;
    int word[MAX_PASSES];
    int shift[MAX_PASSES];
};

template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void radix_hist_kernel(const T* __restrict__ k1,
                                                           const T* __restrict__ k2, uint64_t n,
                                                           HistArgs a,
                                                           unsigned long long* __restrict__ hist) {
    constexpr int PER = 16 / sizeof(T);
    __shared__ unsigned lh[MAX_PASSES * RADIX];
    for (int i = threadIdx.x; i < a.n_pass * RADIX; i += BLOCK) lh[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK * PER;
    for (uint64_t e0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * PER; e0 < n; e0 += stride) {
        T x1[PER], x2[PER];
        load_run<T, PER>(k1, e0, n, x1, (T)0);
        load_run<T, PER>(k2, e0, n, x2, (T)0);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const bool in = e0 + j < n;
            for (int p = 0; p < a.n_pass; ++p) {
                const T w = a.word[p] ? x2[j] : x1[j];
                const unsigned d = (unsigned)(w >> a.shift[p]) & (RADIX - 1);
                wave_hist_add(&lh[p * RADIX], d, in);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.n_pass * RADIX; i += BLOCK) {
        const unsigned c = lh[i];
        if (c) atomicAdd(&hist[i], (unsigned long long)c);
    }
}

// Digit starts of every pass of a small sort, on the device: block p turns the histogram of pass p into its exclusive
// scan (the look-back form used to fetch the histograms, scan them on the host and send the starts back: one host
// synchronisation, two copies and a memset per pass).
// constant[p] (host-pinned memory the device can write) = 1 when one digit of pass p holds all n keys: the host skips it.
template <int DUMMY>
__global__ __launch_bounds__(RADIX) void radix_hist_scan_kernel(const unsigned long long* __restrict__ hist,
                                                                 unsigned long long* __restrict__ base,
                                                                 unsigned long long n, unsigned* __restrict__ constant) {
    __shared__ unsigned long long tmp[RADIX / WAVE + 1];
    const unsigned long long h = hist[(size_t)blockIdx.x * RADIX + threadIdx.x];
    unsigned long long tot;
    const unsigned long long ex = block_scan_exclusive<RADIX, unsigned long long>(h, OpSum(), 0ull, tmp, &tot);
    base[(size_t)blockIdx.x * RADIX + threadIdx.x] = ex;
    const bool all = __syncthreads_or(h == n);
    if (threadIdx.x == 0) constant[blockIdx.x] = all ? 1u : 0u;
}

// ------------------------------------------------------------ scatter pass
__device__ __forceinline__ uint64_t match_any8(unsigned d, bool valid) {
    // lanes whose digit equals this lane's: for every bit, the ballot of the bit XOR this lane's bit spread over a word (0 / all ones) marks the
    // lanes that DIFFER there; OR over the bits, inverted.  Four vector instructions per bit: v_bfe_i32 (the bit as 0 / -1), the compare
    // for the ballot, and one v_bitop3 (acc | (ballot ^ bit)) per half of the mask -- the form `m &= bit ? bal : ~bal` compiled to ten,
    // and the scatter passes are bound by vector issue: 100 instructions per record at 4.4 SIMD cycles each were 12.1 of their 15.2 ms
    // (tools/ubench_valu.hip; profiles/r4d_sq_counters_one_gpu_kernels.txt).
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
        int x = __builtin_amdgcn_sbfe((int)d, (unsigned)b, 1u);            // 0 or -1
        asm("" : "+v"(x));                   // (keeps the compare on x: the optimizer otherwise derives the ballot from d with a shift of its own)
        const uint64_t bal = __ballot(x != 0);
        lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)bal, (uint32_t)x, 0xF6);        // lo | (bal ^ x)
        hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(bal >> 32), (uint32_t)x, 0xF6);
    }
    return ~(((uint64_t)hi << 32) | lo) & __ballot(valid);
}

// T: record word type.  D: look-back descriptor word (uint32_t when n < 2^30).
// kd_*: the key word that carries this pass's digit; ko_*: the other key word;
// v_in may be null, in which case the payload is the record's global index.
// dbg (optional): every 64th tile stores shader-clock stamps of its phases.
// (A variant that found the lanes of a wave with equal digits through a per-wave table of 64-bit lane masks in LDS instead of
//  eight ballots halved the vector instructions of the pass and changed its time by -2 % / +3 %: removed in round 3,
//  DESIGN section 7.)
template <typename T, int TILE, int NW> struct ScatterShared {
    T stage[TILE];
    uint8_t sdig[TILE];          // digit of the record at each tile-sorted position
    unsigned wcnt[NW * RADIX];   // per-wave digit counters -> per-wave exclusive bases
    unsigned bstart[RADIX];      // tile-local start of each digit
    T goff[RADIX];               // global offset of digit run minus bstart (wraps)
    unsigned scan_tmp[NW + 1];
    unsigned s_tile;
};

// FULL: the tile holds exactly TILE records, so no bounds guards are compiled in.
// LB: global offsets by decoupled look-back over `desc`; otherwise they were
// precomputed (tile_excl row of this tile + slab_excl row of its slab).
// EXT: the 8-bit class of a record comes from a separate array (dsrc, not moved) instead of a key
// digit: one such pass partitions records by an externally computed destination.
// NOKO: records of two words (digit word + payload); ko_in / ko_out are not touched.
// VN (64-bit words, payloads below 2^32 -- the suffix indices of a text of at most 2^32 characters): 1 = the payload arrays
// hold 32-bit entries on both sides, 2 = 32-bit entries in, full words out (the last pass of a sort).  The first round's
// prefix sort then moves 12 instead of 16 bytes per record and pass.
// CLSB (EXT only): bytes per entry of the class array dsrc (sizeof(T), or 1 for a byte array).
// voff: added to the payload a pass makes up itself (v_in == nullptr): the records of a rank's block, or of a piece of it.
// DNEXT3 (records of several words, three-kernel form): as dnext below for the passes of radix_scatter3_kernel -- the byte comes from the digit word
// (dnext_ko = 0) or from the other key word (dnext_ko = 1: the next pass sorts on a digit of that word)
template <typename T, typename D, int BLOCK, int ITEMS, bool FULL, bool LB, bool EXT = false, bool NOKO = false, int VN = 0, int CLSB = sizeof(T), bool DNEXT3 = false>
__device__ __forceinline__ void radix_scatter_tile(
    ScatterShared<T, BLOCK * ITEMS, BLOCK / WAVE>& sh, const unsigned tile, const unsigned count,
    const T* __restrict__ kd_in, const T* __restrict__ ko_in, const T* __restrict__ v_in,
    T* __restrict__ kd_out, T* __restrict__ ko_out, T* __restrict__ v_out, int shift,
    const unsigned long long* __restrict__ digit_base, D* __restrict__ desc, unsigned* __restrict__ err,
    unsigned long long* __restrict__ dbg, const uint64_t spec, const uint64_t spec_n,
    const unsigned* __restrict__ tile_excl, const unsigned long long* __restrict__ slab_excl,
    const T* __restrict__ dsrc = nullptr, const unsigned slab_tiles = SLAB_TILES, const uint64_t voff = 0, const unsigned pack = 0,
    const T (*kd_made)[ITEMS] = nullptr, uint8_t* __restrict__ dnext = nullptr, const int dnext_shift = 0, const uint64_t out_pad = 0,
    const int dnext_ko = 0) {
    // out_pad: the records of digit d land d * out_pad places further (the pass on the top digit: 256 output fronts that lie a multiple of
    // 2^27 bytes apart alias in the memory channels -- 14.1 against 7.2 ms for the stores alone at 2^32 records, tools/ubench_fronts.hip)
    // dnext (one-word records out): byte `at` receives bits dnext_shift .. + 7 of the record written to place `at` -- the digit the NEXT pass
    // sorts on, so that its tile histograms read one byte per record instead of the record (radix_tile_hist_bytes_kernel)
    constexpr int TILE = BLOCK * ITEMS;
    constexpr int NW = BLOCK / WAVE;
    // VN 7 / 8 / 9: one-word records (the prefix sort of the first round, most significant digit first; engine.hpp: prefix_sort_1w).
    // 7: the pass on the TOP digit of the prefix; it makes the payload up and writes ONE word per record, (prefix without its top
    //    digit) << 32 | payload (pack = lo1, the bits of word 1 below the prefix); 8: one-word records in and out (a digit of the
    //    upper half; the records of one top digit only: the caller offsets the arrays); 9: one-word records in, word 1
    //    (prefix << lo1, top digit = voff, pack = lo1 | (prefix bits without the top digit) << 8) and the 64-bit payload out.
    // 10: as 7, with word 1 of the thread's records handed over in registers (kd_made[i] = the record (wave, i, lane) of the striped order:
    //     sa_kernels.hpp: key_scatter1w_kernel cuts them out of the tile's packed text instead of reading them from memory)
    // Width of the payload field of a one-word record: bits 16 .. 23 of `pack` (0 = 32).  A text of up to 2^S characters keeps
    // 64 - S prefix bits in the word (the multi-GPU engine: 2^34 characters -> 34 + 30).
    constexpr bool ONEW_IN = VN == 8 || VN == 9;
    constexpr bool ONEW_OUT = VN == 7 || VN == 8 || VN == 10;
    constexpr bool ONEW_MAKE = VN == 7 || VN == 10;
    const unsigned sfield = (ONEW_MAKE || VN == 9) ? (((pack >> 16) & 255u) ? ((pack >> 16) & 255u) : 32u) : 0u;
    T* const stage = sh.stage;
    uint8_t* const sdig = sh.sdig;
    unsigned* const wcnt = sh.wcnt;
    unsigned* const bstart = sh.bstart;
    T* const goff = sh.goff;
    unsigned* const scan_tmp = sh.scan_tmp;

    const unsigned tid = threadIdx.x;
    const unsigned lane = lane_id();
    const unsigned wave = tid / WAVE;
    const bool stamp = dbg != nullptr && (tile & 63u) == 0 && tid == 0;
    unsigned long long* mydbg = dbg + (size_t)(tile >> 6) * 8;
    if (stamp) mydbg[0] = __builtin_amdgcn_s_memtime();
    const uint64_t base = (uint64_t)tile * TILE;

    // wave-striped load: record (wave, i, lane) = base + wave*64*ITEMS + i*64 + lane
    const T* __restrict__ pkd = kd_in + base;
    const T* __restrict__ pko = ko_in + base;
    const T* __restrict__ pv = v_in ? v_in + base : nullptr;
    // offsets of this tile (three-kernel form): fetched now, used after the ranking -- except in the narrow forms, which run
    // under a register cap (engine.hpp: launch_pass3) and rank with two registers less when the fetch comes afterwards; the
    // three-word pass loses 1.5 ms of 20 when the fetch comes late (its latency is then exposed before the barrier)
    constexpr bool LATE_EXCL = NOKO && VN != 0;
    uint64_t pre_excl = 0;
    if (!LATE_EXCL && !LB && tid < RADIX)
        pre_excl = (uint64_t)tile_excl[(uint64_t)tile * RADIX + tid] + (uint64_t)slab_excl[(uint64_t)(tile / slab_tiles) * RADIX + tid] +
                   (uint64_t)digit_base[tid];
    // (payload registers: 32 bits wide whenever the entries are -- the narrow and packed forms of 64-bit words)
    typedef typename std::conditional<(VN != 0) && sizeof(T) == 8, uint32_t, T>::type PV;
    T kd[ITEMS], ko[NOKO ? 1 : ITEMS];
    PV vv[ITEMS];
    unsigned char cls[EXT ? ITEMS : 1];
    const unsigned wbase = wave * (WAVE * ITEMS) + lane;
    if (EXT) {
        const T* __restrict__ pd = dsrc + base;
        const unsigned char* __restrict__ pb = reinterpret_cast<const unsigned char*>(dsrc) + base;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const unsigned loc = wbase + i * WAVE;
            if (CLSB == 1) cls[EXT ? i : 0] = (FULL || loc < count) ? pb[loc] : (unsigned char)0;
            else cls[EXT ? i : 0] = (FULL || loc < count) ? (unsigned char)pd[loc] : (unsigned char)0;
        }
    }
    // (non-temporal loads, so that the streamed input does not push half-written output lines out of L2: 255.3 / 264.4 ms for the
    //  4 GiB construction against 255.9 / 256.3 -- inside the run-to-run spread, dropped)
    // (lane pointer + constant: the loads of a thread differ in their immediate offsets only -- with the 32-bit sum wbase + i * 64
    //  as the index every load had its own address register)
    const T* __restrict__ pkd_l = pkd + wbase;
    if (VN == 10) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) kd[i] = (*kd_made)[i];
    } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned loc = wbase + i * WAVE;
        kd[i] = (FULL || loc < count) ? pkd_l[i * WAVE] : (T)0;
    }
    }
    if (!NOKO) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const unsigned loc = wbase + i * WAVE;
            ko[NOKO ? 0 : i] = (FULL || loc < count) ? (pko + wbase)[i * WAVE] : (T)0;
        }
    }
    // (one uniform branch around the whole group of loads: tested per record, the compiler kept it inside the unrolled loop and
    //  waited for a load in the middle of the group to spill it)
    if (ONEW_IN) {
        // one-word records: the payload sits in the low half of the key word
    } else if (pv) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const unsigned loc = wbase + i * WAVE;
            if (VN == 1 || VN == 2) vv[i] = (FULL || loc < count) ? (PV)(reinterpret_cast<const uint32_t*>(v_in) + base + wbase)[i * WAVE] : (PV)0;
            else vv[i] = (FULL || loc < count) ? (PV)(pv + wbase)[i * WAVE] : (PV)0;
        }
    } else if (!ONEW_MAKE) {         // (one-word records out: the payload is put together when the word is staged)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            // implicit payload: the record index, or the suffix the first-round record stands for
            const uint64_t g = base + wbase + i * WAVE;
            const uint64_t made = (spec_n ? (g < spec ? spec_n - 1 - g : g - spec) : g) + voff;
            vv[i] = (PV)made;
        }
    }

    // rank inside the wave, round by round (keeps the sort stable).  Every lane reads the
    // wave's running count of its digit, then the lowest lane of each digit group adds the
    // group size; LDS operations of one wave execute in program order, so the rounds chain
    // through the counters without any explicit wait between them.
    unsigned rank[ITEMS];
    unsigned* mycnt = wcnt + wave * RADIX;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const bool valid = FULL || (wbase + i * WAVE) < count;
        const unsigned d = EXT ? (unsigned)cls[EXT ? i : 0] : ((unsigned)(kd[i] >> shift) & (RADIX - 1));
        const uint64_t m = match_any8(d, valid);
        const unsigned prior = __hip_atomic_load(&mycnt[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        // lanes of the group below this one (mbcnt: no lane mask held in registers)
        const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (valid && below == 0)
            __hip_atomic_fetch_add(&mycnt[d], (unsigned)__builtin_popcountll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        rank[i] = prior + below;
        // (Experiment, round 3: the whole ranking replaced by ONE returning LDS atomic add per record -- the LDS unit served the
        //  lanes of a wave in ascending order in every run, so the sort stayed stable and all parity tests and the full-size
        //  checker passed -- removes 600 of the 900 vector instructions of a wave's tile and leaves the pass at the same
        //  25.0 ms: the pass is not bound by instruction issue.  Not kept: the order is not a documented property.)
    }
    if (LATE_EXCL && !LB && tid < RADIX)
        pre_excl = (uint64_t)tile_excl[(uint64_t)tile * RADIX + tid] + (uint64_t)slab_excl[(uint64_t)(tile / slab_tiles) * RADIX + tid] +
                   (uint64_t)digit_base[tid];
    __syncthreads();
    if (stamp) mydbg[1] = __builtin_amdgcn_s_memtime();

    // per digit: exclusive bases over waves, tile total, look-back
    unsigned tot = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const unsigned c = wcnt[w * RADIX + tid];
            wcnt[w * RADIX + tid] = tot;
            tot += c;
        }
        // publish the tile aggregate as early as possible
        if (LB) desc_store<D>(desc + (uint64_t)tile * RADIX + tid, tile == 0 ? 2u : 1u, (D)tot);
    }
    unsigned tile_total;
    unsigned bs = block_scan_exclusive<BLOCK, unsigned>(tid < RADIX ? tot : 0u, OpSum(), 0u, scan_tmp, &tile_total);
    if (stamp) mydbg[2] = __builtin_amdgcn_s_memtime();
    if (tid < RADIX) {
        bstart[tid] = bs;
        uint64_t excl = 0;
        if (!LB) {
            excl = pre_excl;
        } else if (tile != 0) {
            long long t = (long long)tile - 1;
            const D* dp = desc + tid;
            // two predecessors per step: the second load is speculative and hides one round trip
            while (t >= 0) {
                D d0 = desc_load<D>(dp + (uint64_t)t * RADIX);
                D d1 = t > 0 ? desc_load<D>(dp + (uint64_t)(t - 1) * RADIX) : (D)0;
                if ((d0 >> Desc<D>::SHIFT) == 0) d0 = desc_wait<D>(dp + (uint64_t)t * RADIX, err);
                excl += (uint64_t)(d0 & Desc<D>::MASK);
                if ((d0 >> Desc<D>::SHIFT) == 2u) break;
                if (t == 0) break;
                if ((d1 >> Desc<D>::SHIFT) == 0) d1 = desc_wait<D>(dp + (uint64_t)(t - 1) * RADIX, err);
                excl += (uint64_t)(d1 & Desc<D>::MASK);
                if ((d1 >> Desc<D>::SHIFT) == 2u) break;
                t -= 2;
            }
            desc_store<D>(desc + (uint64_t)tile * RADIX + tid, 2u, (D)(excl + tot));
        }
        goff[tid] = (T)((LB ? (uint64_t)digit_base[tid] : 0ull) + excl - (uint64_t)bs + (uint64_t)tid * out_pad);
    }
    __syncthreads();
    if (stamp) mydbg[3] = __builtin_amdgcn_s_memtime();

    // final tile-local position of every record; stage the digit word and the digit
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned d = EXT ? (unsigned)cls[EXT ? i : 0] : ((unsigned)(kd[i] >> shift) & (RADIX - 1));
        rank[i] += bstart[d] + mycnt[d];
        T staged = kd[i];
        if (ONEW_MAKE) {
            const uint64_t g = base + wbase + i * WAVE;
            const uint64_t made = (spec_n ? (g < spec ? spec_n - 1 - g : g - spec) : g) + voff;
            staged = (T)((((uint64_t)kd[i] >> (pack & 255u)) << sfield) | made);
        }
        if (FULL || (wbase + i * WAVE) < count) { stage[rank[i]] = staged; sdig[rank[i]] = (uint8_t)d; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (FULL || p < count) {
            const T x = stage[p];
            const T at = (T)(goff[sdig[p]] + (T)p);
            if (VN == 9) {
                kd_out[at] = (T)(((voff << ((pack >> 8) & 255u)) | ((uint64_t)x >> sfield)) << (pack & 255u));
                v_out[at] = (T)((uint64_t)x & ((1ull << sfield) - 1ull));
            } else {
                kd_out[at] = x;
                if (ONEW_OUT && dnext) dnext[at] = (uint8_t)((uint64_t)x >> dnext_shift);
                if (DNEXT3 && !dnext_ko) dnext[at] = (uint8_t)((uint64_t)x >> dnext_shift);
            }
        }
    }
    if (ONEW_OUT || VN == 9) return;            // one word moved: nothing else to stage
    __syncthreads();
    if (stamp) mydbg[4] = __builtin_amdgcn_s_memtime();
    if (!NOKO) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if (FULL || (wbase + i * WAVE) < count) stage[rank[i]] = ko[NOKO ? 0 : i];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const unsigned p = tid + j * BLOCK;
            if (FULL || p < count) {
                const T at = (T)(goff[sdig[p]] + (T)p);
                ko_out[at] = stage[p];
                if (DNEXT3 && dnext_ko) dnext[at] = (uint8_t)((uint64_t)stage[p] >> dnext_shift);
            }
        }
        __syncthreads();
    }
    if (stamp) mydbg[5] = __builtin_amdgcn_s_memtime();
    // (the payload goes through the stage in its own width: 32-bit entries of a 64-bit word's sort are staged as 32 bits)
    PV* const pstage = reinterpret_cast<PV*>(stage);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (FULL || (wbase + i * WAVE) < count) pstage[rank[i]] = vv[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (FULL || p < count) {
            const T at = (T)(goff[sdig[p]] + (T)p);
            if (VN == 1) reinterpret_cast<uint32_t*>(v_out)[at] = (uint32_t)pstage[p];
            else v_out[at] = (T)pstage[p];
        }
    }
    if (stamp) mydbg[6] = __builtin_amdgcn_s_memtime();
}

template <typename T, typename D, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void radix_scatter_kernel(
    const T* __restrict__ kd_in, const T* __restrict__ ko_in, const T* __restrict__ v_in,
    T* __restrict__ kd_out, T* __restrict__ ko_out, T* __restrict__ v_out, uint64_t n, int shift,
    const unsigned long long* __restrict__ digit_base, D* __restrict__ desc,
    unsigned* __restrict__ tile_counter, unsigned* __restrict__ err,
    unsigned long long* __restrict__ dbg, uint64_t spec, uint64_t spec_n, unsigned chunk) {
    constexpr int TILE = BLOCK * ITEMS;
    constexpr int NW = BLOCK / WAVE;
    static_assert(BLOCK >= RADIX, "one thread per digit needed");
    __shared__ ScatterShared<T, TILE, NW> sh;
    if (threadIdx.x == 0) sh.s_tile = claim_tile(tile_counter, gridDim.x, chunk);
    for (int i = threadIdx.x; i < NW * RADIX; i += BLOCK) sh.wcnt[i] = 0;
    __syncthreads();
    const unsigned tile = sh.s_tile;
    const uint64_t remain = n - (uint64_t)tile * TILE;
    if (remain >= (uint64_t)TILE)
        radix_scatter_tile<T, D, BLOCK, ITEMS, true, true>(sh, tile, (unsigned)TILE, kd_in, ko_in, v_in, kd_out, ko_out,
                                                     v_out, shift, digit_base, desc, err, dbg, spec, spec_n, nullptr, nullptr);
    else
        radix_scatter_tile<T, D, BLOCK, ITEMS, false, true>(sh, tile, (unsigned)remain, kd_in, ko_in, v_in, kd_out, ko_out,
                                                      v_out, shift, digit_base, desc, err, dbg, spec, spec_n, nullptr, nullptr);
}

// ---------------------------------------------------------------------------
// Destination class of every record for a splitter-based shuffle (sample sort): the number of
// splitters that do not sort after the record in the total order (k1, k2, rank, index).
struct Splitters { unsigned n; unsigned long long k1[64], k2[64], rank[64], idx[64]; };

template <typename T>
__global__ void classify_kernel(const T* __restrict__ k1, const T* __restrict__ k2, uint64_t n, Splitters sp,
                                unsigned long long my_rank, T* __restrict__ cls) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long a = k1[i], b = k2[i];
        unsigned lo = 0, hi = sp.n;                       // first splitter that sorts after the record
        while (lo < hi) {
            const unsigned mid = (lo + hi) >> 1;
            bool after;                                    // splitter[mid] > record ?
            if (sp.k1[mid] != a) after = sp.k1[mid] > a;
            else if (sp.k2[mid] != b) after = sp.k2[mid] > b;
            else if (sp.rank[mid] != my_rank) after = sp.rank[mid] > my_rank;
            else after = sp.idx[mid] > i;
            if (after) hi = mid; else lo = mid + 1;
        }
        cls[i] = (T)lo;
    }
}

// Destination of a two-word record (word 1, suffix) in a shuffle by the leading bits of word 1: the number of splitters
// that do not exceed its prefix, so that records with equal prefixes never part (their order by the rest of the window is
// decided on the receiving rank).
// The classes leave as bytes; counts[q * 64 + d] += records of piece q (pieces of `piece` records; a workgroup never
// straddles two) that go to destination d.
template <typename T>
__global__ __launch_bounds__(256) void classify_prefix_kernel(const T* __restrict__ k1, uint64_t n, unsigned lo1, Splitters sp, uint8_t* __restrict__ cls,
                                                              uint64_t piece, unsigned long long* __restrict__ counts) {
    __shared__ unsigned lh[64];
    __shared__ unsigned long long skey[64];
    if (threadIdx.x < 64) { lh[threadIdx.x] = 0; skey[threadIdx.x] = threadIdx.x < sp.n ? sp.k1[threadIdx.x] : ~0ull; }
    __syncthreads();
    // workgroup b works on [b * span, (b + 1) * span) of its piece
    constexpr uint64_t SPAN = 256 * 32;
    const uint64_t per_piece = (piece + SPAN - 1) / SPAN;
    const uint64_t q = blockIdx.x / per_piece, b = blockIdx.x % per_piece;
    const uint64_t lo_i = q * piece + b * SPAN;
    uint64_t hi_i = lo_i + SPAN;
    if (hi_i > (q + 1) * piece) hi_i = (q + 1) * piece;
    if (hi_i > n) hi_i = n;
    const unsigned ns = sp.n;
    for (uint64_t i = lo_i + threadIdx.x; i < hi_i; i += 256) {
        const unsigned long long a = (unsigned long long)k1[i] >> lo1;
        unsigned lo = 0, hi = ns;                         // first splitter that exceeds the prefix
        while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (skey[mid] > a) hi = mid; else lo = mid + 1; }
        cls[i] = (uint8_t)lo;
        atomicAdd(&lh[lo], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64 && lh[threadIdx.x]) atomicAdd(&counts[q * 64 + threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

// per-tile class counts of a byte class array (the tile shape of the scatter pass that follows)
template <int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void class_tile_hist_kernel(const uint8_t* __restrict__ cls, uint64_t n, unsigned* __restrict__ tile_hist) {
    constexpr int TILE = BLOCK * ITEMS;
    __shared__ unsigned lh[4][RADIX];
    for (int i = threadIdx.x; i < 4 * RADIX; i += BLOCK) (&lh[0][0])[i] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    unsigned* my = lh[(threadIdx.x / WAVE) & 3];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint64_t e = base + (uint64_t)i * BLOCK + threadIdx.x;
        wave_hist_add(my, e < n ? (unsigned)cls[e] : 0u, e < n);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < RADIX; d += BLOCK)
        tile_hist[(uint64_t)blockIdx.x * RADIX + d] = lh[0][d] + lh[1][d] + lh[2][d] + lh[3][d];
}

// ---------------------------------------------------------------------------
// Three-kernel form of a pass (no inter-workgroup waiting): per-tile digit
// histograms -> exclusive scan over tiles (within slabs of SLAB_TILES tiles, then
// over slabs) -> scatter with known offsets.  Costs one extra read of the digit
// word (w bytes per record) and 1 KiB of counters per tile.
// ---------------------------------------------------------------------------
template <typename T, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void radix_tile_hist_kernel(const T* __restrict__ kd_in, uint64_t n, int shift,
                                                                unsigned* __restrict__ tile_hist) {
    constexpr int TILE = BLOCK * ITEMS;
    constexpr int PER = 16 / sizeof(T);
    static_assert(ITEMS % PER == 0, "ITEMS must cover whole vectors");
    __shared__ unsigned lh[4][RADIX];
    for (int i = threadIdx.x; i < 4 * RADIX; i += BLOCK) (&lh[0][0])[i] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    unsigned* my = lh[(threadIdx.x / WAVE) & 3];
#pragma unroll
    for (int v = 0; v < ITEMS / PER; ++v) {
        const uint64_t e0 = base + ((uint64_t)v * BLOCK + threadIdx.x) * PER;
        T x[PER];
        load_run<T, PER>(kd_in, e0, n, x, (T)0);
#pragma unroll
        for (int j = 0; j < PER; ++j) wave_hist_add(my, (unsigned)(x[j] >> shift) & (RADIX - 1), e0 + j < n);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < RADIX; d += BLOCK)
        tile_hist[(uint64_t)blockIdx.x * RADIX + d] = lh[0][d] + lh[1][d] + lh[2][d] + lh[3][d];
}

// one workgroup per slab: in-place exclusive scan of its tiles' counts per digit, slab totals out
template <int TAG>
__global__ __launch_bounds__(RADIX) void radix_slab_scan_kernel(unsigned* __restrict__ tile_hist, uint64_t ntiles,
                                                                unsigned long long* __restrict__ slab_tot,
                                                                unsigned slab_tiles = SLAB_TILES) {
    const uint64_t t0 = (uint64_t)blockIdx.x * slab_tiles;
    const unsigned d = threadIdx.x;
    unsigned long long run = 0;
    constexpr int B = 16;
    for (unsigned b0 = 0; b0 < slab_tiles; b0 += B) {
        unsigned v[B];
#pragma unroll
        for (int j = 0; j < B; ++j) v[j] = (t0 + b0 + j < ntiles) ? tile_hist[(t0 + b0 + j) * RADIX + d] : 0u;
#pragma unroll
        for (int j = 0; j < B; ++j) {
            if (t0 + b0 + j < ntiles) tile_hist[(t0 + b0 + j) * RADIX + d] = (unsigned)run;
            run += v[j];
        }
    }
    slab_tot[(uint64_t)blockIdx.x * RADIX + d] = run;
}

// one workgroup: exclusive scan of the slab totals per digit (in place)
// also turns the per-digit totals into the global start of every digit (digit_base)
template <int TAG>
__global__ __launch_bounds__(RADIX) void radix_top_scan_kernel(unsigned long long* __restrict__ slab_tot, uint64_t nslabs,
                                                               unsigned long long* __restrict__ digit_base) {
    __shared__ unsigned long long tmp[RADIX / WAVE + 1];
    const unsigned d = threadIdx.x;
    unsigned long long run = 0;
    constexpr int B = 16;
    for (uint64_t b0 = 0; b0 < nslabs; b0 += B) {
        unsigned long long v[B];
#pragma unroll
        for (int j = 0; j < B; ++j) v[j] = (b0 + j < nslabs) ? slab_tot[(b0 + j) * RADIX + d] : 0ull;
#pragma unroll
        for (int j = 0; j < B; ++j) {
            if (b0 + j < nslabs) slab_tot[(b0 + j) * RADIX + d] = run;
            run += v[j];
        }
    }
    unsigned long long total;
    const unsigned long long start = block_scan_exclusive<RADIX, unsigned long long>(run, OpSum(), 0ull, tmp, &total);
    digit_base[d] = start;
}

template <typename T, int BLOCK, int ITEMS, bool EXT = false, int MINW = 1, bool NOKO = false, int VN = 0, int CLSB = sizeof(T), bool DNEXT3 = false>
__global__ __launch_bounds__(BLOCK, MINW) void radix_scatter3_kernel(
    const T* __restrict__ kd_in, const T* __restrict__ ko_in, const T* __restrict__ v_in,
    T* __restrict__ kd_out, T* __restrict__ ko_out, T* __restrict__ v_out, uint64_t n, int shift,
    const unsigned long long* __restrict__ digit_base, const unsigned* __restrict__ tile_excl,
    const unsigned long long* __restrict__ slab_excl, unsigned long long* __restrict__ dbg, uint64_t spec,
    uint64_t spec_n, unsigned* __restrict__ tile_counter, unsigned chunk, const T* __restrict__ dsrc = nullptr,
    unsigned slab_tiles = SLAB_TILES, uint64_t voff = 0, unsigned pack = 0, uint8_t* __restrict__ dnext = nullptr, int dnext_shift = 0, int dnext_ko = 0) {
    // dnext (DNEXT3): one byte per record, at the record's place in the output: the digit the next pass sorts on (radix_tile_hist_bytes_flat_kernel)
    // (a persistent variant, one workgroup looping over tiles with its next ticket prefetched, was
    // measured: the loop raised the register count from 118 to 173 and lost 20 %)
    constexpr int TILE = BLOCK * ITEMS;
    constexpr int NW = BLOCK / WAVE;
    static_assert(BLOCK >= RADIX, "one thread per digit needed");
    static_assert(VN == 0 || sizeof(T) == 8, "narrow payloads exist for records of 64-bit words");
    __shared__ ScatterShared<T, TILE, NW> sh;
    // tiles are handed out in start order so that neighbouring runs of a digit are written
    // close in time (they share cache lines); nothing ever waits on another workgroup
    if (threadIdx.x == 0) sh.s_tile = tile_counter ? claim_tile(tile_counter, gridDim.x, chunk) : blockIdx.x;
    for (int i = threadIdx.x; i < NW * RADIX; i += BLOCK) sh.wcnt[i] = 0;
    __syncthreads();
    const unsigned tile = sh.s_tile;
    const uint64_t remain = n - (uint64_t)tile * TILE;
    if (remain >= (uint64_t)TILE)
        radix_scatter_tile<T, unsigned, BLOCK, ITEMS, true, false, EXT, NOKO, VN, CLSB, DNEXT3>(sh, tile, (unsigned)TILE, kd_in, ko_in, v_in, kd_out,
                                                                        ko_out, v_out, shift, digit_base, nullptr, nullptr, dbg,
                                                                        spec, spec_n, tile_excl, slab_excl, dsrc, slab_tiles, voff, pack,
                                                                        nullptr, dnext, dnext_shift, 0, dnext_ko);
    else
        radix_scatter_tile<T, unsigned, BLOCK, ITEMS, false, false, EXT, NOKO, VN, CLSB, DNEXT3>(sh, tile, (unsigned)remain, kd_in, ko_in, v_in, kd_out,
                                                                         ko_out, v_out, shift, digit_base, nullptr, nullptr, dbg,
                                                                         spec, spec_n, tile_excl, slab_excl, dsrc, slab_tiles, voff, pack,
                                                                         nullptr, dnext, dnext_shift, 0, dnext_ko);
}

// ---------------------------------------------------------------------------
// One-word records, most significant digit first (engine.hpp: prefix_sort_1w).  After the pass on the top digit of the prefix the
// records of top digit b lie in [bucket_off[b], bucket_off[b + 1]) as ONE 64-bit word each (rest of the prefix << 32 | suffix); the
// remaining digits are LSD passes inside every bucket.  All 256 buckets run in one launch.  The tiles of a bucket are numbered
// from a slab boundary: bucket b owns the slabs slab_start[b] .. slab_start[b + 1] - 1 of `slab` tiles each (the tiles beyond
// the end of the bucket in its last slab do nothing), slab_bucket[s] names the bucket of slab s, so a tile index finds its bucket
// with one look-up and the tables have at most n / tile + 256 * slab rows however unevenly the top digit is filled.
// Eight bytes per record read and written per pass instead of twelve, one stage through LDS instead of two, and no register
// of a thread holds payload.
// ---------------------------------------------------------------------------
struct __attribute__((aligned(32))) SlabInfo {      // what a workgroup needs to know about its slab: one 32-byte read
    unsigned long long first;                // global index of the first record of the slab's first tile
    unsigned long long end;                  // end of the bucket
    unsigned bucket, slab0;                  // the bucket and its first slab
    unsigned pad[2];
};
struct OneWordTabs {
    const unsigned long long* bucket_off;    // [257]
    const unsigned long long* slab_start;    // [257]
    const SlabInfo* slab_info;               // [slab_start[256]]
    unsigned slab;                           // tiles per slab
};

template <int TAG>
__global__ void radix_slab_info_kernel(const unsigned long long* __restrict__ bucket_off, const unsigned long long* __restrict__ slab_start,
                                       unsigned total_slabs, unsigned slab_records, SlabInfo* __restrict__ slab_info) {
    const unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= total_slabs) return;
    unsigned lo = 0, hi = RADIX;            // the last b with slab_start[b] <= s (buckets without records own no slab)
    while (hi - lo > 1) { const unsigned mid = (lo + hi) >> 1; if (slab_start[mid] <= s) lo = mid; else hi = mid; }
    SlabInfo si;
    si.bucket = lo; si.slab0 = (unsigned)slab_start[lo];
    si.first = bucket_off[lo] + (unsigned long long)(s - si.slab0) * slab_records;
    si.end = bucket_off[lo + 1];
    si.pad[0] = si.pad[1] = 0;
    slab_info[s] = si;
}

template <int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void radix_tile_hist1w_kernel(const uint64_t* __restrict__ in, OneWordTabs tb, int shift, unsigned* __restrict__ tile_hist,
                                                                  uint64_t in_pad = 0) {
    // in_pad (even): the records of bucket b lie b * in_pad places behind where the tables say (radix_scatter_tile: out_pad)
    constexpr int TILE = BLOCK * ITEMS;
    constexpr int PER = 2;
    const unsigned vt = blockIdx.x;
    const unsigned gs = vt / tb.slab;
    const SlabInfo si = tb.slab_info[gs];
    in += (uint64_t)si.bucket * in_pad;
    const uint64_t g0 = si.first + (uint64_t)(vt - gs * tb.slab) * TILE;      // global index of the tile's first record
    if (g0 >= si.end) return;
    const uint64_t g1 = si.end - g0 < (uint64_t)TILE ? si.end : g0 + TILE;
    __shared__ unsigned lh[4][RADIX];
    for (int i = threadIdx.x; i < 4 * RADIX; i += BLOCK) (&lh[0][0])[i] = 0;
    __syncthreads();
    unsigned* my = lh[(threadIdx.x / WAVE) & 3];
    // a bucket starts anywhere: 16-byte loads at even global indices; the first pair of a tile that starts at an odd index
    // holds a record of the tile before (left out), the last record of such a tile is picked up by itself
    const uint64_t a0 = g0 & ~1ull;
    if (a0 + TILE <= si.end) {
        // every pair lies inside the bucket: all loads of a thread issued together
        uint4 q[ITEMS / PER];
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(in + a0) + threadIdx.x;
#pragma unroll
        for (int v = 0; v < ITEMS / PER; ++v) q[v] = src[v * BLOCK];
#pragma unroll
        for (int v = 0; v < ITEMS / PER; ++v) {
            const uint64_t e0 = a0 + ((uint64_t)v * BLOCK + threadIdx.x) * PER;
            const uint64_t x0 = ((uint64_t)q[v].y << 32) | q[v].x, x1 = ((uint64_t)q[v].w << 32) | q[v].z;
            wave_hist_add(my, (unsigned)(x0 >> shift) & (RADIX - 1), e0 >= g0);
            wave_hist_add(my, (unsigned)(x1 >> shift) & (RADIX - 1), e0 + 1 < g1);
        }
    } else
#pragma unroll
    for (int v = 0; v < ITEMS / PER; ++v) {
        const uint64_t e0 = a0 + ((uint64_t)v * BLOCK + threadIdx.x) * PER;
        uint64_t x0 = 0, x1 = 0;
        if (e0 + 1 < g1) {
            const uint4 q = *reinterpret_cast<const uint4*>(in + e0);
            x0 = ((uint64_t)q.y << 32) | q.x; x1 = ((uint64_t)q.w << 32) | q.z;
        } else if (e0 < g1) x0 = in[e0];                                // (the pair straddles the end of the tile: nothing is read beyond it)
        wave_hist_add(my, (unsigned)(x0 >> shift) & (RADIX - 1), e0 >= g0 && e0 < g1);
        wave_hist_add(my, (unsigned)(x1 >> shift) & (RADIX - 1), e0 + 1 < g1);
    }
    if (a0 != g0 && threadIdx.x == 0 && a0 + TILE < g1) {
        const uint64_t x = in[a0 + TILE];
        atomicAdd(&my[(unsigned)(x >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
    unsigned* row = tile_hist + (uint64_t)vt * RADIX;
    for (int d = threadIdx.x; d < RADIX; d += BLOCK) row[d] = lh[0][d] + lh[1][d] + lh[2][d] + lh[3][d];
}

// The same tile histograms from the digit bytes the pass before left beside its records (radix_scatter_tile: dnext): one byte read per
// record instead of eight.  A WAVE takes a tile (TILE bytes = 64 per lane, read as four aligned 16-byte pieces; a tile's bytes start
// anywhere: the bytes outside the tile are masked off, a lane-0 tail picks up what the shifted pieces miss) and counts into its own
// 256 counters in LDS -- no barrier; eight tiles per workgroup.
template <int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void radix_tile_hist_bytes_kernel(const uint8_t* __restrict__ dig, OneWordTabs tb, unsigned vtiles, unsigned* __restrict__ tile_hist) {
    constexpr int TILE = BLOCK * ITEMS, NW = BLOCK / WAVE;
    static_assert(TILE == 64 * WAVE, "a lane takes 64 bytes of the tile");
    __shared__ unsigned lh[NW][RADIX];
    const unsigned wave = threadIdx.x / WAVE, lane = lane_id();
    const unsigned vt = blockIdx.x * NW + wave;
    if (vt >= vtiles) return;
    const unsigned gs = vt / tb.slab;
    const SlabInfo si = tb.slab_info[gs];
    const uint64_t g0 = si.first + (uint64_t)(vt - gs * tb.slab) * TILE;      // global index of the tile's first record
    if (g0 >= si.end) return;
    const uint64_t g1 = si.end - g0 < (uint64_t)TILE ? si.end : g0 + TILE;
    unsigned* const my = lh[wave];
#pragma unroll
    for (int i = 0; i < RADIX / WAVE; ++i) my[i * WAVE + lane] = 0;
    xrun_order();
    const uint64_t a0 = g0 & ~15ull;
    // pieces a0 + 16 (i * 64 + lane), i = 0 .. 3, and one more piece (lane 0) when the tile does not start on a piece
    auto piece = [&](uint64_t e0) {
        if (e0 >= g1) return;
        const uint4 q = *reinterpret_cast<const uint4*>(dig + e0);           // (the array is padded to whole pieces)
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) if (e0 + k >= g0 && e0 + k < g1) atomicAdd(&my[(w[k >> 2] >> (8 * (k & 3))) & 255u], 1u);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) piece(a0 + 16ull * (uint64_t)(i * WAVE + lane));
    if (lane == 0 && a0 != g0) piece(a0 + 16ull * 4 * WAVE);
    xrun_order();
    unsigned* row = tile_hist + (uint64_t)vt * RADIX;
#pragma unroll
    for (int i = 0; i < RADIX / WAVE; ++i) row[i * WAVE + lane] = my[i * WAVE + lane];
}

// ... and for the flat arrays of the three-kernel passes (engine.hpp: dispatch_pass3): tile t is the TILE bytes from t * TILE on, a wave per tile,
// the array 16-byte aligned and readable up to the next multiple of 16 beyond n.  The keys of a refinement round come in long runs of one digit
// when they are nearly sorted already (64 same-address LDS atomics serialise): a wave whose 1024 bytes of a step agree adds once.
template <int BLOCK, int TILE>
__global__ __launch_bounds__(BLOCK) void radix_tile_hist_bytes_flat_kernel(const uint8_t* __restrict__ dig, uint64_t n, uint64_t ntiles, unsigned* __restrict__ tile_hist) {
    constexpr int NW = BLOCK / WAVE;
    static_assert(TILE % (16 * WAVE) == 0, "whole 16-byte pieces per lane");
    __shared__ unsigned lh[NW][RADIX];
    const unsigned wave = threadIdx.x / WAVE, lane = lane_id();
    const uint64_t t = (uint64_t)blockIdx.x * NW + wave;
    if (t >= ntiles) return;
    unsigned* const my = lh[wave];
#pragma unroll
    for (int i = 0; i < RADIX / WAVE; ++i) my[i * WAVE + lane] = 0;
    xrun_order();
    const uint64_t g0 = t * TILE, g1 = n - g0 < (uint64_t)TILE ? n : g0 + TILE;
    uint4 q[TILE / (16 * WAVE)];
#pragma unroll
    for (int i = 0; i < TILE / (16 * WAVE); ++i) {
        const uint64_t e0 = g0 + 16ull * (uint64_t)(i * WAVE + lane);
        q[i] = e0 < g1 ? *reinterpret_cast<const uint4*>(dig + e0) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TILE / (16 * WAVE); ++i) {
        const uint64_t e0 = g0 + 16ull * (uint64_t)(i * WAVE + lane);
        const uint32_t w[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
        // Keys that are nearly sorted already, or drawn from few distinct values (repeated reads), come in runs of one digit: 64 same-address
        // LDS adds serialise.  A lane whose 16 bytes agree is `pure`; neighbouring pure lanes of one value form a segment and its first lane adds
        // once for all of them; the other lanes add once per run of equal bytes.  (Measured on 2^30 bytes of the first sort of repeated reads
        // with mutations: a plain add per byte 3.3 ms, two ballots and an add per byte -- wave_hist_add -- 2.2, the histogram over the records 1.45.)
        const bool pure = e0 + 16 <= g1 && w[0] == w[1] && w[1] == w[2] && w[2] == w[3] && ((w[0] >> 8) | (w[0] << 24)) == w[0];
        const unsigned v = w[0] & 255u;
        const uint64_t P = __ballot(pure);
        const unsigned vprev = shfl<unsigned>(v, (int)((lane + WAVE - 1) & (WAVE - 1)));
        const bool head = pure && (lane == 0 || !((P >> ((lane - 1) & (WAVE - 1))) & 1ull) || vprev != v);
        const uint64_t H = __ballot(head);
        if (head) {
            const uint64_t stop = ((H | ~P) >> lane) >> 1;          // the lanes behind this one that end its segment
            const unsigned len = stop ? (unsigned)__builtin_ctzll(stop) + 1u : (unsigned)WAVE - lane;
            atomicAdd(&my[v], 16u * len);
        }
        if (!pure && e0 < g1) {
            unsigned prev = v, cnt = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const unsigned b = (w[k >> 2] >> (8 * (k & 3))) & 255u;
                if (b != prev) { if (cnt) atomicAdd(&my[prev], cnt); prev = b; cnt = 0; }
                cnt += e0 + k < g1 ? 1u : 0u;
            }
            if (cnt) atomicAdd(&my[prev], cnt);
        }
    }
    xrun_order();
    unsigned* row = tile_hist + t * RADIX;
#pragma unroll
    for (int i = 0; i < RADIX / WAVE; ++i) row[i * WAVE + lane] = my[i * WAVE + lane];
}

// one workgroup per slab: exclusive scan of the tile counts of the slab per digit (in place), slab totals out
template <int TAG>
__global__ __launch_bounds__(RADIX) void radix_slab_scan1w_kernel(unsigned* __restrict__ tile_hist, OneWordTabs tb, unsigned tile_records,
                                                                  unsigned long long* __restrict__ slab_tot) {
    const unsigned gs = blockIdx.x;
    const SlabInfo si = tb.slab_info[gs];
    const uint64_t ntiles = (si.end - si.first + tile_records - 1) / tile_records;       // tiles from the slab's first to the end of the bucket
    const uint64_t t0 = 0;
    const unsigned d = threadIdx.x;
    unsigned* __restrict__ rows = tile_hist + (uint64_t)gs * tb.slab * RADIX;
    unsigned long long run = 0;
    constexpr int B = 16;
    for (unsigned b0 = 0; b0 < tb.slab && t0 + b0 < ntiles; b0 += B) {
        unsigned v[B];
#pragma unroll
        for (int j = 0; j < B; ++j) v[j] = (b0 + j < tb.slab && t0 + b0 + j < ntiles) ? rows[(uint64_t)(b0 + j) * RADIX + d] : 0u;
#pragma unroll
        for (int j = 0; j < B; ++j) {
            if (b0 + j < tb.slab && t0 + b0 + j < ntiles) rows[(uint64_t)(b0 + j) * RADIX + d] = (unsigned)run;
            run += v[j];
        }
    }
    slab_tot[(uint64_t)gs * RADIX + d] = run;
}

// grid 256: one workgroup per bucket: exclusive scan of its slab totals per digit (in place), start of every digit of the bucket
template <int TAG>
__global__ __launch_bounds__(RADIX) void radix_top_scan1w_kernel(unsigned long long* __restrict__ slab_tot, OneWordTabs tb,
                                                                 unsigned long long* __restrict__ digit_base) {
    __shared__ unsigned long long tmp[RADIX / WAVE + 1];
    const unsigned b = blockIdx.x, d = threadIdx.x;
    const unsigned s_lo = (unsigned)tb.slab_start[b], s_hi = (unsigned)tb.slab_start[b + 1];
    unsigned long long run = 0;
    for (unsigned gs = s_lo; gs < s_hi; ++gs) {
        const unsigned long long v = slab_tot[(uint64_t)gs * RADIX + d];
        slab_tot[(uint64_t)gs * RADIX + d] = run;
        run += v;
    }
    unsigned long long total;
    const unsigned long long start = block_scan_exclusive<RADIX, unsigned long long>(run, OpSum(), 0ull, tmp, &total);
    digit_base[(uint64_t)b * RADIX + d] = tb.bucket_off[b] + start;
}

// one workgroup per (bucket, tile), handed out in order through the per-XCD queues
template <int BLOCK, int ITEMS, int VN>
__global__ __launch_bounds__(BLOCK, ITEMS <= 6 ? 8 : ITEMS <= 8 ? 6 : 4) void radix_scatter1w_kernel(
    const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t* __restrict__ v_out, int shift, OneWordTabs tb,
    const unsigned long long* __restrict__ digit_base, const unsigned* __restrict__ tile_excl,
    const unsigned long long* __restrict__ slab_excl, unsigned* __restrict__ tile_counter, unsigned chunk, unsigned pack,
    uint8_t* __restrict__ dnext = nullptr, int dnext_shift = 0, uint64_t in_pad = 0) {
    constexpr int TILE = BLOCK * ITEMS;
    constexpr int NW = BLOCK / WAVE;
    static_assert(VN == 8 || VN == 9, "one-word records in");
    __shared__ ScatterShared<uint64_t, TILE, NW> sh;
    if (threadIdx.x == 0) sh.s_tile = tile_counter ? claim_tile(tile_counter, gridDim.x, chunk) : blockIdx.x;
    for (int i = threadIdx.x; i < NW * RADIX; i += BLOCK) sh.wcnt[i] = 0;
    __syncthreads();
    const unsigned vt = sh.s_tile;
    const unsigned gs = vt / tb.slab;
    const SlabInfo si = tb.slab_info[gs];
    const unsigned b = si.bucket, s0 = si.slab0;
    const unsigned t = vt - s0 * tb.slab;
    const uint64_t off = si.first - (uint64_t)(gs - s0) * tb.slab * TILE, n = si.end - off;
    in += (uint64_t)b * in_pad;
    if ((uint64_t)t * TILE >= n) return;
    const uint64_t remain = n - (uint64_t)t * TILE;
    const unsigned* te = tile_excl + (uint64_t)s0 * tb.slab * RADIX;
    const unsigned long long* se = slab_excl + (uint64_t)s0 * RADIX;
    const unsigned long long* db = digit_base + (uint64_t)b * RADIX;
    if (remain >= (uint64_t)TILE)
        radix_scatter_tile<uint64_t, unsigned, BLOCK, ITEMS, true, false, false, true, VN>(sh, t, (unsigned)TILE, in + off, nullptr, nullptr, out, nullptr, v_out,
                                                                                          shift, db, nullptr, nullptr, nullptr, 0, 0, te, se, nullptr, tb.slab,
                                                                                          (uint64_t)b, pack, nullptr, dnext, dnext_shift);
    else
        radix_scatter_tile<uint64_t, unsigned, BLOCK, ITEMS, false, false, false, true, VN>(sh, t, (unsigned)remain, in + off, nullptr, nullptr, out, nullptr, v_out,
                                                                                           shift, db, nullptr, nullptr, nullptr, 0, 0, te, se, nullptr, tb.slab,
                                                                                           (uint64_t)b, pack, nullptr, dnext, dnext_shift);
}

} // namespace psacx
