// construct.hpp -- the prefix-doubling loop on one MI355X.
//
// Shape of the reference loop: /root/reference/include/suffix_array.hpp:365-466
// (k-mer bucketing, then per round shift -> rank-pair sort -> LCP -> rebucket ->
// SA->ISA) and its sparse tail :1032-1285.  On the GPU both phases are one
// routine: after the first full (B1,B2) sort only suffixes that still share a
// bucket are gathered, sorted by (bucket, rank of the suffix h further) and
// written back in place, so the work per round follows the number of
// unresolved suffixes.  The final SA / ISA / LCP are identical to the
// reference's because they are uniquely determined by the text.
#pragma once
#include <chrono>
#include <thread>
#include "engine.hpp"
#include "bucket_sort.hpp"
#include "heavy_keys.hpp"

namespace psacx {

// Tile shape of the scan-structured kernels (last head / rebucket / compaction).  64-bit words run 512-thread
// workgroups (two per CU at their register count: rebucket_first_kernel 60.8 -> 51.1 ms at 2^32), 32-bit words
// 768-thread ones (the tile of their radix scatter passes; 512 and 1024 measured the same or slower).
constexpr int SCAN_ITEMS = 8;
#ifndef RB1W_BLOCK
#define RB1W_BLOCK 512      // threads of rebucket_first_kernel on one-word records (the tile stays 4096 records)
#endif
template <typename T> struct ScanCfg {
    static constexpr int BLOCK = sizeof(T) == 8 ? 512 : 768;
    static constexpr int TILE = BLOCK * SCAN_ITEMS;
};
constexpr int SCAN_TILE_MIN = 512 * SCAN_ITEMS;      // sizes per-tile arrays where the word type is not known

constexpr unsigned SCAN_CHUNKS = 4096, SCAN_CHUNK = 1024 * 4;      // (chunks of 4096 tiles; up to 2^24 tiles)
template <typename T> struct Work {
    T *bsa;
    SortBufs<T> x, y;                  // record sets of the first sort (y may alias the output buffers)
    SortBufs<T> ry;                    // second record set of the refinement rounds (cap_active records)
    T *pos_a, *pos_b;                  // active position lists (cap_active entries)
    uint64_t cap_active;
    bool diet;                         // large-n layout: the output buffers double as sort scratch
    Pyramid<T> pyr;
    T *aux_pre[PYR_MAX], *aux_suf[PYR_MAX];   // storage for pyr.pre / pyr.suf of levels >= 1
    unsigned long long* d_hist256;     // char histogram
    uint64_t* d_carry;                 // per scan tile: id of the last head (then its exclusive max-scan)
    uint64_t* d_nact;                  // per scan tile: active positions (then exclusive sum-scan)
    uint64_t* d_nunf;                  // per scan tile: buckets with > 1 member
    uint64_t* d_totals;                // [0] active, [1] unfinished buckets
    uint64_t* d_over;                  // [2] tasks too long for the sort in LDS (bucket_sort.hpp), wide / narrow windows
    uint64_t* d_chunks;                // 2 x SCAN_CHUNKS chunk totals of the long tile scans
    unsigned* d_cursors;               // fill cursors of the destination buckets (ISA inversion)
    size_t n_cursors;
    unsigned* d_gcursors;              // the same for the rank requests of a refinement round (GatherLevels; the ISA levels of a round in slabs stay open meanwhile)
    uint64_t* d_gwin;                  // ... and where the records of every window start in the round's sort input
    uint64_t* d_heavy;                 // tables of the heavy / light split of a round (heavy_keys.hpp: HeavyTabs for HEAVY_MAXB buckets)
    ulonglong2* d_htile;               // ... and per scan tile the key and the shift of the heavy run that holds it (sa_kernels.hpp: heavy_tiles_kernel)
    SortScratch sc;
};

// Normal layout: Bsa + two n-record sets + two n-entry position lists (9 n w bytes).
// Diet layout (when that does not fit in HBM): the second record set of the first sort is the
// output buffers themselves (ISA, LCP, SA are dead until the sort is over), and the refinement
// rounds get `cap` records of room instead of n (4 n w + 5 cap w bytes).
// slack of the first record array for the padded output of the pass on the top digit (engine.hpp: prefix_sort_1w)
constexpr uint64_t ONEW_PAD = 20480;            // places per bucket: 160 KiB
template <typename T> inline uint64_t onew_pad_total(uint64_t n) { return (sizeof(T) == 8 && n >= (1ull << 28)) ? (uint64_t)RADIX * ONEW_PAD : 0; }

constexpr int ISA_NARROW_WB = 14, ISA_NARROW_CB = 9;      // windows of 2^14 positions, 512-way partition levels (invert_permutation, IsaLevels, GatherLevels)
template <typename T>
size_t carve(Arena& a, Work<T>& w, uint64_t n, bool with_lcp, T* d_lcp, bool diet, uint64_t cap, T* d_sa, T* d_isa, const Knobs& kn) {
    w.diet = diet;
    w.cap_active = diet ? cap : n;
    w.bsa = a.take<T>(n);
    w.x.k1 = a.take<T>(n + onew_pad_total<T>(n)); w.x.k2 = a.take<T>(n); w.x.v = a.take<T>(n);
    if (!diet) {
        w.y.k1 = a.take<T>(n); w.y.k2 = a.take<T>(n); w.y.v = a.take<T>(n);
        w.ry = w.y;
        w.pos_a = a.take<T>(n); w.pos_b = a.take<T>(n);
    } else {
        w.y.k1 = d_isa; w.y.k2 = with_lcp ? d_lcp : a.take<T>(n); w.y.v = d_sa;
        w.ry.k1 = a.take<T>(cap); w.ry.k2 = a.take<T>(cap); w.ry.v = a.take<T>(cap);
        w.pos_a = a.take<T>(cap); w.pos_b = a.take<T>(cap);
    }
    w.pyr = Pyramid<T>();
    for (int i = 0; i < PYR_MAX; ++i) { w.aux_pre[i] = nullptr; w.aux_suf[i] = nullptr; }
    if (with_lcp) {
        w.pyr.lvl[0] = d_lcp; w.pyr.len[0] = n; w.pyr.nlev = 1;
        uint64_t len = n;
        while (len > 128 && w.pyr.nlev < PYR_MAX) {
            len = (len + 63) / 64;
            w.pyr.lvl[w.pyr.nlev] = a.take<T>(len);
            w.pyr.len[w.pyr.nlev] = len;
            w.aux_pre[w.pyr.nlev] = a.take<T>(len);      // range-minimum helpers of the upper levels (level 0: ctx->aux)
            w.aux_suf[w.pyr.nlev] = a.take<T>(len);
            w.pyr.nlev++;
        }
    }
    w.d_hist256 = a.take<unsigned long long>(256);
    const uint64_t nt = (n + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE + 1;
    w.d_carry = a.take<uint64_t>(nt);
    w.d_nact = a.take<uint64_t>(nt);
    w.d_nunf = a.take<uint64_t>(nt);
    w.d_totals = a.take<uint64_t>(4);
    w.d_over = a.take<uint64_t>(2);
    w.d_chunks = a.take<uint64_t>(2 * SCAN_CHUNKS);
    w.n_cursors = (size_t)(n >> INV_WINDOW_BITS) + 2 + RADIX_P;
    w.d_cursors = a.take<unsigned>(w.n_cursors);
    w.d_gcursors = a.take<unsigned>(1024 + (size_t)(n >> ISA_NARROW_WB) + 2);
    w.d_gwin = a.take<uint64_t>((size_t)(n >> ISA_NARROW_WB) + 2);
    w.d_heavy = a.take<uint64_t>(HeavyTabs::words(HEAVY_MAXB));
    w.d_htile = a.take<ulonglong2>((n + SCAN_TILE_MIN - 1) / SCAN_TILE_MIN + 1);
    w.sc.d_hist = a.take<unsigned long long>((size_t)MAX_PASSES * RADIX);
    w.sc.d_base = a.take<unsigned long long>((size_t)MAX_PASSES * RADIX);
    w.sc.desc_bytes = sort_desc_bytes(n);
    w.sc.d_desc = a.take<char>(w.sc.desc_bytes);
    w.sc.d_err = a.take<unsigned>(64);
    w.sc.d_summary = a.take<unsigned long long>(8);
    w.sc.d_partials = a.take<unsigned long long>(((size_t)(n / 2048) + 8192) * 4);
    // digit bytes between the three-kernel passes of the sorts of 64-bit words (engine.hpp: dispatch_pass3): one byte per record of the largest sort
    // that has no such array of its own -- the first sort of a repetitive text, the sorts of the refinement rounds: n records
    w.sc.d_dig = nullptr; w.sc.dig_cap = 0;
    // (not in the reduced-memory layout: a byte per record of room is 2.4 % fewer records per slab -- the 4 GiB tandem repeat 6.24 -> 6.36 s)
    if (sizeof(T) == 8 && !kn.no_digit_bytes && n <= (1ull << 32) && !diet) {
        w.sc.dig_cap = n;
        w.sc.d_dig = a.take<uint8_t>(((size_t)w.sc.dig_cap + 64 + 255) & ~(size_t)255);
    }
    return a.off;
}

inline int ensure_pinned(psacx_ctx* c, size_t bytes) {
    if (c->pinned_bytes >= bytes) return PSACX_OK;
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr; c->pinned_bytes = 0; c->pinned_dev = nullptr;
    PSACX_HIP(c, hipHostMalloc((void**)&c->pinned, bytes, hipHostMallocDefault));
    c->pinned_bytes = bytes;
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, c->pinned, 0) == hipSuccess) c->pinned_dev = static_cast<char*>(dp);
    else (void)hipGetLastError();
    return PSACX_OK;
}

// alphabet.hpp:147-164 on the host from the device histogram
// tab gets codes 0..sigma-1 in byte order (the packed sort key needs no end-marker code);
// bits is psac's bits_per_char = ceil(log2(sigma + 1)), bits_packed = max(1, ceil(log2(sigma))).
inline void build_alphabet(const unsigned long long* hist, CodeTable& tab, uint32_t& sigma, uint32_t& bits,
                           uint32_t& bits_packed) {
    uint16_t next = 0;
    for (int ch = 0; ch < 256; ++ch) tab.c[ch] = hist[ch] ? next++ : (uint16_t)0;
    sigma = next;
    uint32_t b = 0;
    while ((1u << b) < sigma + 1u) ++b;          // ceil(log2(sigma + 1))
    bits = b;
    b = 0;
    while ((1u << b) < sigma) ++b;               // ceil(log2(sigma))
    bits_packed = b ? b : 1;
}

// kmer.hpp:26-40 for a single rank
inline uint32_t choose_k(uint32_t word_bits, uint32_t l, uint64_t n, uint32_t k) {
    const uint32_t max_k = word_bits / l;
    if (k == 0 || k > max_k) k = max_k;
    if ((uint64_t)k >= n) { k = (uint32_t)n; if (k > 1) --k; }
    return k;
}

// exclusive max-scan of the per-tile carries (in chunks when there are many tiles: sa_kernels.hpp: chunk_scan_kernel)
template <typename T>
int scan_carries(psacx_ctx* c, Work<T>& w, uint64_t ntiles) {
    const uint64_t nch = (ntiles + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (ntiles >= 8 * SCAN_CHUNK && nch <= SCAN_CHUNKS) {
        hipLaunchKernelGGL((chunk_scan_kernel<1024, OpMax>), dim3((unsigned)nch, 1), dim3(1024), 0, c->stream, w.d_carry, w.d_carry, ntiles, OpMax(), (uint64_t)0, w.d_chunks, SCAN_CHUNKS);
        hipLaunchKernelGGL((tile_scan_kernel<1024, OpMax>), dim3(1), dim3(1024), 0, c->stream, w.d_chunks, nch, OpMax(), (uint64_t)0, (uint64_t*)nullptr);
        hipLaunchKernelGGL((chunk_add_kernel<1024, OpMax>), dim3((unsigned)nch, 1), dim3(1024), 0, c->stream, w.d_carry, w.d_carry, ntiles, OpMax(), (const uint64_t*)w.d_chunks, SCAN_CHUNKS);
    } else
        hipLaunchKernelGGL((tile_scan_kernel<1024, OpMax>), dim3(1), dim3(1024), 0, c->stream, w.d_carry, ntiles, OpMax(), (uint64_t)0, (uint64_t*)nullptr);
    PSACX_HIP(c, hipGetLastError());
    return PSACX_OK;
}

// per-tile carries of the prefix-max: last head of every tile, then an exclusive max-scan
template <typename T, bool REFINE, bool GSA = false>
int run_carries(psacx_ctx* c, Work<T>& w, const T* a1, const T* a2, const T* pos, uint64_t cnt, const T* sa,
                KeyShape ks, const HeavyView<T>* hv = nullptr) {
    // hv (refinement rounds): the sorted records of a split round, read through the view (heavy_keys.hpp)
    const uint64_t ntiles = (cnt + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
    if (REFINE && !GSA && hv)
        hipLaunchKernelGGL((last_head_kernel<T, REFINE, false, true>), dim3((unsigned)ntiles), dim3(256), 0, c->stream,
                           (const T*)nullptr, (const T*)nullptr, pos, cnt, (unsigned)ScanCfg<T>::TILE, ntiles, w.d_carry, sa, ks, cnt, Boundary<T>(), *hv);
    else
    hipLaunchKernelGGL((last_head_kernel<T, REFINE, GSA>), dim3((unsigned)(REFINE ? ntiles : (ntiles + 3) / 4)), dim3(256), 0, c->stream,
                       a1, a2, pos, cnt, (unsigned)ScanCfg<T>::TILE, ntiles, w.d_carry, sa, ks, cnt, Boundary<T>());
    PSACX_HIP(c, hipGetLastError());
    PSACX_TRY(scan_carries<T>(c, w, ntiles));
    return PSACX_OK;
}

// totals of the per-tile activity counts, then the compacted list of still-active positions
template <typename T>
int run_compact(psacx_ctx* c, Work<T>& w, const T* ids, const T* pos_in, uint64_t cnt, T* pos_out,
                uint64_t* active, uint64_t* unf_buckets, uint64_t capacity, unsigned shift = 0,
                const T* payload = nullptr, T* out_id = nullptr, T* out_payload = nullptr, uint64_t pos_off = 0, bool fill_lazy_ids = false,
                uint32_t* ord_out = nullptr, bool* payload32 = nullptr) {
    // ord_out: the list entries' bucket numbers counted from 0 (needs the per-tile counts of unfinished buckets in w.d_nunf)
    // payload32 (in: allowed; out: done): the emitted payloads as 32-bit entries when the list is long enough for the three-kernel sort
    // pos_off: SA position of ids[0] when pos_in is null (a slab of the reduced-memory layout)
    uint64_t* h_cnt = reinterpret_cast<uint64_t*>(c->pinned);   // [2]
    const uint64_t ntiles = (cnt + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
    {
        ProfScope ps(c, TC_COMPACT);
        // the kernel stores the two totals into the pinned host words itself when the device can address them
        const uint64_t nch = (ntiles + SCAN_CHUNK - 1) / SCAN_CHUNK;
        uint64_t* const host_tot = c->pinned_dev ? reinterpret_cast<uint64_t*>(c->pinned_dev) : (uint64_t*)nullptr;
        if (ntiles >= 8 * SCAN_CHUNK && nch <= SCAN_CHUNKS) {
            hipLaunchKernelGGL((chunk_scan_kernel<1024, OpSum>), dim3((unsigned)nch, 2), dim3(1024), 0, c->stream, w.d_nact, w.d_nunf, ntiles, OpSum(), (uint64_t)0, w.d_chunks, SCAN_CHUNKS);
            hipLaunchKernelGGL((tile_scan2_kernel<1024>), dim3(2), dim3(1024), 0, c->stream, w.d_chunks, w.d_chunks + SCAN_CHUNKS, nch, w.d_totals, host_tot);
            hipLaunchKernelGGL((chunk_add_kernel<1024, OpSum>), dim3((unsigned)nch, 2), dim3(1024), 0, c->stream, w.d_nact, w.d_nunf, ntiles, OpSum(), (const uint64_t*)w.d_chunks, SCAN_CHUNKS);
        } else
        hipLaunchKernelGGL((tile_scan2_kernel<1024>), dim3(2), dim3(1024), 0, c->stream, w.d_nact, w.d_nunf, ntiles, w.d_totals, host_tot);
        PSACX_HIP(c, hipGetLastError());
    }
    if (!c->pinned_dev) PSACX_HIP(c, hipMemcpyAsync(h_cnt, w.d_totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    *active = h_cnt[0];
    *unf_buckets = h_cnt[1];
    if (fill_lazy_ids && *active > 0) {
        // (rebucket_first_kernel left the ids of the tiles without unresolved suffixes unwritten: somebody is going to read them now)
        ProfScope ps(c, TC_COMPACT);
        hipLaunchKernelGGL((fill_resolved_ids_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0, c->stream,
                           const_cast<T*>(ids), cnt, (const uint64_t*)w.d_nact, (const uint64_t*)w.d_totals);
        PSACX_HIP(c, hipGetLastError());
    }
    if (*active > 0 && *active <= capacity) {
        ProfScope ps(c, TC_COMPACT);
        const bool p32 = payload32 && *payload32 && sizeof(T) == 8 && *active >= SMALL_SORT_MAX;
        if (payload32) *payload32 = p32;
        if (payload)
            hipLaunchKernelGGL((compact_active_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, true>), dim3((unsigned)ntiles),
                               dim3(ScanCfg<T>::BLOCK), 0, c->stream, ids, pos_in, cnt, pos_out, w.d_nact, pos_off, (T)0, (T)0, shift,
                               payload, out_id, out_payload, (const uint64_t*)w.d_nunf, ord_out, p32 ? 1 : 0);
        else
            hipLaunchKernelGGL((compact_active_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS>), dim3((unsigned)ntiles),
                               dim3(ScanCfg<T>::BLOCK), 0, c->stream, ids, pos_in, cnt, pos_out, w.d_nact, pos_off, (T)0, (T)0, shift,
                               (const T*)nullptr, (T*)nullptr, (T*)nullptr, (const uint64_t*)w.d_nunf, ord_out);
        PSACX_HIP(c, hipGetLastError());
    }
    else if (payload32) *payload32 = false;
    return PSACX_OK;
}

// whether the inversion of n records of T runs its partition levels as radix passes (see invert_permutation)
template <typename T>
inline bool isa_radix_levels(uint64_t n, const Knobs& kn) {
    return sizeof(T) == 4 && n >= (1ull << 22) && n <= (1ull << 30);
}

// ISA[SA[i]] = val[i] - 1 for a full permutation SA (bulk_permute.hpp:14-73).  Large inputs go
// through destination-partition passes + an LDS window scatter (see partition_pairs_kernel);
// t1/t2 are two scratch pair buffers of n entries each.
// koff: the keys are a permutation of [koff, koff + n) (a rank's block in the distributed path)
// sc != nullptr (single-GPU engine, 32-bit words): the partition levels are run as stable LSD passes of
// the two-word radix scatter kernel instead (deterministic tile order keeps neighbouring runs in one
// XCD's L2: 1.3 ms per level including its tile histogram, against 1.5 ms for the reservation kernel;
// with 64-bit words the reservation kernel is the faster one).  Letting the first level make up its payload
// (ISA[SA[i]] = i, members of unresolved buckets repaired afterwards) was measured: the pass is not
// read-bound, no gain.
// 64-bit words, at most 2^32 positions: the inversion moves 32-bit (position, rank) pairs through 512-way partition
// levels down to windows of 2^14 positions.  Returns the number of levels (0: the form does not apply).
// buckets left unresolved, summed over the tiles of the first round's rebucket kernel (one workgroup)
template <int DUMMY>
__global__ __launch_bounds__(1024) void sum_counts_kernel(const uint64_t* __restrict__ per_tile, uint64_t ntiles, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long part[16];
    unsigned long long s = 0;
    for (uint64_t i = threadIdx.x; i < ntiles; i += 1024) s += per_tile[i];
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int i = 0; i < 16; ++i) t += part[i]; *out = t; }
}

template <typename T>
inline int isa_narrow_levels(uint64_t n, const Knobs& kn) {
    if (sizeof(T) != 8 || n < (1ull << 22) || n > (1ull << 32)) return 0;
    const unsigned idx_bits = bits_for(n - 1);
    return (int)((idx_bits - ISA_NARROW_WB + ISA_NARROW_CB - 1) / ISA_NARROW_CB);
}
inline unsigned isa_narrow_shift(int levels, int lv) { return ISA_NARROW_WB + ISA_NARROW_CB * (levels - 1 - lv); }
// the same levels for 32-bit words (their pairs are 32-bit anyway), used when rebucket_first_kernel runs the first one
inline int isa_levels32(uint64_t n) {
    if (n < (1ull << 22) || n > (1ull << 32)) return 0;
    return (int)((bits_for(n - 1) - ISA_NARROW_WB + ISA_NARROW_CB - 1) / ISA_NARROW_CB);
}

// Levels 1.. of the inversion on 32-bit (position, rank) pairs whose first level rebucket_first_kernel has run
// (pairs in k32 / v32, ranks already 0-based), then the window scatter into ISA.  ko / vo: a second pair of arrays.
template <typename T>
int finish_inversion32(psacx_ctx* c, unsigned* d_cursors, uint32_t* k32, uint32_t* v32, uint32_t* ko, uint32_t* vo, uint64_t n,
                       int levels, T* d_isa) {
    constexpr int PB = 512, PI = 16, WB = ISA_NARROW_WB, CB = ISA_NARROW_CB;
    const uint64_t ntiles = (n + PB * PI - 1) / (PB * PI);
    for (int lv = 1; lv < levels; ++lv) {
        const unsigned shift = isa_narrow_shift(levels, lv);
        PSACX_HIP(c, hipMemsetAsync(d_cursors, 0, ((size_t)(n >> shift) + 1) * sizeof(unsigned), c->stream));
        hipLaunchKernelGGL((partition_pairs_kernel<uint32_t, uint32_t, PB, PI, false, CB>), dim3((unsigned)ntiles), dim3(PB), 0, c->stream,
                           (const uint32_t*)k32, (const uint32_t*)v32, ko, vo, n, shift, d_cursors, (uint64_t)0);
        PSACX_HIP(c, hipGetLastError());
        std::swap(k32, ko); std::swap(v32, vo);
    }
    const uint64_t nwin = (n + (1ull << WB) - 1) >> WB;
    hipLaunchKernelGGL((window_scatter_kernel<uint32_t, T, 1024, false, WB>), dim3((unsigned)nwin), dim3(1024), 0, c->stream,
                       (const uint32_t*)k32, (const uint32_t*)v32, n, d_isa);
    PSACX_HIP(c, hipGetLastError());
    return PSACX_OK;
}

// fused_l1: the first narrow level has been run by rebucket_first_kernel (pairs in the two halves of t1.k1)
template <typename T>
int invert_permutation(psacx_ctx* c, unsigned* d_cursors, const T* d_sa, const T* val, uint64_t n, T* d_isa,
                       SortBufs<T> t1, SortBufs<T> t2, const Knobs& kn, uint64_t koff = 0, SortScratch* sc = nullptr,
                       bool have_hist0 = false, bool fused_l1 = false) {
    // have_hist0: the tile histograms of the first radix level are already in sc->d_desc (rebucket_first_kernel)
    constexpr int PB = 512, PI = 16;                      // 8192-record tiles: 32-record runs on average (1024 x 16, 32-bit destinations with staged class bytes, and cursors padded to their own cache lines all measured the same or worse)
    const unsigned idx_bits = bits_for(n - 1);
    if (n < (1ull << 22) || idx_bits > INV_WINDOW_BITS + 24) {
        hipLaunchKernelGGL((isa_scatter_kernel<T>), dim3(grid_for(c, n, 256, 16)), dim3(256), 0, c->stream, d_sa, val, n, d_isa, koff);
        PSACX_HIP(c, hipGetLastError());
        return PSACX_OK;
    }
    const int levels = (int)((idx_bits - INV_WINDOW_BITS + 7) / 8);
    const T* kin = d_sa; const T* vin = val;
    SortBufs<T> bufs[2] = {t1, t2};
    // (measured: 3.07 against 3.46 ms at 2^28, 18.7 against 19.4 ms at 2^30, 75.5 against 68.4 ms at 2^32 - 2)
    const bool radix_levels = sc && koff == 0 && isa_radix_levels<T>(n, kn);
    for (int lv = 0; radix_levels && lv < levels; ++lv) {
        SortBufs<T> o = bufs[lv & 1];
        PSACX_HIP(c, hipMemsetAsync(sc->d_desc, 0, 256, c->stream));
        dispatch_pass3<T>(c, kin, (const T*)nullptr, vin, o.k1, (T*)nullptr, o.k2, n, (int)(INV_WINDOW_BITS + 8 * lv),
                          sc->d_base + (size_t)lv * RADIX, sc->d_desc, 0, 0, lv == 0 && have_hist0);
        PSACX_HIP(c, hipGetLastError());
        c->stats.scatter_launches[2] += 1; c->stats.scatter_records[2] += n; c->stats.scatter_bytes[2] += 4ull * sizeof(T) * n;
        kin = o.k1; vin = o.k2;
    }
    // 64-bit words, at most 2^32 positions: the pairs are narrowed to 32 bits by the first partition level (sa_kernels.hpp)
    const bool narrow = !radix_levels && sizeof(T) == 8 && n <= (1ull << 32) && levels >= 1;
    if (fused_l1 && !(narrow && isa_narrow_levels<T>(n, kn) > 0)) { c->hip_err = "inversion: fused first level without the narrow form"; return PSACX_EINVAL; }
    if (narrow && isa_narrow_levels<T>(n, kn) > 0) {
        // 2^14-entry windows (64 KiB of 32-bit values in LDS) and 512-way levels: 2^32 positions need two partition
        // levels instead of three (24 + 16 + 16 = 56 instead of 72 bytes per record)
        constexpr int WB = ISA_NARROW_WB, CB = ISA_NARROW_CB;
        const uint64_t ntiles = (n + PB * PI - 1) / (PB * PI);
        const int lv9 = isa_narrow_levels<T>(n, kn);
        // packed pairs: one array of (position | rank << 32) entries per level (sa_kernels.hpp: partition_packed_kernel)
        uint64_t* pb[2] = {reinterpret_cast<uint64_t*>(t1.k1), reinterpret_cast<uint64_t*>(t2.k1)};
        const uint64_t* cur = fused_l1 ? pb[0] : nullptr;
        for (int lv = fused_l1 ? 1 : 0; lv < lv9; ++lv) {
            const unsigned shift = isa_narrow_shift(lv9, lv);
            PSACX_HIP(c, hipMemsetAsync(d_cursors, 0, ((size_t)(n >> shift) + 1) * sizeof(unsigned), c->stream));
            uint64_t* o = pb[lv & 1];
            if (lv == 0)
                hipLaunchKernelGGL((partition_packed_kernel<T, PB, PI, 1, CB>), dim3((unsigned)ntiles), dim3(PB), 0, c->stream, d_sa, val,
                                   (const uint64_t*)nullptr, o, n, shift, d_cursors, koff);
            else
                hipLaunchKernelGGL((partition_packed_kernel<T, PB, PI, 0, CB>), dim3((unsigned)ntiles), dim3(PB), 0, c->stream, (const T*)nullptr,
                                   (const T*)nullptr, cur, o, n, shift, d_cursors, (uint64_t)0);
            PSACX_HIP(c, hipGetLastError());
            cur = o;
        }
        if (!cur) { c->hip_err = "inversion: no partition level"; return PSACX_EINVAL; }
        const uint64_t nwin = (n + (1ull << WB) - 1) >> WB;
        hipLaunchKernelGGL((window_scatter_packed_kernel<T, 1024, WB>), dim3((unsigned)nwin), dim3(1024), 0, c->stream, cur, n, d_isa);
        PSACX_HIP(c, hipGetLastError());
        return PSACX_OK;
    }
    for (int lv = 0; !radix_levels && lv < levels; ++lv) {
        const unsigned shift = INV_WINDOW_BITS + 8 * (levels - 1 - lv);
        const size_t ncur = (size_t)(n >> shift) + 1;
        PSACX_HIP(c, hipMemsetAsync(d_cursors, 0, ncur * sizeof(unsigned), c->stream));
        SortBufs<T> o = bufs[lv & 1];
        const uint64_t ntiles = (n + PB * PI - 1) / (PB * PI);
        hipLaunchKernelGGL((partition_pairs_kernel<T, T, PB, PI>), dim3((unsigned)ntiles), dim3(PB), 0, c->stream, kin, vin,
                           o.k1, o.k2, n, shift, d_cursors, lv == 0 ? koff : (uint64_t)0);
        PSACX_HIP(c, hipGetLastError());
        kin = o.k1; vin = o.k2;
    }
    const uint64_t nwin = (n + (1ull << INV_WINDOW_BITS) - 1) >> INV_WINDOW_BITS;
    hipLaunchKernelGGL((window_scatter_kernel<T, T, 512>), dim3((unsigned)nwin), dim3(512), 0, c->stream, kin, vin, n, d_isa);
    PSACX_HIP(c, hipGetLastError());
    return PSACX_OK;
}

// The ISA entries of a refinement round -- (suffix | value << 32) pairs in list order, a subset of the positions -- taken to their
// places like the pairs of the whole inversion: 512-way partition levels by destination (the class regions of a level are filled as far
// as its cursors say), then stores that stay inside windows of 2^14 entries.  64-bit words, 2^22 .. 2^32 characters.  A round that is
// worked off in slabs runs the first level per slab (the class regions fill up slab by slab) and the rest once at its end: the ranks every
// slab reads are those of the round's start, as in psac's rounds, and every line of ISA is written once.
// lvl_a, lvl_b: two arrays of n entries (lvl_a must survive from the first slab to the end of the round); cursors: d_cursors
// (1024 + (n >> 14) + 1 entries).  One random 8-byte store per record instead: 217 ms per round of 2^32 records.
template <typename T> struct IsaLevels {
    static constexpr int PB = 512, PI = 16, WB = ISA_NARROW_WB, CB = ISA_NARROW_CB;
    psacx_ctx* c; unsigned* cursors; uint64_t n; int lv9; uint64_t* lvl_a; bool open;
    unsigned* c0() const { return cursors; }
    unsigned* c1() const { return cursors + 1024; }
    int begin(psacx_ctx* ctx, unsigned* d_cursors, uint64_t n_, uint64_t* a, const Knobs& kn) {
        c = ctx; cursors = d_cursors; n = n_; lvl_a = a; open = true;
        lv9 = isa_narrow_levels<T>(n, kn);
        if (lv9 < 1 || lv9 > 2) { c->hip_err = "ISA update by levels: text size out of range"; return PSACX_EINVAL; }
        PSACX_HIP(c, hipMemsetAsync(cursors, 0, (1024 + (size_t)(n >> WB) + 1) * sizeof(unsigned), c->stream));
        return PSACX_OK;
    }
    // skip (split rounds, heavy_keys.hpp): per scan tile of the list whether it lies inside a heavy run that keeps its rank -- tiles of such entries are left out
    int add(const uint64_t* pairs, uint64_t cnt, const ulonglong2* skip = nullptr) {
        static_assert((PB * PI) % ScanCfg<T>::TILE == 0 || sizeof(T) != 8, "a tile of pairs covers whole scan tiles");
        hipLaunchKernelGGL((partition_packed_kernel<T, PB, PI, 0, CB>), dim3((unsigned)((cnt + PB * PI - 1) / (PB * PI))), dim3(PB), 0, c->stream, (const T*)nullptr,
                           (const T*)nullptr, pairs, lvl_a, cnt, isa_narrow_shift(lv9, 0), lv9 == 1 ? c1() : c0(), (uint64_t)0, (const unsigned*)nullptr, 0u,
                           (const uint32_t*)nullptr, (uint64_t)0, (uint64_t)0, skip, skip ? (unsigned)((PB * PI) / ScanCfg<T>::TILE) : 0u);
        PSACX_HIP(c, hipGetLastError());
        return PSACX_OK;
    }
    int finish(T* d_isa, uint64_t* lvl_b) {
        open = false;
        const uint64_t* last = lvl_a;
        if (lv9 == 2) {
            hipLaunchKernelGGL((partition_packed_kernel<T, PB, PI, 0, CB>), dim3((unsigned)((n + PB * PI - 1) / (PB * PI))), dim3(PB), 0, c->stream, (const T*)nullptr,
                               (const T*)nullptr, (const uint64_t*)lvl_a, lvl_b, n, (unsigned)WB, c1(), (uint64_t)0, (const unsigned*)c0(), isa_narrow_shift(lv9, 0));
            PSACX_HIP(c, hipGetLastError());
            last = lvl_b;
        }
        hipLaunchKernelGGL((window_store_sparse_kernel<T, 1024, WB>), dim3((unsigned)((n + (1ull << WB) - 1) >> WB)), dim3(1024), 0, c->stream, last, (const unsigned*)c1(), n, d_isa);
        PSACX_HIP(c, hipGetLastError());
        return PSACX_OK;
    }
};

// The ranks h further of a refinement round (B2, suffix_array.hpp:972-996) fetched like the ISA entries are stored: the request of list entry j --
// (SA[pos[j]] + h | number of j's bucket << 32) -- goes through the same 512-way partition levels by text position (psac batches its requests by
// owner, bulk_rma.hpp:20-49; this is the one-GPU form of it), the window kernel reads ISA in whole lines and the requests leave it as the records
// of the round's sort, (number << kb2 | rank + 1) with the suffix as a 32-bit entry.  One random 8-byte fetch per record costs 27 - 30 ps on this
// part whatever the distance between neighbouring requests (tools/ubench_gather.hip: 36 G requests/s over 32 GiB, 38 G inside 64 MiB, 58 G inside
// 128 KiB -- the fetch is bound by the number of requests, not by HBM); the levels move 8-byte requests at the bandwidth of a copy.
// 64-bit words, 2^23 < n <= 2^32 (two levels), a list with its buckets' numbers beside it.  lvl_a, lvl_b: two arrays of n entries; keys: cnt
// words (may be lvl_a); v32: cnt 32-bit entries.
template <typename T>
int gather_by_levels(psacx_ctx* c, Work<T>& w, uint64_t n, uint64_t h, const T* plist, uint64_t cnt, const T* d_sa, const uint32_t* ord, const T* d_isa,
                     uint64_t* lvl_a, uint64_t* lvl_b, T* keys, uint32_t* v32, unsigned kb2, const Knobs& kn, unsigned* nblocks) {
    constexpr int PB = 512, PI = 16, WB = ISA_NARROW_WB, CB = ISA_NARROW_CB;
    if (isa_narrow_levels<T>(n, kn) != 2) { c->hip_err = "B2 fetch by levels: text size out of range"; return PSACX_EINVAL; }
    unsigned* const c0 = w.d_gcursors; unsigned* const c1 = w.d_gcursors + 1024;
    const uint64_t nwin = (n + (1ull << WB) - 1) >> WB;
    PSACX_HIP(c, hipMemsetAsync(c0, 0, (1024 + (size_t)nwin + 1) * sizeof(unsigned), c->stream));
    hipLaunchKernelGGL((partition_packed_kernel<T, PB, PI, 2, CB>), dim3((unsigned)((cnt + PB * PI - 1) / (PB * PI))), dim3(PB), 0, c->stream, d_sa, plist,
                       (const uint64_t*)nullptr, lvl_a, cnt, isa_narrow_shift(2, 0), c0, (uint64_t)0, (const unsigned*)nullptr, 0u, ord, h, n);
    hipLaunchKernelGGL((partition_packed_kernel<T, PB, PI, 0, CB>), dim3((unsigned)((n + PB * PI - 1) / (PB * PI))), dim3(PB), 0, c->stream, (const T*)nullptr,
                       (const T*)nullptr, (const uint64_t*)lvl_a, lvl_b, n, (unsigned)WB, c1, (uint64_t)0, (const unsigned*)c0, isa_narrow_shift(2, 0));
    hipLaunchKernelGGL(window_offsets_kernel<1024>, dim3(1), dim3(1024), 0, c->stream, (const unsigned*)c1, nwin, w.d_gwin);
    hipLaunchKernelGGL((window_gather_kernel<T, 1024, WB>), dim3((unsigned)nwin), dim3(1024), 0, c->stream, (const uint64_t*)lvl_b, (const unsigned*)c1,
                       (const uint64_t*)w.d_gwin, n, h, d_isa, kb2, keys, v32, w.sc.d_partials);
    PSACX_HIP(c, hipGetLastError());
    *nblocks = (unsigned)nwin;
    return PSACX_OK;
}

// gather_by_levels with the heavy / light split of heavy_keys.hpp: the requests go through the same two levels; the window kernel writes the
// light records (LK: keys, LV: suffixes as 32-bit entries, compacted in any order) and the heavy suffixes (HB, in their buckets' runs of a
// cnt-entry array in list order).  *light = the number of light records (the stream is drained for it).
inline HeavyTabs heavy_tabs(uint64_t* base, unsigned nb) {
    HeavyTabs ht;
    ht.bstart = base; base += nb + 1;
    ht.value = base; base += nb;
    ht.less = reinterpret_cast<unsigned long long*>(base); base += nb;
    ht.lstart = base; base += nb + 1;
    ht.rank = base; base += nb;
    ht.light = reinterpret_cast<unsigned long long*>(base); base += 8;           // (the reservation counters start on a 64-byte line: d_heavy is 256-byte aligned, nb words above are whole lines only by luck -- the padding keeps the counters apart from each other, which is what matters)
    ht.eq = reinterpret_cast<unsigned long long*>(base);
    return ht;
}
template <typename T>
int gather_heavy_by_levels(psacx_ctx* c, Work<T>& w, uint64_t n, uint64_t h, const T* plist, uint64_t cnt, const T* d_sa, const uint32_t* ord, unsigned nb,
                           const T* d_isa, uint64_t* lvl_a, uint64_t* lvl_b, T* LK, uint32_t* LV, uint32_t* HB, unsigned kb2, const Knobs& kn, HeavyTabs* tabs,
                           uint64_t* light, unsigned* nblocks) {
    constexpr int PB = 512, PI = 16, WB = ISA_NARROW_WB, CB = ISA_NARROW_CB;
    if (isa_narrow_levels<T>(n, kn) != 2 || nb == 0 || nb > HEAVY_MAXB) { c->hip_err = "heavy / light split: out of range"; return PSACX_EINVAL; }
    const HeavyTabs ht = heavy_tabs(w.d_heavy, nb);
    *tabs = ht;
    unsigned* const c0 = w.d_gcursors; unsigned* const c1 = w.d_gcursors + 1024;
    const uint64_t nwin = (n + (1ull << WB) - 1) >> WB;
    hipLaunchKernelGGL((heavy_probe_kernel<T>), dim3((nb + 1 + 255) / 256), dim3(256), 0, c->stream, ord, cnt, (uint32_t)nb, plist, d_sa, d_isa, n, h, ht);
    PSACX_HIP(c, hipMemsetAsync(c0, 0, (1024 + (size_t)nwin + 1) * sizeof(unsigned), c->stream));
    hipLaunchKernelGGL((partition_packed_kernel<T, PB, PI, 2, CB>), dim3((unsigned)((cnt + PB * PI - 1) / (PB * PI))), dim3(PB), 0, c->stream, d_sa, plist,
                       (const uint64_t*)nullptr, lvl_a, cnt, isa_narrow_shift(2, 0), c0, (uint64_t)0, (const unsigned*)nullptr, 0u, ord, h, n);
    hipLaunchKernelGGL((partition_packed_kernel<T, PB, PI, 0, CB>), dim3((unsigned)((n + PB * PI - 1) / (PB * PI))), dim3(PB), 0, c->stream, (const T*)nullptr,
                       (const T*)nullptr, (const uint64_t*)lvl_a, lvl_b, n, (unsigned)WB, c1, (uint64_t)0, (const unsigned*)c0, isa_narrow_shift(2, 0));
    hipLaunchKernelGGL((window_gather_heavy_kernel<T, 1024, WB>), dim3((unsigned)nwin), dim3(1024), (size_t)nb * 2 * sizeof(uint32_t), c->stream, (const uint64_t*)lvl_b,
                       (const unsigned*)c1, n, h, d_isa, kb2, (uint32_t)nb, ht, LK, LV, HB, w.sc.d_partials);
    uint64_t* const h_light = reinterpret_cast<uint64_t*>(c->pinned + 112);
    PSACX_HIP(c, hipGetLastError());
    PSACX_HIP(c, hipMemcpyAsync(h_light, ht.light, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    if (h_light[0] > cnt) { c->hip_err = "heavy / light split: more light records than records"; return PSACX_EDEVICE; }
    *light = h_light[0];
    *nblocks = (unsigned)nwin;
    return PSACX_OK;
}

// Range minima of a refinement round (suffix_array.hpp:1457-1476 issues one per freshly split boundary)
// read LCP values set in earlier rounds only, so per-group running minima can be tabulated once per
// round: a query then costs two loads per level instead of up to 126.  Levels >= 1 are tiny and always
// tabulated when the round has enough queries to matter; level 0 (two arrays of n entries in a lazily
// allocated second workspace) only when a large part of the suffixes is still active.
template <typename T>
int prepare_range_min(psacx_ctx* c, Work<T>& w, uint64_t queries, uint64_t n, const Knobs& kn) {
    for (int L = 0; L < PYR_MAX; ++L) { w.pyr.pre[L] = nullptr; w.pyr.suf[L] = nullptr; }
    if (queries < (1u << 16)) return PSACX_OK;
    ProfScope ps(c, TC_RMQ_BUILD);
    for (int L = 1; L + 1 < w.pyr.nlev; ++L) {
        hipLaunchKernelGGL((pyramid_aux_kernel<T>), dim3(grid_for(c, w.pyr.len[L], 256, 8)), dim3(256), 0, c->stream,
                           w.pyr.lvl[L], w.pyr.len[L], w.aux_pre[L], w.aux_suf[L]);
        PSACX_HIP(c, hipGetLastError());
        w.pyr.pre[L] = w.aux_pre[L]; w.pyr.suf[L] = w.aux_suf[L];
    }
    if (queries >= n / 32 && !w.diet && w.pyr.nlev > 1) {
        const size_t need = 2 * n * sizeof(T);
        if (c->aux_bytes < need) {
            if (c->aux) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->aux); c->aux = nullptr; c->aux_bytes = 0; }
            if (hipMalloc((void**)&c->aux, need) == hipSuccess) c->aux_bytes = need;
            else { (void)hipGetLastError(); c->aux = nullptr; }
        }
        if (c->aux) {
            T* pre0 = reinterpret_cast<T*>(c->aux);
            T* suf0 = pre0 + n;
            hipLaunchKernelGGL((pyramid_aux_kernel<T>), dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream,
                               w.pyr.lvl[0], n, pre0, suf0);
            PSACX_HIP(c, hipGetLastError());
            w.pyr.pre[0] = pre0; w.pyr.suf[0] = suf0;
        }
    }
    return PSACX_OK;
}

// rebucket_first_kernel with the first level of the inversion fused in
template <typename T, bool WITH_LCP>
inline void launch_rebucket_first_fused(psacx_ctx* c, unsigned ntiles, const T* s1, const T* s2, const T* sa, uint64_t n, KeyShape ks,
                                        T* bsa, T* lcp, uint64_t* carry, uint64_t* nact, uint64_t* nunf, T* pyr1,
                                        uint32_t* pk, uint32_t* pv, unsigned shift, unsigned* cursors, int lazy_ids = 0) {
    // pv == nullptr: packed pairs (64-bit words: one array of (position | rank << 32) entries at pk)
    if (pv)
        hipLaunchKernelGGL((rebucket_first_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, WITH_LCP, false, ISA_NARROW_CB>), dim3(ntiles),
                           dim3(ScanCfg<T>::BLOCK), 0, c->stream, s1, s2, sa, n, ks, bsa, lcp, carry, nact, nunf, n, Boundary<T>(), pyr1,
                           (unsigned*)nullptr, 0, pk, pv, shift, cursors, (T*)nullptr, lazy_ids);
    else
        hipLaunchKernelGGL((rebucket_first_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, WITH_LCP, false, ISA_NARROW_CB, true>), dim3(ntiles),
                           dim3(ScanCfg<T>::BLOCK), 0, c->stream, s1, s2, sa, n, ks, bsa, lcp, carry, nact, nunf, n, Boundary<T>(), pyr1,
                           (unsigned*)nullptr, 0, pk, pv, shift, cursors, (T*)nullptr, lazy_ids);
}

// d_slen != nullptr: generalized suffix array of a string set (construct_ss, suffix_array.hpp:267-363);
// d_text holds the strings back to back, d_slen[i] the characters from i to the end of its string.
template <typename T, bool WITH_LCP>
int construct_dev(psacx_ctx* c, const uint8_t* d_text, uint64_t n, uint32_t k_req, uint32_t flags,
                  T* d_sa, T* d_isa, T* d_lcp, const T* d_slen = nullptr) {
    const bool gsa = d_slen != nullptr;
    const bool no_fast = (flags & PSACX_NO_FAST) != 0;
    const Knobs kn = c->knobs;
    psacx_stats& st = c->stats;
    PSACX_TRY(ensure_pinned(c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 4096));

    if (n == 1) {
        PSACX_HIP(c, hipMemsetAsync(d_sa, 0, sizeof(T), c->stream));
        PSACX_HIP(c, hipMemsetAsync(d_isa, 0, sizeof(T), c->stream));
        if (WITH_LCP) PSACX_HIP(c, hipMemsetAsync(d_lcp, 0, sizeof(T), c->stream));
        PSACX_HIP(c, hipStreamSynchronize(c->stream));
        st.sigma = 1; st.bits_per_char = 1; st.k = 1;
        return PSACX_OK;
    }

    // workspace: the 9 n w layout when it fits, otherwise the diet layout (see carve)
    Work<T> w;
    bool diet = false;
    uint64_t cap = n;
    {
        Arena dry(nullptr);
        carve<T>(dry, w, n, WITH_LCP, d_lcp, false, n, d_sa, d_isa, kn);
        size_t need = dry.off + 4096;
        size_t free_b = 0, total_b = 0;
        PSACX_HIP(c, hipMemGetInfo(&free_b, &total_b));
        const size_t margin = (size_t)512 << 20;
        // the device copies an earlier host-pointer call left in the ctx (construct_host keeps them between calls) are
        // reclaimable when this call does not run on them: they go back before the reduced-memory layout is chosen
        const bool io_idle = c->io && !(reinterpret_cast<const char*>(d_sa) >= c->io && reinterpret_cast<const char*>(d_sa) < c->io + c->io_bytes);
        if (io_idle && need > c->slab_bytes && need + margin > free_b + c->slab_bytes) {
            PSACX_HIP(c, hipStreamSynchronize(c->stream));
            PSACX_HIP(c, hipFree(c->io));
            c->io = nullptr; c->io_bytes = 0;
            PSACX_HIP(c, hipMemGetInfo(&free_b, &total_b));
        }
        const size_t avail = free_b + c->slab_bytes > margin ? free_b + c->slab_bytes - margin : 0;
        if ((kn.force_diet && !no_fast) || (need > c->slab_bytes && need > avail)) {
            Arena d0(nullptr);
            carve<T>(d0, w, n, WITH_LCP, d_lcp, true, 0, d_sa, d_isa, kn);
            const size_t base = d0.off + 8192;
            if (no_fast || base >= avail) {
                c->hip_err = "workspace does not fit in HBM";
                return PSACX_ENOMEM;
            }
            cap = std::min<uint64_t>(n, (avail - base) / (5 * sizeof(T)) > 4096 ? (avail - base) / (5 * sizeof(T)) - 4096 : 0);
            if (kn.diet_cap) cap = std::min<uint64_t>(cap, kn.diet_cap);
            if (cap < std::min<uint64_t>(n, 1024)) { c->hip_err = "workspace does not fit in HBM"; return PSACX_ENOMEM; }
            diet = true;
            Arena d1(nullptr);
            carve<T>(d1, w, n, WITH_LCP, d_lcp, true, cap, d_sa, d_isa, kn);
            need = d1.off + 4096;
        }
        PSACX_TRY(ensure_slab(c, need));
    }
    Arena ar(c->slab);
    carve<T>(ar, w, n, WITH_LCP, d_lcp, diet, cap, d_sa, d_isa, kn);
    st.workspace_bytes = c->slab_bytes;
    w.sc.h_hist = reinterpret_cast<unsigned long long*>(c->pinned + 1024);
    w.sc.h_base = w.sc.h_hist + (size_t)MAX_PASSES * RADIX;
    w.sc.h_summary = reinterpret_cast<unsigned long long*>(c->pinned + 256);
    PSACX_HIP(c, hipMemsetAsync(w.sc.d_err, 0, 64 * sizeof(unsigned), c->stream));

    ProfScope* total = new ProfScope(c, TC_TOTAL);
    struct TotalGuard { ProfScope*& p; ~TotalGuard() { delete p; p = nullptr; } } tg{total};

    // ---- alphabet (alphabet.hpp:213-218) and k (kmer.hpp:26-40)
    CodeTable tab;
    {
        ProfScope ps(c, TC_ALPHABET);
        PSACX_HIP(c, hipMemsetAsync(w.d_hist256, 0, 256 * sizeof(unsigned long long), c->stream));
        hipLaunchKernelGGL((char_hist_kernel<256>), dim3(grid_for(c, n / 16 + 1, 256, 8)), dim3(256), 0, c->stream,
                           d_text, n, w.d_hist256);
        PSACX_HIP(c, hipGetLastError());
    }
    unsigned long long* h_hist = reinterpret_cast<unsigned long long*>(c->pinned + 1024);
    PSACX_HIP(c, hipMemcpyAsync(h_hist, w.d_hist256, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    uint32_t lc = 1;
    build_alphabet(h_hist, tab, st.sigma, st.bits_per_char, lc);
    const uint32_t l = st.bits_per_char;
    const uint32_t k = choose_k((uint32_t)sizeof(T) * 8, l, n, k_req);
    st.k = k;
    // the 2k-character window of the first round, packed with lc bits per character
    KeyShape ks;
    if (gsa) {
        // string ends need their own code in the key: psac's codes 1..sigma with l bits, 0 = end
        for (int ch = 0; ch < 256; ++ch) if (h_hist[ch]) tab.c[ch] = (uint16_t)(tab.c[ch] + 1);
        lc = l;
    }
    ks.lc = lc;
    ks.c1 = std::min<uint32_t>(2 * k, (uint32_t)(sizeof(T) * 8) / lc);
    ks.c2 = 2 * k - ks.c1;
    ks.spec = gsa ? 0 : std::min<uint64_t>(2ull * k - 1, n);

    // ---- first rank-pair sort (idxsort.hpp:23-83); payload = text position -> SA.
    // Two stages when the leading bits of word 1 already separate almost every suffix (random DNA,
    // 32-bit words: 4^16 windows for 2^28 suffixes; 64-bit words: the top 40 bits for 2^32): stage 1
    // sorts (word 1, suffix) pairs on those bits only -- two thirds of the bytes per pass and far
    // fewer passes -- and stage 2 orders the suffixes that still tie by the full window.
    const unsigned bits_w1 = ks.c1 * lc, bits_w2 = ks.c2 * lc;
    unsigned lead = bits_for(n - 1) + 3;                      // leading bits stage 1 sorts on (ties: ~ n / 2^lead)
    lead = (lead + RADIX_BITS - 1) / RADIX_BITS * RADIX_BITS;
    // word 1 slightly too short for that (DNA, 32-bit words, 2^29 < n <= 2^30): all of word 1 still leaves
    // fewer than a quarter of the suffixes tied, which is cheaper than carrying word 2 through five passes
    const unsigned slack = 2;
    if (lead > bits_w1 && bits_for(n - 1) + slack <= bits_w1 && bits_w1 % RADIX_BITS == 0) lead = bits_w1;
    bool two_stage = !gsa && n >= (1ull << 21) && !kn.one_stage && lead <= bits_w1 &&
                     lead + RADIX_BITS <= bits_w1 + bits_w2;     // at least one pass less
    psacx_round* r0 = &st.rounds[0];
    SortBufs<T> sorted;
    bool onew_recs = false;          // the sorted records of the first round are still one-word (sa_kernels.hpp: OneWordView); sorted.k1 holds them
    OneWordView onew_view = OneWordView();
    T* onew_w1 = nullptr;            // ... and word 1 of the suffixes that tie on the leading bits is here
    // (second attempt: only when the two-stage form met more ties than the reduced-memory layout has room for)
    for (int attempt = 0; attempt < 2; ++attempt) {
    bool retry_one_stage = false;
    const unsigned lo1 = two_stage ? bits_w1 - lead : 0;
    const bool hist_in_keys = two_stage;
    // one-word records, most significant digit first (engine.hpp: prefix_sort_1w): 64-bit words, suffixes below 2^32, the
    // prefix without its top digit in 32 bits
    bool one_word = two_stage && hist_in_keys && !gsa && sizeof(T) == 8 && n <= (1ull << 32) && n >= (1ull << kn.one_word_min) && lead >= 3 * RADIX_BITS &&
                    lead <= 32 + RADIX_BITS && lead % RADIX_BITS == 0 && !kn.no_one_word && attempt == 0;

    // In the diet layout the second record set is the output buffers (y = ISA, LCP, SA).  One stage: both sorted key
    // words must end up in the workspace set x (word 2 in the LCP buffer would be overwritten while its neighbours are
    // still being read), so an odd number of passes starts from y.  Two stages: only word 1 and the suffix are sorted;
    // word 1 may just as well end up in the ISA buffer (it is dead before the inversion writes ISA), and then the suffixes
    // land in the SA buffer itself instead of being copied there (n w bytes read + written: 13 ms of a 4 GiB / uint64
    // construction) -- so an even number of passes starts from y.
    const unsigned planned = two_stage ? lead / RADIX_BITS
                                       : (bits_w1 + RADIX_BITS - 1) / RADIX_BITS + (bits_w2 + RADIX_BITS - 1) / RADIX_BITS;
    const bool start_y = w.diet && (two_stage ? !(planned & 1u) : (planned & 1u) != 0);
    SortBufs<T> first_in = start_y ? w.y : w.x, first_alt = start_y ? w.x : w.y;
    // word 2 of every record is sorted along in one stage; with two stages it is not kept at all
    // (the few suffixes that need it read it from the text)
    T* const k2rec = two_stage ? (T*)nullptr : first_in.k2;

    // ---- first-round keys: the 2k-character window at every position, packed (kmer.hpp:119-177,
    //      shifting.hpp:33-122; see key_pairs_kernel for the packing)
    // (the one-word prefix sort computes word 1 inside its pass on the top digit: no keys in memory unless it has to give up)
    auto make_keys = [&](bool with_hist) -> int {
        ProfScope ps(c, TC_KMER);
        constexpr int KB = 256, KI = 8;
        uint64_t nb = (n + KB * KI - 1) / (KB * KI);
        if (gsa)
            hipLaunchKernelGGL((key_pairs_kernel<T, KB, KI, true>), dim3((unsigned)nb), dim3(KB), 0, c->stream, d_text, n, n,
                               tab, ks, first_in.k1, k2rec, w.sc.d_partials, d_slen);
        else if (with_hist) {
            // tile shape of the stage-1 sort, pass-1 histograms written on the way
            constexpr int HB = ScatterCfg<T>::BLOCK, HI = ScatterCfg<T>::ITEMS;
            nb = (n + HB * HI - 1) / (HB * HI);
            hipLaunchKernelGGL((key_pairs_kernel<T, HB, HI, false, true>), dim3((unsigned)nb), dim3(HB), 0, c->stream, d_text, n, n,
                               tab, ks, first_in.k1, k2rec, w.sc.d_partials, (const T*)nullptr,
                               reinterpret_cast<unsigned*>(w.sc.d_desc + 256), (int)lo1);
        } else
            hipLaunchKernelGGL((key_pairs_kernel<T, KB, KI>), dim3((unsigned)nb), dim3(KB), 0, c->stream, d_text, n, n,
                               tab, ks, first_in.k1, k2rec, w.sc.d_partials, (const T*)nullptr);
        PSACX_HIP(c, hipGetLastError());
        PSACX_TRY(summary_finish(c, w.sc, (unsigned)nb));
        return PSACX_OK;
    };
    if (!one_word) PSACX_TRY(make_keys(hist_in_keys));

    std::memset(r0, 0, sizeof(*r0));
    if (two_stage) {
        SortBufs<T> in1{first_in.k1, nullptr, first_in.v}, alt1{first_alt.k1, nullptr, first_alt.v};
        bool packed1 = false;            // the one-word sort ran: word 1 comes back without its bits below the prefix
        uint8_t* tie_bytes = nullptr;    // ... and left the lowest byte of every record's prefix bits here (engine.hpp: onew_bucket_passes)
        onew_recs = false;
        if constexpr (sizeof(T) == 8) {
            if (one_word) {
                uint64_t* s1 = nullptr;
                // the records stay one-word through the tie stage and the rebucket kernel when that kernel runs in its fused form
                const bool keep = !kn.widen_last && !kn.ties_radix && isa_narrow_levels<T>(n, kn) > 0;
                // (the digit bytes between the passes live in the array that takes word 2 of the tied suffixes after the sort: idle until then)
                uint8_t* const dig = kn.no_digit_bytes ? (uint8_t*)nullptr : reinterpret_cast<uint8_t*>(w.diet ? w.x.k2 : first_alt.k2);
                // (... and the bytes the tie stage looks for its groups in live in the other second-word array, which nobody writes before the rebucket kernel)
                tie_bytes = (keep && !kn.no_digit_bytes) ? reinterpret_cast<uint8_t*>(w.diet ? w.y.k2 : first_in.k2) : (uint8_t*)nullptr;
                const int rc1 = prefix_sort_1w(c, w.sc, reinterpret_cast<uint64_t*>(in1.k1), reinterpret_cast<uint64_t*>(alt1.k1),
                                               reinterpret_cast<uint64_t*>(d_sa), n, lo1, lead, r0, &s1, d_text, n, tab, ks, !kn.one_word_always,
                                               keep ? &onew_view : nullptr, dig,
                                               (in1.k1 == w.x.k1 && onew_pad_total<T>(n)) ? ONEW_PAD : (uint64_t)0, tie_bytes);
                if (rc1 == PSACX_RETRY_1STAGE) {          // (nearly every suffix ties on the prefix: one sort over both words; nothing was written)
                    two_stage = false; retry_one_stage = true; one_word = false;
                } else if (rc1 == PSACX_RETRY_1W) {          // (a repetitive text, or no room for the bucket tables: nothing was written)
                    one_word = false;
                    PSACX_TRY(make_keys(hist_in_keys));
                }
                else {
                    PSACX_TRY(rc1);
                    sorted.k1 = reinterpret_cast<T*>(s1); sorted.k2 = nullptr; sorted.v = d_sa;
                    packed1 = true;          // (the bits of word 1 below the prefix are gone: ties read word 1 from the text)
                    onew_recs = keep;
                }
            }
        } else one_word = false;
        if (retry_one_stage) continue;
        if (!one_word)
        PSACX_TRY(pair_sort<T>(c, w.sc, in1, alt1, n, /*iota=*/true, bits_w1, 0, w.diet ? (T*)nullptr : d_sa, &sorted, r0,
                               ks.spec, n, /*summary_ready=*/true, lo1, hist_in_keys ? (int)lo1 : -1));
        if (w.diet && sorted.v != d_sa)          // (a skipped pass changed the parity)
            PSACX_HIP(c, hipMemcpyAsync(d_sa, sorted.v, n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        T* S1 = sorted.k1;
        T* const S2 = w.diet ? w.x.k2 : first_alt.k2;        // word 2 in sorted order, filled for the ties only
        T* free_k1 = (S1 == w.x.k1) ? w.y.k1 : w.x.k1;
        // stage 2, common case: all tie groups are tiny and get ordered in place
        // (tile shapes measured: 64-bit words 256 x 32: 10.7 ms at 2^32, 128 x 32: 11.1, 256 x 16: 13.0, 512 x 8: 16.6;
        //  32-bit words 256 x 16: 1.2-1.3 ms at 2^28, 128 x 32: 1.4)
        constexpr int TB = 256, TI = sizeof(T) == 8 ? 32 : 16, TG = 8;
        unsigned long long* d_big = reinterpret_cast<unsigned long long*>(w.d_totals + 2);
        unsigned long long* h_big = reinterpret_cast<unsigned long long*>(c->pinned + 64);
        {
            ProfScope ps(c, TC_GATHER);
            PSACX_HIP(c, hipMemsetAsync(d_big, 0, sizeof(unsigned long long), c->stream));
            const uint64_t nb = (n + (uint64_t)TB * TI - 1) / ((uint64_t)TB * TI);
            if constexpr (sizeof(T) == 8) {
                if (onew_recs)          // S1: the one-word records; the array the last pass did not write takes word 1 of the tied suffixes
                    hipLaunchKernelGGL((tie_resolve_1w_kernel<TB, TI, TG>), dim3((unsigned)nb), dim3(TB), 0, c->stream, reinterpret_cast<uint64_t*>(S1),
                                       reinterpret_cast<uint64_t*>(free_k1), reinterpret_cast<uint64_t*>(S2), n, onew_view, d_text, n, tab, ks, d_big,
                                       (const uint8_t*)tie_bytes);
            }
            if (!kn.ties_radix && !onew_recs)
                hipLaunchKernelGGL((tie_resolve_kernel<T, TB, TI, TG>), dim3((unsigned)nb), dim3(TB), 0, c->stream, S1, d_sa, S2, n, lo1,
                                   d_text, n, tab, ks, d_big, packed1);
            PSACX_HIP(c, hipGetLastError());
        }
        PSACX_HIP(c, hipMemcpyAsync(h_big, d_big, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        PSACX_HIP(c, hipStreamSynchronize(c->stream));
        if constexpr (sizeof(T) == 8) {
            if (onew_recs && *h_big) {
                // a long group: the radix path below works on word 1 and the suffixes as arrays -- written now, as the last pass would have
                ProfScope ps(c, TC_GATHER);
                hipLaunchKernelGGL(onew_widen_kernel<0>, dim3(grid_for(c, n, 256, 8)), dim3(256), 0, c->stream, reinterpret_cast<const uint64_t*>(S1), n, onew_view,
                                   reinterpret_cast<uint64_t*>(free_k1), reinterpret_cast<uint64_t*>(d_sa));
                PSACX_HIP(c, hipGetLastError());
                std::swap(S1, free_k1);
                sorted.k1 = S1;
                onew_recs = false;
            }
        }
        if (*h_big || kn.ties_radix) {
            // some group is long (repetitive text): compact all ties and radix-sort them by the full window
            const uint64_t ntiles = (n + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
            {
                ProfScope ps(c, TC_COMPACT);
                hipLaunchKernelGGL((count_active_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0,
                                   c->stream, S1, n, (T)0, (T)0, w.d_nact, lo1, w.d_nunf);
                PSACX_HIP(c, hipGetLastError());
            }
            // two record sets for the ties out of buffers that are idle until the rebucket step; the
            // compaction already fills word 1 and the suffix of set a
            SortBufs<T> a, b, s2;
            if (w.diet) { a = w.ry; b.k1 = w.bsa; b.k2 = w.x.v; b.v = free_k1; }
            else { a.k1 = free_k1; a.k2 = first_alt.v; a.v = first_in.v; b.k1 = w.bsa; b.k2 = w.pos_b; b.v = d_isa; }
            // (the tie groups counted from 0 beside the list, in the idle second key array of set b: the sort below takes the group's
            //  number in place of the sorted prefix)
            uint32_t* const tie_ord = reinterpret_cast<uint32_t*>(b.k2);
            uint64_t ties = 0, tie_groups = 0;
            bool tie_v32 = sizeof(T) == 8 && n <= (1ull << 32);         // (the suffixes of the ties as 32-bit entries through the sort)
            PSACX_TRY(run_compact<T>(c, w, S1, nullptr, n, w.pos_a, &ties, &tie_groups, w.cap_active, lo1, d_sa, a.k1, a.v, 0, false, tie_ord, &tie_v32));
            const unsigned ord_bits = bits_for(tie_groups > 1 ? tie_groups - 1 : 1);
            const bool by_ord = lo1 > 0 && ties > 0 && tie_groups > 0 && tie_groups < (1ull << 32) && lo1 + ord_bits + RADIX_BITS <= bits_w1;
            if (ties > w.cap_active) {
                // repetitive text in the reduced-memory layout: there is no room to sort all ties at once, so the
                // first round is run again as one sort over both words (the keys were sorted in place: rebuilt)
                two_stage = false; retry_one_stage = true;
                ties = 0;
            }
            if (ties) {
                {
                    ProfScope ps(c, TC_GATHER);
                    const int gg = grid_for(c, ties, 256, 16);
                    hipLaunchKernelGGL((gather_prefix_ties_kernel<T, 256>), dim3(gg), dim3(256), 0, c->stream, ties, a.k1, a.v,
                                       d_text, n, tab, ks, a.k2, w.sc.d_partials, packed1, by_ord ? (const uint32_t*)tie_ord : (const uint32_t*)nullptr, lo1, tie_v32 ? 1 : 0);
                    PSACX_HIP(c, hipGetLastError());
                    PSACX_TRY(summary_finish(c, w.sc, (unsigned)gg));
                }
                psacx_round r1;
                std::memset(&r1, 0, sizeof(r1));
                PSACX_TRY(pair_sort<T>(c, w.sc, a, b, ties, /*iota=*/false, by_ord ? lo1 + ord_bits : bits_w1, bits_w2, nullptr, &s2, &r1, 0, 0,
                                       /*summary_ready=*/true, 0, -1, /*v32_in=*/tie_v32));
                r0->sort_passes += r1.sort_passes; r0->sort_passes_skipped += r1.sort_passes_skipped;
                ProfScope ps(c, TC_GATHER);
                hipLaunchKernelGGL((scatter_prefix_ties_kernel<T>), dim3(grid_for(c, ties, 256, 16)), dim3(256), 0, c->stream,
                                   w.pos_a, ties, lo1 ? s2.k1 : (const T*)nullptr, s2.k2, s2.v, S1, S2, d_sa, by_ord ? lo1 : 0u);
                PSACX_HIP(c, hipGetLastError());
            }
        }
        sorted.k2 = S2; sorted.v = d_sa;
        onew_w1 = onew_recs ? free_k1 : (T*)nullptr;
        if (retry_one_stage) continue;
    } else {
        PSACX_TRY(pair_sort<T>(c, w.sc, first_in, first_alt, n, /*iota=*/true, bits_w1, bits_w2, w.diet ? (T*)nullptr : d_sa,
                               &sorted, r0, ks.spec, n, /*summary_ready=*/true));
        if (w.diet) {
            // keys into the workspace set if a skipped pass changed the parity; SA out of the scratch payload
            if (sorted.k1 != w.x.k1) {
                PSACX_HIP(c, hipMemcpyAsync(w.x.k1, sorted.k1, n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                PSACX_HIP(c, hipMemcpyAsync(w.x.k2, sorted.k2, n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                sorted.k1 = w.x.k1; sorted.k2 = w.x.k2;
            }
            if (sorted.v != d_sa) PSACX_HIP(c, hipMemcpyAsync(d_sa, sorted.v, n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
            sorted.v = d_sa;
        }
    }
    if (!retry_one_stage) break;
    }

    // ---- LCP of the 2k-mers + new bucket ids (suffix_array.hpp:1353-1396, bucketing.hpp:57-123)
    bool isa_hist_ready = false;
    const bool fuse_l1 = !gsa && isa_narrow_levels<T>(n, kn) > 0;
    bool lazy_ids = false;          // the rebucket kernel left out the ids of tiles without unresolved suffixes (filled in by run_compact if needed)
    // 32-bit words, normal layout: the same fusion; the pairs use two payload scratch arrays of the sort, the second level
    // the two position lists (all idle between the sort and the first compaction)
    const bool fuse32 = sizeof(T) == 4 && !gsa && !w.diet && isa_levels32(n) > 0;
    {
        ProfScope ps(c, TC_REBUCKET);
        const uint64_t ntiles = (n + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
        T* const pyr1 = (WITH_LCP && w.pyr.nlev > 1) ? w.pyr.lvl[1] : (T*)nullptr;   // level 1 comes out of the rebucket kernel
        // ... and so do the tile histograms of the inversion's first radix level when the tiles agree
        isa_hist_ready = !fuse32 && isa_radix_levels<T>(n, kn) && (uint64_t)ScanCfg<T>::TILE == (uint64_t)ScatterCfg<T>::TILE;
        unsigned* const sa_hist = isa_hist_ready ? reinterpret_cast<unsigned*>(w.sc.d_desc + 256) : (unsigned*)nullptr;
        if (gsa) {
            PSACX_TRY((run_carries<T, false, true>(c, w, sorted.k1, sorted.k2, nullptr, n, d_sa, ks)));
            hipLaunchKernelGGL((rebucket_first_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, WITH_LCP, true>), dim3((unsigned)ntiles),
                               dim3(ScanCfg<T>::BLOCK), 0, c->stream, sorted.k1, sorted.k2, d_sa, n, ks, w.bsa, d_lcp,
                               w.d_carry, w.d_nact, w.d_nunf, n, Boundary<T>(), pyr1, sa_hist, (int)INV_WINDOW_BITS);
        } else if (fuse32) {
            PSACX_TRY((run_carries<T, false>(c, w, sorted.k1, sorted.k2, nullptr, n, d_sa, ks)));
            PSACX_HIP(c, hipMemsetAsync(w.d_cursors, 0, ((size_t)1 << ISA_NARROW_CB) * sizeof(unsigned) + sizeof(unsigned), c->stream));
            launch_rebucket_first_fused<T, WITH_LCP>(c, (unsigned)ntiles, sorted.k1, sorted.k2, d_sa, n, ks, w.bsa, d_lcp, w.d_carry, w.d_nact,
                                                     w.d_nunf, pyr1, reinterpret_cast<uint32_t*>(w.x.v), reinterpret_cast<uint32_t*>(w.y.v),
                                                     isa_narrow_shift(isa_levels32(n), 0), w.d_cursors);
        } else if (fuse_l1 && onew_recs) {
            // ... on the one-word records of the prefix sort: the kernel takes word 1 and the suffixes out of them and writes the suffix array
            if constexpr (sizeof(T) == 8) {
                hipLaunchKernelGGL(last_head_1w_kernel<0>, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, c->stream, reinterpret_cast<const uint64_t*>(sorted.k1),
                                   reinterpret_cast<const uint64_t*>(onew_w1), reinterpret_cast<const uint64_t*>(sorted.k2), n, (unsigned)ScanCfg<T>::TILE, ntiles,
                                   w.d_carry, ks, onew_view);
                PSACX_HIP(c, hipGetLastError());
                PSACX_TRY(scan_carries<T>(c, w, ntiles));
                PSACX_HIP(c, hipMemsetAsync(w.d_cursors, 0, ((size_t)1 << ISA_NARROW_CB) * sizeof(unsigned) + sizeof(unsigned), c->stream));
                lazy_ids = n >= (1ull << 22);
                hipLaunchKernelGGL((rebucket_first_kernel<T, RB1W_BLOCK, ScanCfg<T>::TILE / RB1W_BLOCK, WITH_LCP, false, ISA_NARROW_CB, true, true>), dim3((unsigned)ntiles),
                                   dim3(RB1W_BLOCK), 0, c->stream, sorted.k1, sorted.k2, (const T*)nullptr, n, ks, w.bsa, d_lcp, w.d_carry, w.d_nact, w.d_nunf, n,
                                   Boundary<T>(), pyr1, (unsigned*)nullptr, 0, reinterpret_cast<uint32_t*>(w.x.v), (uint32_t*)nullptr,
                                   isa_narrow_shift(isa_narrow_levels<T>(n, kn), 0), w.d_cursors, d_sa, lazy_ids ? 1 : 0, onew_view, (const T*)onew_w1);
            }
        } else if (fuse_l1) {
            // the first level of the SA -> ISA inversion rides along: its (position, rank) pairs go to the payload scratch
            // array of the sort, which nobody reads any more (word 1 / word 2 / SA are read from other arrays)
            PSACX_TRY((run_carries<T, false>(c, w, sorted.k1, sorted.k2, nullptr, n, d_sa, ks)));
            PSACX_HIP(c, hipMemsetAsync(w.d_cursors, 0, ((size_t)1 << ISA_NARROW_CB) * sizeof(unsigned) + sizeof(unsigned), c->stream));
            uint32_t* const pk = reinterpret_cast<uint32_t*>(w.x.v);
            lazy_ids = n >= (1ull << 22);
            launch_rebucket_first_fused<T, WITH_LCP>(c, (unsigned)ntiles, sorted.k1, sorted.k2, d_sa, n, ks, w.bsa, d_lcp, w.d_carry, w.d_nact,
                                                     w.d_nunf, pyr1, pk, (uint32_t*)nullptr,
                                                     isa_narrow_shift(isa_narrow_levels<T>(n, kn), 0), w.d_cursors, lazy_ids ? 1 : 0);
        } else {
            PSACX_TRY((run_carries<T, false>(c, w, sorted.k1, sorted.k2, nullptr, n, d_sa, ks)));
            hipLaunchKernelGGL((rebucket_first_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, WITH_LCP>), dim3((unsigned)ntiles),
                               dim3(ScanCfg<T>::BLOCK), 0, c->stream, sorted.k1, sorted.k2, d_sa, n, ks, w.bsa, d_lcp,
                               w.d_carry, w.d_nact, w.d_nunf, n, Boundary<T>(), pyr1, sa_hist, (int)INV_WINDOW_BITS);
        }
        PSACX_HIP(c, hipGetLastError());
    }
    // (a host-pointer call starts SA and LCP on their way out from here: they are final unless refinement rounds follow)
    if (c->first_round_hook && !gsa && c->early_word) {
        const uint64_t ntiles = (n + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
        hipLaunchKernelGGL(sum_counts_kernel<0>, dim3(1), dim3(1024), 0, c->stream, (const uint64_t*)w.d_nunf, ntiles, c->early_word);
        PSACX_HIP(c, hipGetLastError());
        void (*hook)(void*) = c->first_round_hook; c->first_round_hook = nullptr; hook(c->first_round_hook_arg);
    }
    // ---- SA -> ISA (bulk_permute.hpp:14-73)
    {
        ProfScope ps(c, TC_ISA_SCATTER);
        SortBufs<T> t1 = w.x, t2 = w.y;
        if (w.diet) { t2.k1 = w.x.v; t2.k2 = d_isa; }       // the last partition level may write the values into ISA itself
        if (fuse_l1) { t1.k1 = w.x.v; t2.k1 = w.x.k1; }     // level 1 is in x.v; level 2 writes over word 1 (or its twin), dead by now
        if (fuse32)
            PSACX_TRY(finish_inversion32<T>(c, w.d_cursors, reinterpret_cast<uint32_t*>(w.x.v), reinterpret_cast<uint32_t*>(w.y.v),
                                            reinterpret_cast<uint32_t*>(w.pos_a), reinterpret_cast<uint32_t*>(w.pos_b), n, isa_levels32(n), d_isa));
        else
            PSACX_TRY(invert_permutation<T>(c, w.d_cursors, d_sa, w.bsa, n, d_isa, t1, t2, kn, 0, &w.sc, isa_hist_ready, fuse_l1));
    }
    if (WITH_LCP) {
        ProfScope ps(c, TC_RMQ_BUILD);
        for (int L = 2; L < w.pyr.nlev; ++L) {
            hipLaunchKernelGGL((pyramid_level_kernel<T>), dim3(grid_for(c, w.pyr.len[L] * 64, 256, 8)), dim3(256), 0,
                               c->stream, w.pyr.lvl[L - 1], w.pyr.len[L - 1], w.pyr.lvl[L], w.pyr.len[L]);
            PSACX_HIP(c, hipGetLastError());
        }
    }

    // ---- which suffixes still share a bucket (suffix_array.hpp:925-965)
    uint64_t active = 0, unf_b = 0;
    // (64-bit words, at most 2^32 characters: the list comes with its entries' bucket numbers counted from 0, in the idle upper half of the
    //  payload array -- the one-word sort keys of the refinement rounds then need only as many bits as there are buckets)
    // (ALIAS: ord_arr is the upper half of w.x.v.  x.v is also written in full by the last pass of a v32 sort, by shift_keys_kernel in whole
    //  rounds at n = 2^32, by three-word sorts and as level-1 ISA pairs by the fused rebucket kernel.  Every reader of ord_arr --
    //  gather_keys_kernel with by_ord -- runs directly after the run_compact that wrote it and before the round's sort touches x.v; keep it so.)
    uint32_t* const ord_arr = (sizeof(T) == 8 && n <= (1ull << 32)) ? reinterpret_cast<uint32_t*>(w.x.v) + n : (uint32_t*)nullptr;
    PSACX_TRY(run_compact<T>(c, w, w.bsa, nullptr, n, w.pos_a, &active, &unf_b, w.cap_active, 0, nullptr, nullptr, nullptr, 0, lazy_ids, ord_arr));
    r0->h = k; r0->active = n; r0->unfinished_buckets = unf_b; r0->unfinished_elements = active;
    st.n_rounds = 1;
    // The list of unresolved SA positions exists only while it fits the workspace.  In the reduced-memory layout a
    // round with more unresolved suffixes than `cap_active` is worked off in slabs of whole buckets (see below).
    bool have_list = active <= w.cap_active;

    T* pos = w.pos_a;
    T* pos_next = w.pos_b;
    const unsigned id_bits = bits_for(n);
    // one refinement pass over the `cnt` list entries plist (suffix_array.hpp:1092-1157 for one bucket range):
    // B2 = rank of the suffix h further, sort by (bucket, B2), new ids / LCP / ISA written in place
    // whole: the round takes all n suffixes in TEXT order (shift_keys_kernel) and ISA is rebuilt by inverting the new SA
    // (the destination-partition levels of the first round) instead of one random store per record
    IsaLevels<T> isa_lv; isa_lv.open = false;
    // nb_in: buckets in the list (0: not known, or the list has no bucket numbers beside it); list_out == nullptr: only the counters
    auto refine = [&](const T* plist, uint64_t cnt, uint64_t h, psacx_round* rr, T* list_out, uint64_t* nactive, uint64_t* nunf, bool whole = false,
                      uint64_t nb_in = 0) -> int {
        // 64-bit words, at most 2^32 characters: bucket number and rank h further share one word, the suffix is a 32-bit entry --
        // two-word records with a narrow payload (radix.hpp: NOKO, VN 1 / 2), 24 instead of 48 bytes per record and pass.  With a list
        // of unresolved positions the bucket numbers are dense (gather_keys_kernel: list index of the head, halved): 31 bits beside a
        // 33-bit rank at n = 2^32, and fewer digits to sort on whenever few suffixes are left.
        const bool dense = plist != nullptr && !whole;
        const unsigned kb2 = dense ? id_bits : 32u;
        const bool both = sizeof(T) == 8 && cnt >= SMALL_SORT_MAX && (n < (1ull << 32) || (n == (1ull << 32) && dense));
        const bool by_ord = dense && both && ord_arr && nb_in > 0;
        const unsigned num_bits = by_ord ? bits_for(nb_in > 1 ? nb_in - 1 : 1) : dense ? bits_for(cnt > 2 ? (cnt - 1) >> 1 : 1) : id_bits;
        T* const key2 = both ? (T*)nullptr : w.x.k2;
        // large rounds: the ranks h further come through partition levels by text position instead of one random fetch per record
        // (gather_by_levels); the requests then leave the levels as the sort's records, in the order of the text
        // -- for rounds whose buckets are too long for the sort in LDS (bucket_sort.hpp), which needs a bucket's records side by side
        const bool lds_sort_possible = sizeof(T) == 8 && both && dense && !kn.no_bucket_sort && (by_ord ? cnt / nb_in : (uint64_t)0) <= BSORT_CAP / 4;
        const bool by_levels = by_ord && !gsa && !lds_sort_possible && isa_narrow_levels<T>(n, kn) == 2 && cnt >= (1ull << 22) &&
                               (kn.gather == 2 || (kn.gather == 0 && cnt >= n / 8));
        // (the payload entries of the sort's input; in the reduced layout the levels end in x.v, so the records' suffixes start in the other set)
        T* const vin = (by_levels && w.diet) ? w.ry.v : w.x.v;
        T* const valt = (by_levels && w.diet) ? w.x.v : w.ry.v;
        psacx_round rs; std::memset(&rs, 0, sizeof(rs));
        // ... and when the list has few enough buckets for their tables to live in LDS, the records are split where their keys are made: those that
        // carry their bucket's heavy rank skip the sort (heavy_keys.hpp)
        const bool heavy = by_levels && !kn.no_heavy && nb_in <= HEAVY_MAXB && (!w.diet || (2 * cnt + 64 <= n && cnt + 64 <= w.cap_active));
        bool merged = false;
        T* ids_heavy = nullptr; uint64_t* pairs_heavy = nullptr;
        HeavyView<T> hview;
        uint64_t rmq_queries = cnt;
        if constexpr (sizeof(T) == 8) {
            if (heavy) {
                // arrays (normal layout: everything has n entries; reduced layout: x.k1, x.k2, x.v have n, the ry set and the lists cap_active):
                //   levels            ry.k1 -> ry.k2                       | x.k1 -> x.v
                //   light records     x.k1, x.v (32-bit)                   | x.k1, ry.v (32-bit)
                //   heavy suffixes    ry.v (32-bit)                        | ry.k1, lower half (32-bit)
                //   light sort, alt   ry.k1, x.k2                          | x.k1 behind the records, ry.k1 upper half
                //   (the sorted order is read through a HeavyView by the kernels that follow: nothing is merged into arrays)
                //   then ids / pairs  ry.k2 / x.k1 or ry.k1 (the idle one) | ry.k2 / x.v
                const uint64_t cnt_r = (cnt + 63) & ~63ull;
                uint64_t* const la = reinterpret_cast<uint64_t*>(w.diet ? w.x.k1 : w.ry.k1);
                uint64_t* const lb = reinterpret_cast<uint64_t*>(w.diet ? w.x.v : w.ry.k2);
                uint32_t* const LV = reinterpret_cast<uint32_t*>(w.diet ? w.ry.v : w.x.v);
                uint32_t* const HB = reinterpret_cast<uint32_t*>(w.diet ? w.ry.k1 : w.ry.v);
                SortBufs<T> altL;
                altL.k1 = w.diet ? w.x.k1 + cnt_r : w.ry.k1; altL.k2 = nullptr;
                altL.v = w.diet ? reinterpret_cast<T*>(reinterpret_cast<uint32_t*>(w.ry.k1) + cnt_r) : w.x.k2;

                HeavyTabs ht; uint64_t nlight = 0; unsigned nblk = 0;
                {
                    ProfScope ps(c, TC_GATHER);
                    PSACX_TRY(gather_heavy_by_levels<T>(c, w, n, h, plist, cnt, d_sa, ord_arr, (unsigned)nb_in, d_isa, la, lb, w.x.k1, LV, HB, kb2, kn, &ht, &nlight, &nblk));
                    if (nlight) PSACX_TRY(summary_finish(c, w.sc, nblk));
                }
                SortBufs<T> sl{w.x.k1, nullptr, reinterpret_cast<T*>(LV)};
                if (nlight)
                    PSACX_TRY(pair_sort<T>(c, w.sc, sl, altL, nlight, /*iota=*/false, kb2 + num_bits, 0, nullptr, &sl, &rs, 0, 0,
                                           /*summary_ready=*/true, 0, -1, /*v32_in=*/true, /*keep_v32=*/true));
                {
                    ProfScope ps(c, TC_SORT_SCATTER);
                    hipLaunchKernelGGL((heavy_plan_kernel<T>), dim3(((unsigned)nb_in + 1 + 255) / 256), dim3(256), 0, c->stream, (uint32_t)nb_in, ht, kb2, (const T*)sl.k1, nlight, w.sc.d_err,
                                       plist, kn.no_lazy_ranks ? 0 : 1);
                    PSACX_HIP(c, hipGetLastError());
                }
                // the round's sorted records are not written out: the kernels below read them through the view
                hview.bstart = ht.bstart; hview.value = ht.value; hview.less = ht.less; hview.eq = ht.eq; hview.lstart = ht.lstart;
                hview.SLK = sl.k1; hview.SLV = reinterpret_cast<const uint32_t*>(sl.v); hview.HB = HB; hview.nb = (uint32_t)nb_in; hview.kb2 = kb2;
                hview.rank = ht.rank;
                // (what the rebucket kernel writes must not lie where the view reads: the sorted light records are in one array of each pair of the
                //  light sort, the heavy suffixes in ry.v / ry.k1)
                hview.tile_b = w.d_htile;
                {
                    ProfScope ps(c, TC_REBUCKET);
                    const uint64_t nt_ = (cnt + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
                    hipLaunchKernelGGL((heavy_tiles_kernel<T>), dim3((unsigned)((nt_ + 255) / 256)), dim3(256), 0, c->stream, hview, cnt, (unsigned)ScanCfg<T>::TILE, nt_, w.d_htile);
                    PSACX_HIP(c, hipGetLastError());
                }
                ids_heavy = w.ry.k2;
                pairs_heavy = reinterpret_cast<uint64_t*>(w.diet ? w.x.v : (sl.k1 == w.x.k1 ? w.ry.k1 : w.x.k1));
                c->stats.heavy_rounds += 1; c->stats.heavy_records += cnt - nlight; c->stats.light_records += nlight; c->stats.level_gathers += 1;
                sorted.k1 = nullptr; sorted.k2 = nullptr; sorted.v = nullptr;
                rmq_queries = nlight + 2 * nb_in;
                merged = true;
            }
        }
        if (!merged) {
            ProfScope ps(c, TC_GATHER);
            int gg = grid_for(c, cnt, 256, 16);
            if (whole)
                hipLaunchKernelGGL((shift_keys_kernel<T>), dim3(gg), dim3(256), 0, c->stream, d_isa, n, h, w.x.k1, key2, w.x.v, w.sc.d_partials, d_slen);
            else if (by_levels) {
                unsigned nb = 0;
                PSACX_TRY(gather_by_levels<T>(c, w, n, h, plist, cnt, d_sa, ord_arr, d_isa, reinterpret_cast<uint64_t*>(w.diet ? w.x.k1 : w.ry.k1),
                                              reinterpret_cast<uint64_t*>(w.diet ? w.x.v : w.ry.k2), w.x.k1, reinterpret_cast<uint32_t*>(vin), kb2, kn, &nb));
                gg = (int)nb;
                c->stats.level_gathers += 1;
            } else
                hipLaunchKernelGGL((gather_keys_kernel<T>), dim3(gg), dim3(256), 0, c->stream,
                                   plist, cnt, d_sa, w.bsa, d_isa, n, h, w.x.k1, key2, w.x.v, w.sc.d_partials, d_slen, kb2,
                                   by_ord ? (const uint32_t*)ord_arr : (const uint32_t*)nullptr);
            PSACX_HIP(c, hipGetLastError());
            // (only the three-kernel form of the sort reads the key summary)
            if (sort_is_three(cnt, true)) PSACX_TRY(summary_finish(c, w.sc, (unsigned)gg));
        }
        // no bucket longer than a workgroup holds in LDS (asked of the device first): every bucket is sorted there and the records cross
        // memory once (bucket_sort.hpp); a text with a few huge buckets left -- a tandem repeat -- is not worth the question
        bool sorted_in_lds = false;
        if constexpr (sizeof(T) == 8) {
            if (lds_sort_possible) {
                // (wide windows first; both questions are asked before the one synchronisation)
                const uint64_t nt_wide = (cnt + BSORT_W_WIDE - 1) / BSORT_W_WIDE, nt_narrow = (cnt + BSORT_W_NARROW - 1) / BSORT_W_NARROW;
                uint64_t* const st_wide = reinterpret_cast<uint64_t*>(w.sc.d_desc + 256);
                uint64_t* const st_narrow = st_wide + nt_wide + 1;
                unsigned long long* d_over = reinterpret_cast<unsigned long long*>(w.d_over);
                unsigned long long* h_over = reinterpret_cast<unsigned long long*>(c->pinned + 96);
                {
                    ProfScope ps(c, TC_SORT_HIST);
                    PSACX_HIP(c, hipMemsetAsync(d_over, 0, 2 * sizeof(unsigned long long), c->stream));
                    hipLaunchKernelGGL(bucket_task_starts_kernel<0>, dim3((unsigned)((nt_wide + 1 + 3) / 4)), dim3(256), 0, c->stream,
                                       reinterpret_cast<const uint64_t*>(w.x.k1), cnt, kb2, nt_wide, BSORT_W_WIDE, st_wide, d_over);
                    hipLaunchKernelGGL(bucket_task_starts_kernel<0>, dim3((unsigned)((nt_narrow + 1 + 3) / 4)), dim3(256), 0, c->stream,
                                       reinterpret_cast<const uint64_t*>(w.x.k1), cnt, kb2, nt_narrow, BSORT_W_NARROW, st_narrow, d_over + 1);
                    PSACX_HIP(c, hipGetLastError());
                    PSACX_HIP(c, hipMemcpyAsync(h_over, d_over, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
                }
                PSACX_HIP(c, hipStreamSynchronize(c->stream));
                const bool wide = h_over[0] == 0;
                const uint64_t ntasks = wide ? nt_wide : nt_narrow;
                const uint64_t* const starts = wide ? st_wide : st_narrow;
                if (wide || h_over[1] == 0) {
                    ProfScope ps(c, TC_SORT_SCATTER);
                    hipLaunchKernelGGL((bucket_sort_lds_kernel<BSORT_BLOCK, BSORT_ITEMS>), dim3((unsigned)ntasks), dim3(BSORT_BLOCK), 0, c->stream,
                                       reinterpret_cast<const uint64_t*>(w.x.k1), reinterpret_cast<const uint32_t*>(vin), starts, kb2,
                                       reinterpret_cast<uint64_t*>(w.ry.k1), reinterpret_cast<uint64_t*>(valt));
                    PSACX_HIP(c, hipGetLastError());
                    sorted.k1 = w.ry.k1; sorted.k2 = nullptr; sorted.v = valt;
                    rs.sort_passes = 1;
                    sorted_in_lds = true;
                }
            }
        }
        if (sorted_in_lds || merged) {
        } else if (both) {
            SortBufs<T> in2{w.x.k1, nullptr, vin}, alt2{w.ry.k1, nullptr, valt};
            PSACX_TRY(pair_sort<T>(c, w.sc, in2, alt2, cnt, /*iota=*/false, kb2 + num_bits, 0, nullptr, &sorted, &rs, 0, 0,
                                   /*summary_ready=*/true, 0, -1, /*v32_in=*/true));
            sorted.k2 = nullptr;
        } else
        PSACX_TRY(pair_sort<T>(c, w.sc, w.x, w.ry, cnt, /*iota=*/false, id_bits, id_bits, nullptr, &sorted, &rs, 0, 0,
                               /*summary_ready=*/true));
        if (rr) { rr->sort_passes += rs.sort_passes; rr->sort_passes_skipped += rs.sort_passes_skipped; }
        T* ids = merged ? ids_heavy : (sorted.k1 == w.x.k1) ? w.ry.k1 : w.x.k1;    // the set not holding the result is free
        // large rounds of the two-word form: the ISA entries leave the rebucket kernel as pairs (in the idle second key array of the round's
        // second record set) and reach ISA through partition levels once the compaction has read the ids (x.k2 and x.k1 are idle then)
        // (... also at fewer characters when the round's ranks came through the levels: long buckets that stride through the text)
        const bool levels_pay = kn.isa_update == 2 || (kn.isa_update == 0 && cnt >= n / 8 && (n >= (1ull << 31) || by_levels));
        uint64_t* const isa_pairs = (both && !whole && ((cnt >= (1ull << 22) && levels_pay) || isa_lv.open) && isa_narrow_levels<T>(n, kn) > 0)
                                        ? (merged ? pairs_heavy : reinterpret_cast<uint64_t*>(w.ry.k2)) : (uint64_t*)nullptr;
        // (a split round asks for a range minimum at most once per light record and bucket: its heavy runs stay whole)
        if (WITH_LCP) PSACX_TRY(prepare_range_min<T>(c, w, merged ? rmq_queries : cnt, n, kn));
        {
            ProfScope ps(c, TC_REBUCKET);
            const uint64_t ntiles = (cnt + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
            KeyShape split; split.lc = kb2; split.c1 = split.c2 = 0; split.spec = 0;       // (last_head_kernel: where a one-word key divides)
            PSACX_TRY((run_carries<T, true>(c, w, sorted.k1, sorted.k2, plist, cnt, nullptr, split, merged ? &hview : nullptr)));
            if (merged) {
                if constexpr (sizeof(T) == 8)
                    hipLaunchKernelGGL((rebucket_pure_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0, c->stream, hview, plist, cnt,
                                       d_sa, w.bsa, d_isa, ids, (const uint64_t*)w.d_carry, w.d_nact, w.d_nunf, isa_pairs);
            }
            if (merged)
                hipLaunchKernelGGL((rebucket_refine_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, WITH_LCP, false, true>), dim3((unsigned)ntiles),
                                   dim3(ScanCfg<T>::BLOCK), 0, c->stream, (const T*)nullptr, (const T*)nullptr, (const T*)nullptr, plist, cnt, n, h,
                                   d_sa, w.bsa, d_isa, w.pyr, ids, w.d_carry, w.d_nact, w.d_nunf, Boundary<T>(),
                                   (T*)nullptr, (T*)nullptr, (T*)nullptr, (unsigned long long*)nullptr, kb2, isa_pairs, hview);
            else
            hipLaunchKernelGGL((rebucket_refine_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS, WITH_LCP, false>), dim3((unsigned)ntiles),
                               dim3(ScanCfg<T>::BLOCK), 0, c->stream, sorted.k1, sorted.k2, sorted.v, plist, cnt, n, h,
                               d_sa, w.bsa, whole ? (T*)nullptr : d_isa, w.pyr, ids, w.d_carry, w.d_nact, w.d_nunf, Boundary<T>(),
                               (T*)nullptr, (T*)nullptr, (T*)nullptr, (unsigned long long*)nullptr, kb2, isa_pairs);
            PSACX_HIP(c, hipGetLastError());
        }
        PSACX_TRY(run_compact<T>(c, w, ids, plist, cnt, list_out, nactive, nunf, list_out ? w.cap_active : 0, 0, nullptr, nullptr, nullptr, 0, false,
                                 list_out ? ord_arr : (uint32_t*)nullptr));
        if (isa_pairs) {
            ProfScope ps(c, TC_ISA_SCATTER);
            const bool mine = !isa_lv.open;          // (a round in slabs opens the levels itself and closes them after its last slab)
            if (mine) PSACX_TRY(isa_lv.begin(c, w.d_cursors, n, reinterpret_cast<uint64_t*>(w.x.k2), kn));
            PSACX_TRY(isa_lv.add(isa_pairs, cnt, merged ? (const ulonglong2*)w.d_htile : (const ulonglong2*)nullptr));
            if (mine) PSACX_TRY(isa_lv.finish(d_isa, reinterpret_cast<uint64_t*>(w.x.k1)));
        }
        if (whole) {
            ProfScope ps(c, TC_ISA_SCATTER);
            PSACX_TRY(invert_permutation<T>(c, w.d_cursors, d_sa, w.bsa, n, d_isa, w.x, w.ry, kn, 0, &w.sc, false, false));
        }
        return PSACX_OK;
    };

    for (uint64_t h = 2ull * k; unf_b > 0 && h < n; h <<= 1) {
        psacx_round* rr = st.n_rounds < PSACX_MAX_ROUNDS ? &st.rounds[st.n_rounds] : nullptr;
        if (rr) std::memset(rr, 0, sizeof(*rr));
        uint64_t nactive = 0;
        const uint64_t round_cnt = no_fast ? n : active;
        if (!no_fast && !have_list && active <= w.cap_active) {
            // back under the capacity: rebuild the list of unresolved positions from the bucket ids
            const uint64_t ntiles = (n + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
            hipLaunchKernelGGL((count_active_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0,
                               c->stream, w.bsa, n, (T)0, (T)0, w.d_nact, 0u, w.d_nunf);
            PSACX_HIP(c, hipGetLastError());
            uint64_t a2 = 0, u2 = 0;
            PSACX_TRY(run_compact<T>(c, w, w.bsa, nullptr, n, pos, &a2, &u2, w.cap_active, 0, nullptr, nullptr, nullptr, 0, false, ord_arr));
            if (a2 != active || u2 != unf_b) { c->hip_err = "active list rebuild disagrees with the round counters"; return PSACX_EDEVICE; }
            have_list = true;
        }
        // rounds in which at least 7/8 of the suffixes are unresolved (repetitive texts) take all n in text order, as psac's
        // doubling rounds do: the random fetch of the ranks h further and the random ISA stores turn into streams
        bool whole = !no_fast && !w.diet && have_list && w.cap_active >= n && n >= (1ull << 16) &&
                     active >= n - n / 8 && !kn.no_whole;
        // ... unless the list takes the partition levels (long buckets, 64-bit words: gather_by_levels and the heavy / light split in refine):
        // the round then sorts bucket numbers of a few bits instead of ranks of log n, and mostly not even those (heavy_keys.hpp)
        if (whole && sizeof(T) == 8 && !gsa && ord_arr && kn.gather != 1 && isa_narrow_levels<T>(n, kn) == 2 && active >= (1ull << 22) &&
            unf_b > 0 && active / unf_b > BSORT_CAP / 4) whole = false;
        if (whole) {
            // ... unless SA order is nearly text order (sa_locality_kernel): 2^27 equal characters take 9.2 ms per round through
            // the list and 12.0 ms as whole rounds, a period-1024 tandem repeat 18.0 against 14.7 (profiles/r03g_*)
            unsigned long long* d_near = reinterpret_cast<unsigned long long*>(w.d_totals + 2);
            unsigned long long* h_near = reinterpret_cast<unsigned long long*>(c->pinned + 64);
            const uint64_t samples = 1u << 16;
            PSACX_HIP(c, hipMemsetAsync(d_near, 0, sizeof(unsigned long long), c->stream));
            hipLaunchKernelGGL((sa_locality_kernel<T>), dim3(64), dim3(256), 0, c->stream, d_sa, n, samples, d_near);
            PSACX_HIP(c, hipGetLastError());
            PSACX_HIP(c, hipMemcpyAsync(h_near, d_near, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            PSACX_HIP(c, hipStreamSynchronize(c->stream));
            if (*h_near * 2 > samples) whole = false;
        }
        // (Tried: such a round through the list when its sort can stay in LDS -- 2^30 characters of repeated reads with mutations 682 -> 778 ms:
        //  the random fetch of the ranks and the random ISA stores of nearly n suffixes cost more than the eight passes over three words.)
        if (whole) {
            PSACX_TRY(refine((const T*)nullptr, n, h, rr, pos_next, &nactive, &unf_b, true));
            std::swap(pos, pos_next);
        } else if (no_fast || have_list) {
            PSACX_TRY(refine(no_fast ? (const T*)nullptr : pos, round_cnt, h, rr, pos_next, &nactive, &unf_b, false, no_fast ? 0 : unf_b));
            std::swap(pos, pos_next);
        } else {
            // Slabs.  Buckets are contiguous in SA order and a refinement only permutes inside buckets, so any range
            // of SA positions that ends on a bucket boundary can be refined on its own.  The ranks read for later
            // slabs may already carry this round's refinement (Larsson-Sadakane style): the order stays a
            // refinement of the true suffix order and every new LCP is still h + a range minimum over final entries,
            // only the per-round counters may run ahead of the reference's log.
            const uint64_t ntiles = (n + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
            hipLaunchKernelGGL((count_active_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS>), dim3((unsigned)ntiles), dim3(ScanCfg<T>::BLOCK), 0,
                               c->stream, w.bsa, n, (T)0, (T)0, w.d_nact, 0u);
            PSACX_HIP(c, hipGetLastError());
            std::vector<uint64_t> tile_act(ntiles);
            PSACX_HIP(c, hipMemcpyAsync(tile_act.data(), w.d_nact, ntiles * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
            PSACX_HIP(c, hipStreamSynchronize(c->stream));
            const uint64_t room = w.cap_active > 2ull * ScanCfg<T>::TILE ? w.cap_active - 2ull * ScanCfg<T>::TILE : 0;
            uint64_t sum_act = 0, sum_unf = 0, s0 = 0;
            T* h_id = reinterpret_cast<T*>(c->pinned + 128);
            // (64-bit words, at most 2^32 characters: the slabs' ISA entries are collected by destination class and stored at the end of the round)
            const bool collect = sizeof(T) == 8 && n <= (1ull << 32) && isa_narrow_levels<T>(n, kn) > 0 && room >= SMALL_SORT_MAX &&
                                 (kn.isa_update == 2 || (kn.isa_update == 0 && n >= (1ull << 31)));
            while (s0 < n) {
                // furthest tile boundary whose tiles (from the one holding s0) hold at most `room` unresolved positions
                uint64_t t = s0 / ScanCfg<T>::TILE, acc = 0;
                while (t < ntiles && acc + tile_act[t] <= room) acc += tile_act[t++];
                uint64_t e = std::min<uint64_t>(t * (uint64_t)ScanCfg<T>::TILE, n);
                if (e < n) {
                    // back to the head of the bucket that holds position e (bucket id = head position + 1)
                    PSACX_HIP(c, hipMemcpyAsync(h_id, w.bsa + e, sizeof(T), hipMemcpyDeviceToHost, c->stream));
                    PSACX_HIP(c, hipStreamSynchronize(c->stream));
                    e = (uint64_t)*h_id - 1;
                }
                if (e <= s0) {
                    c->hip_err = "a bucket of unresolved suffixes is larger than the reduced-memory layout has room for";
                    return PSACX_ENOMEM;
                }
                const uint64_t len = e - s0, lt = (len + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE;
                hipLaunchKernelGGL((count_active_kernel<T, ScanCfg<T>::BLOCK, SCAN_ITEMS>), dim3((unsigned)lt), dim3(ScanCfg<T>::BLOCK), 0,
                                   c->stream, w.bsa + s0, len, (T)0, (T)0, w.d_nact, 0u, w.d_nunf);
                PSACX_HIP(c, hipGetLastError());
                uint64_t cnt = 0, slab_buckets = 0;
                PSACX_TRY(run_compact<T>(c, w, w.bsa + s0, nullptr, len, pos, &cnt, &slab_buckets, w.cap_active, 0, nullptr, nullptr, nullptr, s0, false, ord_arr));
                if (cnt > w.cap_active) { c->hip_err = "slab larger than planned"; return PSACX_EDEVICE; }
                if (cnt) {
                    uint64_t na = 0, nu = 0;
                    // (a slab too small for the two-word records sorts through x.k2, where the collected pairs live: they go to ISA first)
                    if (isa_lv.open && cnt < SMALL_SORT_MAX) PSACX_TRY(isa_lv.finish(d_isa, reinterpret_cast<uint64_t*>(w.x.k1)));
                    else if (collect && !isa_lv.open && cnt >= SMALL_SORT_MAX) PSACX_TRY(isa_lv.begin(c, w.d_cursors, n, reinterpret_cast<uint64_t*>(w.x.k2), kn));
                    PSACX_TRY(refine(pos, cnt, h, rr, (T*)nullptr, &na, &nu, false, slab_buckets));
                    sum_act += na; sum_unf += nu;
                }
                s0 = e;
            }
            if (isa_lv.open) {
                ProfScope ps(c, TC_ISA_SCATTER);
                PSACX_TRY(isa_lv.finish(d_isa, reinterpret_cast<uint64_t*>(w.x.k1)));
            }
            nactive = sum_act; unf_b = sum_unf;
        }
        if (rr) {
            rr->h = h; rr->active = round_cnt; rr->unfinished_buckets = unf_b; rr->unfinished_elements = nactive;
            st.n_rounds++;
        }
        active = nactive;
    }

    // ISA already holds 0-based ranks (the -1 of suffix_array.hpp:460-464 is applied on every write)
    delete total; total = nullptr;

    unsigned h_err = 0;
    PSACX_HIP(c, hipMemcpyAsync(c->pinned, w.sc.d_err, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    h_err = *reinterpret_cast<unsigned*>(c->pinned);
    if (h_err) return PSACX_EDEVICE;
    return PSACX_OK;
}

template <typename T>
int construct_dispatch(psacx_ctx* c, const uint8_t* d_text, uint64_t n, uint32_t k, uint32_t flags, T* d_sa,
                       T* d_isa, T* d_lcp, uint8_t* d_lc = nullptr) {
    if (!c || !d_text || !d_sa || !d_isa || n == 0) return PSACX_EINVAL;
    if ((flags & PSACX_LCP) && !d_lcp) return PSACX_EINVAL;
    if (d_lc && !(flags & PSACX_LCP)) return PSACX_EINVAL;
    if (sizeof(T) == 4 && n > 0xFFFFFFFEull) return PSACX_ERANGE;
    if (sizeof(T) == 8 && n >= (1ull << 62)) return PSACX_ERANGE;
    PSACX_HIP(c, hipSetDevice(c->device));
    std::memset(&c->stats, 0, sizeof(c->stats));
    c->profile = (flags & PSACX_PROFILE) != 0;
    c->ev_used = 0;
    int rc;
    if (flags & PSACX_LCP) rc = construct_dev<T, true>(c, d_text, n, k, flags, d_sa, d_isa, d_lcp);
    else rc = construct_dev<T, false>(c, d_text, n, k, flags, d_sa, d_isa, (T*)nullptr);
    if (rc == PSACX_OK && d_lc) {
        hipLaunchKernelGGL((left_chars_kernel<T>), dim3(grid_for(c, n, 256, 16)), dim3(256), 0, c->stream, d_text, n, d_sa, d_lcp, d_lc);
        PSACX_HIP(c, hipGetLastError());
        PSACX_HIP(c, hipStreamSynchronize(c->stream));      // like SA / ISA / LCP, Lc is complete when the call returns
    }
    if (rc == PSACX_OK && c->profile) prof_collect(c);
    return rc;
}

// Generalized suffix array (construct_ss, suffix_array.hpp:267-363): d_off[0..m] are the offsets of
// the m strings inside d_text (device memory, ascending, d_off[0] = 0, d_off[m] = n).
template <typename T>
int construct_gsa_dispatch(psacx_ctx* c, const uint8_t* d_text, uint64_t n, const uint64_t* d_off, uint64_t m, uint32_t k,
                           uint32_t flags, T* d_sa, T* d_isa, T* d_lcp) {
    if (!c || !d_text || !d_off || !d_sa || !d_isa || n == 0 || m == 0 || m > n) return PSACX_EINVAL;
    if ((flags & PSACX_LCP) && !d_lcp) return PSACX_EINVAL;
    if (sizeof(T) == 4 && n > 0xFFFFFFFEull) return PSACX_ERANGE;
    if (sizeof(T) == 8 && n >= (1ull << 62)) return PSACX_ERANGE;
    PSACX_HIP(c, hipSetDevice(c->device));
    std::memset(&c->stats, 0, sizeof(c->stats));
    c->profile = (flags & PSACX_PROFILE) != 0;
    c->ev_used = 0;
    // the offsets must describe m non-empty strings covering [0, n) (stringset.hpp:53-72); a malformed array would
    // send the window kernels out of bounds
    {
        PSACX_TRY(ensure_pinned(c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 4096));
        PSACX_TRY(ensure_slab(c, 4096));
        unsigned long long* d_bad = reinterpret_cast<unsigned long long*>(c->slab);
        PSACX_HIP(c, hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), c->stream));
        hipLaunchKernelGGL(check_offsets_kernel<0>, dim3(grid_for(c, m + 1, 256, 8)), dim3(256), 0, c->stream, d_off, m, n, d_bad);
        PSACX_HIP(c, hipGetLastError());
        PSACX_HIP(c, hipMemcpyAsync(c->pinned, d_bad, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        PSACX_HIP(c, hipStreamSynchronize(c->stream));
        if (*reinterpret_cast<unsigned long long*>(c->pinned)) return PSACX_EINVAL;
    }
    T* d_slen = nullptr;
    hipError_t e = hipMalloc((void**)&d_slen, n * sizeof(T));
    if (e != hipSuccess) { c->hip_err = std::string("hipMalloc(string lengths): ") + hipGetErrorString(e); (void)hipGetLastError(); return PSACX_ENOMEM; }
    hipLaunchKernelGGL((string_len_kernel<T>), dim3(grid_for(c, n, 256, 16)), dim3(256), 0, c->stream, d_off, m, n, d_slen);
    int rc = hipGetLastError() == hipSuccess ? PSACX_OK : PSACX_EHIP;
    if (rc == PSACX_OK) {
        if (flags & PSACX_LCP) rc = construct_dev<T, true>(c, d_text, n, k, flags, d_sa, d_isa, d_lcp, d_slen);
        else rc = construct_dev<T, false>(c, d_text, n, k, flags, d_sa, d_isa, (T*)nullptr, d_slen);
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_slen);
    if (rc == PSACX_OK && c->profile) prof_collect(c);
    return rc;
}

template <typename T>
int construct_gsa_host(psacx_ctx* c, const uint8_t* text, uint64_t n, const uint64_t* off, uint64_t m, uint32_t k, uint32_t flags,
                       T* sa, T* isa, T* lcp) {
    if (!c || !text || !off || !sa || !isa || n == 0 || m == 0 || m > n) return PSACX_EINVAL;
    if ((flags & PSACX_LCP) && !lcp) return PSACX_EINVAL;
    if (off[0] != 0 || off[m] != n) return PSACX_EINVAL;
    for (uint64_t t = 0; t < m; ++t) if (off[t + 1] <= off[t]) return PSACX_EINVAL;     // empty strings are not part of a set (stringset.hpp:53-72)
    if (sizeof(T) == 4 && n > 0xFFFFFFFEull) return PSACX_ERANGE;
    PSACX_HIP(c, hipSetDevice(c->device));
    uint8_t* d_text = nullptr; uint64_t* d_off = nullptr; T *d_sa = nullptr, *d_isa = nullptr, *d_lcp = nullptr;
    auto cleanup = [&]() {
        if (d_text) (void)hipFree(d_text); if (d_off) (void)hipFree(d_off); if (d_sa) (void)hipFree(d_sa);
        if (d_isa) (void)hipFree(d_isa); if (d_lcp) (void)hipFree(d_lcp);
    };
    hipError_t e = hipMalloc((void**)&d_text, n);
    if (e == hipSuccess) e = hipMalloc((void**)&d_off, (m + 1) * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_sa, n * sizeof(T));
    if (e == hipSuccess) e = hipMalloc((void**)&d_isa, n * sizeof(T));
    if (e == hipSuccess && (flags & PSACX_LCP)) e = hipMalloc((void**)&d_lcp, n * sizeof(T));
    if (e != hipSuccess) { c->hip_err = std::string("hipMalloc(io): ") + hipGetErrorString(e); (void)hipGetLastError(); cleanup(); return PSACX_ENOMEM; }
    e = hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, off, (m + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { c->hip_err = hipGetErrorString(e); cleanup(); return PSACX_EHIP; }
    int rc = construct_gsa_dispatch<T>(c, d_text, n, d_off, m, k, flags, d_sa, d_isa, d_lcp);
    if (rc == PSACX_OK) {
        e = hipMemcpyAsync(sa, d_sa, n * sizeof(T), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(isa, d_isa, n * sizeof(T), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && d_lcp) e = hipMemcpyAsync(lcp, d_lcp, n * sizeof(T), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { c->hip_err = hipGetErrorString(e); rc = PSACX_EHIP; }
    }
    cleanup();
    return rc;
}

// host-pointer form: stage over PCIe, run, copy back.  The device copies live in the ctx between calls
// (suffix_array<>::construct may be called repeatedly on one object, test/test_psac.cpp:148-170).
// SA and LCP on their way to the caller's arrays while the first round's SA -> ISA inversion still runs (random text: the two arrays are
// final when rebucket_first_kernel has written them, 37 of the 173 ms of a 2^32 construction before it ends).  The hook runs on the
// constructing thread right after that kernel has been enqueued; a second host thread then drives the staged copies, their narrowing
// kernels on a stream of their own behind an event.  Speculative: if refinement rounds follow, the arrays change and are copied again.
template <typename T>
struct EarlyOut {
    psacx_ctx* c; T* sa; const T* d_sa; T* lcp; const T* d_lcp; uint64_t n;
    std::thread th; int rc; bool started;
    std::chrono::steady_clock::time_point t0; double ms[3];
};
template <typename T>
void early_out_hook(void* p) {
    EarlyOut<T>* eo = static_cast<EarlyOut<T>*>(p);
    psacx_ctx* c = eo->c;
    if (hipEventRecord(c->early_ev, c->stream) != hipSuccess) { (void)hipGetLastError(); return; }
    eo->started = true;
    eo->th = std::thread([eo]() {
        psacx_ctx* c = eo->c;
        // nothing leaves unless the first round has resolved every bucket (one word, counted behind the rebucket kernel on its own stream)
        unsigned long long* const h_left = reinterpret_cast<unsigned long long*>(c->stage[0]);
        *h_left = 1;
        if (hipSetDevice(c->device) != hipSuccess || hipStreamWaitEvent(c->early_stream, c->early_ev, 0) != hipSuccess ||
            hipMemcpyAsync(h_left, c->early_word, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->early_stream) != hipSuccess ||
            hipStreamSynchronize(c->early_stream) != hipSuccess) { (void)hipGetLastError(); eo->rc = PSACX_EHIP; return; }
        if (*h_left != 0) { eo->rc = PSACX_EINVAL; return; }          // (refinement rounds follow: the arrays leave when they are through)
        auto since = [eo]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - eo->t0).count(); };
        eo->ms[0] = since();
        D2hJob<T> jobs[2] = {{eo->sa, eo->d_sa, eo->n, eo->n - 1, 0, 0, 0, 0}, {eo->lcp, eo->d_lcp, eo->n, ~0ull, 0, 0, 0, 0}};
        const int rc = staged_d2h_jobs<T>(c, jobs, eo->lcp ? 2 : 1, c->early_stream);          // (SA and LCP chunk by chunk: engine.hpp)
        eo->ms[1] = eo->ms[2] = since();
        eo->rc = rc;
    });
}

template <typename T>
int construct_host(psacx_ctx* c, const uint8_t* text, uint64_t n, uint32_t k, uint32_t flags, T* sa, T* isa, T* lcp,
                   uint8_t* lc = nullptr) {
    if (!c || !text || !sa || !isa || n == 0) return PSACX_EINVAL;
    if ((flags & PSACX_LCP) && !lcp) return PSACX_EINVAL;
    if (lc && !(flags & PSACX_LCP)) return PSACX_EINVAL;
    if (sizeof(T) == 4 && n > 0xFFFFFFFEull) return PSACX_ERANGE;
    PSACX_HIP(c, hipSetDevice(c->device));
    const bool with_lcp = (flags & PSACX_LCP) != 0;
    uint8_t *d_text = nullptr, *d_lc = nullptr; T *d_sa = nullptr, *d_isa = nullptr, *d_lcp = nullptr;
    auto layout = [&](Arena& a) {
        d_sa = a.take<T>(n); d_isa = a.take<T>(n);
        if (with_lcp) d_lcp = a.take<T>(n);
        d_text = a.take<uint8_t>(n);
        if (lc) d_lc = a.take<uint8_t>(n);
    };
    { Arena dry(nullptr); layout(dry); PSACX_TRY(ensure_io(c, dry.off + 4096)); }
    Arena ar(c->io);
    layout(ar);
    typedef std::chrono::steady_clock clk;
    const clk::time_point t0 = clk::now();
    auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
    double ms[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    PSACX_TRY(staged_h2d(c, d_text, text, n));
    ms[0] = ms_since(t0);
    clk::time_point t1 = clk::now();
    // arrays of at least eight staging chunks leave early (smaller ones are on the wire for a few milliseconds)
    EarlyOut<T> eo{c, sa, d_sa, d_lcp ? lcp : (T*)nullptr, d_lcp, n, std::thread(), PSACX_OK, false, t0, {0, 0, 0}};
    if (!c->knobs.no_early_out && n * sizeof(T) >= 8 * STAGE_CHUNK && ensure_stage(c) == PSACX_OK) {
        if (!c->early_stream && hipStreamCreateWithFlags(&c->early_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->early_stream = nullptr; }
        if (!c->early_ev && hipEventCreateWithFlags(&c->early_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); c->early_ev = nullptr; }
        if (!c->early_word && hipMalloc((void**)&c->early_word, sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); c->early_word = nullptr; }
        if (c->early_stream && c->early_ev && c->early_word) { c->first_round_hook = early_out_hook<T>; c->first_round_hook_arg = &eo; }
    }
    int rc = construct_dispatch<T>(c, d_text, n, k, flags, d_sa, d_isa, d_lcp, d_lc);
    c->first_round_hook = nullptr;
    ms[1] = ms_since(t1); t1 = clk::now();
    if (eo.started) eo.th.join();          // (the staging buffers are its until it is through)
    if (rc != PSACX_OK) return rc;
    // what left early stands if the first round was the only one
    const bool early = eo.started && eo.rc == PSACX_OK && c->stats.n_rounds == 1;
    ms[2] = ms_since(t1); t1 = clk::now();          // (what was left of SA and LCP on their early way out)
    {
        // ISA, and SA and LCP unless they are out already, chunk by chunk through one ring (engine.hpp: staged_d2h_jobs)
        D2hJob<T> jobs[3]; int nj = 0;
        jobs[nj++] = D2hJob<T>{isa, d_isa, n, n - 1, 0, 0, 0, 0};
        if (!early) jobs[nj++] = D2hJob<T>{sa, d_sa, n, n - 1, 0, 0, 0, 0};
        if (!early && d_lcp) jobs[nj++] = D2hJob<T>{lcp, d_lcp, n, ~0ull, 0, 0, 0, 0};
        PSACX_TRY(staged_d2h_jobs<T>(c, jobs, nj));
    }
    ms[3] = ms_since(t1); t1 = clk::now();
    if (d_lc) PSACX_TRY(staged_d2h(c, lc, d_lc, n));
    ms[4] = ms_since(t1); ms[5] = ms_since(t0);
    if (early) { ms[6] = eo.ms[0]; ms[7] = eo.ms[1]; ms[8] = eo.ms[2]; }
    for (int i = 0; i < 9; ++i) c->stats.ms_host[i] = ms[i];
    return PSACX_OK;
}

// stand-alone rank-pair sort (idxsort.hpp:23-83): sorts in place, idx receives the permutation
template <typename T>
int pair_sort_dev(psacx_ctx* c, T* d_b1, T* d_b2, T* d_idx, uint64_t n, uint32_t key_bits) {
    if (!c || !d_b1 || !d_b2 || !d_idx || n == 0) return PSACX_EINVAL;
    if (sizeof(T) == 4 && n > 0xFFFFFFFEull) return PSACX_ERANGE;
    PSACX_HIP(c, hipSetDevice(c->device));
    PSACX_TRY(ensure_pinned(c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 4096));
    Arena dry(nullptr);
    auto layout = [&](Arena& a, SortBufs<T>& alt, SortScratch& sc, T*& vtmp) {
        alt.k1 = a.take<T>(n); alt.k2 = a.take<T>(n); alt.v = a.take<T>(n); vtmp = a.take<T>(n);
        sc.d_hist = a.take<unsigned long long>((size_t)MAX_PASSES * RADIX);
        sc.d_base = a.take<unsigned long long>((size_t)MAX_PASSES * RADIX);
        sc.desc_bytes = sort_desc_bytes(n);
        sc.d_desc = a.take<char>(sc.desc_bytes);
        sc.d_err = a.take<unsigned>(64);
        sc.d_summary = a.take<unsigned long long>(8);
        sc.d_partials = a.take<unsigned long long>(((size_t)(n / 2048) + 8192) * 4);
    };
    SortBufs<T> alt; SortScratch sc; T* vtmp;
    layout(dry, alt, sc, vtmp);
    PSACX_TRY(ensure_slab(c, dry.off + 4096));
    Arena ar(c->slab);
    layout(ar, alt, sc, vtmp);
    sc.h_hist = reinterpret_cast<unsigned long long*>(c->pinned + 1024);
    sc.h_base = sc.h_hist + (size_t)MAX_PASSES * RADIX;
    sc.h_summary = reinterpret_cast<unsigned long long*>(c->pinned + 256);
    PSACX_HIP(c, hipMemsetAsync(sc.d_err, 0, 64 * sizeof(unsigned), c->stream));
    std::memset(&c->stats, 0, sizeof(c->stats));
    c->profile = true; c->ev_used = 0;
    SortBufs<T> in{d_b1, d_b2, vtmp}, res;
    psacx_round rs; std::memset(&rs, 0, sizeof(rs));
    PSACX_TRY(pair_sort<T>(c, sc, in, alt, n, true, key_bits ? key_bits : (uint32_t)sizeof(T) * 8, key_bits ? key_bits : (uint32_t)sizeof(T) * 8, nullptr, &res, &rs));
    if (res.k1 != d_b1) {
        PSACX_HIP(c, hipMemcpyAsync(d_b1, res.k1, n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        PSACX_HIP(c, hipMemcpyAsync(d_b2, res.k2, n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    }
    PSACX_HIP(c, hipMemcpyAsync(d_idx, res.v, n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    PSACX_HIP(c, hipMemcpyAsync(c->pinned, sc.d_err, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    c->stats.rounds[0] = rs; c->stats.n_rounds = 1;
    prof_collect(c);
    if (*reinterpret_cast<unsigned*>(c->pinned)) return PSACX_EDEVICE;
    return PSACX_OK;
}

} // namespace psacx
