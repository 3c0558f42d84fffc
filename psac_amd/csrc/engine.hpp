// engine.hpp -- host side of the psacx engine: context, HBM workspace, the
// rank-pair sorter driver and the prefix-doubling loop.  Compiled by hipcc into
// libpsacx.so; the only public surface is include/psacx.h.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/psacx.h"
#include "radix.hpp"
#include "sa_kernels.hpp"

namespace psacx {

enum TimerCat {
    TC_ALPHABET = 0, TC_KMER, TC_SORT_HIST, TC_SORT_SCATTER, TC_SORT_SCATTER3, TC_SORT_SCATTER2, TC_SORT_TILEHIST, TC_REBUCKET, TC_ISA_SCATTER,
    TC_GATHER, TC_COMPACT, TC_RMQ_BUILD, TC_FINALIZE, TC_TOTAL, TC_COUNT
};

} // namespace psacx

struct psacx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    char* slab = nullptr;
    size_t slab_bytes = 0;
    char* aux = nullptr;             // second, lazily allocated workspace (range-minimum helpers of level 0)
    size_t aux_bytes = 0;
    char* pinned = nullptr;          // host-pinned scratch (histograms, counters)
    char* pinned_dev = nullptr;      // the same memory as the device addresses it (kernels store round counters there), or null
    size_t pinned_bytes = 0;
    char* io = nullptr;              // device copies of text / SA / ISA / LCP for the host-pointer entry points (kept between calls)
    size_t io_bytes = 0;
    char* stage[2] = {nullptr, nullptr};   // pinned staging buffers of the host-pointer entry points
    size_t stage_bytes = 0;
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    // Freed device blocks of the multi-GPU path, kept for reuse (size -> pointer).  Every use of such a block is
    // ordered on this ctx's stream (its second stream joins it through events), so a block handed out again is
    // only touched after everything that used it before.  (hipMallocAsync was measured first: its pool stalls for
    // seconds now and then once several streams of one device share it.)
    std::multimap<size_t, void*>* pool = nullptr;
    size_t pool_bytes = 0;           // bytes of cached (free) blocks
    size_t pool_live = 0;            // bytes of blocks handed out
    size_t pool_peak = 0;            // high-water mark of pool_live: what the engine needed at once.  Free blocks kept beyond
                                     // that go back to the device whenever an allocation does not fit (pool_alloc).
    size_t pool_cache_limit = 0;     // > 0: a miss first returns the cached blocks to the device once they exceed this many bytes
    std::string hip_err;
    psacx_stats stats;
    bool profile = false;
    bool profile_ops = false;        // step-level ops accumulate into stats (psacx_profile)
    struct Ev { hipEvent_t a, b; int cat; };
    std::vector<Ev> ev_pool;
    size_t ev_used = 0;
    int n_cu = 256;
};

namespace psacx {

#define PSACX_HIP(ctx, call)                                                              \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            (ctx)->hip_err = std::string(#call) + ": " + hipGetErrorString(e__);          \
            return PSACX_EHIP;                                                            \
        }                                                                                 \
    } while (0)

#define PSACX_TRY(expr)                                                                   \
    do { int rc__ = (expr); if (rc__ != PSACX_OK) return rc__; } while (0)

struct ProfScope {
    psacx_ctx* c; size_t idx; bool on;
    ProfScope(psacx_ctx* ctx, int cat) : c(ctx), idx(0), on(ctx->profile) {
        if (!on) return;
        if (c->ev_used == c->ev_pool.size()) {
            psacx_ctx::Ev e; e.cat = cat;
            if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) { on = false; return; }
            c->ev_pool.push_back(e);
        }
        idx = c->ev_used++;
        c->ev_pool[idx].cat = cat;
        (void)hipEventRecord(c->ev_pool[idx].a, c->stream);
    }
    ~ProfScope() { if (on) (void)hipEventRecord(c->ev_pool[idx].b, c->stream); }
};

// adds the elapsed time of this call's event pairs to the running totals (step-level ops)
inline void prof_accumulate(psacx_ctx* c) {
    double acc[TC_COUNT];
    for (int i = 0; i < TC_COUNT; ++i) acc[i] = 0;
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b) == hipSuccess) acc[c->ev_pool[i].cat] += ms;
    }
    psacx_stats& s = c->stats;
    s.ms_sort_hist += acc[TC_SORT_HIST]; s.ms_sort_scatter += acc[TC_SORT_SCATTER];
    s.ms_sort_scatter3 += acc[TC_SORT_SCATTER3]; s.ms_sort_tilehist += acc[TC_SORT_TILEHIST];
    s.ms_sort_scatter2 += acc[TC_SORT_SCATTER2];
    c->ev_used = 0;
}

inline void prof_collect(psacx_ctx* c) {
    double acc[TC_COUNT];
    for (int i = 0; i < TC_COUNT; ++i) acc[i] = 0;
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b) == hipSuccess) acc[c->ev_pool[i].cat] += ms;
    }
    psacx_stats& s = c->stats;
    s.ms_alphabet = acc[TC_ALPHABET]; s.ms_kmer = acc[TC_KMER]; s.ms_sort_hist = acc[TC_SORT_HIST];
    s.ms_sort_scatter = acc[TC_SORT_SCATTER]; s.ms_sort_scatter3 = acc[TC_SORT_SCATTER3];
    s.ms_sort_tilehist = acc[TC_SORT_TILEHIST]; s.ms_rebucket = acc[TC_REBUCKET];
    s.ms_sort_scatter2 = acc[TC_SORT_SCATTER2];
    s.ms_isa_scatter = acc[TC_ISA_SCATTER]; s.ms_gather = acc[TC_GATHER]; s.ms_compact = acc[TC_COMPACT];
    s.ms_rmq_build = acc[TC_RMQ_BUILD]; s.ms_finalize = acc[TC_FINALIZE]; s.ms_total = acc[TC_TOTAL];
}

// Tuning and test switches of a construction, read from the environment in ONE place at the start of every call (the
// tests flip them between calls of one process, so they are not cached).  Each selects an earlier or fallback form of a
// step that the parity suite keeps covered; none changes the result.
struct Knobs {
    bool force_diet;        // PSACX_FORCE_DIET: reduced-memory layout although the normal one fits
    uint64_t diet_cap;      // PSACX_DIET_CAP: at most this many records of room for the refinement rounds (0 = no limit)
    bool one_stage;         // PSACX_ONE_STAGE: first round as one sort over both key words
    bool ties_radix;        // PSACX_TIES_RADIX: stage 2 of the first round through compaction + radix sort
    bool no_key_hist;       // PSACX_NO_KEY_HIST: no tile histograms out of the key / rebucket kernels
    bool no_one_word;       // PSACX_NO_ONE_WORD: the prefix sort of the first round in (word 1, 32-bit suffix) passes, not one-word records
    bool no_lazy_ids;       // PSACX_NO_LAZY_IDS: rebucket_first_kernel writes the bucket ids of every tile, resolved or not
    bool no_fused_keys;     // PSACX_NO_FUSED_KEYS: one-word prefix sort with word 1 written by key_pairs_kernel and read back by the pass on the top digit
    unsigned one_word_min;  // PSACX_ONE_WORD_MIN: log2 of the smallest text that takes the one-word form (default 24; tests: 21)
    unsigned lead_slack;    // PSACX_LEAD_SLACK (default 2)
    bool isa_partition;     // PSACX_ISA_PARTITION: 32-bit words: reservation levels instead of radix levels / the fused form
    bool isa_wide;          // PSACX_ISA_WIDE: 64-bit words: pairs stay 64-bit
    bool no_fused_l1;       // PSACX_NO_FUSED_L1: first inversion level as its own kernel
    bool no_rmq_aux;        // PSACX_NO_RMQ_AUX: range minima without the running-minimum tables
    bool isa_two_arrays;    // PSACX_ISA_TWO_ARRAYS: the 32-bit pairs of the SA -> ISA levels in two arrays instead of one of packed entries
    bool wide_refine;       // PSACX_WIDE_REFINE: refinement records of 64-bit words keep their three words below 2^32 characters too
    bool no_whole_rounds;   // PSACX_NO_WHOLE_ROUNDS: rounds with almost every suffix unresolved still go through the list of unresolved positions
    bool sort_debug;        // PSACX_SORT_DEBUG: phase stamps of sampled scatter tiles
};
inline Knobs read_knobs() {
    Knobs k;
    k.force_diet = getenv("PSACX_FORCE_DIET") != nullptr;
    const char* e = getenv("PSACX_DIET_CAP");
    k.diet_cap = e ? strtoull(e, nullptr, 10) : 0;
    k.one_stage = getenv("PSACX_ONE_STAGE") != nullptr;
    k.ties_radix = getenv("PSACX_TIES_RADIX") != nullptr;
    k.no_key_hist = getenv("PSACX_NO_KEY_HIST") != nullptr;
    k.no_one_word = getenv("PSACX_NO_ONE_WORD") != nullptr;
    k.no_fused_keys = getenv("PSACX_NO_FUSED_KEYS") != nullptr;
    k.no_lazy_ids = getenv("PSACX_NO_LAZY_IDS") != nullptr;
    k.one_word_min = getenv("PSACX_ONE_WORD_MIN") ? (unsigned)std::max(16, atoi(getenv("PSACX_ONE_WORD_MIN"))) : 24u;
    e = getenv("PSACX_LEAD_SLACK");
    k.lead_slack = e ? (unsigned)atoi(e) : 2u;
    k.isa_partition = getenv("PSACX_ISA_PARTITION") != nullptr;
    k.isa_wide = getenv("PSACX_ISA_WIDE") != nullptr;
    k.no_fused_l1 = getenv("PSACX_NO_FUSED_L1") != nullptr;
    k.no_rmq_aux = getenv("PSACX_NO_RMQ_AUX") != nullptr;
    k.no_whole_rounds = getenv("PSACX_NO_WHOLE_ROUNDS") != nullptr;
    k.isa_two_arrays = getenv("PSACX_ISA_TWO_ARRAYS") != nullptr;
    k.wide_refine = getenv("PSACX_WIDE_REFINE") != nullptr;
    k.sort_debug = getenv("PSACX_SORT_DEBUG") != nullptr;
    return k;
}

// bump allocator over the ctx slab; a first pass with base == nullptr sizes it
struct Arena {
    char* base; size_t off;
    explicit Arena(char* b) : base(b), off(0) {}
    template <typename U> U* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        U* p = base ? reinterpret_cast<U*>(base + off) : nullptr;
        off += count * sizeof(U);
        return p;
    }
};

inline void pool_flush(psacx_ctx* c);

inline int ensure_slab(psacx_ctx* c, size_t bytes) {
    if (c->slab_bytes >= bytes) return PSACX_OK;
    if (c->slab) { (void)hipFree(c->slab); c->slab = nullptr; c->slab_bytes = 0; }
    hipError_t e = hipMalloc((void**)&c->slab, bytes);
    if (e != hipSuccess && c->pool && !c->pool->empty()) {       // the free blocks the multi-GPU path keeps cached go back first
        (void)hipGetLastError();
        pool_flush(c);
        e = hipMalloc((void**)&c->slab, bytes);
    }
    if (e != hipSuccess) {
        c->hip_err = std::string("hipMalloc(workspace): ") + hipGetErrorString(e);
        (void)hipGetLastError();
        return PSACX_ENOMEM;
    }
    c->slab_bytes = bytes;
    return PSACX_OK;
}

inline int ensure_io(psacx_ctx* c, size_t bytes) {
    if (c->io_bytes >= bytes && c->io_bytes / 4 <= bytes) return PSACX_OK;      // a much smaller request gives the rest back
    if (c->io) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->io); c->io = nullptr; c->io_bytes = 0; }
    hipError_t e = hipMalloc((void**)&c->io, bytes);
    if (e != hipSuccess) {
        c->hip_err = std::string("hipMalloc(io): ") + hipGetErrorString(e);
        (void)hipGetLastError();
        return PSACX_ENOMEM;
    }
    c->io_bytes = bytes;
    return PSACX_OK;
}

// Host <-> device copies of the host-pointer entry points.  hipMemcpy on pageable memory stages through the
// runtime's own bounce buffer with one copying thread (measured 9.5 GB/s device -> host); here the DMA engine
// fills one pinned buffer while a few host threads empty the other into the caller's memory.
constexpr size_t STAGE_CHUNK = (size_t)64 << 20;
constexpr int STAGE_THREADS = 8;

inline int ensure_stage(psacx_ctx* c) {
    if (c->stage[0]) return PSACX_OK;
    for (int i = 0; i < 2; ++i) {
        if (hipHostMalloc((void**)&c->stage[i], STAGE_CHUNK, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            c->hip_err = "pinned staging buffers could not be allocated";
            return PSACX_ENOMEM;
        }
    }
    c->stage_bytes = STAGE_CHUNK;
    return PSACX_OK;
}

inline void parallel_memcpy(char* dst, const char* src, size_t bytes) {
    const size_t min_per_thread = (size_t)4 << 20;
    int nt = (int)std::min<size_t>(STAGE_THREADS, (bytes + min_per_thread - 1) / min_per_thread);
    if (nt <= 1) { std::memcpy(dst, src, bytes); return; }
    const size_t per = ((bytes + nt - 1) / nt + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) {
        const size_t o = per * t;
        if (o >= bytes) break;
        th.emplace_back([=]() { std::memcpy(dst + o, src + o, std::min(per, bytes - o)); });
    }
    std::memcpy(dst, src, std::min(per, bytes));
    for (auto& x : th) x.join();
}

inline int staged_d2h(psacx_ctx* c, void* dst_, const void* src_, size_t bytes) {
    PSACX_TRY(ensure_stage(c));
    char* dst = static_cast<char*>(dst_); const char* src = static_cast<const char*>(src_);
    size_t issued = 0, drained = 0; int qi = 0, qd = 0;
    size_t len[2] = {0, 0};
    while (drained < bytes) {
        while (issued < bytes && len[qi] == 0) {
            const size_t m = std::min(STAGE_CHUNK, bytes - issued);
            PSACX_HIP(c, hipMemcpyAsync(c->stage[qi], src + issued, m, hipMemcpyDeviceToHost, c->stream));
            PSACX_HIP(c, hipEventRecord(c->stage_ev[qi], c->stream));
            len[qi] = m; issued += m; qi ^= 1;
        }
        PSACX_HIP(c, hipEventSynchronize(c->stage_ev[qd]));
        parallel_memcpy(dst + drained, c->stage[qd], len[qd]);
        drained += len[qd]; len[qd] = 0; qd ^= 1;
    }
    return PSACX_OK;
}

inline int staged_h2d(psacx_ctx* c, void* dst_, const void* src_, size_t bytes) {
    PSACX_TRY(ensure_stage(c));
    char* dst = static_cast<char*>(dst_); const char* src = static_cast<const char*>(src_);
    int q = 0; bool used[2] = {false, false};
    for (size_t off = 0; off < bytes; off += STAGE_CHUNK, q ^= 1) {
        const size_t m = std::min(STAGE_CHUNK, bytes - off);
        if (used[q]) PSACX_HIP(c, hipEventSynchronize(c->stage_ev[q]));       // the DMA out of this buffer is over
        parallel_memcpy(c->stage[q], src + off, m);
        PSACX_HIP(c, hipMemcpyAsync(dst + off, c->stage[q], m, hipMemcpyHostToDevice, c->stream));
        PSACX_HIP(c, hipEventRecord(c->stage_ev[q], c->stream));
        used[q] = true;
    }
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    return PSACX_OK;
}

inline void pool_flush(psacx_ctx* c) {
    if (!c->pool) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto& kv : *c->pool) (void)hipFree(kv.second);
    c->pool->clear();
    c->pool_bytes = 0;
}
// *cap receives the size of the block actually handed out (pass it back to pool_free)
inline void* pool_alloc(psacx_ctx* c, size_t bytes, size_t* cap) {
    if (!c->pool) c->pool = new std::multimap<size_t, void*>();
    const size_t want = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    auto it = c->pool->lower_bound(want);
    if (it != c->pool->end() && it->first <= want + want / 4 + 4096) {
        void* p = it->second; *cap = it->first;
        c->pool_bytes -= it->first;
        c->pool_live += it->first;
        c->pool_peak = std::max(c->pool_peak, c->pool_live);
        c->pool->erase(it);
        return p;
    }
    if (c->pool_cache_limit && c->pool_bytes > c->pool_cache_limit) pool_flush(c);
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError();
        pool_flush(c);                                   // give the cached blocks back and try once more
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    *cap = want;
    c->pool_live += want;
    c->pool_peak = std::max(c->pool_peak, c->pool_live);
    return p;
}
inline void pool_free(psacx_ctx* c, void* p, size_t cap) {
    if (!p) return;
    if (!c->pool) c->pool = new std::multimap<size_t, void*>();
    c->pool->emplace(cap, p);
    c->pool_bytes += cap;
    c->pool_live -= std::min(c->pool_live, cap);
}

inline int grid_for(const psacx_ctx* c, uint64_t work_items, int block, int per_cu = 8) {
    uint64_t want = (work_items + block - 1) / block;
    uint64_t cap = (uint64_t)c->n_cu * per_cu;
    if (want < 1) want = 1;
    return (int)std::min<uint64_t>(want, cap);
}

inline unsigned bits_for(uint64_t max_value) {      // bits needed to hold values 0..max_value
    unsigned b = 0;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b ? b : 1;
}

// ----------------------------------------------------------------------------
// Rank-pair sorter
// ----------------------------------------------------------------------------
template <typename T> struct SortBufs { T* k1; T* k2; T* v; };

struct SortScratch {
    unsigned long long* d_hist;    // [MAX_PASSES][RADIX]
    unsigned long long* d_base;    // [MAX_PASSES][RADIX]
    char* d_desc;                  // counter (256 B) + descriptors
    size_t desc_bytes;
    unsigned* d_err;
    unsigned long long* h_hist;    // pinned
    unsigned long long* h_base;    // pinned
    unsigned long long* d_dbg;     // phase stamps of sampled tiles (PSACX_SORT_DEBUG), may be null
    unsigned long long* d_summary; // OR/AND of the keys (see key_summary_add), 4 words
    unsigned long long* h_summary; // pinned, 4 words
    unsigned long long* d_partials;// per-workgroup key summaries of the producer kernel
};

constexpr int SORT_TILE_MIN = 2048;   // smallest tile of any scatter configuration
constexpr uint64_t SMALL_SORT_MAX = 1ull << 21;   // below: single-sweep scatter passes with decoupled look-back

inline size_t sort_desc_bytes(uint64_t n) {
    const uint64_t nt = (n + SORT_TILE_MIN - 1) / SORT_TILE_MIN + 1;
    // look-back descriptors, or (three-kernel form) per-tile counters + per-slab totals
    size_t b = 1024 + nt * RADIX * sizeof(uint64_t) + (nt / SLAB_TILES + 2) * RADIX * sizeof(uint64_t);
    // small sorts (look-back form): one descriptor region per pass, so that one memset serves the whole sort
    if (n < SMALL_SORT_MAX) b = std::max<size_t>(b, (size_t)MAX_PASSES * (512 + nt * RADIX * sizeof(uint32_t)));
    return b;
}

inline bool sort_host_scan_env() { static const bool on = getenv("PSACX_SORT_HOST_SCAN") != nullptr; return on; }

inline unsigned sort_chunk_env() {   // tiles per XCD-local chunk (0 = plain ticket order)
    static int v = -2;
    if (v == -2) { const char* e = getenv("PSACX_SORT_CHUNK"); v = e ? atoi(e) : -1; }
    return (unsigned)v;
}
inline unsigned sort_chunk_for(uint64_t n, bool three) {
    const unsigned e = sort_chunk_env();
    if (e != (unsigned)-1) return e;
    if (n < (1ull << 21)) return 0;          // few tiles: plain start order
    return three ? 64u : 16u;
}

template <typename T, typename D, int BLOCK, int ITEMS>
inline void launch_scatter(psacx_ctx* c, const T* kd_in, const T* ko_in, const T* v_in, T* kd_out,
                           T* ko_out, T* v_out, uint64_t n, int shift, const unsigned long long* base,
                           char* desc, unsigned* err, unsigned long long* dbg, uint64_t spec, uint64_t spec_n) {
    constexpr int TILE = BLOCK * ITEMS;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
    hipLaunchKernelGGL((radix_scatter_kernel<T, D, BLOCK, ITEMS>), dim3((unsigned)ntiles), dim3(BLOCK), 0,
                       c->stream, kd_in, ko_in, v_in, kd_out, ko_out, v_out, n, shift, base,
                       reinterpret_cast<D*>(desc + 256), reinterpret_cast<unsigned*>(desc), err, dbg, spec, spec_n,
                       sort_chunk_for(n, false));
}

template <typename T> struct ScatterCfg;
// DEF: three-word records, DEF2: two-word records.  Measured per word size: uint32 512 x 12; uint64 512 x 8
// (two-word: 3.8 ms per 2^29-record pass against 5.2 ms with 256 x 8; three-word: 2.35 against 2.60 ms at 2^28,
// 22 against 32 ms at 2^31, 47 against 78 ms at 2^32 where the smaller tiles hit a stride artefact)
template <> struct ScatterCfg<uint32_t> { static constexpr int DEF = 7; static constexpr int DEF2 = 7; };
template <> struct ScatterCfg<uint64_t> { static constexpr int DEF = 2; static constexpr int DEF2 = 2; };

inline int sort_cfg_env() {
    static int v = -2;
    if (v == -2) { const char* e = getenv("PSACX_SORT_CFG"); v = e ? atoi(e) : -1; }
    return v;
}

// Packed payload of the first round's prefix sort (radix.hpp: VN 3 .. 6): the low `bits` bits of word 1 lie below the sorted
// prefix and carry the low bits of the suffix a record stands for, `bytes` (1 or 2) = size of the entries that hold the rest.
// bytes == 0: not packed.
struct PackedForm {
    unsigned bits = 0, bytes = 0;
    bool local = true;      // records that arrive packed stay packed between the passes of the local sort (else its first pass widens them)
    bool on() const { return bytes != 0; }
};
// the form for payloads below `count` when the low lo1 bits of a 64-bit word 1 are free; none when more than 16 bits remain.
// Measured on one GPU (4 GiB DNA, uint64, profiles/r03c_*): a pass over (8-byte word, 1-byte entry) records takes 32.4 ms
// against 28.0 ms for (8-byte word, 32-bit entry) records although it moves 18 instead of 24 bytes per record, and 33.5 ms
// with 2-byte entries (profiles/r03d_*): entries narrower than 32 bits cost more than they save in this scatter pattern --
// so the one-GPU engine keeps its 32-bit payloads and the packed form is for the multi-GPU shuffle of texts beyond 2^32
// characters, where the alternative is a 64-bit payload on the wire; there the first pass of the local sort widens the
// entries again (two ranks of 2^31 + 2^20 characters: local sort 214 ms packed throughout against 184 ms)
// (PSACX_PACKED=1 / 0 forces the form on / off wherever it applies, PSACX_PACKED_LOCAL=1 keeps the local passes packed).
inline PackedForm packed_form_for(uint64_t count, unsigned lo1, size_t word_bytes, bool by_default) {
    PackedForm pf;
    const char* e = getenv("PSACX_PACKED");
    const bool on = e ? atoi(e) != 0 : by_default;
    if (!on || word_bytes != 8 || lo1 == 0) return pf;
    const unsigned need = bits_for(count > 1 ? count - 1 : 1);
    const unsigned rest = need > lo1 ? need - lo1 : 0;
    if (rest > 16) return pf;
    pf.bits = lo1 < 63 ? lo1 : 63; pf.bytes = rest > 8 ? 2 : 1;
    if (const char* b = getenv("PSACX_PACKED_BYTES")) if (atoi(b) == 2) pf.bytes = 2;      // (measurements: 16-bit entries where 8 would do)
    pf.local = getenv("PSACX_PACKED_LOCAL") != nullptr;
    return pf;
}

struct ScatterShape { int block, items; };
// scatter configurations selectable with PSACX_SORT_CFG (tuning aid)
static const ScatterShape kShapes[] = {{256, 8}, {256, 16}, {512, 8}, {512, 16}, {256, 12}, {1024, 4}, {1024, 8}, {512, 12}};
constexpr int N_SHAPES = 8;   // (register caps through __launch_bounds__ were measured: spills cost 1.6-3x)

inline uint64_t cfg_tile(int cfg) {
    if (cfg < 0 || cfg >= N_SHAPES) cfg = 1;
    return (uint64_t)kShapes[cfg].block * kShapes[cfg].items;
}

template <typename T, typename D>
inline void dispatch_scatter(psacx_ctx* c, int cfg, const T* kd_in, const T* ko_in, const T* v_in,
                             T* kd_out, T* ko_out, T* v_out, uint64_t n, int shift,
                             const unsigned long long* base, char* desc, unsigned* err, unsigned long long* dbg,
                             uint64_t spec, uint64_t spec_n) {
#define PSACX_SC(B, I) launch_scatter<T, D, B, I>(c, kd_in, ko_in, v_in, kd_out, ko_out, v_out, n, shift, base, desc, err, dbg, spec, spec_n)
    switch (cfg) {
        case 0: PSACX_SC(256, 8); break;
        case 2: PSACX_SC(512, 8); break;
        case 3: PSACX_SC(512, 16); break;
        case 4: PSACX_SC(256, 12); break;
        case 5: PSACX_SC(1024, 4); break;
        case 6: PSACX_SC(1024, 8); break;
        case 7: PSACX_SC(512, 12); break;
        case 1:
        default: PSACX_SC(256, 16); break;
    }
#undef PSACX_SC
}

inline int sort_mode_env() {     // 0 = single-sweep with look-back, 1 = three kernels per pass
    static int v = -2;
    if (v == -2) { const char* e = getenv("PSACX_SORT_MODE"); v = e ? atoi(e) : -1; }
    return v;
}

template <typename T, int BLOCK, int ITEMS, int MINW = 1>
inline void launch_pass3(psacx_ctx* c, const T* kd_in, const T* ko_in, const T* v_in, T* kd_out, T* ko_out, T* v_out,
                         uint64_t n, int shift, const unsigned long long* base, char* scratch,
                         unsigned long long* dbg, uint64_t spec, uint64_t spec_n, bool have_hist = false, int vn = 0, unsigned pack = 0) {
    constexpr int TILE = BLOCK * ITEMS;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
    const unsigned slab_tiles = slab_tiles_for(ntiles);
    const uint64_t nslabs = (ntiles + slab_tiles - 1) / slab_tiles;
    unsigned* tile_hist = reinterpret_cast<unsigned*>(scratch + 256);
    unsigned long long* slab_tot = reinterpret_cast<unsigned long long*>(scratch + 256 + ((ntiles * RADIX * sizeof(unsigned) + 255) & ~(size_t)255));
    {
        ProfScope ps(c, TC_SORT_TILEHIST);
        if (!have_hist)        // (the producer of the keys may have left this pass's tile histograms in place)
            hipLaunchKernelGGL((radix_tile_hist_kernel<T, BLOCK, ITEMS>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in, n,
                               shift, tile_hist);
        hipLaunchKernelGGL(radix_slab_scan_kernel<0>, dim3((unsigned)nslabs), dim3(RADIX), 0, c->stream, tile_hist, ntiles, slab_tot, slab_tiles);
        hipLaunchKernelGGL(radix_top_scan_kernel<0>, dim3(1), dim3(RADIX), 0, c->stream, slab_tot, nslabs,
                           const_cast<unsigned long long*>(base));
    }
    ProfScope ps(c, ko_in ? TC_SORT_SCATTER3 : TC_SORT_SCATTER2);
    constexpr bool DEF_SHAPE = BLOCK == 512 && ITEMS == (sizeof(T) == 4 ? 12 : 8);
    constexpr bool NARROW_OK = DEF_SHAPE && sizeof(T) == 8;       // 32-bit payload arrays (radix.hpp: VN), default shape only (8192-record tiles measured: 33.6-34.9 against 28.1 ms per pass)
    if (NARROW_OK && vn && !ko_in) {
        // (radix.hpp: VN -- 1 / 2: 32-bit payload entries; 3 .. 6: payload packed into the low bits of the key word + 8- or 16-bit entries)
        // registers capped for six waves per SIMD = three workgroups per CU: the narrow forms need 82, the cap costs them a
        // few spilled registers and gains a third tile in flight (the pass is bound by the latency chain of a tile)
#define PSACX_VN(V)                                                                                                                          \
        hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, false, (NARROW_OK ? 6 : MINW), true, NARROW_OK ? V : 0>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in, \
                           ko_in, v_in, kd_out, ko_out, v_out, n, shift, base, tile_hist, slab_tot, dbg, spec, spec_n,                    \
                           reinterpret_cast<unsigned*>(scratch), sort_chunk_for(n, true), (const T*)nullptr, slab_tiles, (uint64_t)0, pack)
        switch (vn) {
            case 1: PSACX_VN(1); break;
            case 2: PSACX_VN(2); break;
            case 3: PSACX_VN(3); break;
            case 4: PSACX_VN(4); break;
            case 5: PSACX_VN(5); break;
            default: PSACX_VN(6); break;
        }
#undef PSACX_VN
        return;
    }
    if (ko_in)
        hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, false, MINW>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in,
                           ko_in, v_in, kd_out, ko_out, v_out, n, shift, base, tile_hist, slab_tot, dbg, spec, spec_n,
                           reinterpret_cast<unsigned*>(scratch), sort_chunk_for(n, true), (const T*)nullptr, slab_tiles);
    else            // two-word records (k1, v): the prefix sort of the first round
        hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, false, MINW, true>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in,
                           ko_in, v_in, kd_out, ko_out, v_out, n, shift, base, tile_hist, slab_tot, dbg, spec, spec_n,
                           reinterpret_cast<unsigned*>(scratch), sort_chunk_for(n, true), (const T*)nullptr, slab_tiles);
}

template <typename T>
inline void dispatch_pass3(psacx_ctx* c, int cfg, const T* kd_in, const T* ko_in, const T* v_in, T* kd_out, T* ko_out,
                           T* v_out, uint64_t n, int shift, const unsigned long long* base, char* scratch,
                           unsigned long long* dbg, uint64_t spec, uint64_t spec_n, bool have_hist = false, int vn = 0, unsigned pack = 0) {
#define PSACX_P3(B, I) launch_pass3<T, B, I>(c, kd_in, ko_in, v_in, kd_out, ko_out, v_out, n, shift, base, scratch, dbg, spec, spec_n, have_hist, vn, pack)
    switch (cfg) {
        case 0: PSACX_P3(256, 8); break;
        case 2: PSACX_P3(512, 8); break;
        case 3: PSACX_P3(512, 16); break;
        case 4: PSACX_P3(256, 12); break;
        case 5: PSACX_P3(1024, 4); break;
        case 6: PSACX_P3(1024, 8); break;
        case 7: PSACX_P3(512, 12); break;
        case 1:
        default: PSACX_P3(256, 16); break;
    }
#undef PSACX_P3
}

// One pass that groups records by an externally supplied 8-bit class (cls[i] < 256), stable.
// Result in `out`; class_start_host[0..256] receives the start of every class (host array).
template <typename T>
int class_partition(psacx_ctx* c, SortScratch& sc, SortBufs<T> in, SortBufs<T> out, const T* cls, uint64_t n,
                    unsigned long long* class_start_host) {
    constexpr int BLOCK = 512, ITEMS = 12, TILE = BLOCK * ITEMS;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
    const unsigned slab_tiles = slab_tiles_for(ntiles);
    const uint64_t nslabs = (ntiles + slab_tiles - 1) / slab_tiles;
    char* scratch = sc.d_desc;
    unsigned* tile_hist = reinterpret_cast<unsigned*>(scratch + 256);
    unsigned long long* slab_tot = reinterpret_cast<unsigned long long*>(scratch + 256 + ((ntiles * RADIX * sizeof(unsigned) + 255) & ~(size_t)255));
    PSACX_HIP(c, hipMemsetAsync(scratch, 0, 256, c->stream));
    hipLaunchKernelGGL((radix_tile_hist_kernel<T, BLOCK, ITEMS>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, cls, n, 0, tile_hist);
    hipLaunchKernelGGL(radix_slab_scan_kernel<0>, dim3((unsigned)nslabs), dim3(RADIX), 0, c->stream, tile_hist, ntiles, slab_tot, slab_tiles);
    hipLaunchKernelGGL(radix_top_scan_kernel<0>, dim3(1), dim3(RADIX), 0, c->stream, slab_tot, nslabs, sc.d_base);
    if (in.k2)
        hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, true>), dim3((unsigned)ntiles), dim3(BLOCK), 0,
                           c->stream, in.k1, in.k2, in.v, out.k1, out.k2, out.v, n, 0, sc.d_base, tile_hist, slab_tot,
                           (unsigned long long*)nullptr, (uint64_t)0, (uint64_t)0, reinterpret_cast<unsigned*>(scratch),
                           sort_chunk_for(n, true), cls, slab_tiles);
    else            // two-word records (k1, v): routing (position, value) pairs to their owners
        hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, true, 1, true>), dim3((unsigned)ntiles), dim3(BLOCK), 0,
                           c->stream, in.k1, in.k2, in.v, out.k1, out.k2, out.v, n, 0, sc.d_base, tile_hist, slab_tot,
                           (unsigned long long*)nullptr, (uint64_t)0, (uint64_t)0, reinterpret_cast<unsigned*>(scratch),
                           sort_chunk_for(n, true), cls, slab_tiles);
    PSACX_HIP(c, hipGetLastError());
    PSACX_HIP(c, hipMemcpyAsync(class_start_host, sc.d_base, RADIX * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    class_start_host[RADIX] = n;
    return PSACX_OK;
}

// The shuffle pass of the two-word first round (multi.hpp: sort_first_two_word): the records k1[0 .. n) of one piece of a
// rank's block are grouped by the destination in the byte array cls (stable), their payload -- the suffix a record stands
// for -- is made up on the way (spec / spec_n / voff as in radix_scatter_tile) and leaves as 32-bit entries when v32.
// The per-class totals of the piece are known already (classify_prefix_kernel), so nothing comes back to the host.
template <typename T>
int piece_partition(psacx_ctx* c, char* scratch, unsigned long long* d_base, const T* k1, const uint8_t* cls, uint64_t n, T* k1_out, void* v_out,
                    bool v32, uint64_t spec, uint64_t spec_n, uint64_t voff, PackedForm pf = PackedForm()) {
    constexpr int BLOCK = 512, ITEMS = sizeof(T) == 4 ? 12 : 8, TILE = BLOCK * ITEMS;
    if (n == 0) return PSACX_OK;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
    const unsigned slab_tiles = slab_tiles_for(ntiles);
    const uint64_t nslabs = (ntiles + slab_tiles - 1) / slab_tiles;
    unsigned* tile_hist = reinterpret_cast<unsigned*>(scratch + 256);
    unsigned long long* slab_tot = reinterpret_cast<unsigned long long*>(scratch + 256 + ((ntiles * RADIX * sizeof(unsigned) + 255) & ~(size_t)255));
    PSACX_HIP(c, hipMemsetAsync(scratch, 0, 256, c->stream));
    hipLaunchKernelGGL((class_tile_hist_kernel<BLOCK, ITEMS>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, cls, n, tile_hist);
    hipLaunchKernelGGL(radix_slab_scan_kernel<0>, dim3((unsigned)nslabs), dim3(RADIX), 0, c->stream, tile_hist, ntiles, slab_tot, slab_tiles);
    hipLaunchKernelGGL(radix_top_scan_kernel<0>, dim3(1), dim3(RADIX), 0, c->stream, slab_tot, nslabs, d_base);
    const T* dsrc = reinterpret_cast<const T*>(cls);
    if (sizeof(T) == 8 && pf.bytes == 1)
        hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, true, 1, true, (sizeof(T) == 8 ? 3 : 0), 1>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, k1,
                           (const T*)nullptr, (const T*)nullptr, k1_out, (T*)nullptr, static_cast<T*>(v_out), n, 0, d_base, tile_hist, slab_tot,
                           (unsigned long long*)nullptr, spec, spec_n, reinterpret_cast<unsigned*>(scratch), sort_chunk_for(n, true), dsrc, slab_tiles, voff, pf.bits);
    else if (sizeof(T) == 8 && pf.bytes == 2)
        hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, true, 1, true, (sizeof(T) == 8 ? 5 : 0), 1>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, k1,
                           (const T*)nullptr, (const T*)nullptr, k1_out, (T*)nullptr, static_cast<T*>(v_out), n, 0, d_base, tile_hist, slab_tot,
                           (unsigned long long*)nullptr, spec, spec_n, reinterpret_cast<unsigned*>(scratch), sort_chunk_for(n, true), dsrc, slab_tiles, voff, pf.bits);
    else if (sizeof(T) == 8 && v32)
        hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, true, 1, true, (sizeof(T) == 8 ? 1 : 0), 1>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, k1,
                           (const T*)nullptr, (const T*)nullptr, k1_out, (T*)nullptr, static_cast<T*>(v_out), n, 0, d_base, tile_hist, slab_tot,
                           (unsigned long long*)nullptr, spec, spec_n, reinterpret_cast<unsigned*>(scratch), sort_chunk_for(n, true), dsrc, slab_tiles, voff);
    else
        hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, true, 1, true, 0, 1>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, k1,
                           (const T*)nullptr, (const T*)nullptr, k1_out, (T*)nullptr, static_cast<T*>(v_out), n, 0, d_base, tile_hist, slab_tot,
                           (unsigned long long*)nullptr, spec, spec_n, reinterpret_cast<unsigned*>(scratch), sort_chunk_for(n, true), dsrc, slab_tiles, voff);
    PSACX_HIP(c, hipGetLastError());
    return PSACX_OK;
}

// whether a sort of n records runs its passes in the three-kernel form (else: single sweep with look-back).
// Default: three kernels for large inputs (no workgroup ever waits on another), look-back for small ones where the
// launch count matters more; records without a second key word exist only in the three-kernel form.
inline bool sort_is_three(uint64_t n, bool has_k2) {
    return !has_k2 || (sort_mode_env() >= 0 ? sort_mode_env() == 1 : n >= SMALL_SORT_MAX);
}

// folds the per-workgroup key summaries a producer kernel left in sc.d_partials into sc.d_summary
inline int summary_finish(psacx_ctx* c, SortScratch& sc, unsigned nblocks) {
    hipLaunchKernelGGL(summary_reduce_kernel<0>, dim3(1), dim3(1024), 0, c->stream, sc.d_partials, nblocks, sc.d_summary);
    PSACX_HIP(c, hipGetLastError());
    return PSACX_OK;
}

// Sorts `n` records by (k1, k2); bits1/bits2 = significant low bits of each word.
// With `iota` the payload read by the first pass is the record index, or, when spec_n is
// set, the suffix start the first-round record stands for (in.v is only scratch).  The sorted arrays end up in
// *res (the `in` or the `alt` set); when final_v is given the payload of the last
// executed pass is written there instead and res->v == final_v.
template <typename T>
int pair_sort(psacx_ctx* c, SortScratch& sc, SortBufs<T> in, SortBufs<T> alt, uint64_t n, bool iota,
              unsigned bits1, unsigned bits2, T* final_v, SortBufs<T>* res, psacx_round* rs,
              uint64_t spec = 0, uint64_t spec_n = 0, bool summary_ready = false, unsigned lo1 = 0,
              int ready_hist_shift = -1, bool v32_in = false, PackedForm pf = PackedForm(), bool packed_in = false, bool* ran_packed = nullptr) {
    // pf.on(): two-word records of 64-bit words whose payload travels packed (radix.hpp: VN 3 .. 6): made up and packed by
    // the first pass (iota), or arriving that way in in.k1 / in.v (packed_in: after the multi-GPU shuffle); put together
    // again by the last pass
    // v32_in (64-bit words, two-word records): in.v holds 32-bit entries (payloads below 2^32 that arrived that way, the
    // suffixes of a text of at most 2^32 characters after the multi-GPU shuffle); they stay 32-bit between the passes and the
    // last pass widens them, as for a payload the first pass makes up
    // ready_hist_shift >= 0: the tile histograms of word 1 at that bit position are already in the scratch
    // (written by key_pairs_kernel<..., HIST> with the tile shape of this sort)
    if (bits1 > sizeof(T) * 8) bits1 = sizeof(T) * 8;
    if (bits2 > sizeof(T) * 8) bits2 = sizeof(T) * 8;
    if (!in.k2) bits2 = 0;
    const PassPlan plan = make_plan((int)bits1, (int)bits2, (int)lo1);
    // default: three-kernel passes (no workgroup ever waits on another) for large inputs, the
    // single-sweep look-back form for small ones where launch count matters more
    // (records without a second key word exist only in the three-kernel form)
    const bool three = sort_is_three(n, in.k2 != nullptr);
    bool skip[MAX_PASSES];
    int n_exec = 0;
    // look-back form: digit starts scanned on the device, one descriptor region per pass (zeroed by one memset together
    // with the histograms); the host only learns which passes have a constant digit (flags the scan kernel stores into
    // pinned host memory).  Needs the scratch laid out as carve() lays it out.
    int cfg = sort_cfg_env();
    if (cfg < 0) cfg = in.k2 ? ScatterCfg<T>::DEF : ScatterCfg<T>::DEF2;
    // (2048-record tiles for the small sorts were measured: three times the look-back chain, 0.27 -> 0.29 ms per round at 2^20)
    const bool small_desc = n < (1ull << 30);
    const size_t hist_bytes = sizeof(unsigned long long) * MAX_PASSES * RADIX;
    const size_t desc_stride = (256 + ((n + cfg_tile(cfg) - 1) / cfg_tile(cfg)) * RADIX * (small_desc ? sizeof(uint32_t) : sizeof(uint64_t)) + 255) & ~(size_t)255;
    const bool dev_scan = !three && !sort_host_scan_env() && !sc.d_dbg && c->pinned_dev && c->pinned_bytes >= 512 &&
                          reinterpret_cast<char*>(sc.d_base) == reinterpret_cast<char*>(sc.d_hist) + hist_bytes &&
                          sc.d_desc == reinterpret_cast<char*>(sc.d_base) + hist_bytes &&
                          (size_t)plan.n_pass * desc_stride <= sc.desc_bytes;
    if (three) {
        // constant digits from the OR/AND summary of the keys; digit starts come from each pass's own scan
        if (!summary_ready) {
            ProfScope ps(c, TC_SORT_HIST);
            const int g = grid_for(c, n, 256, 8);
            hipLaunchKernelGGL((key_summary_kernel<T>), dim3(g), dim3(256), 0, c->stream, in.k1, in.k2, n, sc.d_partials);
            PSACX_HIP(c, hipGetLastError());
            PSACX_TRY(summary_finish(c, sc, (unsigned)g));
            c->stats.hist_bytes += 2ull * sizeof(T) * n;
        }
        PSACX_HIP(c, hipMemcpyAsync(sc.h_summary, sc.d_summary, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        PSACX_HIP(c, hipStreamSynchronize(c->stream));
        for (int p = 0; p < plan.n_pass; ++p) {
            const unsigned long long diff = plan.word[p] ? (sc.h_summary[2] ^ sc.h_summary[3]) : (sc.h_summary[0] ^ sc.h_summary[1]);
            skip[p] = ((diff >> plan.shift[p]) & (RADIX - 1)) == 0;
            if (!skip[p]) ++n_exec;
        }
    } else if (dev_scan) {
        HistArgs ha;
        ha.n_pass = plan.n_pass;
        for (int p = 0; p < plan.n_pass; ++p) { ha.word[p] = plan.word[p]; ha.shift[p] = plan.shift[p]; }
        {
            ProfScope ps(c, TC_SORT_HIST);
            PSACX_HIP(c, hipMemsetAsync(sc.d_hist, 0, 2 * hist_bytes + (size_t)plan.n_pass * desc_stride, c->stream));
            const int grid = grid_for(c, (n + 3) / 4, 256, 8);
            hipLaunchKernelGGL((radix_hist_kernel<T, 256>), dim3(grid), dim3(256), 0, c->stream, in.k1, in.k2, n, ha, sc.d_hist);
            hipLaunchKernelGGL(radix_hist_scan_kernel<0>, dim3(plan.n_pass), dim3(RADIX), 0, c->stream, sc.d_hist, sc.d_base,
                               (unsigned long long)n, reinterpret_cast<unsigned*>(c->pinned_dev + 384));
            PSACX_HIP(c, hipGetLastError());
        }
        PSACX_HIP(c, hipStreamSynchronize(c->stream));
        const unsigned* constant = reinterpret_cast<const unsigned*>(c->pinned + 384);
        for (int p = 0; p < plan.n_pass; ++p) { skip[p] = constant[p] != 0; if (!skip[p]) ++n_exec; }
        c->stats.hist_bytes += 2ull * sizeof(T) * n;
    } else {
        HistArgs ha;
        ha.n_pass = plan.n_pass;
        for (int p = 0; p < plan.n_pass; ++p) { ha.word[p] = plan.word[p]; ha.shift[p] = plan.shift[p]; }
        {
            ProfScope ps(c, TC_SORT_HIST);
            PSACX_HIP(c, hipMemsetAsync(sc.d_hist, 0, sizeof(unsigned long long) * MAX_PASSES * RADIX, c->stream));
            const int grid = grid_for(c, (n + 3) / 4, 256, 8);
            hipLaunchKernelGGL((radix_hist_kernel<T, 256>), dim3(grid), dim3(256), 0, c->stream, in.k1, in.k2, n,
                               ha, sc.d_hist);
            PSACX_HIP(c, hipGetLastError());
        }
        PSACX_HIP(c, hipMemcpyAsync(sc.h_hist, sc.d_hist, sizeof(unsigned long long) * plan.n_pass * RADIX,
                                    hipMemcpyDeviceToHost, c->stream));
        PSACX_HIP(c, hipStreamSynchronize(c->stream));
        c->stats.hist_bytes += 2ull * sizeof(T) * n;
        for (int p = 0; p < plan.n_pass; ++p) {
            const unsigned long long* h = sc.h_hist + (size_t)p * RADIX;
            unsigned long long run = 0;
            skip[p] = false;
            for (int d = 0; d < RADIX; ++d) {
                if (h[d] == n) skip[p] = true;
                sc.h_base[(size_t)p * RADIX + d] = run;
                run += h[d];
            }
            if (!skip[p]) ++n_exec;
        }
        if (n_exec) {
            PSACX_HIP(c, hipMemcpyAsync(sc.d_base, sc.h_base, sizeof(unsigned long long) * plan.n_pass * RADIX,
                                        hipMemcpyHostToDevice, c->stream));
        }
    }
    if (rs) { rs->sort_passes = (uint32_t)n_exec; rs->sort_passes_skipped = (uint32_t)(plan.n_pass - n_exec); }

    // two-word records of 64-bit words whose payload is made up by the first pass (suffix indices < n <= 2^32): the payload
    // travels as 32-bit entries between the passes and is widened by the last one (radix.hpp: VN)
    const bool narrow = three && sizeof(T) == 8 && !in.k2 && ((iota && n <= (1ull << 32)) || v32_in) && cfg == ScatterCfg<T>::DEF2;
    if (v32_in && !narrow) return PSACX_EINVAL;
    const bool packed = pf.on() && three && sizeof(T) == 8 && !in.k2 && (iota || packed_in) && cfg == ScatterCfg<T>::DEF2 && pf.bits <= lo1;
    if (packed_in && !packed) return PSACX_EINVAL;
    // (a sort of one executed pass makes its payload up in full and leaves word 1 as it is)
    if (ran_packed) *ran_packed = packed && (packed_in || n_exec > 1);
    SortBufs<T> cur = in, oth = alt;
    int done = 0;
    for (int p = 0; p < plan.n_pass; ++p) {
        if (skip[p]) continue;
        const bool first = (done == 0);
        ++done;
        const bool last = (done == n_exec);
        const T* kd_in = plan.word[p] ? cur.k2 : cur.k1;
        const T* ko_in = plan.word[p] ? cur.k1 : cur.k2;
        const T* v_in = (first && iota) ? nullptr : cur.v;
        T* kd_out = plan.word[p] ? oth.k2 : oth.k1;
        T* ko_out = plan.word[p] ? oth.k1 : oth.k2;
        T* v_out = (last && final_v) ? final_v : oth.v;
        const uint64_t tile = cfg_tile(cfg);
        const uint64_t ntiles = (n + tile - 1) / tile;
        const size_t dbytes = 256 + ntiles * RADIX * (small_desc ? sizeof(uint32_t) : sizeof(uint64_t));
        char* const desc = dev_scan ? sc.d_desc + (size_t)(done - 1) * desc_stride : sc.d_desc;
        if (!dev_scan) PSACX_HIP(c, hipMemsetAsync(sc.d_desc, 0, three ? 256 : dbytes, c->stream));
        if (three) {
            const unsigned long long* base = sc.d_base + (size_t)p * RADIX;
            const bool have_hist = first && plan.word[p] == 0 && plan.shift[p] == ready_hist_shift && sort_cfg_env() < 0;
            int vn = narrow ? (last ? ((first && !v32_in) ? 0 : 2) : 1) : 0;
            const bool widen_first = packed && packed_in && !pf.local;      // packed on the wire only
            if (widen_first) vn = first ? (pf.bytes == 1 ? 4 : 6) : 0;
            else if (packed) vn = last ? ((first && !packed_in) ? 0 : (pf.bytes == 1 ? 4 : 6)) : (pf.bytes == 1 ? 3 : 5);
            dispatch_pass3<T>(c, cfg, kd_in, ko_in, v_in, kd_out, ko_out, v_out, n, plan.shift[p], base, sc.d_desc, sc.d_dbg, spec, spec_n,
                              have_hist, vn, pf.bits);
            PSACX_HIP(c, hipGetLastError());
        } else {
            ProfScope ps(c, TC_SORT_SCATTER);
            const unsigned long long* base = sc.d_base + (size_t)p * RADIX;
            if (small_desc)
                dispatch_scatter<T, uint32_t>(c, cfg, kd_in, ko_in, v_in, kd_out, ko_out, v_out, n, plan.shift[p], base, desc, sc.d_err, sc.d_dbg, spec, spec_n);
            else
                dispatch_scatter<T, uint64_t>(c, cfg, kd_in, ko_in, v_in, kd_out, ko_out, v_out, n, plan.shift[p], base, desc, sc.d_err, sc.d_dbg, spec, spec_n);
            PSACX_HIP(c, hipGetLastError());
        }
        if (sc.d_dbg && ntiles >= 64) {
            // tuning aid: average shader-clock span of each phase over the sampled tiles
            const size_t ns = (size_t)(ntiles / 64);
            std::vector<unsigned long long> h(ns * 8);
            PSACX_HIP(c, hipMemcpyAsync(h.data(), sc.d_dbg, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
            PSACX_HIP(c, hipStreamSynchronize(c->stream));
            double acc[6] = {0, 0, 0, 0, 0, 0};
            for (size_t i = 0; i < ns; ++i) for (int q = 0; q < 6; ++q) acc[q] += (double)(h[i * 8 + q + 1] - h[i * 8 + q]);
            fprintf(stderr, "[psacx sort dbg] pass %d n=%llu tiles=%llu cycles/tile: load+rank %.0f scan %.0f lookback %.0f key %.0f key2 %.0f val %.0f\n",
                    p, (unsigned long long)n, (unsigned long long)ntiles, acc[0] / ns, acc[1] / ns, acc[2] / ns, acc[3] / ns, acc[4] / ns, acc[5] / ns);
        }
        const int form = !in.k2 ? 2 : (three ? 1 : 0);
        c->stats.scatter_launches[form] += 1;
        c->stats.scatter_records[form] += n;
        // words read + written per record; a pass that makes up its payload (iota) reads one word less
        if (packed && packed_in && !pf.local) c->stats.scatter_bytes[form] += (2ull * sizeof(T) + (first ? (uint64_t)pf.bytes : sizeof(T)) + sizeof(T)) * n;
        else if (packed) c->stats.scatter_bytes[form] += (2ull * sizeof(T) + (v_in ? (uint64_t)pf.bytes : 0ull) + (last ? sizeof(T) : (uint64_t)pf.bytes)) * n;
        else if (narrow) c->stats.scatter_bytes[form] += (2ull * sizeof(T) + (v_in ? ((first && !v32_in) ? sizeof(T) : 4ull) : 0ull) + (last ? sizeof(T) : 4ull)) * n;
        else c->stats.scatter_bytes[form] += ((in.k2 ? 6ull : 4ull) - (v_in ? 0ull : 1ull)) * sizeof(T) * n;
        std::swap(cur, oth);
        cur.v = v_out;
    }
    if (n_exec == 0) {
        // every digit constant: the input order is already sorted
        T* dst = final_v ? final_v : cur.v;
        if (iota) {
            hipLaunchKernelGGL((iota_kernel<T>), dim3(grid_for(c, n, 256)), dim3(256), 0, c->stream, dst, n, spec, spec_n);
            PSACX_HIP(c, hipGetLastError());
        } else if (packed_in) {
            T* const w = (dst == cur.v) ? oth.v : dst;
            hipLaunchKernelGGL((unpack_payload_kernel<T>), dim3(grid_for(c, n, 256)), dim3(256), 0, c->stream, (const T*)cur.k1, (const void*)cur.v, n, pf.bits, pf.bytes, w);
            PSACX_HIP(c, hipGetLastError());
            if (w != dst) PSACX_HIP(c, hipMemcpyAsync(dst, w, n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        } else if (v32_in) {
            // (32-bit entries in, words out: through the other payload array when the widening would run in place)
            T* const w = (dst == cur.v) ? oth.v : dst;
            hipLaunchKernelGGL((widen32_kernel<T>), dim3(grid_for(c, n, 256)), dim3(256), 0, c->stream, reinterpret_cast<const uint32_t*>(cur.v), n, w);
            PSACX_HIP(c, hipGetLastError());
            if (w != dst) PSACX_HIP(c, hipMemcpyAsync(dst, w, n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        } else if (dst != cur.v) {
            PSACX_HIP(c, hipMemcpyAsync(dst, cur.v, n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        }
        cur.v = dst;
    }
    *res = cur;
    return PSACX_OK;
}

// ----------------------------------------------------------------------------
// Prefix sort of the first round in one-word records, most significant digit first (radix.hpp: VN 7 .. 9, *1w kernels).
// The two-stage first round sorts (word 1, suffix) on the `lead` bits of word 1 above bit lo1 and reads nothing below them
// afterwards (ties get their windows from the text again).  With lead - 8 <= 32 and suffixes below 2^32 a record fits ONE
// 64-bit word once the top digit of the prefix is known from the record's place: pass 0 partitions k1 by the top digit and writes
// (rest of the prefix) << 32 | suffix; the remaining digits are LSD passes inside the 256 buckets, all buckets in one launch; the
// last of them writes word 1 (prefix << lo1, low bits zero: the packed form of the ties machinery) and the suffixes as words.
// Bytes per record: 16 + (lead / 8 - 2) x 16 + 24 instead of (lead / 8 - 1) x 24 + 28 (and 8 per pass for the histograms as before).
// k0: word 1 of every record in record order (destroyed); a: scratch of n words; *s1: whichever of the two holds the sorted word 1; sa_out: sorted
// suffixes.  The tile histograms of the top digit (shift lo1 + lead - 8) must be in the scratch (key_pairs_kernel<..., HIST>).
// Returns PSACX_RETRY_1W without having touched k0 when the scratch has no room for the bucket tables.
constexpr int PSACX_RETRY_1W = 1001;
#ifndef PSACX_1W_ITEMS
#define PSACX_1W_ITEMS 8
#endif
// Scratch layout of the bucket passes over one-word records.  h_tabs: bucket_off[0 .. 256] filled in (start of every bucket's records;
// buckets without records own nothing), slab_start[0 .. 256] behind it is written here.
struct OneWordLayout { unsigned slab; uint64_t total_slabs, vtiles; size_t hist_bytes, slab_bytes, tabs_bytes, need; };
template <int TILE>
inline OneWordLayout onew_layout(unsigned long long* h_tabs, uint64_t ntiles_hint) {
    OneWordLayout lay;
    lay.slab = ntiles_hint >= (1u << 16) ? 64u : 16u;
    uint64_t total_slabs = 0;
    for (int d = 0; d < RADIX; ++d) {
        const uint64_t cnt = h_tabs[d + 1] - h_tabs[d];
        h_tabs[RADIX + 1 + d] = total_slabs;
        total_slabs += ((cnt + TILE - 1) / TILE + lay.slab - 1) / lay.slab;
    }
    h_tabs[2 * RADIX + 1] = total_slabs;
    lay.total_slabs = total_slabs;
    lay.vtiles = total_slabs * lay.slab;
    lay.hist_bytes = ((size_t)lay.vtiles * RADIX * sizeof(unsigned) + 255) & ~(size_t)255;
    lay.slab_bytes = ((size_t)total_slabs * RADIX * sizeof(unsigned long long) + 255) & ~(size_t)255;
    lay.tabs_bytes = (2 * (RADIX + 1) * sizeof(unsigned long long) + 64 + total_slabs * sizeof(SlabInfo) + 255) & ~(size_t)255;
    lay.need = 256 + lay.hist_bytes + lay.slab_bytes + (size_t)RADIX * RADIX * 8 + lay.tabs_bytes;
    return lay;
}
// The LSD passes inside the buckets: one-word records (rest of the prefix << sfield | suffix) of bucket b at [bucket_off[b], bucket_off[b + 1]) of `cur`,
// `low` prefix bits in the word.  All but the last pass ping-pong between cur and oth; the last one writes word 1 ((b << low | rest) << lo1) into the
// array it does not read (*s1 tells which) and the suffixes as words into sa_out.  h_tabs must stay untouched until the stream has passed the copy.
inline int onew_bucket_passes(psacx_ctx* c, char* scratch, const unsigned long long* h_tabs, const OneWordLayout& lay, uint64_t* cur, uint64_t* oth, uint64_t* sa_out,
                              unsigned sfield, unsigned low, unsigned lo1, uint64_t nrec, uint64_t** s1) {
    constexpr int BLOCK = 512, ITEMS_B = PSACX_1W_ITEMS, TILE = BLOCK * ITEMS_B;
    unsigned* tile_hist = reinterpret_cast<unsigned*>(scratch + 256);
    unsigned long long* slab_tot = reinterpret_cast<unsigned long long*>(scratch + 256 + lay.hist_bytes);
    unsigned long long* base2 = reinterpret_cast<unsigned long long*>(scratch + 256 + lay.hist_bytes + lay.slab_bytes);
    unsigned long long* d_tabs = base2 + (size_t)RADIX * RADIX;
    OneWordTabs tb;
    tb.bucket_off = d_tabs; tb.slab_start = d_tabs + RADIX + 1; tb.slab = lay.slab;
    SlabInfo* slab_info = reinterpret_cast<SlabInfo*>((reinterpret_cast<uintptr_t>(d_tabs + 2 * (RADIX + 1)) + 31) & ~(uintptr_t)31);
    tb.slab_info = slab_info;
    *s1 = cur;
    if (lay.total_slabs == 0) return PSACX_OK;
    const uint64_t total_slabs = lay.total_slabs, vtiles = lay.vtiles;
    PSACX_HIP(c, hipMemcpyAsync(d_tabs, h_tabs, 2 * (RADIX + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(radix_slab_info_kernel<0>, dim3((unsigned)((total_slabs + 255) / 256)), dim3(256), 0, c->stream, tb.bucket_off, tb.slab_start,
                       (unsigned)total_slabs, (unsigned)(lay.slab * TILE), slab_info);
    PSACX_HIP(c, hipGetLastError());
    const int npass = (int)((low + RADIX_BITS - 1) / RADIX_BITS);
    for (int j = 0; j < npass; ++j) {
        const bool last = j + 1 == npass;
        const int shift = (int)sfield + j * RADIX_BITS;
        {
            ProfScope ps(c, TC_SORT_TILEHIST);
            hipLaunchKernelGGL((radix_tile_hist1w_kernel<BLOCK, ITEMS_B>), dim3((unsigned)vtiles), dim3(BLOCK), 0, c->stream, (const uint64_t*)cur, tb, shift, tile_hist);
            hipLaunchKernelGGL(radix_slab_scan1w_kernel<0>, dim3((unsigned)total_slabs), dim3(RADIX), 0, c->stream, tile_hist, tb, (unsigned)TILE, slab_tot);
            hipLaunchKernelGGL(radix_top_scan1w_kernel<0>, dim3(RADIX), dim3(RADIX), 0, c->stream, slab_tot, tb, base2);
            PSACX_HIP(c, hipGetLastError());
        }
        ProfScope ps(c, TC_SORT_SCATTER2);
        PSACX_HIP(c, hipMemsetAsync(scratch, 0, 256, c->stream));
        if (!last)
            hipLaunchKernelGGL((radix_scatter1w_kernel<BLOCK, ITEMS_B, 8>), dim3((unsigned)vtiles), dim3(BLOCK), 0, c->stream, (const uint64_t*)cur, oth, (uint64_t*)nullptr, shift,
                               tb, base2, tile_hist, slab_tot, reinterpret_cast<unsigned*>(scratch), sort_chunk_for(nrec, true), 0u);
        else {
            // the last pass reads `cur` and writes word 1 into the other array and the suffixes into sa_out
            hipLaunchKernelGGL((radix_scatter1w_kernel<BLOCK, ITEMS_B, 9>), dim3((unsigned)vtiles), dim3(BLOCK), 0, c->stream, (const uint64_t*)cur, oth, sa_out, shift,
                               tb, base2, tile_hist, slab_tot, reinterpret_cast<unsigned*>(scratch), sort_chunk_for(nrec, true), lo1 | (low << 8) | (sfield << 16));
        }
        PSACX_HIP(c, hipGetLastError());
        c->stats.scatter_launches[2] += 1; c->stats.scatter_records[2] += nrec; c->stats.scatter_bytes[2] += (last ? 24ull : 16ull) * nrec;
        std::swap(cur, oth);
    }
    *s1 = cur;
    return PSACX_OK;
}
// text != nullptr (fused front end, sa_kernels.hpp: key_scatter1w_kernel): k0 holds nothing yet -- the histograms of the top digit come
// from the text and pass 0 computes word 1 of its tile in registers (no key_pairs_kernel launch, 16 bytes per record less).
inline int prefix_sort_1w(psacx_ctx* c, SortScratch& sc, uint64_t* k0, uint64_t* a, uint64_t* sa_out, uint64_t n, unsigned lo1, unsigned lead,
                          uint64_t spec, uint64_t spec_n, psacx_round* rs, uint64_t** s1, const uint8_t* text = nullptr, uint64_t n_text = 0,
                          const CodeTable* tab = nullptr, const KeyShape* ks = nullptr) {
    constexpr int BLOCK = 512, ITEMS = 8, TILE0 = BLOCK * ITEMS;          // pass 0 (the tile key_pairs_kernel / key_scatter1w_kernel use)
    // (bucket passes with other tiles, measured at 2^32 records: 512 x 6 -- 62 VGPRs, four workgroups per CU -- 106 ms for the five
    //  passes against 89 ms; 512 x 12 -- two workgroups per CU -- 89 ms: the run length gained is the occupancy lost)
    constexpr int ITEMS_B = PSACX_1W_ITEMS, TILE = BLOCK * ITEMS_B;        // bucket passes
    const unsigned low = lead - RADIX_BITS;            // prefix bits that stay in the word
    const unsigned sfield = 64 - low;                  // the payload field takes the rest (32 bits when lead = 40)
    const uint64_t ntiles = (n + TILE0 - 1) / TILE0;
    char* const scratch = sc.d_desc;
    // pass 0: offsets from the histograms key_pairs_kernel left, top digit
    const unsigned slab0 = slab_tiles_for(ntiles);
    const uint64_t nslabs0 = (ntiles + slab0 - 1) / slab0;
    unsigned* tile_hist0 = reinterpret_cast<unsigned*>(scratch + 256);
    unsigned long long* slab_tot0 = reinterpret_cast<unsigned long long*>(scratch + 256 + ((ntiles * RADIX * sizeof(unsigned) + 255) & ~(size_t)255));
    unsigned long long* base0 = sc.d_base;
    if (text) {
        // a text that repeats itself massively (every sampled prefix seen before) keeps the two-array passes: see prefix_dup_probe_kernel
        ProfScope ps(c, TC_KMER);
        const uint64_t stride = std::max<uint64_t>(64, n >> 20), samples = n / stride;
        uint64_t slots = 1; while (slots < 4 * samples) slots <<= 1;
        unsigned long long* table = reinterpret_cast<unsigned long long*>(scratch + 256);
        unsigned long long* d_dups = reinterpret_cast<unsigned long long*>(scratch + 128);
        if (256 + slots * 8 <= sc.desc_bytes && samples >= 1024 && !getenv("PSACX_ONE_WORD_ALWAYS")) {      // (the switch: tests of the tie paths)
            PSACX_HIP(c, hipMemsetAsync(scratch, 0, 256 + slots * 8, c->stream));
            hipLaunchKernelGGL((prefix_dup_probe_kernel<uint64_t>), dim3((unsigned)((samples + 255) / 256)), dim3(256), 0, c->stream, text, n_text, *tab, *ks, lo1,
                               stride, samples, table, slots, d_dups);
            PSACX_HIP(c, hipGetLastError());
            PSACX_HIP(c, hipMemcpyAsync(sc.h_base, d_dups, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            PSACX_HIP(c, hipStreamSynchronize(c->stream));
            if (getenv("PSACX_SORT_DEBUG")) fprintf(stderr, "[psacx 1w] %llu of %llu sampled prefixes seen before\n", (unsigned long long)sc.h_base[0], (unsigned long long)samples);
            if (sc.h_base[0] * 8 > samples) return PSACX_RETRY_1W;
        }
        hipLaunchKernelGGL((top_digit_hist_kernel<uint64_t, BLOCK, ITEMS>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, text, n, n_text, *tab, *ks, tile_hist0);
        PSACX_HIP(c, hipGetLastError());
    }
    {
        ProfScope ps(c, TC_SORT_TILEHIST);
        hipLaunchKernelGGL(radix_slab_scan_kernel<0>, dim3((unsigned)nslabs0), dim3(RADIX), 0, c->stream, tile_hist0, ntiles, slab_tot0, slab0);
        hipLaunchKernelGGL(radix_top_scan_kernel<0>, dim3(1), dim3(RADIX), 0, c->stream, slab_tot0, nslabs0, base0);
        PSACX_HIP(c, hipGetLastError());
    }
    PSACX_HIP(c, hipMemcpyAsync(sc.h_base, base0, RADIX * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    // tables of the buckets: every bucket owns whole slabs of tiles (radix.hpp: OneWordTabs)
    unsigned long long* h_tabs = sc.h_base + RADIX;             // pinned: bucket_off[257], slab_start[257]
    for (int d = 0; d < RADIX; ++d) h_tabs[d] = sc.h_base[d];
    h_tabs[RADIX] = n;
    const OneWordLayout lay = onew_layout<TILE>(h_tabs, ntiles);
    if (getenv("PSACX_SORT_DEBUG"))
        fprintf(stderr, "[psacx 1w] n=%llu lead=%u lo1=%u: %llu slabs of %u tiles for %llu tiles\n", (unsigned long long)n, lead, lo1,
                (unsigned long long)lay.total_slabs, lay.slab, (unsigned long long)ntiles);
    if (lay.need > sc.desc_bytes || lay.vtiles >= (1ull << 31)) return PSACX_RETRY_1W;
    if (text) {
        ProfScope ps(c, TC_KMER);           // (key generation and the partition by the top digit in one kernel: timed with the keys)
        PSACX_HIP(c, hipMemsetAsync(scratch, 0, 256, c->stream));
        hipLaunchKernelGGL((key_scatter1w_kernel<BLOCK, ITEMS>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, text, n, n_text, *tab, *ks, a, (int)(lo1 + low),
                           base0, tile_hist0, slab_tot0, reinterpret_cast<unsigned*>(scratch), sort_chunk_for(n, true), slab0, lo1 | (sfield << 16), (uint64_t)0);
        PSACX_HIP(c, hipGetLastError());
    } else {
        ProfScope ps(c, TC_SORT_SCATTER2);
        PSACX_HIP(c, hipMemsetAsync(scratch, 0, 256, c->stream));
        hipLaunchKernelGGL((radix_scatter3_kernel<uint64_t, BLOCK, ITEMS, false, 6, true, 7>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream,
                           (const uint64_t*)k0, (const uint64_t*)nullptr, (const uint64_t*)nullptr, a, (uint64_t*)nullptr, (uint64_t*)nullptr, n,
                           (int)(lo1 + low), base0, tile_hist0, slab_tot0, (unsigned long long*)nullptr, spec, spec_n,
                           reinterpret_cast<unsigned*>(scratch), sort_chunk_for(n, true), (const uint64_t*)nullptr, slab0, (uint64_t)0, lo1 | (sfield << 16));
        PSACX_HIP(c, hipGetLastError());
    }
    if (!text) { c->stats.scatter_launches[2] += 1; c->stats.scatter_records[2] += n; c->stats.scatter_bytes[2] += 16ull * n; }
    // the buckets (the tables of pass 0 in the scratch are dead once its scatter has run: same stream)
    uint64_t* cur = nullptr;
    PSACX_TRY(onew_bucket_passes(c, scratch, h_tabs, lay, a, k0, sa_out, sfield, low, lo1, n, &cur));
    *s1 = cur;          // (k0 after an odd number of bucket passes, `a` after an even number)
    if (rs) { rs->sort_passes = (uint32_t)((low + RADIX_BITS - 1) / RADIX_BITS + 1); rs->sort_passes_skipped = 0; }
    return PSACX_OK;
}

} // namespace psacx
