// engine.hpp -- host side of the psacx engine: context, HBM workspace, the
// rank-pair sorter driver and the prefix-doubling loop.  Compiled by hipcc into
// libpsacx.so; the only public surface is include/psacx.h.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <sched.h>
#include <pthread.h>
#include <cctype>
#include <cstdio>
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

namespace psacx { struct HostPool; }

namespace psacx {
// Switches of a construction (psacx_configure, include/psacx.h): each selects the fallback form of one stage, which the parity suite keeps
// covered; none changes the result.  The library never reads the environment: tests and A/B runs go through psacx_configure, or through
// the debug shim psacx_configure_from_env that maps PSACX_* variables to these options.
struct Knobs {
    bool force_diet = false;      // PSACX_OPT_FORCE_DIET: reduced-memory layout although the normal one fits
    uint64_t diet_cap = 0;        // PSACX_OPT_DIET_CAP: at most this many records of room for the refinement rounds (0 = no limit)
    bool one_stage = false;       // PSACX_OPT_ONE_STAGE: first round as one sort over both key words
    bool ties_radix = false;      // PSACX_OPT_TIES_RADIX: stage 2 of the first round through compaction + radix sort
    bool no_one_word = false;     // PSACX_OPT_NO_ONE_WORD: the prefix sort of the first round in (word 1, 32-bit suffix) passes, not one-word records
    bool one_word_always = false; // PSACX_OPT_ONE_WORD_ALWAYS: no repetition probe before the one-word prefix sort (tests of its tie paths)
    unsigned one_word_min = 24;   // PSACX_OPT_ONE_WORD_MIN: log2 of the smallest text that takes the one-word form (default 24; tests: 21)
    int isa_update = 0;           // PSACX_OPT_ISA_UPDATE: 1 = stores, 2 = levels: how large refinement rounds update ISA -- one random store per record, or pairs through
                                  // partition levels (construct.hpp: IsaLevels); default (0): levels from 2^31 characters on, where the random stores
                                  // into 16 GiB and more cost three times as much per record (2^30: 16 against 18 ps, 2^32: 32 against 11), and for rounds of long buckets
    bool no_digit_bytes = false;  // PSACX_OPT_NO_DIGIT_BYTES: the tile histograms of the bucket passes read the records, not the digit bytes the pass before left (A/B runs)
    bool no_bucket_sort = false;  // PSACX_OPT_NO_BUCKET_SORT: the refinement rounds always take the global radix sort (A/B runs; bucket_sort.hpp)
    int gather = 0;               // PSACX_OPT_GATHER: 1 = fetch, 2 = levels: how a refinement round gets the ranks h further -- one random fetch per record, or requests through
                                  // partition levels (construct.hpp: gather_by_levels); default (0): levels for rounds of at least n / 8 records in long buckets
    bool no_heavy = false;        // PSACX_OPT_NO_HEAVY: no split of a round's records into heavy and light ones (heavy_keys.hpp; A/B runs)
    bool no_early_out = false;    // PSACX_OPT_NO_EARLY_OUT: host-pointer calls copy SA / LCP out only after the construction (construct.hpp: construct_host)
    bool no_lazy_ranks = false;   // PSACX_OPT_NO_LAZY_RANKS: every heavy run of a split round takes the rank of its head and stores it (heavy_keys.hpp; A/B runs)
    bool no_whole = false;        // PSACX_OPT_NO_WHOLE: rounds in which nearly every suffix is unresolved take the list of positions too (A/B runs)
    bool widen_last = false;      // PSACX_OPT_WIDEN_LAST: the last pass of the one-word prefix sort writes word 1 and the suffixes as two arrays (the form the
                                  // tie stage's radix path and the multi-GPU engine read) although the kernels after the sort could read one-word records
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
    // host-pointer entry points: a ring of pinned staging buffers, device-side bounce buffers of the same size (results leave the device
    // in the narrowest entries that hold them) and a pool of host threads that moves the staged chunks to / from the caller's memory
    static constexpr int STAGE_SLOTS = 4;
    char* stage[STAGE_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    char* dstage = nullptr;          // STAGE_SLOTS chunks of device memory
    size_t stage_bytes = 0;
    hipEvent_t stage_ev[STAGE_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t copy_stream[2] = {nullptr, nullptr};     // the device -> host copies of the narrowed chunks alternate between two streams (two DMA engines)
    hipEvent_t narrow_ev[STAGE_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    // host-pointer calls: called by the construction once its first round has written SA and LCP (final unless refinement rounds follow);
    // construct_host starts their way out from there, on early_stream, under the SA -> ISA inversion
    void (*first_round_hook)(void*) = nullptr;
    void* first_round_hook_arg = nullptr;
    hipStream_t early_stream = nullptr;
    hipEvent_t early_ev = nullptr;
    unsigned long long* early_word = nullptr;       // device: buckets the first round left unresolved (0: SA and LCP are final)
    psacx::HostPool* hpool = nullptr;
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
    psacx::Knobs knobs;              // psacx_configure
    uint64_t slab_gen = 0;           // bumped whenever an operation claims the slab as its scratch (ensure_slab)
    struct { const void* s1 = nullptr; uint64_t cnt = 0; unsigned lo1 = 0; uint64_t gen = 0; } tie_stamp;      // dist_ops.hpp: op_compact_ties(counted)
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
    ++c->slab_gen;
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
    if (c->io_bytes >= bytes) return PSACX_OK;      // (kept between calls; it goes back to the device when a construction needs the room, construct_dev, or with psacx_trim)
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
// runtime's own bounce buffer with one copying thread (measured 9.5 GB/s device -> host); here the DMA engine works through a
// ring of pinned buffers while a pool of host threads moves the chunks between them and the caller's memory.  Results leave the
// device narrowed to the fewest bytes per entry that hold their largest value (a suffix array of 2^32 positions: 4 of its 8 bytes;
// the LCP array of random text: 1) and are widened again by the host threads: PCIe carries 36 instead of 96 GiB for the headline
// workload, the caller sees the same arrays.
constexpr size_t STAGE_CHUNK = (size_t)64 << 20;
inline int grid_for(const psacx_ctx* c, uint64_t work_items, int block, int per_cu);

struct HostPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    const std::function<void(int, int)>* job = nullptr;
    uint64_t gen = 0;
    int pending = 0, n = 0;
    bool stop = false;
    std::vector<cpu_set_t> homes;          // where the threads run: thread i on the CPUs of homes[i % homes.size()] (empty: anywhere)
    explicit HostPool(int threads, const std::vector<cpu_set_t>& where = std::vector<cpu_set_t>()) : n(threads), homes(where) {
        for (int i = 0; i < n; ++i) th.emplace_back([this, i]() { work(i); });
    }
    void work(int i) {
        if (!homes.empty()) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &homes[(size_t)i % homes.size()]);
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int, int)>* f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&]() { return stop || gen != seen; });
                if (stop) return;
                seen = gen; f = job;
            }
            (*f)(i, n);
            std::unique_lock<std::mutex> lk(mu);
            if (--pending == 0) cv_done.notify_all();
        }
    }
    // f(t, n) on every thread t of the pool; returns when all are done
    void run(const std::function<void(int, int)>& f) {
        std::unique_lock<std::mutex> lk(mu);
        job = &f; pending = n; ++gen;
        cv_job.notify_all();
        cv_done.wait(lk, [this]() { return pending == 0; });
    }
    ~HostPool() {
        { std::unique_lock<std::mutex> lk(mu); stop = true; cv_job.notify_all(); }
        for (auto& t : th) t.join();
    }
};

// The CPUs of ONE NUMA node of the host -- the device's own by sysfs, else the one this thread runs on --, cut down to what the process may
// use (nothing if fewer than four are left: a process confined elsewhere keeps its threads where they are).  The threads that widen the
// staged chunks into the caller's arrays stay there: the pages of arrays the library is the first to write then lie on that node, and
// every later call writes them from there.  Left to float over both sockets of the GPU box, the same call took 0.96 s in one process and
// 1.12 s in the next (profiles/r6_host_path_early_out.txt) -- the copy engines deliver 56 GB/s into pinned memory on either node
// (tools/ubench_numa.hip); it was the widening that crossed the socket link.  (Threads spread over both nodes, each always on its own
// part of a chunk: 0.96 - 1.04 s; all on one: 0.955 - 0.967 in five processes.)
inline std::vector<cpu_set_t> host_home_cpus(int device) {
    std::vector<cpu_set_t> home;
    cpu_set_t allowed; CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return home;
    int node = -1;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) == hipSuccess) {
        for (char* p = bdf; *p; ++p) *p = (char)std::tolower((unsigned char)*p);
        const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
        if (FILE* f = std::fopen(path.c_str(), "r")) { if (std::fscanf(f, "%d", &node) != 1) node = -1; std::fclose(f); }
    } else (void)hipGetLastError();
    const int cpu_now = sched_getcpu();
    for (int nd = 0; nd < 64; ++nd) {
        const std::string path = "/sys/devices/system/node/node" + std::to_string(nd) + "/cpulist";
        FILE* f = std::fopen(path.c_str(), "r");
        if (!f) break;
        cpu_set_t set; CPU_ZERO(&set);
        bool here = false;
        int a = 0, b = 0;
        for (;;) {
            if (std::fscanf(f, "%d", &a) != 1) break;
            b = a;
            int ch = std::fgetc(f);
            if (ch == '-') { if (std::fscanf(f, "%d", &b) != 1) break; ch = std::fgetc(f); }
            for (int x = a; x <= b && x < CPU_SETSIZE; ++x) { CPU_SET(x, &set); if (x == cpu_now) here = true; }
            if (ch != ',') break;
        }
        std::fclose(f);
        if (node >= 0 ? nd != node : !here) continue;
        cpu_set_t mine; CPU_ZERO(&mine);
        CPU_AND(&mine, &set, &allowed);
        if (CPU_COUNT(&mine) >= 4) home.push_back(mine);
        break;
    }
    return home;
}

inline int ensure_stage(psacx_ctx* c) {
    if (c->stage[0] && c->hpool) return PSACX_OK;
    for (int i = 0; i < psacx_ctx::STAGE_SLOTS; ++i) {
        if (c->stage[i]) continue;
        if (hipHostMalloc((void**)&c->stage[i], STAGE_CHUNK, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            c->hip_err = "pinned staging buffers could not be allocated";
            return PSACX_ENOMEM;
        }
    }
    c->stage_bytes = STAGE_CHUNK;
    for (int i = 0; i < 2; ++i)
        if (!c->copy_stream[i] && hipStreamCreateWithFlags(&c->copy_stream[i], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->copy_stream[i] = nullptr; }
    for (int i = 0; i < psacx_ctx::STAGE_SLOTS; ++i)
        if (!c->narrow_ev[i] && hipEventCreateWithFlags(&c->narrow_ev[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); c->narrow_ev[i] = nullptr; }
    if (!c->hpool) {
        // (tools/ubench_pcie.hip on the GPU box: 8 - 16 threads with streaming stores widen at 280 - 300 GB/s, 48 threads at 98 -- more
        //  threads than memory channels lose; the DMA engine delivers 51 - 57 GB/s)
        const unsigned hw = std::thread::hardware_concurrency();
        c->hpool = new HostPool((int)std::max(1u, std::min(16u, hw ? hw : 8u)), host_home_cpus(c->device));
    }
    return PSACX_OK;
}

// bytes [0, bytes) of src to dst by all threads of the pool (pieces of whole pages)
inline void pool_memcpy(HostPool* hp, char* dst, const char* src, size_t bytes) {
    if (bytes < ((size_t)1 << 20)) { std::memcpy(dst, src, bytes); return; }
    hp->run([=](int t, int nt) {
        const size_t per = ((bytes + nt - 1) / nt + 4095) & ~(size_t)4095, o = per * (size_t)t;
        if (o < bytes) std::memcpy(dst + o, src + o, std::min(per, bytes - o));
    });
}

inline int staged_d2h(psacx_ctx* c, void* dst_, const void* src_, size_t bytes, hipStream_t on = nullptr) {
    PSACX_TRY(ensure_stage(c));
    const hipStream_t S = on ? on : c->stream;
    constexpr int NS = psacx_ctx::STAGE_SLOTS;
    char* dst = static_cast<char*>(dst_); const char* src = static_cast<const char*>(src_);
    size_t issued = 0, drained = 0; int qi = 0, qd = 0, inflight = 0;
    size_t len[NS] = {0, 0, 0, 0};
    while (drained < bytes) {
        while (issued < bytes && inflight < NS) {
            const size_t m = std::min(STAGE_CHUNK, bytes - issued);
            PSACX_HIP(c, hipMemcpyAsync(c->stage[qi], src + issued, m, hipMemcpyDeviceToHost, S));
            PSACX_HIP(c, hipEventRecord(c->stage_ev[qi], S));
            len[qi] = m; issued += m; qi = (qi + 1) % NS; ++inflight;
        }
        PSACX_HIP(c, hipEventSynchronize(c->stage_ev[qd]));
        pool_memcpy(c->hpool, dst + drained, c->stage[qd], len[qd]);
        drained += len[qd]; len[qd] = 0; qd = (qd + 1) % NS; --inflight;
    }
    return PSACX_OK;
}

template <typename T, typename E>
__global__ void narrow_entries_kernel(const T* __restrict__ in, uint64_t cnt, E* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) out[i] = (E)in[i];
}
template <typename T>
__global__ void max_entry_kernel(const T* __restrict__ in, uint64_t cnt, unsigned long long* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    T m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) { const T x = in[i]; m = x > m ? x : m; }
    for (int d = 32; d > 0; d >>= 1) { const T o = (T)__shfl_xor((unsigned long long)m, d, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, (unsigned long long)m);
}

// `count` entries of T per job from device memory into the caller's arrays, travelling as the narrowest of 1 / 2 / 4-byte entries that hold
// max_value (UINT64_MAX: found here with one pass over the array; wide values travel as they are).  Several arrays share the ring of
// staging buffers chunk by chunk, the one with the largest part still to go next: the 1-byte LCP entries of a random text widen to 8 on
// the host -- 512 MiB of stores per 64 MiB chunk off the wire, more than one socket's memory takes in the 1.15 ms the chunk is on the
// wire -- while SA entries widen from 4 to 8; mixed, the stores keep up with the wire (SA then LCP: 305 + 130 ms; together: 21.5 GB at
// the wire's 56 GB/s).
template <typename T>
struct D2hJob { T* dst; const T* src; uint64_t count; uint64_t max_value; size_t e; uint64_t per, issued, drained; };

template <typename T>
int staged_d2h_jobs(psacx_ctx* c, D2hJob<T>* jobs, int nj, hipStream_t on = nullptr) {
    // on: the stream of the narrowing kernels (default: the ctx's own)
    PSACX_TRY(ensure_stage(c));
    const hipStream_t S = on ? on : c->stream;
    constexpr int NS = psacx_ctx::STAGE_SLOTS;
    if (!c->dstage && hipMalloc((void**)&c->dstage, NS * STAGE_CHUNK) != hipSuccess) { (void)hipGetLastError(); c->dstage = nullptr; }
    // widths; what is small, or as wide as it is already, goes as it is
    bool pipeline = false;
    for (int j = 0; j < nj; ++j) {
        D2hJob<T>& J = jobs[j];
        J.e = sizeof(T); J.issued = J.drained = 0; J.per = 0;
        if (J.count == 0 || !c->dstage || J.count * sizeof(T) < 4 * STAGE_CHUNK) continue;
        if (J.max_value == ~0ull) {
            unsigned long long* d_max = reinterpret_cast<unsigned long long*>(c->dstage);
            PSACX_HIP(c, hipMemsetAsync(d_max, 0, 8, S));
            hipLaunchKernelGGL((max_entry_kernel<T>), dim3(grid_for(c, J.count, 256, 8)), dim3(256), 0, S, J.src, J.count, d_max);
            PSACX_HIP(c, hipGetLastError());
            PSACX_HIP(c, hipMemcpyAsync(c->stage[0], d_max, 8, hipMemcpyDeviceToHost, S));
            PSACX_HIP(c, hipStreamSynchronize(S));
            J.max_value = *reinterpret_cast<unsigned long long*>(c->stage[0]);
        }
        J.e = J.max_value < (1ull << 8) ? 1 : J.max_value < (1ull << 16) ? 2 : J.max_value < (1ull << 32) ? 4 : 8;
        if (J.e >= sizeof(T)) { J.e = sizeof(T); continue; }
        J.per = STAGE_CHUNK / J.e;
        pipeline = true;
    }
    for (int j = 0; j < nj; ++j)
        if (jobs[j].count && !jobs[j].per) PSACX_TRY(staged_d2h(c, jobs[j].dst, jobs[j].src, jobs[j].count * sizeof(T), S));
    if (!pipeline) return PSACX_OK;
    int qi = 0, qd = 0, inflight = 0;
    uint64_t len[NS] = {0, 0, 0, 0}, at[NS] = {0, 0, 0, 0}; int of[NS] = {0, 0, 0, 0};
    HostPool* hp = c->hpool;
    auto next_job = [&]() -> int {          // the array with the largest part still to be issued
        int best = -1; double most = 0.0;
        for (int j = 0; j < nj; ++j) {
            const D2hJob<T>& J = jobs[j];
            if (!J.per || J.issued >= J.count) continue;
            const double part = (double)(J.count - J.issued) / (double)J.count;
            if (part > most) { most = part; best = j; }
        }
        return best;
    };
    for (;;) {
        int j;
        while (inflight < NS && (j = next_job()) >= 0) {
            D2hJob<T>& J = jobs[j];
            const uint64_t m = std::min(J.per, J.count - J.issued);
            char* const bounce = c->dstage + (size_t)qi * STAGE_CHUNK;
            const int grid = grid_for(c, m, 256, 8);
            if (J.e == 1) hipLaunchKernelGGL((narrow_entries_kernel<T, uint8_t>), dim3(grid), dim3(256), 0, S, J.src + J.issued, m, reinterpret_cast<uint8_t*>(bounce));
            else if (J.e == 2) hipLaunchKernelGGL((narrow_entries_kernel<T, uint16_t>), dim3(grid), dim3(256), 0, S, J.src + J.issued, m, reinterpret_cast<uint16_t*>(bounce));
            else hipLaunchKernelGGL((narrow_entries_kernel<T, uint32_t>), dim3(grid), dim3(256), 0, S, J.src + J.issued, m, reinterpret_cast<uint32_t*>(bounce));
            PSACX_HIP(c, hipGetLastError());
            hipStream_t cs = S;
            if (c->copy_stream[qi & 1] && c->narrow_ev[qi]) {
                cs = c->copy_stream[qi & 1];
                PSACX_HIP(c, hipEventRecord(c->narrow_ev[qi], S));
                PSACX_HIP(c, hipStreamWaitEvent(cs, c->narrow_ev[qi], 0));
            }
            PSACX_HIP(c, hipMemcpyAsync(c->stage[qi], bounce, (size_t)m * J.e, hipMemcpyDeviceToHost, cs));
            PSACX_HIP(c, hipEventRecord(c->stage_ev[qi], cs));
            len[qi] = m; at[qi] = J.issued; of[qi] = j; J.issued += m; qi = (qi + 1) % NS; ++inflight;
        }
        if (!inflight) break;
        PSACX_HIP(c, hipEventSynchronize(c->stage_ev[qd]));
        {
            D2hJob<T>& J = jobs[of[qd]];
            const uint64_t m = len[qd];
            const size_t e = J.e;
            T* const out = J.dst + at[qd];
            const char* const in = c->stage[qd];
            hp->run([=](int t, int nt) {
                const uint64_t a = m * (uint64_t)t / nt, b = m * (uint64_t)(t + 1) / nt;
                // (streaming stores: the caller's array is written once and not read here -- no read-for-ownership traffic on the host side,
                //  which is what bounds this path: 128 MiB written per 64 MiB chunk that PCIe delivers)
                if (e == 1) { const uint8_t* p = reinterpret_cast<const uint8_t*>(in); for (uint64_t i = a; i < b; ++i) __builtin_nontemporal_store((T)p[i], out + i); }
                else if (e == 2) { const uint16_t* p = reinterpret_cast<const uint16_t*>(in); for (uint64_t i = a; i < b; ++i) __builtin_nontemporal_store((T)p[i], out + i); }
                else { const uint32_t* p = reinterpret_cast<const uint32_t*>(in); for (uint64_t i = a; i < b; ++i) __builtin_nontemporal_store((T)p[i], out + i); }
            });
            J.drained += m;
        }
        len[qd] = 0; qd = (qd + 1) % NS; --inflight;
    }
    return PSACX_OK;
}

template <typename T>
int staged_d2h_entries(psacx_ctx* c, T* dst, const T* src, uint64_t count, uint64_t max_value, hipStream_t on = nullptr) {
    if (count == 0) return PSACX_OK;
    D2hJob<T> job{dst, src, count, max_value, 0, 0, 0, 0};
    return staged_d2h_jobs<T>(c, &job, 1, on);
}

inline int staged_h2d(psacx_ctx* c, void* dst_, const void* src_, size_t bytes) {
    PSACX_TRY(ensure_stage(c));
    constexpr int NS = psacx_ctx::STAGE_SLOTS;
    char* dst = static_cast<char*>(dst_); const char* src = static_cast<const char*>(src_);
    int q = 0; bool used[NS] = {false, false, false, false};
    for (size_t off = 0; off < bytes; off += STAGE_CHUNK, q = (q + 1) % NS) {
        const size_t m = std::min(STAGE_CHUNK, bytes - off);
        if (used[q]) PSACX_HIP(c, hipEventSynchronize(c->stage_ev[q]));       // the DMA out of this buffer is over
        pool_memcpy(c->hpool, c->stage[q], src + off, m);
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

inline int grid_for(const psacx_ctx* c, uint64_t work_items, int block, int per_cu = 8);
inline int grid_for(const psacx_ctx* c, uint64_t work_items, int block, int per_cu) {
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
    unsigned long long* d_summary; // OR/AND of the keys (see key_summary_add), 4 words
    unsigned long long* h_summary; // pinned, 4 words
    unsigned long long* d_partials;// per-workgroup key summaries of the producer kernel
    uint8_t* d_dig = nullptr;      // digit bytes between the three-kernel passes of a sort of 64-bit words with 32-bit payloads (dispatch_pass3), or null
    uint64_t dig_cap = 0;          // ... records it has room for (the array 16-byte aligned and readable up to the next multiple of 16 beyond them: the histograms read 16-byte pieces)
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

// tiles per XCD-local chunk of the tile queue (dev_common.hpp: claim_tile; 0 = plain ticket order)
inline unsigned sort_chunk_for(uint64_t n, bool three) {
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

// Tile shape of the scatter passes, measured per word size: uint32 512 x 12; uint64 512 x 8 (two-word: 3.8 ms per 2^29-record
// pass against 5.2 ms with 256 x 8; three-word: 2.35 against 2.60 ms at 2^28, 22 against 32 ms at 2^31, 47 against 78 ms at 2^32
// where the smaller tiles hit a stride artefact; register caps through __launch_bounds__ were measured: spills cost 1.6-3x)
template <typename T> struct ScatterCfg { static constexpr int BLOCK = 512, ITEMS = sizeof(T) == 4 ? 12 : 8, TILE = BLOCK * ITEMS; };

template <typename T, typename D>
inline void dispatch_scatter(psacx_ctx* c, const T* kd_in, const T* ko_in, const T* v_in, T* kd_out, T* ko_out, T* v_out, uint64_t n, int shift,
                             const unsigned long long* base, char* desc, unsigned* err, uint64_t spec, uint64_t spec_n) {
    launch_scatter<T, D, ScatterCfg<T>::BLOCK, ScatterCfg<T>::ITEMS>(c, kd_in, ko_in, v_in, kd_out, ko_out, v_out, n, shift, base, desc, err, nullptr, spec, spec_n);
}

// One pass of the three-kernel form: tile histograms (unless the producer of the keys left them), slab / top scans, scatter.
// vn (64-bit words, two-word records): 0 = word payloads, 1 = 32-bit payload entries on both sides, 2 = 32-bit entries in and
// words out (radix.hpp: VN).
template <typename T>
inline void dispatch_pass3(psacx_ctx* c, const T* kd_in, const T* ko_in, const T* v_in, T* kd_out, T* ko_out, T* v_out,
                           uint64_t n, int shift, const unsigned long long* base, char* scratch, uint64_t spec, uint64_t spec_n,
                           bool have_hist = false, int vn = 0, const uint8_t* dig_in = nullptr, uint8_t* dig_out = nullptr, int dig_shift = 0, int dig_ko = 0) {
    // dig_in: the digit of this pass of the record at every place, one byte each, left by the pass before -- the tile histograms read it instead of
    // the records (8 bytes per record less); dig_out (vn == 1 only): this pass leaves the byte of the NEXT pass (bits dig_shift .. + 7 of the digit
    // word, or of the other key word when dig_ko) beside its records.  One array serves both: the histograms are done when the scatter starts.
    constexpr int BLOCK = ScatterCfg<T>::BLOCK, ITEMS = ScatterCfg<T>::ITEMS, TILE = BLOCK * ITEMS;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
    const unsigned slab_tiles = slab_tiles_for(ntiles);
    const uint64_t nslabs = (ntiles + slab_tiles - 1) / slab_tiles;
    unsigned* tile_hist = reinterpret_cast<unsigned*>(scratch + 256);
    unsigned long long* slab_tot = reinterpret_cast<unsigned long long*>(scratch + 256 + ((ntiles * RADIX * sizeof(unsigned) + 255) & ~(size_t)255));
    {
        ProfScope ps(c, TC_SORT_TILEHIST);
        if (have_hist) {       // (the producer of the keys may have left this pass's tile histograms in place)
        } else if (dig_in)
            hipLaunchKernelGGL((radix_tile_hist_bytes_flat_kernel<BLOCK, TILE>), dim3((unsigned)((ntiles + BLOCK / WAVE - 1) / (BLOCK / WAVE))), dim3(BLOCK), 0, c->stream,
                               dig_in, n, ntiles, tile_hist);
        else
            hipLaunchKernelGGL((radix_tile_hist_kernel<T, BLOCK, ITEMS>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in, n,
                               shift, tile_hist);
        hipLaunchKernelGGL(radix_slab_scan_kernel<0>, dim3((unsigned)nslabs), dim3(RADIX), 0, c->stream, tile_hist, ntiles, slab_tot, slab_tiles);
        hipLaunchKernelGGL(radix_top_scan_kernel<0>, dim3(1), dim3(RADIX), 0, c->stream, slab_tot, nslabs,
                           const_cast<unsigned long long*>(base));
    }
    ProfScope ps(c, ko_in ? TC_SORT_SCATTER3 : TC_SORT_SCATTER2);
    unsigned* const counter = reinterpret_cast<unsigned*>(scratch);
    const unsigned chunk = sort_chunk_for(n, true);
    if constexpr (sizeof(T) == 8) {
        // 32-bit payload arrays: registers capped for six waves per SIMD = three workgroups per CU: the narrow forms need 82, the cap
        // costs them a few spilled registers and gains a third tile in flight (the pass is bound by the latency chain of a tile)
        if (vn == 1 && !ko_in && dig_out) {
            hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, false, 6, true, 1, sizeof(T), true>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in, ko_in, v_in, kd_out,
                               ko_out, v_out, n, shift, base, tile_hist, slab_tot, (unsigned long long*)nullptr, spec, spec_n, counter, chunk, (const T*)nullptr, slab_tiles,
                               (uint64_t)0, 0u, dig_out, dig_shift, 0);
            return;
        }
        if (vn == 1 && !ko_in) {
            hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, false, 6, true, 1>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in, ko_in, v_in, kd_out,
                               ko_out, v_out, n, shift, base, tile_hist, slab_tot, (unsigned long long*)nullptr, spec, spec_n, counter, chunk, (const T*)nullptr, slab_tiles);
            return;
        }
        if (vn == 2 && !ko_in) {
            hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, false, 6, true, 2>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in, ko_in, v_in, kd_out,
                               ko_out, v_out, n, shift, base, tile_hist, slab_tot, (unsigned long long*)nullptr, spec, spec_n, counter, chunk, (const T*)nullptr, slab_tiles);
            return;
        }
    }
    if constexpr (sizeof(T) == 8) {
        // three-word records whose payload stays below 2^32 (the suffixes of a text of at most 2^32 characters): 32-bit payload entries
        // between the passes here too -- 40 instead of 48 bytes per record and pass (the one-stage first round of repetitive texts)
        if (ko_in && vn == 1 && dig_out) {
            hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, false, 1, false, 1, sizeof(T), true>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in,
                               ko_in, v_in, kd_out, ko_out, v_out, n, shift, base, tile_hist, slab_tot, (unsigned long long*)nullptr, spec, spec_n,
                               counter, chunk, (const T*)nullptr, slab_tiles, (uint64_t)0, 0u, dig_out, dig_shift, dig_ko);
            return;
        }
        if (ko_in && vn == 1) {
            hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, false, 1, false, 1>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in,
                               ko_in, v_in, kd_out, ko_out, v_out, n, shift, base, tile_hist, slab_tot, (unsigned long long*)nullptr, spec, spec_n,
                               counter, chunk, (const T*)nullptr, slab_tiles);
            return;
        }
        if (ko_in && vn == 2) {
            hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, false, 1, false, 2>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in,
                               ko_in, v_in, kd_out, ko_out, v_out, n, shift, base, tile_hist, slab_tot, (unsigned long long*)nullptr, spec, spec_n,
                               counter, chunk, (const T*)nullptr, slab_tiles);
            return;
        }
    }
    if (ko_in)
        hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, false, 1>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in,
                           ko_in, v_in, kd_out, ko_out, v_out, n, shift, base, tile_hist, slab_tot, (unsigned long long*)nullptr, spec, spec_n,
                           counter, chunk, (const T*)nullptr, slab_tiles);
    else            // two-word records (k1, v): the prefix sort of the first round
        hipLaunchKernelGGL((radix_scatter3_kernel<T, BLOCK, ITEMS, false, 1, true>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, kd_in,
                           ko_in, v_in, kd_out, ko_out, v_out, n, shift, base, tile_hist, slab_tot, (unsigned long long*)nullptr, spec, spec_n,
                           counter, chunk, (const T*)nullptr, slab_tiles);
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

// The shuffle pass of the two-word first round (multi.hpp: sort_first_two_word): the records k1[0 .. n) of a rank's block are
// grouped by the destination in the byte array cls (stable), their payload -- the suffix a record stands for -- is made up on the
// way (spec / spec_n / voff as in radix_scatter_tile) and leaves as 32-bit entries when v32.
// The per-class totals are known already (classify_prefix_kernel), so nothing comes back to the host.
template <typename T>
int piece_partition(psacx_ctx* c, char* scratch, unsigned long long* d_base, const T* k1, const uint8_t* cls, uint64_t n, T* k1_out, void* v_out,
                    bool v32, uint64_t spec, uint64_t spec_n, uint64_t voff) {
    constexpr int BLOCK = ScatterCfg<T>::BLOCK, ITEMS = ScatterCfg<T>::ITEMS, TILE = BLOCK * ITEMS;
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
    if (sizeof(T) == 8 && v32)
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
    return !has_k2 || n >= SMALL_SORT_MAX;
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
// v32_in (64-bit words, two-word records): in.v holds 32-bit entries (payloads below 2^32: the suffixes of a text of at most 2^32
// characters after the multi-GPU shuffle, the suffixes of refinement records); they stay 32-bit between the passes and the last
// pass widens them, as for a payload the first pass makes up.
// ready_hist_shift >= 0: the tile histograms of word 1 at that bit position are already in the scratch (written by
// key_pairs_kernel<..., HIST> with the tile shape of this sort).
template <typename T>
int pair_sort(psacx_ctx* c, SortScratch& sc, SortBufs<T> in, SortBufs<T> alt, uint64_t n, bool iota,
              unsigned bits1, unsigned bits2, T* final_v, SortBufs<T>* res, psacx_round* rs,
              uint64_t spec = 0, uint64_t spec_n = 0, bool summary_ready = false, unsigned lo1 = 0,
              int ready_hist_shift = -1, bool v32_in = false, bool keep_v32 = false) {
    // keep_v32 (with v32_in): the payloads stay 32-bit entries after the last pass too (res->v holds n 32-bit entries)
    if (bits1 > sizeof(T) * 8) bits1 = sizeof(T) * 8;
    if (bits2 > sizeof(T) * 8) bits2 = sizeof(T) * 8;
    if (!in.k2) bits2 = 0;
    const PassPlan plan = make_plan((int)bits1, (int)bits2, (int)lo1);
    // three-kernel passes (no workgroup ever waits on another) for large inputs, the single-sweep look-back form for small ones
    // where the launch count matters more (records without a second key word exist only in the three-kernel form)
    const bool three = sort_is_three(n, in.k2 != nullptr);
    bool skip[MAX_PASSES];
    int n_exec = 0;
    // look-back form: digit starts scanned on the device, one descriptor region per pass (zeroed by one memset together
    // with the histograms); the host only learns which passes have a constant digit (flags the scan kernel stores into
    // pinned host memory).  Needs the scratch laid out as carve() lays it out.
    // (2048-record tiles for the small sorts were measured: three times the look-back chain, 0.27 -> 0.29 ms per round at 2^20)
    constexpr uint64_t TILE = ScatterCfg<T>::TILE;
    const bool small_desc = n < (1ull << 30);
    const size_t hist_bytes = sizeof(unsigned long long) * MAX_PASSES * RADIX;
    const size_t desc_stride = (256 + ((n + TILE - 1) / TILE) * RADIX * (small_desc ? sizeof(uint32_t) : sizeof(uint64_t)) + 255) & ~(size_t)255;
    const bool dev_scan = !three && c->pinned_dev && c->pinned_bytes >= 512 &&
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
    } else {
        HistArgs ha;
        ha.n_pass = plan.n_pass;
        for (int p = 0; p < plan.n_pass; ++p) { ha.word[p] = plan.word[p]; ha.shift[p] = plan.shift[p]; }
        if (dev_scan) {
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
        } else {
            // (the scratch is not laid out for the device-side scan, or the device cannot store into the pinned words: digit starts on the host)
            {
                ProfScope ps(c, TC_SORT_HIST);
                PSACX_HIP(c, hipMemsetAsync(sc.d_hist, 0, sizeof(unsigned long long) * MAX_PASSES * RADIX, c->stream));
                const int grid = grid_for(c, (n + 3) / 4, 256, 8);
                hipLaunchKernelGGL((radix_hist_kernel<T, 256>), dim3(grid), dim3(256), 0, c->stream, in.k1, in.k2, n, ha, sc.d_hist);
                PSACX_HIP(c, hipGetLastError());
            }
            PSACX_HIP(c, hipMemcpyAsync(sc.h_hist, sc.d_hist, sizeof(unsigned long long) * plan.n_pass * RADIX, hipMemcpyDeviceToHost, c->stream));
            PSACX_HIP(c, hipStreamSynchronize(c->stream));
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
            if (n_exec) PSACX_HIP(c, hipMemcpyAsync(sc.d_base, sc.h_base, sizeof(unsigned long long) * plan.n_pass * RADIX, hipMemcpyHostToDevice, c->stream));
        }
        c->stats.hist_bytes += 2ull * sizeof(T) * n;
    }
    if (rs) { rs->sort_passes = (uint32_t)n_exec; rs->sort_passes_skipped = (uint32_t)(plan.n_pass - n_exec); }

    // two-word records of 64-bit words whose payload is made up by the first pass (suffix indices < n <= 2^32): the payload
    // travels as 32-bit entries between the passes and is widened by the last one (radix.hpp: VN)
    const bool narrow = three && sizeof(T) == 8 && ((iota && n <= (1ull << 32)) || v32_in);
    if (v32_in && !narrow) return PSACX_EINVAL;
    if (keep_v32 && (!v32_in || final_v)) return PSACX_EINVAL;
    SortBufs<T> cur = in, oth = alt;
    int done = 0;
    bool dig_ready = false;          // the pass before left this pass's digit bytes in sc.d_dig
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
        const uint64_t ntiles = (n + TILE - 1) / TILE;
        const size_t dbytes = 256 + ntiles * RADIX * (small_desc ? sizeof(uint32_t) : sizeof(uint64_t));
        char* const desc = dev_scan ? sc.d_desc + (size_t)(done - 1) * desc_stride : sc.d_desc;
        if (!dev_scan) PSACX_HIP(c, hipMemsetAsync(sc.d_desc, 0, three ? 256 : dbytes, c->stream));
        const unsigned long long* base = sc.d_base + (size_t)p * RADIX;
        bool dig_next = false;
        if (three) {
            const bool have_hist = first && plan.word[p] == 0 && plan.shift[p] == ready_hist_shift;
            const int vn = narrow ? ((last && !keep_v32) ? ((first && !v32_in) ? 0 : 2) : 1) : 0;
            // the digit of the next executed pass travels as a byte beside the records this pass writes (not out of a pass that writes padded
            // or widened output, and only when the caller gave the array)
            int q = p + 1;
            while (q < plan.n_pass && skip[q]) ++q;
            dig_next = sizeof(T) == 8 && vn == 1 && !last && q < plan.n_pass && sc.d_dig && n <= sc.dig_cap;
            dispatch_pass3<T>(c, kd_in, ko_in, v_in, kd_out, ko_out, v_out, n, plan.shift[p], base, sc.d_desc, spec, spec_n, have_hist, vn,
                              dig_ready ? sc.d_dig : (const uint8_t*)nullptr, dig_next ? sc.d_dig : (uint8_t*)nullptr, dig_next ? plan.shift[q] : 0,
                              dig_next && plan.word[q] != plan.word[p] ? 1 : 0);
            dig_ready = dig_next;
            PSACX_HIP(c, hipGetLastError());
        } else {
            ProfScope ps(c, TC_SORT_SCATTER);
            if (small_desc) dispatch_scatter<T, uint32_t>(c, kd_in, ko_in, v_in, kd_out, ko_out, v_out, n, plan.shift[p], base, desc, sc.d_err, spec, spec_n);
            else dispatch_scatter<T, uint64_t>(c, kd_in, ko_in, v_in, kd_out, ko_out, v_out, n, plan.shift[p], base, desc, sc.d_err, spec, spec_n);
            PSACX_HIP(c, hipGetLastError());
        }
        const int form = !in.k2 ? 2 : (three ? 1 : 0);
        c->stats.scatter_launches[form] += 1;
        c->stats.scatter_records[form] += n;
        // words read + written per record; a pass that makes up its payload (iota) reads one word less
        if (narrow) c->stats.scatter_bytes[form] += ((in.k2 ? 4ull : 2ull) * sizeof(T) + (v_in ? ((first && !v32_in) ? sizeof(T) : 4ull) : 0ull) + (last ? sizeof(T) : 4ull) + (dig_next ? 1ull : 0ull)) * n;
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
        } else if (v32_in && keep_v32) {
            dst = cur.v;               // (32-bit entries in and out: nothing to do)
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
// Prefix sort of the first round in one-word records, most significant digit first (radix.hpp: VN 8 .. 10, *1w kernels).
// The two-stage first round sorts (word 1, suffix) on the `lead` bits of word 1 above bit lo1 and reads nothing below them
// afterwards (ties get their windows from the text again).  With lead - 8 <= 32 and suffixes below 2^32 a record fits ONE
// 64-bit word once the top digit of the prefix is known from the record's place: the pass on the top digit computes word 1 of its
// tile in registers straight from the text (sa_kernels.hpp: key_scatter1w_kernel; the tile histograms of that digit come from the
// text too, top_digit_hist_kernel) and writes (rest of the prefix) << sfield | suffix; the remaining digits are LSD passes inside the
// 256 buckets, all buckets in one launch; the last of them writes word 1 (prefix << lo1, low bits zero) and the suffixes as words.
// Bytes per record: 1 + 8, then (lead / 8 - 2) x 16 + 24, and 8 per bucket pass for its histograms.
// k0, a: two scratch arrays of n words; *s1: whichever holds the sorted word 1; sa_out: sorted suffixes.
// Returns PSACX_RETRY_1W without having written anything when the text repeats itself massively (probe) or the scratch has no room
// for the bucket tables.
constexpr int PSACX_RETRY_1W = 1001;
// ... and PSACX_RETRY_1STAGE when at least three of four sampled prefixes were seen before: nearly every suffix would tie on the sorted
// prefix and be sorted a second time by its full window -- the first round as ONE sort over both words is the cheaper form then
// (2^30 characters of 20 symbols with geometric frequencies: 448 -> 218 ms; repeated reads with mutations: 1.19 -> 1.00 s)
constexpr int PSACX_RETRY_1STAGE = 1002;
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
// view_out != nullptr: the last pass leaves one-word records like the others (*s1: where) and *view_out describes them (sa_kernels.hpp: OneWordView;
// its table lives in the scratch until the next sort) -- the kernels after the sort read them where they lie.
// dig != nullptr: one byte per record (the array padded to a multiple of 16) that holds, at every place, the digit of the record there that the NEXT pass sorts on
// -- written by the pass on the top digit and by every bucket pass but the last; the tile histograms then read it instead of the records.
inline int onew_bucket_passes(psacx_ctx* c, char* scratch, const unsigned long long* h_tabs, const OneWordLayout& lay, uint64_t* cur, uint64_t* oth, uint64_t* sa_out,
                              unsigned sfield, unsigned low, unsigned lo1, uint64_t nrec, uint64_t** s1, OneWordView* view_out = nullptr, uint8_t* dig = nullptr,
                              uint64_t in_pad = 0, uint8_t* tie_bytes = nullptr) {
    // tie_bytes (with view_out): the last pass leaves the LOWEST byte of the prefix bits of the record at every place there: two neighbours that
    // tie on the prefix agree in it, two that do not agree in it once in 256 -- the tie stage reads these bytes instead of the records
    // in_pad: in the input of the FIRST pass the records of bucket b lie b * in_pad places further than bucket_off says (prefix_sort_1w)
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
    if (view_out) { view_out->off = d_tabs; view_out->low = low; view_out->sfield = sfield; view_out->lo1 = lo1; }
    PSACX_HIP(c, hipMemcpyAsync(d_tabs, h_tabs, 2 * (RADIX + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    if (lay.total_slabs == 0) return PSACX_OK;
    const uint64_t total_slabs = lay.total_slabs, vtiles = lay.vtiles;
    hipLaunchKernelGGL(radix_slab_info_kernel<0>, dim3((unsigned)((total_slabs + 255) / 256)), dim3(256), 0, c->stream, tb.bucket_off, tb.slab_start,
                       (unsigned)total_slabs, (unsigned)(lay.slab * TILE), slab_info);
    PSACX_HIP(c, hipGetLastError());
    const int npass = (int)((low + RADIX_BITS - 1) / RADIX_BITS);
    for (int j = 0; j < npass; ++j) {
        const bool last = j + 1 == npass;
        const int shift = (int)sfield + j * RADIX_BITS;
        {
            ProfScope ps(c, TC_SORT_TILEHIST);
            if (dig && j > 0) hipLaunchKernelGGL((radix_tile_hist_bytes_kernel<BLOCK, ITEMS_B>), dim3((unsigned)((vtiles + BLOCK / WAVE - 1) / (BLOCK / WAVE))), dim3(BLOCK), 0, c->stream,
                                                 (const uint8_t*)dig, tb, (unsigned)vtiles, tile_hist);
            else
            hipLaunchKernelGGL((radix_tile_hist1w_kernel<BLOCK, ITEMS_B>), dim3((unsigned)vtiles), dim3(BLOCK), 0, c->stream, (const uint64_t*)cur, tb, shift, tile_hist,
                               j == 0 ? in_pad : (uint64_t)0);
            hipLaunchKernelGGL(radix_slab_scan1w_kernel<0>, dim3((unsigned)total_slabs), dim3(RADIX), 0, c->stream, tile_hist, tb, (unsigned)TILE, slab_tot);
            hipLaunchKernelGGL(radix_top_scan1w_kernel<0>, dim3(RADIX), dim3(RADIX), 0, c->stream, slab_tot, tb, base2);
            PSACX_HIP(c, hipGetLastError());
        }
        ProfScope ps(c, TC_SORT_SCATTER2);
        PSACX_HIP(c, hipMemsetAsync(scratch, 0, 256, c->stream));
        if (!last || view_out)
            hipLaunchKernelGGL((radix_scatter1w_kernel<BLOCK, ITEMS_B, 8>), dim3((unsigned)vtiles), dim3(BLOCK), 0, c->stream, (const uint64_t*)cur, oth, (uint64_t*)nullptr, shift,
                               tb, base2, tile_hist, slab_tot, reinterpret_cast<unsigned*>(scratch), sort_chunk_for(nrec, true), 0u,
                               last ? tie_bytes : dig, last ? (int)sfield : shift + RADIX_BITS, j == 0 ? in_pad : (uint64_t)0);
        else {
            // the last pass reads `cur` and writes word 1 into the other array and the suffixes into sa_out
            hipLaunchKernelGGL((radix_scatter1w_kernel<BLOCK, ITEMS_B, 9>), dim3((unsigned)vtiles), dim3(BLOCK), 0, c->stream, (const uint64_t*)cur, oth, sa_out, shift,
                               tb, base2, tile_hist, slab_tot, reinterpret_cast<unsigned*>(scratch), sort_chunk_for(nrec, true), lo1 | (low << 8) | (sfield << 16),
                               (uint8_t*)nullptr, 0, j == 0 ? in_pad : (uint64_t)0);
        }
        PSACX_HIP(c, hipGetLastError());
        c->stats.scatter_launches[2] += 1; c->stats.scatter_records[2] += nrec; c->stats.scatter_bytes[2] += ((last && !view_out ? 24ull : 16ull) + (((dig && !last) || (last && view_out && tie_bytes)) ? 1ull : 0ull)) * nrec;      // (+ the digit byte for the next pass's histograms)
        c->stats.onew_passes += 1;
        std::swap(cur, oth);
    }
    *s1 = cur;
    return PSACX_OK;
}
// text != nullptr (fused front end, sa_kernels.hpp: key_scatter1w_kernel): k0 holds nothing yet -- the histograms of the top digit come
// from the text and pass 0 computes word 1 of its tile in registers (no key_pairs_kernel launch, 16 bytes per record less).
// dig: n bytes (rounded up to 16) of scratch for the digit bytes between the passes (onew_bucket_passes), or null
inline int prefix_sort_1w(psacx_ctx* c, SortScratch& sc, uint64_t* k0, uint64_t* a, uint64_t* sa_out, uint64_t n, unsigned lo1, unsigned lead,
                          psacx_round* rs, uint64_t** s1, const uint8_t* text, uint64_t n_text, const CodeTable& tab, const KeyShape& ks, bool probe,
                          OneWordView* view_out = nullptr, uint8_t* dig = nullptr, uint64_t pad = 0, uint8_t* tie_bytes = nullptr) {
    // pad (even; k0 must hold n + 256 * pad words): the pass on the top digit writes bucket b's records b * pad places further -- into k0, so that
    // only the first bucket pass reads the padded layout -- because output fronts a multiple of 2^27 bytes apart (2^32 records of 8 bytes in 256
    // equal buckets) alias in the memory channels (tools/ubench_fronts.hip)
    constexpr int BLOCK = 512, ITEMS = 8, TILE0 = BLOCK * ITEMS;          // the pass on the top digit
    // (bucket passes with other tiles, measured at 2^32 records: 512 x 6 -- 62 VGPRs, four workgroups per CU -- 106 ms for the five
    //  passes against 89 ms; 512 x 12 -- two workgroups per CU -- 89 ms: the run length gained is the occupancy lost)
    constexpr int TILE = BLOCK * PSACX_1W_ITEMS;        // bucket passes
    const unsigned low = lead - RADIX_BITS;            // prefix bits that stay in the word
    const unsigned sfield = 64 - low;                  // the payload field takes the rest (32 bits when lead = 40)
    const uint64_t ntiles = (n + TILE0 - 1) / TILE0;
    char* const scratch = sc.d_desc;
    const unsigned slab0 = slab_tiles_for(ntiles);
    const uint64_t nslabs0 = (ntiles + slab0 - 1) / slab0;
    unsigned* tile_hist0 = reinterpret_cast<unsigned*>(scratch + 256);
    unsigned long long* slab_tot0 = reinterpret_cast<unsigned long long*>(scratch + 256 + ((ntiles * RADIX * sizeof(unsigned) + 255) & ~(size_t)255));
    unsigned long long* base0 = sc.d_base;
    {
        // a text that repeats itself massively (every sampled prefix seen before) keeps the two-array passes: see prefix_dup_probe_kernel
        ProfScope ps(c, TC_KMER);
        const uint64_t stride = std::max<uint64_t>(64, n >> 20), samples = n / stride;
        uint64_t slots = 1; while (slots < 4 * samples) slots <<= 1;
        unsigned long long* table = reinterpret_cast<unsigned long long*>(scratch + 256);
        unsigned long long* d_dups = reinterpret_cast<unsigned long long*>(scratch + 128);
        if (256 + slots * 8 <= sc.desc_bytes && samples >= 1024 && probe) {
            PSACX_HIP(c, hipMemsetAsync(scratch, 0, 256 + slots * 8, c->stream));
            hipLaunchKernelGGL((prefix_dup_probe_kernel<uint64_t>), dim3((unsigned)((samples + 255) / 256)), dim3(256), 0, c->stream, text, n_text, tab, ks, lo1,
                               stride, samples, table, slots, d_dups);
            PSACX_HIP(c, hipGetLastError());
            PSACX_HIP(c, hipMemcpyAsync(sc.h_base, d_dups, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            PSACX_HIP(c, hipStreamSynchronize(c->stream));
            if (sc.h_base[0] * 4 >= samples * 3) return PSACX_RETRY_1STAGE;
            if (sc.h_base[0] * 8 > samples) return PSACX_RETRY_1W;
        }
        hipLaunchKernelGGL((top_digit_hist_kernel<uint64_t, BLOCK, ITEMS>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, text, n, n_text, tab, ks, tile_hist0);
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
    if (lay.need > sc.desc_bytes || lay.vtiles >= (1ull << 31)) return PSACX_RETRY_1W;
    {
        ProfScope ps(c, TC_KMER);           // (key generation and the partition by the top digit in one kernel: timed with the keys)
        PSACX_HIP(c, hipMemsetAsync(scratch, 0, 256, c->stream));
        hipLaunchKernelGGL((key_scatter1w_kernel<BLOCK, ITEMS>), dim3((unsigned)ntiles), dim3(BLOCK), 0, c->stream, text, n, n_text, tab, ks, pad ? k0 : a, (int)(lo1 + low),
                           base0, tile_hist0, slab_tot0, reinterpret_cast<unsigned*>(scratch), sort_chunk_for(n, true), slab0, lo1 | (sfield << 16), (uint64_t)0,
                           (uint8_t*)nullptr, 0, pad);  // (no digit bytes out of this pass: their stores cost it 4.9 ms -- also at the padded places -- and save the first bucket pass's histogram 4.6)
        PSACX_HIP(c, hipGetLastError());
    }
    // the buckets (the tables of the first pass in the scratch are dead once its scatter has run: same stream)
    uint64_t* cur = nullptr;
    if (pad) PSACX_TRY(onew_bucket_passes(c, scratch, h_tabs, lay, k0, a, sa_out, sfield, low, lo1, n, &cur, view_out, dig, pad, tie_bytes));
    else PSACX_TRY(onew_bucket_passes(c, scratch, h_tabs, lay, a, k0, sa_out, sfield, low, lo1, n, &cur, view_out, dig, 0, tie_bytes));
    *s1 = cur;          // (without pad: k0 after an odd number of bucket passes, `a` after an even number; with pad the other way round)
    if (rs) { rs->sort_passes = (uint32_t)((low + RADIX_BITS - 1) / RADIX_BITS + 1); rs->sort_passes_skipped = 0; }
    return PSACX_OK;
}

} // namespace psacx
