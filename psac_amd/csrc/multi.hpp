// multi.hpp -- the block-distributed construction on several GPUs, host side in C++ behind include/psacx.h
// (psacx_multi_*).  One rank per GPU, text / SA / ISA / LCP block-partitioned the way psac partitions them over
// MPI ranks (suffix_array.hpp:183-194, mxx::blk_dist).  Two ways to run it:
//   * one process, one host thread driving all GPUs of the node (psacx_multi_create; `psac --gpus N`): the ranks are
//     local objects, every collective is issued for all of them inside one RCCL group;
//   * one process per GPU (psacx_multi_create_rank; psac's own model of one MPI rank per device, and what
//     `bench.py --gpus N` runs under torchrun): the communicator is built from a unique id the host broadcasts.
//
//   psac step (MPI through mxx)                          here
//   ---------------------------------------------------  ----------------------------------------------------------
//   alphabet allreduce            alphabet.hpp:98        all-gather of the 256-bin histograms, summed on the host
//   k-mer left_shift              kmer.hpp:142           first 2k characters sent to the left rank
//   mxx::sort (sample sort)       idxsort.hpp:60-62      regular samples -> splitters -> classify + one stable
//                                                        partition pass -> grouped ncclSend/ncclRecv of the three record
//                                                        arrays -> local radix sort -> exact re-balance
//   right_shift / exscan(max)     bucketing.hpp:39,77    all-gather of boundary records and last bucket heads
//   bulk_permute_inplace          bulk_permute.hpp:14    partition (index, value) by owner -> all-to-all -> local scatter
//   bulk_rma / sparse_get_b2      suffix_array.hpp:972   queries to owners, answers back, un-permute
//   bulk_rmq_v2                   par_rmq.hpp:199-332    edge sub-queries to owners + all-gathered block minima
//
// Exchanges run on a second HIP stream per GPU (events order them against the compute stream), so that the local
// work which does not depend on an exchange proceeds while it is in flight.  Small per-round scalars travel in one
// fixed-size all-gather through pinned host memory (or not at all when every rank lives in this process).
// Ranks that share one device (dev_ids with repeats: the test configuration on a one-GPU box) exchange by
// device-to-device copies, because RCCL refuses two ranks on one device.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only: librccl is opened with dlopen when a communicator is needed

#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "dist_ops.hpp"
#include "ansv_wave.hpp"
#include "shm_link.hpp"
#include "slice_inv.hpp"
#include "multi_kernels.hpp"

namespace psacx {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;

    bool load(std::string& err) {
        if (handle) return true;
        // the copy PyTorch (or the host program) already mapped, else the ROCm one
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names) { handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (handle) break; }
        for (const char* nm : names) { if (handle) break; handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); }
        if (!handle) { err = std::string("librccl not found: ") + dlerror(); return false; }
#define PSACX_SYM(f) f = reinterpret_cast<decltype(f)>(dlsym(handle, "nccl" #f)); if (!f) { err = "librccl lacks nccl" #f; return false; }
        PSACX_SYM(GetUniqueId) PSACX_SYM(CommInitRank) PSACX_SYM(CommInitAll) PSACX_SYM(CommDestroy) PSACX_SYM(GroupStart)
        PSACX_SYM(GroupEnd) PSACX_SYM(Send) PSACX_SYM(Recv) PSACX_SYM(AllGather) PSACX_SYM(GetErrorString)
#undef PSACX_SYM
        return true;
    }
};
inline RcclApi& rccl() { static RcclApi a; return a; }

struct MRank {
    int grank = 0;
    psacx_ctx* ctx = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    ncclComm_t comm = nullptr;
    uint64_t* d_scal = nullptr;      // device staging of the scalar all-gather (process-per-GPU mode)
    size_t scal_words = 0;
    // event pairs bracketing every exchange on comm_stream; their elapsed times are summed at the end of a call
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ex_ev;
    size_t ex_used = 0;
    double exchange_ms = 0;          // time the last call's exchanges occupied comm_stream
};

} // namespace psacx

constexpr int PSACX_MULTI_EPEER = -7;     // RCCL failure

namespace psacx {
// one worker thread per local rank, alive as long as the communicator
struct RankPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    const std::function<int(int)>* job = nullptr;
    const std::function<void(int)>* prep = nullptr;
    uint64_t gen = 0;
    int pending = 0;
    std::vector<int> rc;
    bool stop = false;
    int run(int n, const std::function<int(int)>& f, const std::function<void(int)>& p) {
        std::unique_lock<std::mutex> lk(mu);
        if ((int)th.size() != n) {
            rc.assign(n, 0);
            for (int i = (int)th.size(); i < n; ++i) th.emplace_back([this, i]() { work(i); });
        }
        job = &f; prep = &p; pending = n; ++gen;
        cv_job.notify_all();
        cv_done.wait(lk, [this]() { return pending == 0; });
        for (int i = 0; i < n; ++i) if (rc[i] != 0) return rc[i];
        return 0;
    }
    void work(int i) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<int(int)>* f; const std::function<void(int)>* p;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&]() { return stop || gen != seen; });
                if (stop) return;
                seen = gen; f = job; p = prep;
            }
            (*p)(i);
            const int r = (*f)(i);
            std::unique_lock<std::mutex> lk(mu);
            rc[i] = r;
            if (--pending == 0) cv_done.notify_all();
        }
    }
    ~RankPool() {
        { std::unique_lock<std::mutex> lk(mu); stop = true; cv_job.notify_all(); }
        for (auto& t : th) t.join();
    }
};
} // namespace psacx

// how the ranks of a communicator reach each other
enum { PSACX_TR_COPY = 0,      // every rank in this process: device-to-device copies (ranks may share a device)
       PSACX_TR_RCCL = 1,      // grouped ncclSend / ncclRecv + ncclAllGather (over xGMI between the GPUs of a node)
       PSACX_TR_SHM = 2 };     // one process per rank on one host, staged through POSIX shared memory (shm_link.hpp)

struct psacx_multi {
    int nranks = 0, nlocal = 0, first = 0;
    bool use_rccl = false;
    int transport = PSACX_TR_COPY;
    bool force_wire = false;          // PSACX_MULTI_FORCE_WIRE: no shortcut for data a rank sends to itself or for scalars that
                                      // are already on this host -- every ncclSend / ncclRecv / ncclAllGather is really issued
    psacx::ShmLink shm;
    std::vector<std::pair<std::string, double>> phases;     // wall time of the phases of the last construction (host clock)
    std::vector<psacx::MRank> R;
    std::string err;
    std::mutex err_mu;
    psacx::RankPool pool;
    psacx_stats stats;
    uint64_t bytes_sent = 0;          // payload bytes this process sent to other ranks in the last call
    uint64_t n_exchanges = 0, n_gathers = 0;
    uint64_t wire_sends = 0, wire_recvs = 0, wire_gathers = 0;   // ncclSend / ncclRecv / ncclAllGather calls really issued in the last call
    // psacx_multi_configure
    int opt_layout = 0;               // 0: choose by free device memory, 1: normal, 2: reduced-memory
    uint64_t opt_slab = 0;            // unresolved suffixes per refinement slab of the reduced-memory layout (0: block / 16)
    uint64_t out_slack = 0;           // the output arrays given to construct_dev hold this many elements beyond the block
    bool opt_trace = false;           // PSACX_MULTI_OPT_*: forms of single stages (tests, A/B runs); see include/psacx.h
    uint64_t opt_wire_piece = 0, opt_check_chunks = 0, opt_slice_step = 0;
    int opt_pieces = 0, opt_two_word = 0, opt_one_word = 0;
    bool opt_global_refine_sort = false, opt_one_stage = false, opt_no_slices = false, opt_slice_wide = false;
    unsigned opt_slice_wb = 0, opt_slice_s1 = 0;
    bool last_reduced = false;        // layout the last construction ran in
    bool last_two_word = false;       // the first round ran in two-word form (sort_first_two_word)
    bool last_one_word = false;       // the first round ran in one-word records dealt by top digit (sort_first_one_word)
    bool last_slice_inversion = false;   // SA -> ISA ran slice by slice through the partition levels + window scatter
    uint32_t last_slab_rounds = 0;    // refinement rounds it worked off in more than one slab
    uint32_t last_tie_slabs = 0;      // slabs beyond the first in which the tie stage of the first round ran (reduced-memory layout, repetitive text)
};

namespace psacx {

inline void mg_set_err(psacx_multi* g, const std::string& m) { std::lock_guard<std::mutex> lk(g->err_mu); g->err = m; }
#define MG_HIP(g, call)                                                                   \
    do { hipError_t e__ = (call); if (e__ != hipSuccess) { mg_set_err(g, std::string(#call) + ": " + hipGetErrorString(e__)); return PSACX_EHIP; } } while (0)
#define MG_NCCL(g, call)                                                                  \
    do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) { mg_set_err(g, std::string(#call) + ": " + rccl().GetErrorString(r__)); return PSACX_MULTI_EPEER; } } while (0)
#define MG_OP(g, c, call)                                                                 \
    do { int rc__ = (call); if (rc__ != PSACX_OK) { mg_set_err(g, std::string(#call) + ": " + psacx_strerror(rc__) + " [" + (c)->hip_err + "]"); return rc__; } } while (0)

// device array owned by one rank; blocks come from and return to the rank's cache (engine.hpp: pool_alloc)
template <typename E> struct DBuf {
    E* p = nullptr; uint64_t n = 0; psacx_ctx* c = nullptr;
    DBuf() {}
    DBuf(const DBuf&) = delete; DBuf& operator=(const DBuf&) = delete;
    DBuf(DBuf&& o) noexcept : p(o.p), n(o.n), c(o.c), cap_(o.cap_), lead_(o.lead_), own_(o.own_) { o.p = nullptr; o.n = 0; o.lead_ = 0; }
    DBuf& operator=(DBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; c = o.c; cap_ = o.cap_; lead_ = o.lead_; own_ = o.own_; o.p = nullptr; o.n = 0; o.lead_ = 0; } return *this; }
    ~DBuf() { release(); }
    // `count` elements; the block is sized for max(count, reserve) so that arrays of slightly different lengths reuse
    // one cached block (reduced-memory layout)
    int alloc(psacx_ctx* ctx, uint64_t count, uint64_t reserve = 0) {
        release();
        c = ctx; n = count; own_ = true; lead_ = 0;
        if (hipSetDevice(c->device) != hipSuccess) return PSACX_EHIP;
        p = static_cast<E*>(pool_alloc(c, (size_t)std::max(count, reserve) * sizeof(E), &cap_));
        if (!p) { c->hip_err = "device allocation failed"; n = 0; return PSACX_ENOMEM; }
        return PSACX_OK;
    }
    // a view of memory somebody else owns (an output array used as scratch)
    void borrow(psacx_ctx* ctx, E* ptr, uint64_t count) { release(); c = ctx; p = ptr; n = count; own_ = false; cap_ = 0; lead_ = 0; }
    bool owned() const { return own_; }
    // the array starts k elements further into its block (records that were placed behind a headroom)
    void advance(uint64_t k) { p += k; lead_ += k; }
    void rewind(uint64_t k) { p -= k; lead_ -= k; }
    void release() {
        if (!p) return;
        if (own_) pool_free(c, p - lead_, cap_);
        p = nullptr; n = 0; lead_ = 0;
    }
private:
    size_t cap_ = 0;
    uint64_t lead_ = 0;
    bool own_ = true;
};

template <typename T> struct Rec { DBuf<T> k1, k2, v; uint64_t cnt = 0; };

using plan::prefix_of;

template <typename T> __global__ void gather_at_kernel(const T* __restrict__ a, const uint64_t* __restrict__ idx, unsigned cnt, uint64_t* __restrict__ out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) out[i] = (uint64_t)a[idx[i]];
}
template <typename T> __global__ void reverse_copy_kernel(const T* __restrict__ in, uint64_t cnt, T* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) out[i] = in[cnt - 1 - i];
}

// positions j of [lo, hi) at which a new group of equal prefixes (k1 >> lo1) starts (j >= 1): the last of them into *last (0 = none), the
// first into *first (~0 = none) -- where the tie stage of the reduced-memory layout may cut its slabs (MultiRun::first_sort_ties)
template <typename T> __global__ void prefix_cut_kernel(const T* __restrict__ k1, uint64_t lo, uint64_t hi, unsigned lo1, unsigned long long* __restrict__ last,
                                                        unsigned long long* __restrict__ first) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long mx = 0, mn = ~0ull;
    for (uint64_t j = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < hi; j += stride)
        if (j >= 1 && (k1[j] >> lo1) != (k1[j - 1] >> lo1)) { mx = j > mx ? j : mx; mn = j < mn ? j : mn; }
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long ox = __shfl_xor(mx, d, 64), on = __shfl_xor(mn, d, 64);
        mx = ox > mx ? ox : mx; mn = on < mn ? on : mn;
    }
    if ((threadIdx.x & 63) == 0) { if (mx) atomicMax(last, mx); if (mn != ~0ull) atomicMin(first, mn); }
}

// out[q] = the number of entries of the ascending array a[0 .. cnt) that are below key[q] (one thread per question)
template <typename T> __global__ void lower_bound_kernel(const T* __restrict__ a, uint64_t cnt, const uint64_t* __restrict__ key, unsigned nq, uint64_t* __restrict__ out) {
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint64_t lo = 0, hi = cnt;
    while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if ((uint64_t)a[mid] < key[q]) lo = mid + 1; else hi = mid; }
    out[q] = lo;
}

// The key of record j of a run of whole buckets in list order (pos[j]: SA position, k1[j]: bucket id = SA position of the bucket's head + 1,
// k2[j]: rank h further, below 2^kb): (index of the bucket's head in the run, halved) << kb | k2[j], written over k2 -- the members of a
// bucket are neighbours in the list, so the head's index is j - (pos[j] - head position); buckets have two members at least, so halving
// keeps the numbers of different buckets apart (gather_keys_kernel's dense bucket numbers, sa_kernels.hpp)
template <typename T> __global__ void refine_key_kernel(const T* __restrict__ pos, const T* __restrict__ k1, T* __restrict__ k2, uint64_t len, unsigned kb) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < len; j += stride) {
        const uint64_t hidx = j - ((uint64_t)pos[j] - ((uint64_t)k1[j] - 1));
        k2[j] = (T)(((hidx >> 1) << kb) | (uint64_t)k2[j]);
    }
}
template <typename T> __global__ void mask_low_kernel(T* __restrict__ a, uint64_t len, unsigned kb) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const T m = kb >= sizeof(T) * 8 ? ~(T)0 : (T)(((T)1 << kb) - 1);
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < len; j += stride) a[j] &= m;
}

template <typename T> __global__ void widen_text_kernel(const uint8_t* __restrict__ t, uint64_t cnt, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) out[i] = (T)t[i];
}

template <typename T>
struct MultiRun {
    psacx_multi* g;
    const int P, L;
    uint64_t n = 0;
    std::vector<uint64_t> sizes, offs;
    bool want_lcp = true;
    struct St {
        psacx_ctx* c; int r; uint64_t m, off; const uint8_t* text; T *SA, *ISA, *LCP;
        DBuf<T> Bsa, pos;
        uint64_t out_cap = 0;          // elements every output array holds
        bool out_busy = true;          // the output arrays hold results (or records): not available as scratch
    };
    std::vector<St> S;
    // Reduced-memory layout: the records of the first round alternate between the rank's three
    // output arrays and ONE allocated set, SA -> ISA runs in chunks, and a refinement round with more unresolved
    // suffixes than `slab_cap` on some rank is worked off in slabs of whole buckets.
    bool diet = false, first_round_ = false;
    // string set (construct_ss): the nstr + 1 global string offsets on the host; per local rank the offset of every position
    // of its block inside its string
    const uint64_t* gsa_off_ = nullptr; uint64_t gsa_nstr_ = 0;
    std::vector<DBuf<T>> soff_;
    // one rank and no request to exercise the wire anyway (PSACX_MULTI_FORCE_WIRE): the distributed primitives take their
    // local shortcuts
    bool solo_ = false;
    uint64_t sort_calls_ = 0;
    uint64_t slab_cap = 0;

    // first round of the reduced-memory layout: every record array is cut from a block of one size (a little more than
    // the text block), so the cached blocks serve each other's successors whatever the sample sort's imbalance
    uint64_t reserve_of(int i) const { return diet && first_round_ ? S[i].m + S[i].m / 8 + 256 : 0; }
    // three record arrays of cnt entries: the output arrays of the rank while they are free and large enough, else its cache
    // want_k2 = false: two-word records; the second key array is left out unless it comes for free (an output array)
    int take3(int i, Rec<T>& r, uint64_t cnt, bool want_k2 = true) {
        psacx_ctx* c = ctx(i);
        r = Rec<T>();
        r.cnt = cnt;
        if (diet && !S[i].out_busy && cnt <= S[i].out_cap) {
            r.v.borrow(c, S[i].SA, cnt); r.k1.borrow(c, S[i].ISA, cnt);
            if (S[i].LCP) { r.k2.borrow(c, S[i].LCP, cnt); invalidate_lcp_pyramid(i); } else if (want_k2) MG_OP(g, c, r.k2.alloc(c, cnt, reserve_of(i)));
            S[i].out_busy = true;
            return PSACX_OK;
        }
        MG_OP(g, c, r.k1.alloc(c, cnt, reserve_of(i)));
        if (want_k2) MG_OP(g, c, r.k2.alloc(c, cnt, reserve_of(i)));
        MG_OP(g, c, r.v.alloc(c, cnt, reserve_of(i)));
        return PSACX_OK;
    }
    // the second key array of a two-word record set, when word 2 of the tied records is about to be written
    int need_k2(int i, Rec<T>& r) {
        if (r.k2.p) return PSACX_OK;
        // (records behind a headroom, to be re-balanced in place: word 2 lies the same way)
        const uint64_t head = i < (int)head_.size() ? head_[i] : 0, room = i < (int)room_.size() ? room_[i] : 0;
        MG_OP(g, ctx(i), r.k2.alloc(ctx(i), std::max(r.cnt + head, room), reserve_of(i)));
        r.k2.advance(head); r.k2.n = r.cnt;
        return PSACX_OK;
    }
    // sort_first_one_word: the sorted records of local rank i lie head_[i] elements into arrays of room_[i] elements, so that the pieces of
    // its block that other ranks hold can be received in front of / behind them (rebalance_in_place); empty = no such layout
    std::vector<uint64_t> head_, room_;
    void drop3(int i, Rec<T>& r) {
        // (any member that lives in an output array: sort_first_one_word lends the SA array to the suffixes while word 1 is the engine's own)
        if ((r.k1.p && !r.k1.owned()) || (r.k2.p && !r.k2.owned()) || (r.v.p && !r.v.owned())) S[i].out_busy = false;
        r = Rec<T>();
    }
    void swap3(Rec<T>& a, Rec<T>& b) { std::swap(a.k1, b.k1); std::swap(a.k2, b.k2); std::swap(a.v, b.v); }
    // records that live in the output arrays move to allocated ones (before the outputs receive results)
    int own3(int i, Rec<T>& r) {
        if (!r.k1.p || (r.k1.owned() && r.k2.owned() && r.v.owned())) return PSACX_OK;
        psacx_ctx* c = ctx(i);
        MG_HIP(g, hipSetDevice(c->device));
        DBuf<T>* a[3] = {&r.k1, &r.k2, &r.v};
        bool kept = false;
        for (int q = 0; q < 3; ++q) {
            if (a[q]->owned()) continue;
            // the suffixes may stay where they are when that is the SA array itself: the rebucket step only reads them, and
            // they would be copied there next anyway (two copies of the block less)
            if (q == 2 && a[q]->p == S[i].SA && a[q]->n == S[i].m) { kept = true; continue; }
            DBuf<T> o; MG_OP(g, c, o.alloc(c, a[q]->n, reserve_of(i)));
            MG_HIP(g, hipMemcpyAsync(o.p, a[q]->p, a[q]->n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
            *a[q] = std::move(o);
        }
        S[i].out_busy = kept;          // (the output arrays stay off limits as scratch while the suffixes sit in one)
        return PSACX_OK;
    }

    explicit MultiRun(psacx_multi* mg) : g(mg), P(mg->nranks), L(mg->nlocal), trace_(mg->opt_trace) {
        // (the forms of single stages: psacx_multi_configure)
        if (mg->opt_wire_piece) wire_piece_ = std::max<size_t>(256, (size_t)mg->opt_wire_piece);
        pieces_env_ = std::max(0, mg->opt_pieces);                    // ranges per destination of the first round's shuffle (tests)
        global_refine_sort_env_ = mg->opt_global_refine_sort;         // refinement rounds sort all their records across the ranks (tests, A/B runs)
        one_stage_env_ = mg->opt_one_stage;                           // first round as one sort over both key words
        slab_env_ = 0;
        check_chunks_env_ = mg->opt_check_chunks;
        slice_wb_env_ = mg->opt_slice_wb; slice_s1_env_ = mg->opt_slice_s1; slice_step_env_ = mg->opt_slice_step;      // (tests: the levels of the slice inversion on small inputs)
        solo_ = P == 1 && !mg->force_wire;
        t_last_ = t_phase_ = std::chrono::steady_clock::now();
    }
    // PSACX_MULTI_OPT_TRACE: wall time of every phase on stderr (all local streams drained at each mark)
    bool trace_;
    std::chrono::steady_clock::time_point t_last_, t_phase_;
    // every phase leaves its host wall time in g->phases (accumulated by name: the refinement rounds repeat theirs)
    void mark(const char* what) {
        if (trace_) for (int i = 0; i < L; ++i) { (void)hipSetDevice(ctx(i)->device); (void)hipStreamSynchronize(ctx(i)->stream); (void)hipStreamSynchronize(g->R[i].comm_stream); }
        {
            const auto now = std::chrono::steady_clock::now();
            const double ms = std::chrono::duration<double, std::milli>(now - t_phase_).count();
            std::string key(what);
            const size_t a = key.find_first_not_of(' ');
            key = a == std::string::npos ? key : key.substr(a);
            bool found = false;
            for (auto& ph : g->phases) if (ph.first == key) { ph.second += ms; found = true; break; }
            if (!found) g->phases.emplace_back(key, ms);
            t_phase_ = now;
        }
        if (!trace_) return;
        const auto now = std::chrono::steady_clock::now();
        size_t fr = 0, tot = 0;
        (void)hipMemGetInfo(&fr, &tot);
        int w = 0;                                       // the local rank whose block cache peaked highest
        for (int i = 1; i < L; ++i) if (ctx(i)->pool_peak > ctx(w)->pool_peak) w = i;
        if (g->first == 0) fprintf(stderr, "[psacx multi] %-28s %9.3f ms   (device memory in use %.1f GiB; rank %d's cache: %.1f MiB live, %.1f cached, peak %.1f)\n", what,
                                   std::chrono::duration<double, std::milli>(now - t_last_).count(), (double)(tot - fr) / (1 << 30), rank(w),
                                   ctx(w)->pool_live / 1048576.0, ctx(w)->pool_bytes / 1048576.0, ctx(w)->pool_peak / 1048576.0);
        t_last_ = t_phase_ = std::chrono::steady_clock::now();
    }
    psacx_ctx* ctx(int i) const { return g->R[i].ctx; }
    // body(i) for every local rank.  The step ops synchronise their stream with the host, so a single host thread would
    // run the GPUs of a one-process communicator one after the other: every local rank has its own worker thread, the
    // collectives in between stay on the calling thread.
    int par(const std::function<int(int)>& body) {
        if (L == 1) return body(0);
        return g->pool.run(L, body, [this](int i) { (void)hipSetDevice(g->R[i].ctx->device); });
    }
    int rank(int i) const { return g->R[i].grank; }

    // ---------------------------------------------------------------- collectives
    // every rank contributes k words; all[r * k + j] = word j of rank r
    int gather(int k, const std::vector<std::vector<uint64_t>>& mine, std::vector<uint64_t>& all) {
        all.assign((size_t)P * k, 0);
        g->n_gathers++;
        if (L == P && !(g->force_wire && g->transport == PSACX_TR_RCCL)) {   // every rank lives in this process: nothing has to travel
            for (int i = 0; i < L; ++i) std::memcpy(&all[(size_t)rank(i) * k], mine[i].data(), (size_t)k * 8);
            return PSACX_OK;
        }
        if (g->transport == PSACX_TR_SHM) {
            ShmLink& sh = g->shm;
            std::string e;
            for (size_t at = 0; at < (size_t)k; ) {        // in pieces of one slot
                const size_t part = std::min<size_t>((size_t)k - at, sh.slot_bytes / 8);
                std::memcpy(sh.slot(rank(0)), mine[0].data() + at, part * 8);
                if (!sh.barrier(e)) { mg_set_err(g, e); return PSACX_MULTI_EPEER; }
                for (int r = 0; r < P; ++r) std::memcpy(&all[(size_t)r * k + at], sh.slot(r), part * 8);
                if (!sh.barrier(e)) { mg_set_err(g, e); return PSACX_MULTI_EPEER; }
                at += part;
            }
            return PSACX_OK;
        }
        if (g->transport != PSACX_TR_RCCL) { mg_set_err(g, "scalar all-gather without a transport between the processes"); return PSACX_EINVAL; }
        RcclApi& nc = rccl();
        for (int i = 0; i < L; ++i) {
            MRank& R = g->R[i];
            MG_HIP(g, hipSetDevice(R.ctx->device));
            if (R.scal_words < (size_t)(P + 1) * k) {
                if (R.d_scal) { MG_HIP(g, hipStreamSynchronize(R.ctx->stream)); MG_HIP(g, hipFree(R.d_scal)); }
                R.scal_words = (size_t)(P + 1) * k * 2;
                MG_HIP(g, hipMalloc((void**)&R.d_scal, R.scal_words * 8));
            }
            MG_OP(g, R.ctx, ensure_pinned(R.ctx, (size_t)(P + 1) * k * 8 + 65536));
            uint64_t* h = reinterpret_cast<uint64_t*>(R.ctx->pinned + 32768);
            std::memcpy(h, mine[i].data(), (size_t)k * 8);
            MG_HIP(g, hipMemcpyAsync(R.d_scal, h, (size_t)k * 8, hipMemcpyHostToDevice, R.ctx->stream));
        }
        {
            MG_NCCL(g, nc.GroupStart());
            ncclResult_t bad = ncclSuccess;
            for (int i = 0; i < L && bad == ncclSuccess; ++i) {
                MRank& R = g->R[i];
                bad = nc.AllGather(R.d_scal, R.d_scal + k, (size_t)k, ncclUint64, R.comm, R.ctx->stream);
            }
            const ncclResult_t end = nc.GroupEnd();           // the group is closed on every path
            if (bad != ncclSuccess) { mg_set_err(g, std::string("ncclAllGather: ") + nc.GetErrorString(bad)); return PSACX_MULTI_EPEER; }
            MG_NCCL(g, end);
        }
        g->wire_gathers++;
        for (int i = 0; i < L; ++i) {
            MRank& R = g->R[i];
            MG_HIP(g, hipSetDevice(R.ctx->device));
            uint64_t* h = reinterpret_cast<uint64_t*>(R.ctx->pinned + 32768);
            MG_HIP(g, hipMemcpyAsync(h + k, R.d_scal + k, (size_t)P * k * 8, hipMemcpyDeviceToHost, R.ctx->stream));
            MG_HIP(g, hipStreamSynchronize(R.ctx->stream));
            if (i == 0) std::memcpy(all.data(), h + k, (size_t)P * k * 8);
        }
        return PSACX_OK;
    }
    int gather1(const std::vector<uint64_t>& one_per_local, std::vector<uint64_t>& all) {
        std::vector<std::vector<uint64_t>> mine(L);
        for (int i = 0; i < L; ++i) mine[i] = {one_per_local[i]};
        return gather(1, mine, all);
    }
    // The ranks of different processes agree on a status: a rank that failed locally (an allocation, a kernel launch)
    // would otherwise leave the next collective while its peers block in it.  Returns the first non-zero code of any rank.
    int agree(int rc_local) {
        if (L == P) return rc_local;
        std::vector<uint64_t> one(L, (uint64_t)(int64_t)rc_local), all;
        const int rc = gather1(one, all);
        if (rc != PSACX_OK) return rc;
        for (int r = 0; r < P; ++r) if ((int64_t)all[r] != 0) {
            if (rc_local == PSACX_OK) mg_set_err(g, "rank " + std::to_string(r) + " reported error " + std::to_string((long long)(int64_t)all[r]) + ": all ranks leave the step");
            return (int)(int64_t)all[r];
        }
        return PSACX_OK;
    }

    // event pair around the work an exchange puts on a rank's second stream
    int ex_begin(MRank& R) {
        if (R.ex_used == R.ex_ev.size()) {
            std::pair<hipEvent_t, hipEvent_t> e;
            MG_HIP(g, hipEventCreate(&e.first)); MG_HIP(g, hipEventCreate(&e.second));
            R.ex_ev.push_back(e);
        }
        MG_HIP(g, hipEventRecord(R.ex_ev[R.ex_used].first, R.comm_stream));
        return PSACX_OK;
    }
    int ex_end(MRank& R) { MG_HIP(g, hipEventRecord(R.ex_ev[R.ex_used].second, R.comm_stream)); R.ex_used++; return PSACX_OK; }
    void ex_collect() {
        for (int i = 0; i < L; ++i) {
            MRank& R = g->R[i];
            (void)hipSetDevice(R.ctx->device);
            (void)hipStreamSynchronize(R.comm_stream);
            double ms = 0;
            for (size_t q = 0; q < R.ex_used; ++q) { float t = 0; if (hipEventElapsedTime(&t, R.ex_ev[q].first, R.ex_ev[q].second) == hipSuccess) ms += t; }
            R.exchange_ms = ms; R.ex_used = 0;
        }
    }

    // One message on the wire, in pieces of at most wire_piece_ bytes (sender and receiver cut a message of one length at the
    // same places, and messages between a pair of ranks match in order): a single ncclSend / ncclRecv of 2^31 bytes or more
    // arrived damaged in this stack (seen with a rank's message to itself: half the entries wrong at 2^28 64-bit records;
    // profiles/r04k: 2^30-byte pieces arrive whole, 2^31 - 1 do not), and pieces keep the channels' staging independent of the message length.  PSACX_MULTI_WIRE_PIECE: bytes.
    size_t wire_piece_ = (size_t)1 << 28;
    int pieces_env_ = 0;
    bool one_stage_env_ = false;
    bool global_refine_sort_env_ = false;
    uint64_t slab_env_ = 0, check_chunks_env_ = 0, slice_step_env_ = 0;
    unsigned slice_wb_env_ = 0, slice_s1_env_ = 0;
    ncclResult_t wire_send(RcclApi& nc, MRank& R, const void* p, size_t bytes, int peer) {
        for (size_t o = 0; o < bytes; o += wire_piece_) {
            const ncclResult_t r = nc.Send(static_cast<const char*>(p) + o, std::min(wire_piece_, bytes - o), ncclUint8, peer, R.comm, R.comm_stream);
            g->wire_sends++;
            if (r != ncclSuccess) return r;
        }
        return ncclSuccess;
    }
    ncclResult_t wire_recv(RcclApi& nc, MRank& R, void* p, size_t bytes, int peer) {
        for (size_t o = 0; o < bytes; o += wire_piece_) {
            const ncclResult_t r = nc.Recv(static_cast<char*>(p) + o, std::min(wire_piece_, bytes - o), ncclUint8, peer, R.comm, R.comm_stream);
            g->wire_recvs++;
            if (r != ncclSuccess) return r;
        }
        return ncclSuccess;
    }

    // All-to-all of `na` arrays per rank that share one partition: elements bounds[i][d] .. bounds[i][d+1] of every
    // array of local rank i go to rank d.  out[i][a] receives the elements ordered by source rank; rcnt[i][s] =
    // elements received from rank s.  One exchange of the counts serves all arrays; the transfers of all arrays,
    // ranks and peers form one RCCL group on the ranks' second streams.
    // recv (optional): provides the `na` receive arrays of local rank i for `total` elements instead of the cache
    template <typename E>
    int exchange(int na, const std::vector<std::vector<const E*>>& in, const std::vector<std::vector<uint64_t>>& bounds,
                 std::vector<std::vector<DBuf<E>>>& out, std::vector<std::vector<uint64_t>>& rcnt,
                 const std::function<int(int, uint64_t, std::vector<DBuf<E>>&)>& recv = nullptr) {
        std::vector<std::vector<uint64_t>> mine(L);
        for (int i = 0; i < L; ++i) { mine[i].resize(P); for (int d = 0; d < P; ++d) mine[i][d] = bounds[i][d + 1] - bounds[i][d]; }
        std::vector<uint64_t> all;
        PSACX_TRY(gather(P, mine, all));
        g->n_exchanges++;
        out.clear(); out.resize(L);
        rcnt.assign(L, std::vector<uint64_t>(P, 0));
        std::vector<std::vector<uint64_t>> roff(L);
        int rc_alloc = PSACX_OK;
        for (int i = 0; i < L && rc_alloc == PSACX_OK; ++i) {
            for (int s = 0; s < P; ++s) rcnt[i][s] = all[(size_t)s * P + rank(i)];
            roff[i] = prefix_of(rcnt[i]);
            out[i].resize(na);
            if (recv) { rc_alloc = recv(i, roff[i][P], out[i]); if (rc_alloc == PSACX_OK && (int)out[i].size() != na) rc_alloc = PSACX_EINVAL; }
            else for (int a = 0; a < na && rc_alloc == PSACX_OK; ++a) { rc_alloc = out[i][a].alloc(ctx(i), roff[i][P]); if (rc_alloc != PSACX_OK) mg_set_err(g, "receive array of " + std::to_string(roff[i][P]) + " x " + std::to_string(sizeof(E)) + " bytes: " + ctx(i)->hip_err); }
        }
        PSACX_TRY(agree(rc_alloc));       // (process-per-GPU: a rank without its receive arrays must not leave its peers in the group)
        // the sources are complete when the compute streams reach this point; the receive buffers exist by then too
        for (int i = 0; i < L; ++i) {
            MRank& R = g->R[i];
            MG_HIP(g, hipSetDevice(R.ctx->device));
            MG_HIP(g, hipEventRecord(R.ev_ready, R.ctx->stream));
        }
        if (g->transport == PSACX_TR_RCCL) {
            RcclApi& nc = rccl();
            const bool self_wire = g->force_wire;
            for (int i = 0; i < L; ++i) {
                MG_HIP(g, hipSetDevice(g->R[i].ctx->device)); MG_HIP(g, hipStreamWaitEvent(g->R[i].comm_stream, g->R[i].ev_ready, 0));
                PSACX_TRY(ex_begin(g->R[i]));
            }
            MG_NCCL(g, nc.GroupStart());
            ncclResult_t bad = ncclSuccess;
            for (int i = 0; i < L && bad == ncclSuccess; ++i) {
                MRank& R = g->R[i];
                for (int a = 0; a < na && bad == ncclSuccess; ++a) {
                    for (int d = 0; d < P && bad == ncclSuccess; ++d) {
                        const uint64_t sc = mine[i][d], rc = rcnt[i][d];
                        if (d == R.grank && !self_wire) continue;
                        if (sc) { bad = wire_send(nc, R, in[i][a] + bounds[i][d], (size_t)sc * sizeof(E), d); if (d != R.grank) g->bytes_sent += sc * sizeof(E); }
                        if (rc && bad == ncclSuccess) bad = wire_recv(nc, R, out[i][a].p + roff[i][d], (size_t)rc * sizeof(E), d);
                    }
                }
            }
            const ncclResult_t end = nc.GroupEnd();           // closed on every path: an open group would swallow the next collective
            if (bad != ncclSuccess) { mg_set_err(g, std::string("ncclSend / ncclRecv: ") + nc.GetErrorString(bad)); return PSACX_MULTI_EPEER; }
            MG_NCCL(g, end);
            for (int i = 0; i < L; ++i) {
                MRank& R = g->R[i];
                MG_HIP(g, hipSetDevice(R.ctx->device));
                const uint64_t sc = mine[i][R.grank];
                for (int a = 0; a < na && sc && !self_wire; ++a)
                    MG_HIP(g, hipMemcpyAsync(out[i][a].p + roff[i][R.grank], in[i][a] + bounds[i][R.grank], (size_t)sc * sizeof(E), hipMemcpyDeviceToDevice, R.comm_stream));
                PSACX_TRY(ex_end(R));
                MG_HIP(g, hipEventRecord(R.ev_done, R.comm_stream));
                MG_HIP(g, hipStreamWaitEvent(R.ctx->stream, R.ev_done, 0));
            }
        } else if (g->transport == PSACX_TR_SHM) {
            // one process per rank on one host: the sender's stream of every array (its segments for ranks 0 .. P-1 are
            // contiguous) goes through its box of the shared segment in rounds of one box; after each round's barrier
            // every receiver picks the part of each sender's window that is addressed to it
            ShmLink& sh = g->shm;
            MRank& R = g->R[0];
            const int me = R.grank;
            MG_HIP(g, hipSetDevice(R.ctx->device));
            MG_HIP(g, hipStreamWaitEvent(R.comm_stream, R.ev_ready, 0));
            PSACX_TRY(ex_begin(R));
            std::vector<std::vector<uint64_t>> sb(P);           // sb[s][d]: start (elements) of s's segment for d inside s's stream
            uint64_t longest = 0;
            for (int s = 0; s < P; ++s) {
                std::vector<uint64_t> row(P);
                for (int d = 0; d < P; ++d) row[d] = all[(size_t)s * P + d];
                sb[s] = prefix_of(row);
                longest = std::max(longest, sb[s][P]);
            }
            const uint64_t per = std::max<uint64_t>(sh.box_bytes / sizeof(E), 1);
            std::string e;
            for (int a = 0; a < na; ++a) {
                for (uint64_t w0 = 0; w0 < longest; w0 += per) {
                    const uint64_t w1 = w0 + per;
                    if (w0 < sb[me][P]) {
                        const uint64_t len = std::min(w1, sb[me][P]) - w0;
                        MG_HIP(g, hipMemcpyAsync(sh.box(me), in[0][a] + bounds[0][0] + w0, (size_t)len * sizeof(E), hipMemcpyDeviceToHost, R.comm_stream));
                        MG_HIP(g, hipStreamSynchronize(R.comm_stream));
                    }
                    if (!sh.barrier(e)) { mg_set_err(g, e); return PSACX_MULTI_EPEER; }
                    for (int s = 0; s < P; ++s) {
                        const uint64_t lo = std::max(w0, sb[s][me]), hi = std::min(w1, sb[s][me + 1]);
                        if (lo >= hi) continue;
                        MG_HIP(g, hipMemcpyAsync(out[0][a].p + roff[0][s] + (lo - sb[s][me]), sh.box(s) + (size_t)(lo - w0) * sizeof(E), (size_t)(hi - lo) * sizeof(E),
                                                 hipMemcpyHostToDevice, R.comm_stream));
                        if (s != me) g->bytes_sent += (hi - lo) * sizeof(E);      // (counted on the receiving side: the volumes are symmetric over a step)
                    }
                    MG_HIP(g, hipStreamSynchronize(R.comm_stream));
                    if (!sh.barrier(e)) { mg_set_err(g, e); return PSACX_MULTI_EPEER; }
                }
            }
            PSACX_TRY(ex_end(R));
            MG_HIP(g, hipEventRecord(R.ev_done, R.comm_stream));
            MG_HIP(g, hipStreamWaitEvent(R.ctx->stream, R.ev_done, 0));
        } else {
            // every rank is local (possibly several on one device): the receiver's second stream pulls each piece
            // once the sender's compute stream has produced it
            if (L != P) { mg_set_err(g, "exchange without a transport between the processes"); return PSACX_EINVAL; }
            for (int i = 0; i < L; ++i) {
                MRank& R = g->R[i];
                MG_HIP(g, hipSetDevice(R.ctx->device));
                for (int s = 0; s < L; ++s) MG_HIP(g, hipStreamWaitEvent(R.comm_stream, g->R[s].ev_ready, 0));
                PSACX_TRY(ex_begin(R));
                for (int s = 0; s < L; ++s) {
                    const uint64_t rc = rcnt[i][rank(s)];
                    if (!rc) continue;
                    for (int a = 0; a < na; ++a)
                        MG_HIP(g, hipMemcpyAsync(out[i][a].p + roff[i][rank(s)], in[s][a] + bounds[s][R.grank], (size_t)rc * sizeof(E), hipMemcpyDefault, R.comm_stream));
                    if (s != i) g->bytes_sent += rc * sizeof(E) * na;
                }
                PSACX_TRY(ex_end(R));
                MG_HIP(g, hipEventRecord(R.ev_done, R.comm_stream));
            }
            // a sender may not release or overwrite its arrays before every receiver has pulled its piece
            for (int i = 0; i < L; ++i) {
                MG_HIP(g, hipSetDevice(g->R[i].ctx->device));
                for (int s = 0; s < L; ++s) MG_HIP(g, hipStreamWaitEvent(g->R[i].ctx->stream, g->R[s].ev_done, 0));
            }
        }
        return PSACX_OK;
    }

    // ---------------------------------------------------------------- message lists
    // The general form of an exchange: local rank i sends the elements [off, off + cnt) of each of its arrays in[i][a] to rank
    // `peer`, one message per list entry, and receives its recvs[i] entries (peer = source rank, off = place in out[i][a]).
    // Several messages between one pair of ranks are matched in list order.  esz[a]: element size of array a (bytes); the
    // receive arrays exist already.  done (optional): one event per local rank that is recorded on its second stream when its
    // messages have arrived; the compute streams are then NOT made to wait (the caller waits on the events when it needs the
    // data, so that later exchanges run under earlier local work).
    typedef plan::Msg Msg;          // (cnt elements at element offset off of the local array, to / from rank peer)
    int transfer(const std::vector<std::vector<const void*>>& in, const std::vector<std::vector<void*>>& out, const std::vector<size_t>& esz,
                 const std::vector<std::vector<Msg>>& sends, const std::vector<std::vector<Msg>>& recvs, std::vector<hipEvent_t>* done = nullptr) {
        const int na = (int)esz.size();
        g->n_exchanges++;
        for (int i = 0; i < L; ++i) {
            MRank& R = g->R[i];
            MG_HIP(g, hipSetDevice(R.ctx->device));
            MG_HIP(g, hipEventRecord(R.ev_ready, R.ctx->stream));
        }
        auto finish = [&](int i) -> int {
            MRank& R = g->R[i];
            PSACX_TRY(ex_end(R));
            if (done) MG_HIP(g, hipEventRecord((*done)[i], R.comm_stream));
            else { MG_HIP(g, hipEventRecord(R.ev_done, R.comm_stream)); MG_HIP(g, hipStreamWaitEvent(R.ctx->stream, R.ev_done, 0)); }
            return PSACX_OK;
        };
        if (g->transport == PSACX_TR_RCCL) {
            RcclApi& nc = rccl();
            for (int i = 0; i < L; ++i) {
                MG_HIP(g, hipSetDevice(g->R[i].ctx->device)); MG_HIP(g, hipStreamWaitEvent(g->R[i].comm_stream, g->R[i].ev_ready, 0));
                PSACX_TRY(ex_begin(g->R[i]));
            }
            MG_NCCL(g, nc.GroupStart());
            ncclResult_t bad = ncclSuccess;
            for (int i = 0; i < L && bad == ncclSuccess; ++i) {
                MRank& R = g->R[i];
                for (int a = 0; a < na && bad == ncclSuccess; ++a) {
                    for (const Msg& m : sends[i]) {
                        if (!m.cnt || (m.peer == R.grank && !g->force_wire)) continue;
                        bad = wire_send(nc, R, static_cast<const char*>(in[i][a]) + m.off * esz[a], (size_t)m.cnt * esz[a], m.peer);
                        if (m.peer != R.grank) g->bytes_sent += m.cnt * esz[a];
                        if (bad != ncclSuccess) break;
                    }
                    for (const Msg& m : recvs[i]) {
                        if (bad != ncclSuccess) break;
                        if (!m.cnt || (m.peer == R.grank && !g->force_wire)) continue;
                        bad = wire_recv(nc, R, static_cast<char*>(out[i][a]) + m.off * esz[a], (size_t)m.cnt * esz[a], m.peer);
                    }
                }
            }
            const ncclResult_t end = nc.GroupEnd();
            if (bad != ncclSuccess) { mg_set_err(g, std::string("ncclSend / ncclRecv: ") + nc.GetErrorString(bad)); return PSACX_MULTI_EPEER; }
            MG_NCCL(g, end);
            for (int i = 0; i < L; ++i) {
                MRank& R = g->R[i];
                MG_HIP(g, hipSetDevice(R.ctx->device));
                if (!g->force_wire) {                      // messages to itself: the k-th send pairs with the k-th receive
                    std::vector<const Msg*> ss, rr;
                    for (const Msg& m : sends[i]) if (m.peer == R.grank && m.cnt) ss.push_back(&m);
                    for (const Msg& m : recvs[i]) if (m.peer == R.grank && m.cnt) rr.push_back(&m);
                    if (ss.size() != rr.size()) { mg_set_err(g, "transfer: a rank's messages to itself do not pair up"); return PSACX_EINVAL; }
                    for (size_t q = 0; q < ss.size(); ++q)
                        for (int a = 0; a < na; ++a)
                            MG_HIP(g, hipMemcpyAsync(static_cast<char*>(out[i][a]) + rr[q]->off * esz[a], static_cast<const char*>(in[i][a]) + ss[q]->off * esz[a],
                                                     (size_t)ss[q]->cnt * esz[a], hipMemcpyDeviceToDevice, R.comm_stream));
                }
                PSACX_TRY(finish(i));
            }
        } else if (g->transport == PSACX_TR_SHM) {
            // every rank publishes its list of (destination, count); a rank's stream of an array is its messages back to back
            ShmLink& sh = g->shm;
            MRank& R = g->R[0];
            const int me = R.grank;
            MG_HIP(g, hipSetDevice(R.ctx->device));
            MG_HIP(g, hipStreamWaitEvent(R.comm_stream, R.ev_ready, 0));
            PSACX_TRY(ex_begin(R));
            std::string e;
            if ((sends[0].size() * 2 + 1) * 8 > sh.slot_bytes) { mg_set_err(g, "transfer: message list too long for the shared-memory slot"); return PSACX_EINVAL; }
            {
                uint64_t* sl = reinterpret_cast<uint64_t*>(sh.slot(me));
                sl[0] = sends[0].size();
                for (size_t q = 0; q < sends[0].size(); ++q) { sl[1 + 2 * q] = (uint64_t)sends[0][q].peer; sl[2 + 2 * q] = sends[0][q].cnt; }
            }
            if (!sh.barrier(e)) { mg_set_err(g, e); return PSACX_MULTI_EPEER; }
            // for every source: the stream offsets (elements) of its messages to me, in order
            std::vector<std::vector<std::pair<uint64_t, uint64_t>>> from(P);
            std::vector<uint64_t> slen(P, 0);
            for (int s = 0; s < P; ++s) {
                const uint64_t* sl = reinterpret_cast<const uint64_t*>(sh.slot(s));
                uint64_t at = 0;
                for (uint64_t q = 0; q < sl[0]; ++q) { if ((int)sl[1 + 2 * q] == me) from[s].emplace_back(at, sl[2 + 2 * q]); at += sl[2 + 2 * q]; }
                slen[s] = at;
            }
            if (!sh.barrier(e)) { mg_set_err(g, e); return PSACX_MULTI_EPEER; }
            // my receives from s, in order, take those pieces
            std::vector<std::vector<const Msg*>> mine_from(P);
            for (const Msg& m : recvs[0]) mine_from[m.peer].push_back(&m);
            for (int s = 0; s < P; ++s) {
                if (mine_from[s].size() != from[s].size()) { mg_set_err(g, "transfer: send and receive lists of a pair of ranks differ in length"); return PSACX_EINVAL; }
                for (size_t q = 0; q < from[s].size(); ++q) if (mine_from[s][q]->cnt != from[s][q].second) { mg_set_err(g, "transfer: send and receive counts differ"); return PSACX_EINVAL; }
            }
            uint64_t longest = 0;
            for (int s = 0; s < P; ++s) longest = std::max(longest, slen[s]);
            for (int a = 0; a < na; ++a) {
                const uint64_t per = std::max<uint64_t>(sh.box_bytes / esz[a], 1);
                for (uint64_t w0 = 0; w0 < longest; w0 += per) {
                    const uint64_t w1 = w0 + per;
                    // my stream: messages back to back (their places in the source array are arbitrary)
                    {
                        uint64_t at = 0;
                        for (const Msg& m : sends[0]) {
                            const uint64_t lo = std::max(w0, at), hi = std::min(w1, at + m.cnt);
                            if (lo < hi) MG_HIP(g, hipMemcpyAsync(sh.box(me) + (size_t)(lo - w0) * esz[a], static_cast<const char*>(in[0][a]) + (m.off + (lo - at)) * esz[a],
                                                                  (size_t)(hi - lo) * esz[a], hipMemcpyDeviceToHost, R.comm_stream));
                            at += m.cnt;
                        }
                        MG_HIP(g, hipStreamSynchronize(R.comm_stream));
                    }
                    if (!sh.barrier(e)) { mg_set_err(g, e); return PSACX_MULTI_EPEER; }
                    for (int s = 0; s < P; ++s)
                        for (size_t q = 0; q < from[s].size(); ++q) {
                            const uint64_t a0 = from[s][q].first, lo = std::max(w0, a0), hi = std::min(w1, a0 + from[s][q].second);
                            if (lo >= hi) continue;
                            MG_HIP(g, hipMemcpyAsync(static_cast<char*>(out[0][a]) + (mine_from[s][q]->off + (lo - a0)) * esz[a], sh.box(s) + (size_t)(lo - w0) * esz[a],
                                                     (size_t)(hi - lo) * esz[a], hipMemcpyHostToDevice, R.comm_stream));
                            if (s != me) g->bytes_sent += (hi - lo) * esz[a];
                        }
                    MG_HIP(g, hipStreamSynchronize(R.comm_stream));
                    if (!sh.barrier(e)) { mg_set_err(g, e); return PSACX_MULTI_EPEER; }
                }
            }
            PSACX_TRY(finish(0));
        } else {
            if (L != P) { mg_set_err(g, "transfer without a transport between the processes"); return PSACX_EINVAL; }
            // per sender and destination: its messages in order
            std::vector<std::vector<std::vector<const Msg*>>> to(L, std::vector<std::vector<const Msg*>>(P));
            for (int s = 0; s < L; ++s) for (const Msg& m : sends[s]) to[s][m.peer].push_back(&m);
            for (int i = 0; i < L; ++i) {
                MRank& R = g->R[i];
                MG_HIP(g, hipSetDevice(R.ctx->device));
                for (int s = 0; s < L; ++s) MG_HIP(g, hipStreamWaitEvent(R.comm_stream, g->R[s].ev_ready, 0));
                PSACX_TRY(ex_begin(R));
                std::vector<size_t> taken(L, 0);
                for (const Msg& m : recvs[i]) {
                    int ls = -1;
                    for (int s = 0; s < L; ++s) if (rank(s) == m.peer) ls = s;
                    if (ls < 0 || taken[ls] >= to[ls][R.grank].size() || to[ls][R.grank][taken[ls]]->cnt != m.cnt) { mg_set_err(g, "transfer: send and receive lists do not match"); return PSACX_EINVAL; }
                    const Msg* sm = to[ls][R.grank][taken[ls]++];
                    if (!m.cnt) continue;
                    for (int a = 0; a < na; ++a)
                        MG_HIP(g, hipMemcpyAsync(static_cast<char*>(out[i][a]) + m.off * esz[a], static_cast<const char*>(in[ls][a]) + sm->off * esz[a], (size_t)m.cnt * esz[a],
                                                 hipMemcpyDefault, R.comm_stream));
                    if (ls != i) for (int a = 0; a < na; ++a) g->bytes_sent += m.cnt * esz[a];
                }
                PSACX_TRY(ex_end(R));
                MG_HIP(g, hipEventRecord(done ? (*done)[i] : R.ev_done, R.comm_stream));
            }
            if (!done)       // a sender may not release or overwrite its arrays before every receiver has pulled its piece
                for (int i = 0; i < L; ++i) {
                    MG_HIP(g, hipSetDevice(g->R[i].ctx->device));
                    for (int s = 0; s < L; ++s) MG_HIP(g, hipStreamWaitEvent(g->R[i].ctx->stream, g->R[s].ev_done, 0));
                }
        }
        return PSACX_OK;
    }

    // ---------------------------------------------------------------- small helpers
    int fetch(int i, const T* a, const std::vector<uint64_t>& idx, std::vector<uint64_t>& out) {
        out.assign(idx.size(), 0);
        if (idx.empty()) return PSACX_OK;
        psacx_ctx* c = ctx(i);
        MG_HIP(g, hipSetDevice(c->device));
        const size_t k = idx.size();
        MG_OP(g, c, ensure_pinned(c, 2 * k * 8 + 65536));
        DBuf<uint64_t> d; MG_OP(g, c, d.alloc(c, 2 * k));
        uint64_t* h = reinterpret_cast<uint64_t*>(c->pinned + 32768);
        std::memcpy(h, idx.data(), k * 8);
        MG_HIP(g, hipMemcpyAsync(d.p, h, k * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL((gather_at_kernel<T>), dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, a, d.p, (unsigned)k, d.p + k);
        MG_HIP(g, hipGetLastError());
        MG_HIP(g, hipMemcpyAsync(h, d.p + k, k * 8, hipMemcpyDeviceToHost, c->stream));
        MG_HIP(g, hipStreamSynchronize(c->stream));
        std::memcpy(out.data(), h, k * 8);
        return PSACX_OK;
    }

    // stable local sort of (k1, k2, v) by the low bits1 / bits2 bits; the arrays are replaced by the sorted ones
    int local_sort(int i, Rec<T>& rec, unsigned bits1, unsigned bits2) {
        psacx_ctx* c = ctx(i);
        if (rec.cnt < 2) return PSACX_OK;
        Rec<T> alt;
        PSACX_TRY(take3(i, alt, rec.cnt));
        int32_t where = 0;
        MG_OP(g, c, op_pair_sort<T>(c, rec.k1.p, rec.k2.p, rec.v.p, alt.k1.p, alt.k2.p, alt.v.p, rec.cnt, bits1, bits2, &where));
        if (where) swap3(rec, alt);
        drop3(i, alt);
        return PSACX_OK;
    }

    // The local sort of the first round in two stages, as the one-GPU engine does it (construct.hpp): when the leading
    // `lead` bits of word 1 separate almost every suffix of the whole text, the records are sorted on those bits only
    // (DNA, 64-bit words, 2^34 characters: 5 instead of 11 passes) and the few groups that still tie are ordered by
    // (word 1, word 2) in registers (tie_resolve_kernel reading word 2 from the record).  A group longer than 8
    // (repetitive text) falls back to the full stable sort, which is correct on the partly ordered arrays.
    int local_sort_first(int i, Rec<T>& rec, unsigned bits1, unsigned bits2) {
        psacx_ctx* c = ctx(i);
        unsigned lead = (bits_for(n - 1) + 3 + RADIX_BITS - 1) / RADIX_BITS * RADIX_BITS;
        const bool two_stage = rec.cnt >= (1ull << 21) && lead <= bits1 && lead + RADIX_BITS <= bits1 + bits2 && !one_stage_env_;
        if (!two_stage) return local_sort(i, rec, bits1, bits2);
        const unsigned lo1 = bits1 - lead;
        {
            Rec<T> alt;
            PSACX_TRY(take3(i, alt, rec.cnt));
            int32_t where = 0;
            MG_OP(g, c, op_pair_sort<T>(c, rec.k1.p, rec.k2.p, rec.v.p, alt.k1.p, alt.k2.p, alt.v.p, rec.cnt, bits1, 0, &where, lo1));
            if (where) swap3(rec, alt);
            drop3(i, alt);
        }
        constexpr int TB_ = 256, TI_ = 16, TG_ = 8;
        DBuf<unsigned long long> big; MG_OP(g, c, big.alloc(c, 1));
        MG_HIP(g, hipSetDevice(c->device));
        MG_HIP(g, hipMemsetAsync(big.p, 0, 8, c->stream));
        const uint64_t nb = (rec.cnt + (uint64_t)TB_ * TI_ - 1) / ((uint64_t)TB_ * TI_);
        CodeTable tab; std::memset(&tab, 0, sizeof(tab));
        KeyShape ks; std::memset(&ks, 0, sizeof(ks));
        hipLaunchKernelGGL((tie_resolve_kernel<T, TB_, TI_, TG_, true>), dim3((unsigned)nb), dim3(TB_), 0, c->stream, rec.k1.p, rec.v.p, rec.k2.p,
                           rec.cnt, lo1, (const uint8_t*)nullptr, (uint64_t)0, tab, ks, big.p);
        MG_HIP(g, hipGetLastError());
        MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, big.p, 8, hipMemcpyDeviceToHost, c->stream));
        MG_HIP(g, hipStreamSynchronize(c->stream));
        if (*reinterpret_cast<unsigned long long*>(c->pinned + 32768)) return local_sort(i, rec, bits1, bits2);
        return PSACX_OK;
    }

    // first / last record of every rank's block (has, 3 + 3 words) -> nearest non-empty neighbours of each local rank
    int neighbours(const std::vector<const T*>& a1, const std::vector<const T*>& a2, const std::vector<const T*>& a3,
                   const std::vector<uint64_t>& cnt, int words, std::vector<psacx_boundary>& bd) {
        std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(7, 0));
        PSACX_TRY(par([&](int i) -> int {
            if (!cnt[i]) return PSACX_OK;
            mine[i][0] = 1;
            const T* arr[3] = {a1[i], a2[i], a3[i]};
            for (int w = 0; w < words; ++w) {
                std::vector<uint64_t> o;
                PSACX_TRY(fetch(i, arr[w], {0, cnt[i] - 1}, o));
                mine[i][1 + w] = o[0]; mine[i][4 + w] = o[1];
            }
            return PSACX_OK;
        }));
        std::vector<uint64_t> all;
        PSACX_TRY(gather(7, mine, all));
        bd.assign(L, psacx_boundary());
        for (int i = 0; i < L; ++i) {
            std::memset(&bd[i], 0, sizeof(psacx_boundary));
            const int r = rank(i);
            for (int s = r - 1; s >= 0; --s) if (all[(size_t)s * 7]) { bd[i].has_prev = 1; for (int w = 0; w < 3; ++w) bd[i].prev[w] = all[(size_t)s * 7 + 4 + w]; break; }
            for (int s = r + 1; s < P; ++s) if (all[(size_t)s * 7]) { bd[i].has_next = 1; for (int w = 0; w < 3; ++w) bd[i].next[w] = all[(size_t)s * 7 + 1 + w]; break; }
        }
        return PSACX_OK;
    }

    // ---------------------------------------------------------------- distributed primitives (see the table above)
    // Sorts the records of all ranks by (k1, k2); rank r ends with exactly targets[r] records, the concatenation
    // over ranks being sorted -- the contract psac needs from mxx::sort (idxsort.hpp:67-79).
    int dist_sort(std::vector<Rec<T>>& rec, const std::vector<uint64_t>& targets, unsigned bits1, unsigned bits2, bool first_round = false) {
        ++sort_calls_;
        if (solo_) return first_round ? local_sort_first(0, rec[0], bits1, bits2) : local_sort(0, rec[0], bits1, bits2);
        // 8192 samples per rank: with P ranks a rank's share deviates by about sqrt(P) / sqrt(8192 P) of a block (1.1 %), so the
        // 12.5 % slack of the reduced-memory layout's record arrays is nine standard deviations away
        constexpr int SAMPLES = 8192;
        // One sample from a pseudo-random place in each of SAMPLES equal strata of the local records, made unique by
        // (rank, index) so that ties are divided.  (Evenly spaced samples alias with periodic text: in a tandem repeat whose
        // period divides the spacing every sample of every rank carries the same key, and one rank received 2.8 blocks.)
        std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(1 + 3 * SAMPLES, 0));
        PSACX_TRY(par([&](int i) -> int {
            const std::vector<uint64_t> pos = plan::sample_positions(rec[i].cnt, rank(i), sort_calls_, SAMPLES);
            std::vector<uint64_t> a, b;
            PSACX_TRY(fetch(i, rec[i].k1.p, pos, a));
            PSACX_TRY(fetch(i, rec[i].k2.p, pos, b));
            mine[i][0] = pos.size();
            for (size_t s = 0; s < pos.size(); ++s) { mine[i][1 + 3 * s] = a[s]; mine[i][2 + 3 * s] = b[s]; mine[i][3 + 3 * s] = pos[s]; }
            return PSACX_OK;
        }));
        std::vector<uint64_t> all;
        PSACX_TRY(gather(1 + 3 * SAMPLES, mine, all));
        typedef plan::Smp Smp;
        std::vector<Smp> flat;
        for (int r = 0; r < P; ++r) {
            const uint64_t* row = &all[(size_t)r * (1 + 3 * SAMPLES)];
            for (uint64_t s = 0; s < row[0]; ++s) flat.push_back(Smp{row[1 + 3 * s], row[2 + 3 * s], (uint64_t)r, row[3 + 3 * s]});
        }
        const std::vector<Smp> spl = plan::choose_splitters(std::move(flat), P);
        const uint32_t ns = (uint32_t)spl.size();
        std::vector<uint64_t> s1(ns + 1), s2(ns + 1), sr(ns + 1), sp(ns + 1);
        for (uint32_t s = 0; s < ns; ++s) { s1[s] = spl[s].k1; s2[s] = spl[s].k2; sr[s] = spl[s].r; sp[s] = spl[s].p; }
        // classify + one stable partition pass by destination
        std::vector<Rec<T>> grp(L);
        std::vector<std::vector<uint64_t>> bounds(L);
        std::vector<std::vector<const T*>> in(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            const uint64_t cn = rec[i].cnt;
            PSACX_TRY(take3(i, grp[i], cn));
            std::vector<uint64_t> cs(ns + 2, 0);
            MG_OP(g, c, op_split_by<T>(c, rec[i].k1.p, rec[i].k2.p, rec[i].v.p, cn, s1.data(), s2.data(), sr.data(), sp.data(), ns,
                                      (uint64_t)rank(i), grp[i].k1.p, grp[i].k2.p, grp[i].v.p, cs.data()));
            bounds[i].assign(P + 1, cn);
            for (uint32_t d = 0; d <= ns; ++d) bounds[i][d] = cs[d];
            drop3(i, rec[i]);
            in[i] = {grp[i].k1.p, grp[i].k2.p, grp[i].v.p};
            return PSACX_OK;
        }));
        mark("    sort: samples + partition");
        std::vector<std::vector<DBuf<T>>> got;
        std::vector<std::vector<uint64_t>> rc;
        const std::function<int(int, uint64_t, std::vector<DBuf<T>>&)> recv3 = [this](int i, uint64_t tot, std::vector<DBuf<T>>& o) -> int {
            Rec<T> r;
            PSACX_TRY(take3(i, r, tot));
            o.clear(); o.resize(3);
            o[0] = std::move(r.k1); o[1] = std::move(r.k2); o[2] = std::move(r.v);
            return PSACX_OK;
        };
        PSACX_TRY(exchange<T>(3, in, bounds, got, rc, recv3));
        mark("    sort: shuffle");
        std::vector<uint64_t> c2(L);
        PSACX_TRY(par([&](int i) -> int {
            drop3(i, grp[i]);
            rec[i].k1 = std::move(got[i][0]); rec[i].k2 = std::move(got[i][1]); rec[i].v = std::move(got[i][2]);
            rec[i].cnt = c2[i] = rec[i].k1.n;
            if (first_round) PSACX_TRY(local_sort_first(i, rec[i], bits1, bits2));
            else PSACX_TRY(local_sort(i, rec[i], bits1, bits2));
            return PSACX_OK;
        }));
        mark("    sort: local sort");
        return rebalance(rec, targets);
    }


    // exact re-balance of globally sorted records to the block sizes: the j-th record of rank r has global index G[r] + j
    int rebalance(std::vector<Rec<T>>& rec, const std::vector<uint64_t>& targets) {
        std::vector<uint64_t> c2(L), counts;
        for (int i = 0; i < L; ++i) c2[i] = rec[i].cnt;
        PSACX_TRY(gather1(c2, counts));
        if (counts == targets) return PSACX_OK;
        const std::vector<uint64_t> G = prefix_of(counts), TP = prefix_of(targets);
        std::vector<std::vector<uint64_t>> bounds(L), rc;
        std::vector<std::vector<const T*>> in(L);
        std::vector<std::vector<DBuf<T>>> got;
        const std::function<int(int, uint64_t, std::vector<DBuf<T>>&)> recv3 = [this](int i, uint64_t tot, std::vector<DBuf<T>>& o) -> int {
            Rec<T> r;
            PSACX_TRY(take3(i, r, tot));
            o.clear(); o.resize(3);
            o[0] = std::move(r.k1); o[1] = std::move(r.k2); o[2] = std::move(r.v);
            return PSACX_OK;
        };
        for (int i = 0; i < L; ++i) {
            bounds[i] = plan::rebalance_bounds(G[rank(i)], c2[i], TP);
            in[i] = {rec[i].k1.p, rec[i].k2.p, rec[i].v.p};
        }
        PSACX_TRY(exchange<T>(3, in, bounds, got, rc, recv3));
        for (int i = 0; i < L; ++i) {
            drop3(i, rec[i]);
            rec[i].k1 = std::move(got[i][0]); rec[i].k2 = std::move(got[i][1]); rec[i].v = std::move(got[i][2]);
            rec[i].cnt = rec[i].k1.n;
        }
        return PSACX_OK;
    }

    // Re-balance without a copy (after sort_first_one_word): local rank i holds the globally sorted records held_from_[r] .. + held_cnt_[r]
    // (r its rank) head_[i] elements into arrays of room_[i] elements, and its block starts at most head_[i] records before them: the pieces
    // other ranks hold of it are received in front of and behind its own records, where the arrays have room, and the block then begins
    // at the start of the arrays.  All three arrays of a record set travel in one group of messages.
    std::vector<uint64_t> held_from_, held_cnt_;
    int rebalance_in_place(std::vector<Rec<T>>& rec, const std::vector<uint64_t>& targets) {
        const std::vector<uint64_t> TP = prefix_of(targets);
        std::vector<std::vector<Msg>> sends(L), recvs(L);
        std::vector<std::vector<const void*>> in(L);
        std::vector<std::vector<void*>> out(L);
        for (int i = 0; i < L; ++i) {
            if (!plan::in_place_messages(rank(i), P, held_from_, held_cnt_, TP, head_[i], sends[i], recvs[i])) {
                mg_set_err(g, "re-balance in place: a rank does not hold the tail of its block"); return PSACX_EINVAL;
            }
            T* b1 = rec[i].k1.p - head_[i]; T* b2 = rec[i].k2.p - head_[i]; T* b3 = rec[i].v.p - head_[i];
            in[i] = {b1, b2, b3}; out[i] = {b1, b2, b3};
        }
        PSACX_TRY(transfer(in, out, {sizeof(T), sizeof(T), sizeof(T)}, sends, recvs));
        for (int i = 0; i < L; ++i) {
            rec[i].k1.rewind(head_[i]); rec[i].k2.rewind(head_[i]); rec[i].v.rewind(head_[i]);
            rec[i].cnt = targets[rank(i)];
            rec[i].k1.n = rec[i].k2.n = rec[i].v.n = rec[i].cnt;
        }
        return PSACX_OK;
    }

    // ---------------------------------------------------------------- the first round in two-word / one-word records: multi_first_round.hpp
    static constexpr int PSACX_RETRY_ = 1;
    int dist_windows(const std::vector<DBuf<uint8_t>>& tbuf, uint32_t two_k, const CodeTable& tab, const KeyShape& ks, const std::vector<const T*>& gidx, const std::vector<uint64_t>& cnt, std::vector<DBuf<T>>& w1, std::vector<DBuf<T>>& w2);
    int sort_first_two_word(std::vector<Rec<T>>& rec, const std::vector<uint64_t>& targets, unsigned bits1, unsigned bits2, unsigned lo1, const std::vector<DBuf<uint8_t>>& tbuf, uint32_t two_k, const CodeTable& tab, const KeyShape& ks, bool trust, uint64_t spec_front);
    // (defined in multi_refine.hpp)
    int refine_sort(std::vector<Rec<T>>& rec, const std::vector<const T*>& plist, const std::vector<uint64_t>& counts, unsigned bits1, unsigned bits2);
    int dist_range_min(const std::vector<const T*>& lo, const std::vector<const T*>& hi, const std::vector<uint64_t>& cnt,
                       std::vector<DBuf<T>>& out);
    int first_sort_ties(std::vector<Rec<T>>& rec, const std::vector<uint64_t>& targets, unsigned bits1, unsigned bits2, unsigned lo1, const std::vector<DBuf<uint8_t>>& tbuf, uint32_t two_k, const CodeTable& tab, const KeyShape& ks, bool word1_gone);
    int sort_first_one_word(std::vector<Rec<T>>& rec, const std::vector<uint64_t>& targets, unsigned bits1, unsigned bits2, const std::vector<DBuf<uint8_t>>& tbuf, uint32_t two_k, const CodeTable& tab, const KeyShape& ks, bool trust, uint64_t spec, unsigned* lo1_out);

    // Stable partition of global positions `gidx` and one payload array by owner rank: the owner of every position
    // is computed into a class array, one pass of the radix scatter kernel over two-word records (position, payload)
    // groups them by that class.  out.k2 = positions, out.v = payloads, bounds[d] = start of the records for rank d.
    int route(int i, const T* gidx, const T* payload, uint64_t cnt, Rec<T>& out, std::vector<uint64_t>& bounds) {
        psacx_ctx* c = ctx(i);
        out.cnt = cnt;
        MG_OP(g, c, out.k2.alloc(c, cnt)); MG_OP(g, c, out.v.alloc(c, cnt));
        bounds.assign(P + 1, cnt);
        bounds[0] = 0;
        if (cnt == 0) return PSACX_OK;
        MG_OP(g, c, ensure_pinned(c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
        SortScratch sc;
        T* cls = nullptr;
        auto layout = [&](Arena& a) {
            cls = a.take<T>(cnt);
            sc.d_base = a.take<unsigned long long>((size_t)MAX_PASSES * RADIX);
            sc.desc_bytes = sort_desc_bytes(cnt);
            sc.d_desc = a.take<char>(sc.desc_bytes);
        };
        { Arena dry(nullptr); layout(dry); MG_OP(g, c, ensure_slab(c, dry.off + 4096)); }
        Arena ar(c->slab);
        layout(ar);
        MG_OP(g, c, psacx_op_owners(c, gidx, cnt, n, (uint32_t)P, cls));
        SortBufs<T> in{const_cast<T*>(gidx), nullptr, const_cast<T*>(payload)}, o{out.k2.p, nullptr, out.v.p};
        unsigned long long* starts = reinterpret_cast<unsigned long long*>(c->pinned + 1024);
        MG_OP(g, c, class_partition<T>(c, sc, in, o, cls, cnt, starts));
        for (int d = 0; d < P; ++d) bounds[d] = starts[d];
        return PSACX_OK;
    }
    // the same pass with the classes given (cls[j] < P)
    int route_by(int i, const T* cls, const T* key, const T* payload, uint64_t cnt, Rec<T>& out, std::vector<uint64_t>& bounds) {
        psacx_ctx* c = ctx(i);
        out.cnt = cnt;
        MG_OP(g, c, out.k2.alloc(c, cnt)); MG_OP(g, c, out.v.alloc(c, cnt));
        bounds.assign(P + 1, cnt);
        bounds[0] = 0;
        if (cnt == 0) return PSACX_OK;
        MG_OP(g, c, ensure_pinned(c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
        SortScratch sc;
        auto layout = [&](Arena& a) {
            sc.d_base = a.take<unsigned long long>((size_t)MAX_PASSES * RADIX);
            sc.desc_bytes = sort_desc_bytes(cnt);
            sc.d_desc = a.take<char>(sc.desc_bytes);
        };
        { Arena dry(nullptr); layout(dry); MG_OP(g, c, ensure_slab(c, dry.off + 4096)); }
        Arena ar(c->slab);
        layout(ar);
        SortBufs<T> in{const_cast<T*>(key), nullptr, const_cast<T*>(payload)}, o{out.k2.p, nullptr, out.v.p};
        unsigned long long* starts = reinterpret_cast<unsigned long long*>(c->pinned + 1024);
        MG_OP(g, c, class_partition<T>(c, sc, in, o, cls, cnt, starts));
        for (int d = 0; d <= P; ++d) bounds[d] = starts[d];       // bounds[P]: start of class P ("to nobody"), cnt if there is none
        return PSACX_OK;
    }
    static int psacx_op_owners(psacx_ctx* c, const T* gi, uint64_t cnt, uint64_t n, uint32_t P, T* out) {
        OP_PROLOGUE(c); SIMPLE_LAUNCH(c, (owners_kernel<T>), cnt, gi, cnt, make_dist(n, P), out); return PSACX_OK;
    }
    static int op_take(psacx_ctx* c, const T* b, const T* gi, uint64_t cnt, uint64_t off, uint64_t n, T* o) {
        OP_PROLOGUE(c); SIMPLE_LAUNCH(c, (take_kernel<T>), cnt, b, gi, cnt, off, n, o); return PSACX_OK;
    }
    static int op_put(psacx_ctx* c, T* b, const T* gi, uint64_t cnt, uint64_t off, const T* v, int64_t delta) {
        OP_PROLOGUE(c); SIMPLE_LAUNCH(c, (put_kernel<T>), cnt, b, gi, cnt, off, v, delta); return PSACX_OK;
    }

    // block[gidx - off_owner] = vals + delta on the owner of every global position (bulk_permute.hpp:14-73)
    int dist_put(const std::vector<T*>& block, const std::vector<const T*>& gidx, const std::vector<const T*>& vals,
                 const std::vector<uint64_t>& cnt, int64_t delta, bool permutation) {
        std::vector<Rec<T>> routed(L);
        std::vector<std::vector<DBuf<T>>> got;
        std::vector<const T*> gi(L), vi(L);
        std::vector<uint64_t> rc_tot(L);
        if (solo_) { gi[0] = gidx[0]; vi[0] = vals[0]; rc_tot[0] = cnt[0]; }
        else {
            std::vector<std::vector<uint64_t>> bounds(L), rc;
            std::vector<std::vector<const T*>> in(L);
            for (int i = 0; i < L; ++i) { PSACX_TRY(route(i, gidx[i], vals[i], cnt[i], routed[i], bounds[i])); in[i] = {routed[i].k2.p, routed[i].v.p}; }
            PSACX_TRY(exchange<T>(2, in, bounds, got, rc));
            for (int i = 0; i < L; ++i) { gi[i] = got[i][0].p; vi[i] = got[i][1].p; rc_tot[i] = got[i][0].n; }
        }
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            if (permutation && delta == -1 && rc_tot[i]) {
                DBuf<T> s[4];
                for (int q = 0; q < 4; ++q) MG_OP(g, c, s[q].alloc(c, rc_tot[i]));
                MG_OP(g, c, op_put_perm<T>(c, block[i], gi[i], rc_tot[i], S[i].off, vi[i], s[0].p, s[1].p, s[2].p, s[3].p));
            } else MG_OP(g, c, op_put(c, block[i], gi[i], rc_tot[i], S[i].off, vi[i], delta));
            return PSACX_OK;
        }));
        return PSACX_OK;
    }

    // ISA[SA[j]] = Bsa[j] - 1 for the full permutation of the first round (bulk_permute_inplace, bulk_permute.hpp:14-73),
    // slice by slice: see slice_inv.hpp.  Needs blocks of at most 2^32 positions (32-bit block-relative keys).
    // V: type the ranks travel in (32 bits while the whole text has at most 2^32 characters).
    // ids_in_isa: the bucket ids still sit in the ISA array (reduced-memory layout); the first level copies them to S[i].Bsa
    template <typename V>
    int isa_by_slices_t(bool ids_in_isa) {
        constexpr unsigned WBMAX = sizeof(V) == 4 ? 14 : 13;
        constexpr int PB = 512, PI = 16;
        uint64_t max_m = 0;
        for (int r = 0; r < P; ++r) max_m = std::max(max_m, sizes[r]);
        // slice / window / level widths (multi_plan.hpp: slice_shape)
        const plan::SliceShape shape = plan::slice_shape(max_m, (unsigned)P, WBMAX, (unsigned)SLICE_MAX_CLASSES, slice_wb_env_, slice_s1_env_);
        const unsigned sb = shape.sb, spo = shape.spo, wb = shape.wb, rbits = shape.rbits, levels2 = shape.levels2, C = shape.C;
        const std::vector<unsigned>& cbs = shape.cbs;
        SliceMap map;
        map.div = n / P; map.mod = n % P; map.P = (unsigned)P; map.sb = sb; map.spo = spo; map.dshift = -1;
        if (map.mod == 0 && map.div && (map.div & (map.div - 1)) == 0) { map.dshift = 0; while ((1ull << map.dshift) < map.div) ++map.dshift; }
        const uint64_t slice = 1ull << sb;
        // ranks below 2^32: a pair is one 64-bit entry (position | rank << 32) on the wire and in every level (slice_inv.hpp:
        // *_packed_kernel; PSACX_SLICE_TWO_ARRAYS=1 keeps the two-array form)
        const bool pack = sizeof(V) == 4;
        // ranks beyond 2^32: on the wire as 32 bits relative to the end of the sender's block, packed with the position (8 instead
        // of 12 bytes per pair; slice_inv.hpp: SliceDecode), when no bucket of unresolved suffixes reaches further back than 2^32
        // positions from the end of its rank's block; the first owner-side kernel widens them.  PSACX_SLICE_ABS=1: 64-bit ranks.
        bool wpack = false;
        if (sizeof(V) == 8) {
            std::vector<uint64_t> okv(L, 1), all;
            PSACX_TRY(par([&](int i) -> int {
                if (!S[i].m) return PSACX_OK;
                std::vector<uint64_t> o;
                PSACX_TRY(fetch(i, ids_in_isa ? (const T*)S[i].ISA : (const T*)S[i].Bsa.p, {0}, o));
                okv[i] = (o[0] - 1) + (1ull << 32) >= S[i].off + S[i].m ? 1 : 0;       // (ranks ascend along a block: the first is the smallest)
                return PSACX_OK;
            }));
            PSACX_TRY(gather1(okv, all));
            wpack = true;
            for (uint64_t v : all) if (!v) wpack = false;
        }

        // 1. pairs per class on every rank; every rank learns the whole table
        std::vector<std::vector<uint64_t>> counts(L, std::vector<uint64_t>(C, 0));
        std::vector<DBuf<unsigned long long>> d_cnt(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, d_cnt[i].alloc(c, SLICE_MAX_CLASSES));
            MG_HIP(g, hipSetDevice(c->device));
            MG_HIP(g, hipMemsetAsync(d_cnt[i].p, 0, SLICE_MAX_CLASSES * 8, c->stream));
            if (S[i].m) {
                hipLaunchKernelGGL((slice_hist_kernel<T>), dim3(grid_for(c, S[i].m, 512, 8)), dim3(512), 0, c->stream, (const T*)S[i].SA, S[i].m, map, d_cnt[i].p);
                MG_HIP(g, hipGetLastError());
            }
            MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, d_cnt[i].p, (size_t)C * 8, hipMemcpyDeviceToHost, c->stream));
            MG_HIP(g, hipStreamSynchronize(c->stream));
            std::memcpy(counts[i].data(), c->pinned + 32768, (size_t)C * 8);
            return PSACX_OK;
        }));
        std::vector<uint64_t> table;
        PSACX_TRY(gather((int)C, counts, table));
        auto slice_len = [&](int owner, unsigned sl) -> uint64_t { const uint64_t lo = (uint64_t)sl << sb; return sizes[owner] > lo ? std::min(slice, sizes[owner] - lo) : 0; };
        for (int o = 0; o < P; ++o)
            for (unsigned sl = 0; sl < spo; ++sl) {
                uint64_t tot = 0;
                for (int r = 0; r < P; ++r) tot += table[(size_t)r * C + o * spo + sl];
                if (tot != slice_len(o, sl)) { mg_set_err(g, "SA -> ISA: the suffix array is not a permutation (a destination slice receives the wrong number of entries)"); return PSACX_EDEVICE; }
            }
        // 2. first level on every rank.  All arrays of this routine are cut from a few byte blocks; in the reduced-memory layout
        //    the blocks have the one size every record array of the first round had, so the rank's cache serves them (a miss
        //    there means hipFree + hipMalloc of tens of GB: about a second each)
        //    (normal layout: every array its own block of the usual array size, which the cache holds from the sort)
        struct Cut {
            DBuf<uint8_t> b; size_t used = 0, cap = 0;
            void* take(size_t bytes) { used = (used + 255) & ~(size_t)255; void* q = b.p + used; used += bytes; return q; }
        };
        struct Ptrs { uint32_t* k; V* v; };
        std::vector<std::vector<Cut>> blocks(L);
        std::vector<Ptrs> pk_(L), A0(L), A1(L), Bb(L);
        std::vector<unsigned*> cur(L, nullptr);
        const uint64_t G = plan::slices_per_step(shape, max_m, diet, slice_step_env_);     // (one rank without the wire: the levels work on a step's part of the class array)
        const uint64_t nsteps = (spo + G - 1) / G;
        const uint64_t step_cap = G << sb;
        std::vector<std::vector<uint64_t>> cstart(L);
        const int rc_part = par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            const uint64_t m = S[i].m;
            cstart[i] = prefix_of(counts[i]);
            const size_t std_bytes = diet ? (size_t)(m + m / 8 + 256) * sizeof(T) : (size_t)m * sizeof(T);
            int rc_a = PSACX_OK;
            auto arr = [&](size_t bytes) -> void* {
                auto& v = blocks[i];
                bytes = std::max<size_t>(bytes, 256);        // (exactly the usual array size when that is what an array needs: the cache holds such blocks)
                if (diet && !v.empty() && ((v.back().used + 255) & ~(size_t)255) + bytes <= v.back().cap) return v.back().take(bytes);
                v.emplace_back();
                Cut& ct = v.back();
                const size_t res = bytes <= std_bytes ? std_bytes : 0;
                rc_a = ct.b.alloc(c, bytes, res);
                if (rc_a != PSACX_OK) { mg_set_err(g, "SA -> ISA arrays: " + c->hip_err); return nullptr; }
                ct.cap = std::max(bytes, res); ct.used = 0;
                return ct.take(bytes);
            };
            auto pair = [&](Ptrs& q, uint64_t cnt, bool packed = false) {
                if (pack || packed) { q.k = (uint32_t*)arr((size_t)cnt * 8); q.v = nullptr; }
                else { q.k = (uint32_t*)arr((size_t)cnt * 4); q.v = rc_a == PSACX_OK ? (V*)arr((size_t)cnt * sizeof(V)) : nullptr; }
            };
            pair(pk_[i], m, wpack && !solo_);        // (one rank: the levels write back into this set, so it keeps both arrays; the packed entries go to its value array)
            const uint64_t cap = std::min<uint64_t>(step_cap, std::max<uint64_t>(m, 1));
            if (diet && rc_a == PSACX_OK && cap * 4 <= m) {
                // the step arrays are small beside the block: one block of exactly their size (a block of the usual size for them
                // would count a whole array against the rank's memory)
                const size_t per = (pack ? (size_t)cap * 8 : (size_t)cap * (4 + sizeof(V))) + 512;
                const size_t small = per * ((solo_ ? 0 : 1) + ((!solo_ && nsteps > 1) ? 1 : 0) + (levels2 ? 1 : 0)) + (levels2 ? ((cap >> wb) + 2) * sizeof(unsigned) + 256 : 0) + 256;
                auto& v = blocks[i];
                v.emplace_back();
                Cut& ct = v.back();
                rc_a = ct.b.alloc(c, small, 0);
                if (rc_a != PSACX_OK) mg_set_err(g, "SA -> ISA step arrays: " + c->hip_err);
                ct.cap = small; ct.used = 0;
            }
            if (rc_a == PSACX_OK && !solo_) pair(A0[i], cap);
            if (rc_a == PSACX_OK && !solo_ && nsteps > 1) pair(A1[i], cap);
            if (rc_a == PSACX_OK && levels2) { pair(Bb[i], cap); if (rc_a == PSACX_OK) cur[i] = (unsigned*)arr(((cap >> wb) + 2) * sizeof(unsigned)); }
            if (rc_a != PSACX_OK) return rc_a;
            if (!m) return PSACX_OK;
            MG_HIP(g, hipSetDevice(c->device));
            std::memset(c->pinned + 32768, 0, SLICE_MAX_CLASSES * 8);
            std::memcpy(c->pinned + 32768, cstart[i].data(), (size_t)C * 8);
            MG_HIP(g, hipMemcpyAsync(d_cnt[i].p, c->pinned + 32768, SLICE_MAX_CLASSES * 8, hipMemcpyHostToDevice, c->stream));
            if (pack || wpack)
                hipLaunchKernelGGL((slice_partition_packed_kernel<T, PB, 8>), dim3((unsigned)((m + PB * 8 - 1) / (PB * 8))), dim3(PB), 0, c->stream, (const T*)S[i].SA,
                                   ids_in_isa ? (const T*)S[i].ISA : (const T*)S[i].Bsa.p, m, map, d_cnt[i].p,
                                   (wpack && solo_) ? reinterpret_cast<uint64_t*>(pk_[i].v) : reinterpret_cast<uint64_t*>(pk_[i].k),
                                   ids_in_isa ? S[i].Bsa.p : (T*)nullptr, wpack ? 1 : 0, S[i].off + m - 1);
            else
            hipLaunchKernelGGL((slice_partition_kernel<T, V, PB, PI>), dim3((unsigned)((m + PB * PI - 1) / (PB * PI))), dim3(PB), 0, c->stream, (const T*)S[i].SA,
                               ids_in_isa ? (const T*)S[i].ISA : (const T*)S[i].Bsa.p, m, map, d_cnt[i].p, pk_[i].k, pk_[i].v, ids_in_isa ? S[i].Bsa.p : (T*)nullptr);
            MG_HIP(g, hipGetLastError());
            MG_HIP(g, hipStreamSynchronize(c->stream));       // (the pinned words are reused)
            return PSACX_OK;
        });
        PSACX_TRY(agree(rc_part));      // (a rank without its arrays must not leave its peers in the transfers below)
        mark("    ISA: classes");
        // 3. slices to their owners, G per step; the remaining levels + the window scatter on the owner
        std::vector<hipEvent_t> done[2];
        done[0].assign(L, nullptr); done[1].assign(L, nullptr);
        auto drop_events = [&]() { for (int q = 0; q < 2; ++q) for (int i = 0; i < L; ++i) if (done[q][i]) { (void)hipSetDevice(ctx(i)->device); (void)hipEventDestroy(done[q][i]); done[q][i] = nullptr; } };
        for (int i = 0; i < L; ++i) {
            MG_HIP(g, hipSetDevice(ctx(i)->device));
            for (int q = 0; q < 2; ++q) MG_HIP(g, hipEventCreateWithFlags(&done[q][i], hipEventDisableTiming));
        }
        auto issue = [&](uint64_t t) -> int {
            const unsigned s0 = (unsigned)(t * G), s1e = (unsigned)std::min<uint64_t>(spo, (t + 1) * G);
            std::vector<std::vector<Msg>> sends(L), recvs(L);
            std::vector<std::vector<const void*>> in(L);
            std::vector<std::vector<void*>> out(L);
            for (int i = 0; i < L; ++i) {
                const int me = rank(i);
                for (int d = 0; d < P; ++d)
                    for (unsigned sl = s0; sl < s1e; ++sl) { const unsigned cl = (unsigned)d * spo + sl; sends[i].push_back(Msg{d, cstart[i][cl], counts[i][cl]}); }
                for (unsigned sl = s0; sl < s1e; ++sl) {
                    uint64_t at = (uint64_t)(sl - s0) << sb;
                    for (int r = 0; r < P; ++r) { const uint64_t cn = table[(size_t)r * C + (unsigned)me * spo + sl]; recvs[i].push_back(Msg{r, at, cn}); at += cn; }
                }
                Ptrs& A = (t & 1) ? A1[i] : A0[i];
                if (pack) { in[i] = {pk_[i].k}; out[i] = {A.k}; }
                else if (wpack) { in[i] = {pk_[i].k}; out[i] = {A.v}; }           // (the packed entries land in the value array: 8 bytes per pair)
                else { in[i] = {pk_[i].k, pk_[i].v}; out[i] = {A.k, A.v}; }
            }
            if (pack || wpack) return transfer(in, out, {sizeof(uint64_t)}, sends, recvs, &done[t & 1]);
            return transfer(in, out, {sizeof(uint32_t), sizeof(V)}, sends, recvs, &done[t & 1]);
        };
        std::vector<std::vector<DBuf<uint64_t>>> keep_dec(L);         // segment tables of the steps (alive until the streams have drained)
        int rc = PSACX_OK;
        if (!solo_) rc = issue(0);
        for (uint64_t t = 0; t < nsteps && rc == PSACX_OK; ++t) {
            if (!solo_ && t + 1 < nsteps) rc = issue(t + 1);          // the next slices travel while these are worked on
            if (rc != PSACX_OK) break;
            rc = par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_HIP(g, hipSetDevice(c->device));
                const uint64_t lo = (t * G) << sb;
                const uint64_t len = S[i].m > lo ? std::min<uint64_t>(S[i].m - lo, step_cap) : 0;
                if (!solo_) MG_HIP(g, hipStreamWaitEvent(c->stream, done[t & 1][i], 0));
                if (!len) return PSACX_OK;
                Ptrs& A = (t & 1) ? A1[i] : A0[i];
                // (one rank without the wire: the classes of the first level ARE the slices, in place: the step's part of them)
                const uint64_t so = solo_ ? lo : 0;
                const uint32_t* ks = solo_ ? pk_[i].k + ((pack || !pk_[i].v) ? 2 * so : so) : A.k; const V* vs = solo_ ? (pk_[i].v ? pk_[i].v + so : nullptr) : A.v;
                uint32_t* ka = const_cast<uint32_t*>(ks); V* va = const_cast<V*>(vs);
                unsigned below = rbits;                        // bits still to partition on beneath the current level
                if (pack) {
                    const uint64_t* cur_in = reinterpret_cast<const uint64_t*>(ks);
                    uint64_t* mine = reinterpret_cast<uint64_t*>(ka);
                    for (unsigned j = 0; j < levels2; ++j) {
                        below -= cbs[j];
                        const unsigned shift = wb + below;
                        MG_HIP(g, hipMemsetAsync(cur[i], 0, ((len >> shift) + 2) * sizeof(unsigned), c->stream));
                        uint64_t* o = (j & 1) ? mine : reinterpret_cast<uint64_t*>(Bb[i].k);
                        hipLaunchKernelGGL((pairs_partition_packed_kernel<PB, PI>), dim3((unsigned)((len + PB * PI - 1) / (PB * PI))), dim3(PB), 0, c->stream, cur_in, o, len,
                                           shift, cbs[j], cur[i], j == 0 ? (uint32_t)lo : 0u);
                        MG_HIP(g, hipGetLastError());
                        cur_in = o;
                    }
                    hipLaunchKernelGGL((pairs_window_packed_kernel<T, 1024, WBMAX>), dim3((unsigned)((len + (1ull << wb) - 1) >> wb)), dim3(1024), 0, c->stream, cur_in, len, wb,
                                       levels2 ? 0u : (uint32_t)lo, S[i].ISA + lo);
                    MG_HIP(g, hipGetLastError());
                    return PSACX_OK;
                }
                SliceDecode dec; dec.seg = nullptr; dec.base = nullptr; dec.P = (unsigned)P; dec.sb = sb;
                if (wpack) {
                    // where every sender's segment lies inside the slices of this step, and the last position of every sender's block
                    const unsigned s0 = (unsigned)(t * G), s1e = (unsigned)std::min<uint64_t>(spo, (t + 1) * G);
                    std::vector<uint64_t> h((size_t)(s1e - s0) * (P + 1) + P, 0);
                    for (unsigned sl = s0; sl < s1e; ++sl) {
                        uint64_t at = 0;
                        for (int r = 0; r < P; ++r) { h[(size_t)(sl - s0) * (P + 1) + r] = at; at += solo_ ? counts[i][(size_t)rank(i) * spo + sl] : table[(size_t)r * C + (unsigned)rank(i) * spo + sl]; }
                        h[(size_t)(sl - s0) * (P + 1) + P] = at;
                    }
                    for (int r = 0; r < P; ++r) h[(size_t)(s1e - s0) * (P + 1) + r] = offs[r] + sizes[r] - 1;
                    DBuf<uint64_t> d; MG_OP(g, c, d.alloc(c, h.size()));
                    MG_HIP(g, hipMemcpy(d.p, h.data(), h.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
                    dec.seg = d.p; dec.base = d.p + (size_t)(s1e - s0) * (P + 1);
                    keep_dec[i].push_back(std::move(d));
                    ks = solo_ ? reinterpret_cast<const uint32_t*>(pk_[i].v + so) : reinterpret_cast<const uint32_t*>(A.v);     // the packed entries as they arrived
                }
                for (unsigned j = 0; j < levels2; ++j) {
                    below -= cbs[j];
                    const unsigned shift = wb + below;
                    MG_HIP(g, hipMemsetAsync(cur[i], 0, ((len >> shift) + 2) * sizeof(unsigned), c->stream));
                    uint32_t* ko = (j & 1) ? ka : Bb[i].k; V* vo = (j & 1) ? va : Bb[i].v;
                    if (wpack && j == 0)
                        hipLaunchKernelGGL((pairs_partition_kernel<V, PB, PI, true>), dim3((unsigned)((len + PB * PI - 1) / (PB * PI))), dim3(PB), 0, c->stream, ks, vs, ko, vo, len,
                                           shift, cbs[j], cur[i], (uint32_t)lo, dec);
                    else
                        hipLaunchKernelGGL((pairs_partition_kernel<V, PB, PI>), dim3((unsigned)((len + PB * PI - 1) / (PB * PI))), dim3(PB), 0, c->stream, ks, vs, ko, vo, len,
                                           shift, cbs[j], cur[i], j == 0 ? (uint32_t)lo : 0u);
                    MG_HIP(g, hipGetLastError());
                    ks = ko; vs = vo;
                }
                if (wpack && levels2 == 0)
                    hipLaunchKernelGGL((pairs_window_kernel<V, T, 1024, WBMAX, true>), dim3((unsigned)((len + (1ull << wb) - 1) >> wb)), dim3(1024), 0, c->stream, ks, vs, len, wb,
                                       (uint32_t)lo, S[i].ISA + lo, dec);
                else
                    hipLaunchKernelGGL((pairs_window_kernel<V, T, 1024, WBMAX>), dim3((unsigned)((len + (1ull << wb) - 1) >> wb)), dim3(1024), 0, c->stream, ks, vs, len, wb,
                                       levels2 ? 0u : (uint32_t)lo, S[i].ISA + lo);
                MG_HIP(g, hipGetLastError());
                return PSACX_OK;
            });
        }
        // nobody releases its classes before every receiver has pulled its pieces
        for (int i = 0; i < L; ++i) {
            (void)hipSetDevice(ctx(i)->device);
            for (int q = 0; q < 2 && !solo_; ++q) for (int s2 = 0; s2 < L; ++s2) (void)hipStreamWaitEvent(ctx(i)->stream, done[q][s2], 0);
        }
        for (int i = 0; i < L; ++i) { (void)hipSetDevice(ctx(i)->device); (void)hipStreamSynchronize(ctx(i)->stream); }
        drop_events();
        if (rc != PSACX_OK) return rc;
        g->last_slice_inversion = true;
        return PSACX_OK;
    }
    int isa_by_slices(bool ids_in_isa = false) {
        if (sizeof(T) == 4 || (n <= (1ull << 32) && !g->opt_slice_wide)) return isa_by_slices_t<uint32_t>(ids_in_isa);      // (PSACX_SLICE_WIDE: tests)
        return isa_by_slices_t<T>(ids_in_isa);
    }

    // out[i][j] = block_owner[gidx[i][j] - off_owner] in the order of gidx (bulk_rma.hpp:13-135); positions >= n are clamped
    int dist_take(const std::vector<const T*>& block, const std::vector<const T*>& gidx, const std::vector<uint64_t>& cnt,
                  std::vector<DBuf<T>>& out) {
        out.clear(); out.resize(L);
        if (solo_) {
            MG_OP(g, ctx(0), out[0].alloc(ctx(0), cnt[0]));
            MG_OP(g, ctx(0), op_take(ctx(0), block[0], gidx[0], cnt[0], S[0].off, n, out[0].p));
            return PSACX_OK;
        }
        std::vector<Rec<T>> routed(L);
        std::vector<std::vector<uint64_t>> bounds(L), rc, rc2;
        std::vector<std::vector<const T*>> in(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            DBuf<T> idx; MG_OP(g, c, idx.alloc(c, cnt[i]));
            MG_OP(g, c, psacx_op_iota(c, idx.p, cnt[i], 0));
            PSACX_TRY(route(i, gidx[i], idx.p, cnt[i], routed[i], bounds[i]));
            in[i] = {routed[i].k2.p};
            return PSACX_OK;
        }));
        std::vector<std::vector<DBuf<T>>> q, got;
        PSACX_TRY(exchange<T>(1, in, bounds, q, rc));
        std::vector<DBuf<T>> ans(L);
        std::vector<std::vector<uint64_t>> back_bounds(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, ans[i].alloc(c, q[i][0].n));
            MG_OP(g, c, op_take(c, block[i], q[i][0].p, q[i][0].n, S[i].off, n, ans[i].p));
            back_bounds[i] = prefix_of(rc[i]);
            in[i] = {ans[i].p};
            return PSACX_OK;
        }));
        PSACX_TRY(exchange<T>(1, in, back_bounds, got, rc2));
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, out[i].alloc(c, cnt[i]));
            MG_OP(g, c, op_put(c, out[i].p, routed[i].v.p, cnt[i], 0, got[i][0].p, 0));      // undo the routing permutation
            return PSACX_OK;
        }));
        return PSACX_OK;
    }
    static int psacx_op_iota(psacx_ctx* c, T* out, uint64_t m, uint64_t start) {
        OP_PROLOGUE(c); SIMPLE_LAUNCH(c, (iota_from_kernel<T>), m, out, m, start); return PSACX_OK;
    }


    // 64-ary min-pyramid over this rank's LCP block (levels >= 1 in a buffer of the run; level 0 is the block), and the block minimum.
    // Built once, when the first range minimum of the run is asked for (the first round has written every LCP entry by then), and kept
    // up to date by whoever lowers an entry afterwards (pyramid_set: rebucket_refine_kernel, lcp_apply_pyr_kernel) -- a refinement
    // round in slabs asks for range minima once per slab, and a pyramid per question read the whole block every time (a 2^32 block:
    // 34 GB, 1700 times per construction of a tandem repeat: 8.7 of its 33 s).
    std::vector<Pyramid<T>> lcp_pyr_;
    std::vector<DBuf<T>> lcp_pyr_mem_;
    // whoever lends the LCP block out as scratch or writes it without pyramid_set calls this: the upper levels are rebuilt when next asked for
    void invalidate_lcp_pyramid(int i) { if (i < (int)lcp_pyr_.size()) lcp_pyr_[i] = Pyramid<T>(); }
    int block_pyramid(int i, Pyramid<T>& Pm, uint64_t* block_min) {
        psacx_ctx* c = ctx(i);
        const uint64_t m = S[i].m;
        if ((int)lcp_pyr_.size() != L) { lcp_pyr_.assign(L, Pyramid<T>()); lcp_pyr_mem_.clear(); lcp_pyr_mem_.resize(L); }
        *block_min = (uint64_t)(T)~(T)0;
        Pm = Pyramid<T>();
        if (m == 0) return PSACX_OK;
        OP_PROLOGUE(c);
        if (lcp_pyr_[i].nlev == 0 || lcp_pyr_[i].lvl[0] != S[i].LCP) {
            Pyramid<T>& Q = lcp_pyr_[i];
            DBuf<T>& mem = lcp_pyr_mem_[i];
            Q = Pyramid<T>();
            uint64_t total = 0, len = m;
            int nlev = 1;
            while (len > 128 && nlev < PYR_MAX) { len = (len + 63) / 64; total += (len + 63) & ~63ull; ++nlev; }
            mem.release();
            MG_OP(g, c, mem.alloc(c, total + 64));
            Q.lvl[0] = S[i].LCP; Q.len[0] = m; Q.nlev = 1;
            len = m;
            uint64_t at = 0;
            while (len > 128 && Q.nlev < PYR_MAX) {
                len = (len + 63) / 64;
                Q.lvl[Q.nlev] = mem.p + at; Q.len[Q.nlev] = len; at += (len + 63) & ~63ull;
                hipLaunchKernelGGL((pyramid_level_kernel<T>), dim3(grid_for(c, len * 64, 256, 8)), dim3(256), 0, c->stream, Q.lvl[Q.nlev - 1],
                                   Q.len[Q.nlev - 1], Q.lvl[Q.nlev], len);
                MG_HIP(g, hipGetLastError());
                Q.nlev++;
            }
        }
        Pm = lcp_pyr_[i];
        unsigned long long* d = reinterpret_cast<unsigned long long*>(lcp_pyr_mem_[i].p + (lcp_pyr_mem_[i].n - 64));      // 64 spare entries at the end
        hipLaunchKernelGGL((top_min_kernel<T>), dim3(1), dim3(256), 0, c->stream, Pm.lvl[Pm.nlev - 1], Pm.len[Pm.nlev - 1], d);
        MG_HIP(g, hipGetLastError());
        MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, d, 8, hipMemcpyDeviceToHost, c->stream));
        MG_HIP(g, hipStreamSynchronize(c->stream));
        *block_min = *reinterpret_cast<uint64_t*>(c->pinned + 32768);
        return PSACX_OK;
    }
    // the pyramid of rank i's LCP block if the run keeps one (else one of level 0 only): for the kernels that lower LCP entries
    Pyramid<T> lcp_pyramid_or_block(int i) const {
        if (i < (int)lcp_pyr_.size() && lcp_pyr_[i].nlev > 0 && lcp_pyr_[i].lvl[0] == S[i].LCP) return lcp_pyr_[i];
        Pyramid<T> Q = Pyramid<T>();
        Q.lvl[0] = S[i].LCP; Q.len[0] = S[i].m; Q.nlev = S[i].LCP ? 1 : 0;
        return Q;
    }

    // One refinement pass (suffix_array.hpp:1092-1157, :1181-1285) over the list entries plist[i][0 .. cnt[i]) of every local
    // rank -- global SA positions inside its block, whole buckets: B2 = rank of the suffix h further, sort by (bucket, B2), new bucket ids / SA /
    // ISA / LCP written in place.  kept[i]: the entries that still share a bucket; counts = cnt of every rank.
    int refine_step(uint64_t h, const std::vector<const T*>& plist, const std::vector<uint64_t>& cnt, const std::vector<uint64_t>& counts,
                    unsigned id_bits, std::vector<DBuf<T>>& kept, uint64_t* unf_b, uint64_t* unf_e) {
        std::vector<Rec<T>> rec(L);
        std::vector<DBuf<T>> q(L);
        std::vector<psacx_boundary> bd;
        std::vector<uint64_t> lh(L), heads, nact(L), nunf(L);
        // B2 = rank of the suffix h further (sparse_get_b2, suffix_array.hpp:972-996)
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            rec[i].cnt = cnt[i];
            MG_OP(g, c, rec[i].k1.alloc(c, cnt[i])); MG_OP(g, c, rec[i].v.alloc(c, cnt[i])); MG_OP(g, c, q[i].alloc(c, cnt[i]));
            MG_OP(g, c, op_take(c, S[i].SA, plist[i], cnt[i], S[i].off, n, rec[i].v.p));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (add_scalar_kernel<T>), cnt[i], rec[i].v.p, cnt[i], h, n, q[i].p);      // saturates at n
            MG_OP(g, c, op_take(c, S[i].Bsa.p, plist[i], cnt[i], S[i].off, n, rec[i].k1.p));
            return PSACX_OK;
        }));
        {
            std::vector<const T*> blk(L), gi(L);
            std::vector<DBuf<T>> masked(L);
            const bool gsa = gsa_off_ != nullptr;
            if (gsa)
                PSACX_TRY(par([&](int i) -> int {
                    psacx_ctx* c = ctx(i);
                    MG_OP(g, c, masked[i].alloc(c, S[i].m));
                    OP_PROLOGUE(c);
                    SIMPLE_LAUNCH(c, (mask_by_string_kernel<T>), S[i].m, (const T*)S[i].ISA, (const T*)soff_[i].p, S[i].m, h, masked[i].p);
                    return PSACX_OK;
                }));
            for (int i = 0; i < L; ++i) { blk[i] = gsa ? masked[i].p : S[i].ISA; gi[i] = q[i].p; }
            std::vector<DBuf<T>> ans;
            PSACX_TRY(dist_take(blk, gi, cnt, ans));
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, rec[i].k2.alloc(c, cnt[i]));
                OP_PROLOGUE(c);
                if (gsa) SIMPLE_LAUNCH(c, (finish_b2_masked_kernel<T>), cnt[i], (const T*)ans[i].p, (const T*)q[i].p, cnt[i], n, rec[i].k2.p);
                else SIMPLE_LAUNCH(c, (finish_b2_kernel<T>), cnt[i], ans[i].p, q[i].p, cnt[i], n, rec[i].k2.p);
                return PSACX_OK;
            }));
        }
        q.clear();
        mark("  B2 fetch");
        PSACX_TRY(refine_sort(rec, plist, counts, id_bits, id_bits));
        mark("  sort");
        {
            std::vector<const T*> a1(L), a2(L), a3(L);
            for (int i = 0; i < L; ++i) { a1[i] = rec[i].k1.p; a2[i] = rec[i].k2.p; a3[i] = rec[i].v.p; }
            PSACX_TRY(neighbours(a1, a2, a3, cnt, 3, bd));
        }
        PSACX_TRY(par([&](int i) -> int {
            bd[i].off = 0; bd[i].base = 0;
            psacx_boundary b0 = bd[i]; b0.has_next = 0;
            MG_OP(g, ctx(i), op_last_head<T>(ctx(i), 1, rec[i].k1.p, rec[i].k2.p, plist[i], cnt[i], 0, 1, 1, 0, &b0, &lh[i]));
            return PSACX_OK;
        }));
        PSACX_TRY(gather1(lh, heads));
        std::vector<DBuf<T>> ids(L), qa(L), ql(L), qh(L);
        std::vector<uint64_t> nq(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            uint64_t base = 0;
            for (int s = 0; s < rank(i); ++s) base = std::max(base, heads[s]);
            bd[i].off = S[i].off; bd[i].base = base;
            MG_OP(g, c, ids[i].alloc(c, cnt[i])); MG_OP(g, c, qa[i].alloc(c, cnt[i])); MG_OP(g, c, ql[i].alloc(c, cnt[i])); MG_OP(g, c, qh[i].alloc(c, cnt[i]));
            const Pyramid<T> lp = lcp_pyramid_or_block(i);
            MG_OP(g, c, op_rebucket_refine<T>(c, rec[i].k1.p, rec[i].k2.p, rec[i].v.p, plist[i], cnt[i], n, h, &bd[i], S[i].SA, S[i].Bsa.p,
                                              S[i].LCP, ids[i].p, qa[i].p, ql[i].p, qh[i].p, &nq[i], &nact[i], &nunf[i], &lp));
            return PSACX_OK;
        }));
        {
            std::vector<T*> blk(L); std::vector<const T*> gi(L), va(L);
            for (int i = 0; i < L; ++i) { blk[i] = S[i].ISA; gi[i] = rec[i].v.p; va[i] = ids[i].p; }
            PSACX_TRY(dist_put(blk, gi, va, cnt, -1, false));
        }
        mark("  refine + ISA");
        rec.clear(); rec.resize(L);                       // (the sorted records are not read again: their room serves the range minima)
        if (want_lcp) {
            std::vector<const T*> lo(L), hi(L);
            for (int i = 0; i < L; ++i) { lo[i] = ql[i].p; hi[i] = qh[i].p; }
            std::vector<DBuf<T>> mins;
            PSACX_TRY(dist_range_min(lo, hi, nq, mins));
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (lcp_apply_pyr_kernel<T>), nq[i], lcp_pyramid_or_block(i), qa[i].p, nq[i], S[i].off, mins[i].p, h);
                return PSACX_OK;
            }));
        }
        mark("  range minima");
        PSACX_TRY(next_active(&ids, &plist, nact, nunf, kept, unf_b, unf_e));
        return PSACX_OK;
    }

    // Piece boundaries e[0 .. steps] of the list of unresolved positions of local rank i: about equal pieces, every cut moved
    // back to the head of the bucket it falls into; a bucket that enters from the previous rank stays whole in piece 0, one
    // that leaves to the next rank in the last piece.
    int slab_bounds(int i, uint64_t steps, std::vector<uint64_t>& e, uint64_t* whole) {
        const uint64_t a = S[i].pos.n, m = S[i].m;
        e.assign(steps + 1, 0);
        e[steps] = a;
        if (a == 0) return PSACX_OK;
        psacx_ctx* c = ctx(i);
        std::vector<uint64_t> t(steps + 1), p, id;
        for (uint64_t j = 0; j < steps; ++j) t[j] = (uint64_t)(((unsigned __int128)a * j) / steps);
        t[steps] = a - 1;
        PSACX_TRY(fetch(i, S[i].pos.p, t, p));                  // the list holds global SA positions
        for (auto& x : p) x -= S[i].off;
        PSACX_TRY(fetch(i, S[i].Bsa.p, p, id));
        uint64_t lead = 0;
        if (id[0] - 1 < S[i].off) {                       // the first bucket began on an earlier rank: its members are positions 0 .. lead - 1
            DBuf<uint64_t> d; MG_OP(g, c, d.alloc(c, 1));
            MG_HIP(g, hipSetDevice(c->device));
            hipLaunchKernelGGL((upper_bound_kernel<T>), dim3(1), dim3(1), 0, c->stream, S[i].Bsa.p, m, id[0], d.p);
            MG_HIP(g, hipGetLastError());
            MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, d.p, 8, hipMemcpyDeviceToHost, c->stream));
            MG_HIP(g, hipStreamSynchronize(c->stream));
            lead = std::min<uint64_t>(*reinterpret_cast<uint64_t*>(c->pinned + 32768), a);
        }
        // a bucket that covers this whole block would need three ranks' pieces in one step: the round then runs unsliced
        if (lead == a && p[steps] == m - 1 && P > 1) *whole = 1;
        for (uint64_t j = 1; j < steps; ++j) {
            uint64_t cut;
            if (t[j] < lead) cut = lead;
            else { const uint64_t head = id[j] - 1 - S[i].off; cut = t[j] - (p[j] - head); }     // every member of the bucket is a list entry
            e[j] = std::max(cut, e[j - 1]);
        }
        return PSACX_OK;
    }

    // ---------------------------------------------------------------- the construction (suffix_array.hpp:365-466, :1032-1285)
    // str_off / nstr: a string set (construct_ss, suffix_array.hpp:267-363) -- the nstr + 1 ascending global offsets of the
    // strings, which lie back to back in the block-distributed text (host array, the same on every rank)
    int construct(const std::vector<const uint8_t*>& text, const std::vector<uint64_t>& m_local, uint32_t k_req, uint32_t flags,
                  const std::vector<T*>& d_sa, const std::vector<T*>& d_isa, const std::vector<T*>& d_lcp,
                  const uint64_t* str_off = nullptr, uint64_t nstr = 0) {
        gsa_off_ = str_off; gsa_nstr_ = nstr; soff_.clear();
        const bool gsa = str_off != nullptr;
        want_lcp = (flags & PSACX_LCP) != 0;
        psacx_stats& st = g->stats;
        std::memset(&st, 0, sizeof(st));
        g->bytes_sent = 0; g->n_exchanges = 0; g->n_gathers = 0;
        g->wire_sends = g->wire_recvs = g->wire_gathers = 0;
        g->phases.clear();
        t_phase_ = std::chrono::steady_clock::now();
        for (int i = 0; i < L; ++i) g->R[i].ex_used = 0;
        S.resize(L);
        PSACX_TRY(par([&](int i) -> int {
            S[i].c = ctx(i); S[i].r = rank(i); S[i].m = m_local[i]; S[i].text = text[i];
            S[i].SA = d_sa[i]; S[i].ISA = d_isa[i]; S[i].LCP = want_lcp ? d_lcp[i] : nullptr;
            S[i].out_cap = m_local[i] + g->out_slack; S[i].out_busy = false;
            S[i].c->pool_peak = S[i].c->pool_live;
            MG_OP(g, S[i].c, ensure_pinned(S[i].c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
            return PSACX_OK;
        }));
        diet = false; slab_cap = 0; first_round_ = true;
        g->last_reduced = false; g->last_slab_rounds = 0; g->last_tie_slabs = 0;
        // sizes + alphabet (alphabet.hpp:98: allreduce of the character histograms)
        {
            std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(258, 0));
            // the normal layout holds up to ~14 words per character beside the outputs; a rank whose share of the free
            // device memory is smaller asks for the reduced-memory layout, and then every rank uses it
            for (int i = 0; i < L; ++i) {
                int same = 0;
                for (int j = 0; j < L; ++j) same += ctx(j)->device == ctx(i)->device;
                size_t fr = 0, tot = 0;
                MG_HIP(g, hipSetDevice(ctx(i)->device));
                MG_HIP(g, hipMemGetInfo(&fr, &tot));
                const double avail = ((double)fr + (double)ctx(i)->pool_bytes * same) / same;
                const bool tight = 14.0 * (double)S[i].m * sizeof(T) > 0.9 * avail;
                mine[i][257] = g->opt_layout == 2 || (g->opt_layout == 0 && tight) ? 1 : 0;
            }
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                DBuf<uint64_t> h; MG_OP(g, c, h.alloc(c, 256));
                MG_OP(g, c, psacx_op_char_hist(c, text[i], S[i].m, h.p));
                MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, h.p, 256 * 8, hipMemcpyDeviceToHost, c->stream));
                MG_HIP(g, hipStreamSynchronize(c->stream));
                mine[i][0] = S[i].m;
                std::memcpy(&mine[i][1], c->pinned + 32768, 256 * 8);
                return PSACX_OK;
            }));
            std::vector<uint64_t> all;
            PSACX_TRY(gather(258, mine, all));
            sizes.assign(P, 0);
            uint64_t hist[256] = {0};
            for (int r = 0; r < P; ++r) {
                sizes[r] = all[(size_t)r * 258];
                for (int ch = 0; ch < 256; ++ch) hist[ch] += all[(size_t)r * 258 + 1 + ch];
                if (all[(size_t)r * 258 + 257]) diet = true;
            }
            offs = prefix_of(sizes);
            n = offs[P];
            if (!plan::follows_blk_dist(sizes)) { g->err = "The input string must be equally block decomposed accross all MPI processes."; return PSACX_EINVAL; }      // suffix_array.hpp:226-227
            for (int i = 0; i < L; ++i) S[i].off = offs[rank(i)];
            if (n == 0) return PSACX_EINVAL;
            if (diet) {
                // (a refinement step holds up to seventeen arrays of a slab's length at once -- its records, their new ids and the queries and
                //  answers of the range minima on both sides of an exchange -- beside the bucket ids and the list of unresolved positions:
                //  with 1/32 of a block per step that stays below three words per character, BASELINE.json configs[4])
                slab_cap = g->opt_slab ? g->opt_slab : slab_env_ ? slab_env_ : std::max<uint64_t>(sizes[0] / 32, 1u << 16);
                if (slab_cap < 64) slab_cap = 64;
                // (free blocks stay cached -- hipFree / hipMalloc of a 36 GB block cost about a second each -- and go back to the
                //  device only when an allocation does not fit: pool_alloc)
                for (int i = 0; i < L; ++i) ctx(i)->pool_cache_limit = 0;
                g->last_reduced = true;
            } else for (int i = 0; i < L; ++i) ctx(i)->pool_cache_limit = std::max<size_t>((size_t)S[i].m * sizeof(T) * 16, (size_t)64 << 20);   // free blocks kept for reuse: at most sixteen block-sized arrays (a flush is hipFree + hipMalloc of everything: seconds with eight ranks)
            if (sizeof(T) == 4 && n > 0xFFFFFFFEull) return PSACX_ERANGE;
            uint32_t sigma = 0;
            for (int ch = 0; ch < 256; ++ch) sigma += hist[ch] != 0;
            uint32_t l = 0; while ((1u << l) < sigma + 1u) ++l;
            st.sigma = sigma; st.bits_per_char = l;
            for (int ch = 0, nx = 0; ch < 256; ++ch) codes_[ch] = hist[ch] ? (uint16_t)(nx++) : (uint16_t)0;   // packed codes 0..sigma-1
            if (gsa) {
                // string ends need their own code in the key: psac's codes 1 .. sigma with l bits, 0 = end (kmer.hpp:269-355)
                for (int ch = 0; ch < 256; ++ch) if (hist[ch]) codes_[ch] = (uint16_t)(codes_[ch] + 1);
                if (nstr == 0 || nstr > n || str_off[0] != 0 || str_off[nstr] != n) { g->err = "string set: the offsets do not cover the text"; return PSACX_EINVAL; }
                for (uint64_t t = 0; t < nstr; ++t) if (str_off[t + 1] <= str_off[t]) { g->err = "string set: empty string or offsets not ascending"; return PSACX_EINVAL; }
            }
        }
        mark("alphabet");
        const uint32_t l = st.bits_per_char;
        const uint32_t word_bits = (uint32_t)sizeof(T) * 8;
        uint64_t min_local = sizes[0];
        for (int r = 1; r < P; ++r) min_local = std::min(min_local, sizes[r]);
        uint32_t k;                                    // kmer.hpp:26-40
        {
            const uint32_t max_k = word_bits / l;
            k = (k_req == 0 || k_req > max_k) ? max_k : k_req;
            if ((uint64_t)k >= min_local) { k = (uint32_t)min_local; if (P == 1 && k > 1) --k; }
        }
        st.k = k;
        const uint32_t two_k = 2 * k;
        // (blocks shorter than 2k characters: k was shrunk to the smallest block as kmer.hpp:33-39 does; the 2k-character halo
        //  then spans several right neighbours, see below)
        const bool tiny_blocks = P > 1 && min_local < two_k;
        // the 2k-character window packed without an end-marker code (key_pairs_kernel): lc bits per character
        uint32_t lc = 0; while ((1u << lc) < st.sigma) ++lc; if (!lc) lc = 1;
        if (gsa) lc = l;
        const uint32_t c1 = std::min<uint32_t>(two_k, word_bits / lc), c2 = two_k - c1;

        // ---- halo: the first 2k characters of the right neighbour (kmer.hpp:142)
        std::vector<DBuf<uint8_t>> tbuf(L);
        {
            std::vector<std::vector<DBuf<uint8_t>>> got;
            std::vector<std::vector<uint8_t>> halo_host(L);
            if (tiny_blocks) {
                // the halo [end of block, end of block + 2k) reaches beyond the right neighbour: the whole text is at most
                // 2k P characters, so every rank gets all of it through the scalar all-gather and cuts its halo on the host
                const uint64_t words = (sizes[0] + 7) / 8 + 1;
                std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(words, 0));
                for (int i = 0; i < L; ++i) {
                    MG_HIP(g, hipSetDevice(ctx(i)->device));
                    if (S[i].m) MG_HIP(g, hipMemcpy(mine[i].data(), text[i], S[i].m, hipMemcpyDeviceToHost));
                }
                std::vector<uint64_t> all;
                PSACX_TRY(gather((int)words, mine, all));
                std::vector<uint8_t> whole(n);
                for (int r = 0; r < P; ++r) std::memcpy(whole.data() + offs[r], reinterpret_cast<const uint8_t*>(&all[(size_t)r * words]), sizes[r]);
                for (int i = 0; i < L; ++i) {
                    const uint64_t e = S[i].off + S[i].m;
                    halo_host[i].assign(two_k, 0);
                    for (uint64_t t = 0; t < two_k && e + t < n; ++t) halo_host[i][t] = whole[e + t];
                }
            } else if (!solo_) {
                std::vector<std::vector<uint64_t>> bounds(L), rc;
                std::vector<std::vector<const uint8_t*>> in(L);
                for (int i = 0; i < L; ++i) {
                    const int r = rank(i);
                    bounds[i].assign(P + 1, 0);
                    // the piece [0, 2k) goes to rank r - 1: destinations < r - 1 get nothing, r - 1 gets 2k
                    for (int d = 0; d <= P; ++d) bounds[i][d] = (r > 0 && d >= r) ? two_k : 0;
                    in[i] = {text[i]};
                }
                PSACX_TRY(exchange<uint8_t>(1, in, bounds, got, rc));
            }
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, tbuf[i].alloc(c, S[i].m + two_k));
                MG_HIP(g, hipSetDevice(c->device));
                MG_HIP(g, hipMemsetAsync(tbuf[i].p + S[i].m, 0, two_k, c->stream));
                MG_HIP(g, hipMemcpyAsync(tbuf[i].p, text[i], S[i].m, hipMemcpyDeviceToDevice, c->stream));
                if (tiny_blocks) { MG_HIP(g, hipMemcpyAsync(tbuf[i].p + S[i].m, halo_host[i].data(), two_k, hipMemcpyHostToDevice, c->stream)); MG_HIP(g, hipStreamSynchronize(c->stream)); }
                else if (!solo_ && got[i][0].n) MG_HIP(g, hipMemcpyAsync(tbuf[i].p + S[i].m, got[i][0].p, std::min<uint64_t>(got[i][0].n, two_k), hipMemcpyDeviceToDevice, c->stream));
                return PSACX_OK;
            }));
        }
        // ---- first-round keys; the suffixes shorter than 2k (the last 2k - 1 positions) are moved to the very front
        //      of the record order (rank 0, shortest first): see key_pairs_kernel for why that replaces the end marker
        const uint64_t spec = gsa ? 0 : std::min<uint64_t>(two_k - 1, n);       // (string sets: the end markers are in the keys)
        std::vector<Rec<T>> rec(L);
        std::vector<DBuf<T>> slen(L);
        if (gsa) {
            soff_.resize(L);
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                DBuf<uint64_t> d_off; MG_OP(g, c, d_off.alloc(c, nstr + 1));
                MG_OP(g, c, slen[i].alloc(c, S[i].m)); MG_OP(g, c, soff_[i].alloc(c, S[i].m));
                MG_HIP(g, hipSetDevice(c->device));
                MG_HIP(g, hipMemcpyAsync(d_off.p, str_off, (nstr + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (string_pos_kernel<T>), S[i].m, (const uint64_t*)d_off.p, nstr, n, S[i].off, S[i].m, slen[i].p, soff_[i].p);
                MG_HIP(g, hipStreamSynchronize(c->stream));        // (the offsets leave with this scope)
                return PSACX_OK;
            }));
        }
        // both: word 2 of every record is generated and carried (three-word records); otherwise the records are (word 1, suffix)
        auto make_records = [&](bool both) -> int {
            std::vector<Rec<T>> tails(L);
            std::vector<uint64_t> mine_cnt(L);
            const int na = both ? 3 : 1;
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                const uint64_t m = S[i].m, front = rank(i) == 0 ? spec : 0;
                drop3(i, rec[i]);
                PSACX_TRY(take3(i, rec[i], front + m, both));
                MG_OP(g, c, op_make_keys<T>(c, tbuf[i].p, m, m + two_k, codes_, lc, c1, c2, rec[i].k1.p + front, both ? rec[i].k2.p + front : (T*)nullptr,
                                            gsa ? (const T*)slen[i].p : (const T*)nullptr));
                if (both) MG_OP(g, c, psacx_op_iota(c, rec[i].v.p + front, m, S[i].off));       // (two-word form: the shuffle or the sort makes the suffixes up)
                const uint64_t end = S[i].off + m, first_short = n - spec;
                const uint64_t mine = std::min<uint64_t>(m, end > first_short ? end - first_short : 0);     // short suffixes in this block (its tail)
                mine_cnt[i] = mine;
                tails[i].cnt = mine;
                MG_OP(g, c, tails[i].k1.alloc(c, mine)); MG_OP(g, c, tails[i].k2.alloc(c, mine)); MG_OP(g, c, tails[i].v.alloc(c, mine));
                if (mine) {
                    const T* src[3] = {rec[i].k1.p, rec[i].v.p, both ? rec[i].k2.p : (T*)nullptr}; T* dst[3] = {tails[i].k1.p, tails[i].v.p, tails[i].k2.p};
                    for (int q = 0; q < na; ++q) {
                        hipLaunchKernelGGL((reverse_copy_kernel<T>), dim3((unsigned)((mine + 255) / 256)), dim3(256), 0, c->stream, src[q] + front + m - mine, mine, dst[q]);
                        MG_HIP(g, hipGetLastError());
                    }
                }
                rec[i].cnt = front + m - mine;
                return PSACX_OK;
            }));
            // everything to rank 0, which places the pieces of higher ranks first
            std::vector<std::vector<DBuf<T>>> got;
            std::vector<std::vector<uint64_t>> rc;
            if (!solo_) {
                std::vector<std::vector<uint64_t>> bounds(L);
                std::vector<std::vector<const T*>> in(L);
                for (int i = 0; i < L; ++i) {
                    bounds[i].assign(P + 1, mine_cnt[i]); bounds[i][0] = 0;
                    in[i] = {tails[i].k1.p};
                    if (both) { in[i].push_back(tails[i].v.p); in[i].push_back(tails[i].k2.p); }
                }
                PSACX_TRY(exchange<T>(na, in, bounds, got, rc));
            }
            PSACX_TRY(par([&](int i) -> int {
                if (rank(i) != 0) return PSACX_OK;
                psacx_ctx* c = ctx(i);
                MG_HIP(g, hipSetDevice(c->device));
                T* dst[3] = {rec[i].k1.p, rec[i].v.p, rec[i].k2.p};
                if (solo_) {
                    const T* src[3] = {tails[i].k1.p, tails[i].v.p, tails[i].k2.p};
                    for (int q = 0; q < na && spec; ++q) MG_HIP(g, hipMemcpyAsync(dst[q], src[q], spec * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                } else {
                    const std::vector<uint64_t> cuts = prefix_of(rc[i]);
                    uint64_t at = 0;
                    for (int s = P - 1; s >= 0; --s) {
                        const uint64_t len = rc[i][s];
                        for (int q = 0; q < na && len; ++q) MG_HIP(g, hipMemcpyAsync(dst[q] + at, got[i][q].p + cuts[s], len * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                        at += len;
                    }
                }
                return PSACX_OK;
            }));
            // the pieces are consumed before `got` and `tails` go back to the cache: a block handed out again is only touched
            // in stream order (engine.hpp: pool)
            return PSACX_OK;
        };
        // Two-word form (sort_first_two_word) when the leading bits of word 1 separate almost every suffix and sorting on them
        // saves a pass -- the rule of the one-GPU engine (construct.hpp) with n the length of the WHOLE text.
        // PSACX_MULTI_TWO_WORD: 0 = never, 1 = also below 2^21 records per rank, 2 = additionally ignore what the samples say
        // (tests: repetitive texts through the tie machinery).
        const unsigned bits_w1 = c1 * lc, bits_w2 = c2 * lc;
        const unsigned lead = (bits_for(n - 1) + 3 + RADIX_BITS - 1) / RADIX_BITS * RADIX_BITS;
        const int tw_mode = g->opt_two_word - 1;          // (-1: the engine decides)
        bool two_word = !gsa && tw_mode != 0 && lead <= bits_w1 && lead + RADIX_BITS <= bits_w1 + bits_w2 && (tw_mode >= 1 || min_local >= (1ull << 21)) &&
                        !one_stage_env_;
        // One-word records dealt by the top digit of the prefix (sort_first_one_word): 64-bit words, blocks of at least 2^21 characters
        // (PSACX_MULTI_ONE_WORD: 0 = never, 1 = also for small blocks: tests)
        CodeTable tab; for (int ch = 0; ch < 256; ++ch) tab.c[ch] = codes_[ch];
        KeyShape ks; ks.lc = lc; ks.c1 = c1; ks.c2 = c2; ks.spec = 0;
        const int ow_mode = g->opt_one_word - 1;
        bool one_word = two_word && sizeof(T) == 8 && ow_mode != 0 && !tiny_blocks && (ow_mode >= 1 || min_local >= (1ull << 21));
        unsigned lo1_first = bits_w1 - lead;
        g->last_one_word = false;
        if (one_word) {
            const int rc1 = sort_first_one_word(rec, sizes, bits_w1, bits_w2, tbuf, two_k, tab, ks, tw_mode == 2, spec, &lo1_first);
            if (rc1 == PSACX_RETRY_) one_word = false; else PSACX_TRY(rc1);
        }
        if (!one_word) {
        PSACX_TRY(make_records(!two_word));
        mark("keys");
        }
        if (two_word && !one_word) {
            const int rc2 = sort_first_two_word(rec, sizes, bits_w1, bits_w2, bits_w1 - lead, tbuf, two_k, tab, ks, tw_mode == 2, spec);
            if (rc2 == PSACX_RETRY_) {
                two_word = false;
                PSACX_TRY(make_records(true));
                mark("keys");
            } else PSACX_TRY(rc2);
        }
        if (!two_word) PSACX_TRY(dist_sort(rec, sizes, bits_w1, bits_w2, true));
        tbuf.clear();
        g->last_two_word = two_word;
        PSACX_TRY(par([&](int i) -> int { return own3(i, rec[i]); }));
        mark("first sort");

        // ---- LCP of the 2k-mers, bucket ids (suffix_array.hpp:1353-1396, bucketing.hpp:57-123)
        std::vector<psacx_boundary> bd;
        {
            std::vector<const T*> a1(L), a2(L), a3(L); std::vector<uint64_t> cn(L);
            for (int i = 0; i < L; ++i) { a1[i] = rec[i].k1.p; a2[i] = rec[i].k2.p; a3[i] = rec[i].v.p; cn[i] = rec[i].cnt; }
            PSACX_TRY(neighbours(a1, a2, a3, cn, 3, bd));
        }
        std::vector<uint64_t> lh(L), heads;
        PSACX_TRY(par([&](int i) -> int {
            bd[i].off = S[i].off; bd[i].base = 0;
            psacx_boundary b0 = bd[i]; b0.has_next = 0;
            MG_OP(g, ctx(i), op_last_head<T>(ctx(i), 0, rec[i].k1.p, rec[i].k2.p, rec[i].v.p, rec[i].cnt, n, lc, c1, c2, &b0, &lh[i], gsa));
            return PSACX_OK;
        }));
        PSACX_TRY(gather1(lh, heads));
        std::vector<uint64_t> nact(L), nunf(L);
        std::vector<DBuf<uint64_t>> tile_act(L);      // per-tile counts of unresolved positions out of the rebucket kernel
        const bool slices = !g->opt_no_slices && sizes[0] <= (1ull << 32);      // SA -> ISA by destination slices (below)
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            uint64_t base = 0;
            for (int s = 0; s < rank(i); ++s) base = std::max(base, heads[s]);
            bd[i].base = base;
            // reduced-memory layout: the bucket ids are written into the (still unused) ISA array and move to their own
            // array once the records are gone
            T* bsa_out = S[i].ISA;
            if (!diet) { MG_OP(g, c, S[i].Bsa.alloc(c, S[i].m)); bsa_out = S[i].Bsa.p; }
            // (the suffixes go to the rank's SA block on the way, unless they sit there already)
            MG_OP(g, c, tile_act[i].alloc(c, (rec[i].cnt + ScanCfg<T>::TILE - 1) / ScanCfg<T>::TILE + 1));
            MG_OP(g, c, op_rebucket_first<T>(c, rec[i].k1.p, rec[i].k2.p, rec[i].v.p, rec[i].cnt, n, lc, c1, c2, &bd[i], bsa_out, S[i].LCP, &nact[i], &nunf[i], gsa,
                                             rec[i].v.p != S[i].SA ? S[i].SA : (T*)nullptr, tile_act[i].p));
            drop3(i, rec[i]);
            S[i].out_busy = true;
            if (diet) {
                // the bucket ids move from the ISA array to their own: inside the first level of the slice inversion, which
                // reads them anyway (isa_by_slices), else by a copy
                MG_OP(g, c, S[i].Bsa.alloc(c, S[i].m));
                if (!slices) MG_HIP(g, hipMemcpyAsync(S[i].Bsa.p, S[i].ISA, S[i].m * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
            }
            return PSACX_OK;
        }));
        first_round_ = false;
        mark("rebucket");
        // ---- SA -> ISA (bulk_permute.hpp:14-73): by destination slices (slice_inv.hpp); the earlier form (pairs routed by owner,
        //      plain scatter in chunks of the block in the reduced-memory layout) with PSACX_MULTI_NO_SLICES=1 or blocks beyond 2^32
        g->last_slice_inversion = false;
        if (slices) PSACX_TRY(isa_by_slices(diet));
        else {
            uint64_t chunks = 1;
            if (diet) for (int r = 0; r < P; ++r) chunks = std::max<uint64_t>(chunks, (sizes[r] + slab_cap - 1) / slab_cap);
            chunks = std::min<uint64_t>(chunks, 64);
            for (uint64_t q = 0; q < chunks; ++q) {
                std::vector<T*> blk(L); std::vector<const T*> gi(L), va(L); std::vector<uint64_t> cn(L);
                for (int i = 0; i < L; ++i) {
                    const uint64_t a = (uint64_t)(((unsigned __int128)S[i].m * q) / chunks), b = (uint64_t)(((unsigned __int128)S[i].m * (q + 1)) / chunks);
                    blk[i] = S[i].ISA; gi[i] = S[i].SA + a; va[i] = S[i].Bsa.p + a; cn[i] = b - a;
                }
                PSACX_TRY(dist_put(blk, gi, va, cn, -1, chunks == 1));
            }
        }
        mark("SA -> ISA");
        uint64_t unf_b = 0, unf_e = 0;
        {
            std::vector<DBuf<T>> kept;
            PSACX_TRY(next_active(nullptr, nullptr, nact, nunf, kept, &unf_b, &unf_e, &tile_act));
            for (int i = 0; i < L; ++i) S[i].pos = std::move(kept[i]);
        }
        mark("active list");
        st.rounds[0].h = k; st.rounds[0].active = n; st.rounds[0].unfinished_buckets = unf_b; st.rounds[0].unfinished_elements = unf_e;
        st.n_rounds = 1;

        const unsigned id_bits = bits_for(n);
        for (uint64_t h = two_k; unf_b > 0 && h < n; h <<= 1) {
            std::vector<uint64_t> cnt(L), counts;
            for (int i = 0; i < L; ++i) cnt[i] = S[i].pos.n;
            PSACX_TRY(gather1(cnt, counts));
            uint64_t steps = 1;
            if (diet) for (int r = 0; r < P; ++r) steps = std::max<uint64_t>(steps, (counts[r] + slab_cap - 1) / slab_cap);
            steps = std::min<uint64_t>(steps, 64);
            std::vector<std::vector<uint64_t>> e(L);
            if (steps > 1) {
                // (a bucket longer than a block -- a homopolymer run of the length of a rank's share -- cannot be cut at bucket heads
                //  so that its parts on three ranks meet in one step: such a round takes every unresolved suffix in one step, as in
                //  the normal layout, and fails only if the device really has no room for its records)
                std::vector<uint64_t> whole(L, 0), whole_all;
                PSACX_TRY(par([&](int i) -> int { return slab_bounds(i, steps, e[i], &whole[i]); }));
                PSACX_TRY(gather1(whole, whole_all));
                for (uint64_t w : whole_all) if (w) steps = 1;
            }
            if (steps == 1) {
                std::vector<const T*> pl(L);
                for (int i = 0; i < L; ++i) pl[i] = S[i].pos.p;
                std::vector<DBuf<T>> kept;
                PSACX_TRY(refine_step(h, pl, cnt, counts, id_bits, kept, &unf_b, &unf_e));
                for (int i = 0; i < L; ++i) S[i].pos = std::move(kept[i]);
            } else {
                // Slabs.  Every rank cuts its list of unresolved positions into `steps` pieces at bucket heads; in step t rank
                // r works on its piece (t + r) mod steps, so the last piece of rank r and the first of rank r + 1 -- the two
                // halves of a bucket that crosses the block boundary -- meet in one step.  Later steps may read ranks that
                // earlier steps of this round already refined (Larsson-Sadakane style, see construct.hpp): the result is
                // the same, only the per-round counters may run ahead of the one-step log.
                g->last_slab_rounds++;
                std::vector<std::vector<uint64_t>> kept_n(L, std::vector<uint64_t>(steps, 0));
                uint64_t sum_b = 0, sum_e = 0;
                for (uint64_t t = 0; t < steps; ++t) {
                    std::vector<const T*> pl(L);
                    std::vector<uint64_t> c2(L), counts2;
                    for (int i = 0; i < L; ++i) { const uint64_t j = (t + (uint64_t)rank(i)) % steps; pl[i] = S[i].pos.p + e[i][j]; c2[i] = e[i][j + 1] - e[i][j]; }
                    PSACX_TRY(gather1(c2, counts2));
                    std::vector<DBuf<T>> kept;
                    uint64_t sb = 0, se = 0;
                    PSACX_TRY(refine_step(h, pl, c2, counts2, id_bits, kept, &sb, &se));
                    sum_b += sb; sum_e += se;
                    // the positions that stay unresolved replace the piece they came from
                    PSACX_TRY(par([&](int i) -> int {
                        const uint64_t j = (t + (uint64_t)rank(i)) % steps;
                        kept_n[i][j] = kept[i].n;
                        if (kept[i].n) {
                            MG_HIP(g, hipSetDevice(ctx(i)->device));
                            MG_HIP(g, hipMemcpyAsync(S[i].pos.p + e[i][j], kept[i].p, kept[i].n * sizeof(T), hipMemcpyDeviceToDevice, ctx(i)->stream));
                        }
                        kept[i].release();
                        return PSACX_OK;
                    }));
                }
                PSACX_TRY(par([&](int i) -> int {                  // close the gaps between the pieces
                    psacx_ctx* c = ctx(i);
                    MG_HIP(g, hipSetDevice(c->device));
                    uint64_t dst = 0;
                    for (uint64_t j = 0; j < steps; ++j) {
                        const uint64_t len = kept_n[i][j], src = e[i][j];
                        if (len && src != dst) {
                            DBuf<T> tmp; MG_OP(g, c, tmp.alloc(c, len));
                            MG_HIP(g, hipMemcpyAsync(tmp.p, S[i].pos.p + src, len * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                            MG_HIP(g, hipMemcpyAsync(S[i].pos.p + dst, tmp.p, len * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                        }
                        dst += len;
                    }
                    S[i].pos.n = dst;
                    return PSACX_OK;
                }));
                unf_b = sum_b; unf_e = sum_e;
            }
            mark("  active list");
            if (st.n_rounds < PSACX_MAX_ROUNDS) {
                psacx_round& rr = st.rounds[st.n_rounds++];
                std::memset(&rr, 0, sizeof(rr));
                rr.h = h; rr.unfinished_buckets = unf_b; rr.unfinished_elements = unf_e;
                for (int r = 0; r < P; ++r) rr.active += counts[r];
            }
        }
        PSACX_TRY(par([&](int i) -> int {
            S[i].Bsa.release(); S[i].pos.release();
            MG_HIP(g, hipSetDevice(ctx(i)->device));
            MG_HIP(g, hipStreamSynchronize(ctx(i)->stream));
            return PSACX_OK;
        }));
        ex_collect();
        return PSACX_OK;
    }

    // ---------------------------------------------------------------- queries on a finished result: defined in multi_queries.hpp
    // (all nearest smaller values, left-branching characters, suffix-tree node table, distributed checker)
    struct AnsvState { std::vector<const T*> block; std::vector<uint64_t> m; std::vector<Pyramid<T>> pyr; std::vector<DBuf<T>> pyr_mem; std::vector<uint64_t> mins; };
    int ansv_pyramid(int i, const T* block, uint64_t m, Pyramid<T>& Pm, DBuf<T>& mem, uint64_t* block_min);
    int ansv_ask(AnsvState& A, const std::vector<const T*>& cls, const std::vector<const T*>& start1, const std::vector<const T*>& thr, const std::vector<uint64_t>& cnt, bool strict, bool left, std::vector<DBuf<T>>& idx, std::vector<DBuf<T>>& val);
    int ansv_search(AnsvState& A, const std::vector<const T*>& start1, const std::vector<const T*>& thr, const std::vector<uint64_t>& cnt, bool strict, bool left, bool have_local, std::vector<DBuf<T>>& idx, std::vector<DBuf<T>>& val);
    int ansv(const std::vector<const T*>& block, const std::vector<uint64_t>& m_local, int left_type, int right_type, uint64_t nonsv, const std::vector<uint64_t*>& out_left, const std::vector<uint64_t*>& out_right);
    int left_chars(const std::vector<const uint8_t*>& text, const std::vector<uint64_t>& m_local, const std::vector<T*>& d_sa, const std::vector<T*>& d_lcp, const std::vector<uint8_t*>& d_lc);
    int suffix_tree(const std::vector<const uint8_t*>& text, const std::vector<uint64_t>& m_local, const std::vector<T*>& d_sa, const std::vector<T*>& d_lcp, const std::vector<unsigned long long*>* d_nodes, uint32_t* sigma);
    int check(const std::vector<const uint8_t*>& text, const std::vector<uint64_t>& m_local, const std::vector<T*>& d_sa, const std::vector<T*>& d_isa, const std::vector<T*>& d_lcp, bool with_lcp, uint64_t errors[4]);

    // boundary bucket ids of every block, the list of positions that still share a bucket (suffix_array.hpp:925-965)
    // and the global counters.  ids == nullptr: first round (ids = Bsa, every position is a list entry).
    // tile_counts (first round): the per-tile counts the rebucket kernel left (op_rebucket_first); the compaction then reads the
    // ids once instead of twice and writes straight into a list of the known length
    int next_active(std::vector<DBuf<T>>* ids, const std::vector<const T*>* plist, const std::vector<uint64_t>& nact, const std::vector<uint64_t>& nunf,
                    std::vector<DBuf<T>>& kept_out, uint64_t* unf_b, uint64_t* unf_e, std::vector<DBuf<uint64_t>>* tile_counts = nullptr) {
        std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(5, 0));
        std::vector<uint64_t> cnt(L);
        PSACX_TRY(par([&](int i) -> int {
            const T* a = ids ? (*ids)[i].p : S[i].Bsa.p;
            cnt[i] = ids ? (*ids)[i].n : S[i].m;
            if (cnt[i]) {
                std::vector<uint64_t> o;
                PSACX_TRY(fetch(i, a, {0, cnt[i] - 1}, o));
                mine[i][0] = 1; mine[i][1] = o[0]; mine[i][2] = o[1];
            }
            mine[i][3] = nact[i]; mine[i][4] = nunf[i];
            return PSACX_OK;
        }));
        std::vector<uint64_t> all;
        PSACX_TRY(gather(5, mine, all));
        *unf_b = *unf_e = 0;
        for (int r = 0; r < P; ++r) { *unf_e += all[(size_t)r * 5 + 3]; *unf_b += all[(size_t)r * 5 + 4]; }
        kept_out.clear(); kept_out.resize(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            const int r = rank(i);
            uint64_t pid = 0, nid = 0;
            for (int s = r - 1; s >= 0; --s) if (all[(size_t)s * 5]) { pid = all[(size_t)s * 5 + 2]; break; }
            for (int s = r + 1; s < P; ++s) if (all[(size_t)s * 5]) { nid = all[(size_t)s * 5 + 1]; break; }
            if (tile_counts && !ids && (*tile_counts)[i].p) {
                MG_OP(g, c, kept_out[i].alloc(c, nact[i]));
                MG_OP(g, c, op_compact_counted<T>(c, S[i].Bsa.p, cnt[i], S[i].off, pid, nid, (*tile_counts)[i].p, kept_out[i].p));
                return PSACX_OK;
            }
            DBuf<T> out; MG_OP(g, c, out.alloc(c, cnt[i]));
            uint64_t kept = 0;
            MG_OP(g, c, op_compact<T>(c, ids ? (*ids)[i].p : S[i].Bsa.p, ids ? (*plist)[i] : (const T*)nullptr, cnt[i], S[i].off, pid, nid, out.p, &kept));
            MG_OP(g, c, kept_out[i].alloc(c, kept));
            MG_HIP(g, hipSetDevice(c->device));
            if (kept) MG_HIP(g, hipMemcpyAsync(kept_out[i].p, out.p, kept * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
            return PSACX_OK;
        }));
        return PSACX_OK;
    }

    uint16_t codes_[256];
};

} // namespace psacx
