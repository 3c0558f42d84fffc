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
#include "ansv_seq.hpp"
#include "shm_link.hpp"
#include "slice_inv.hpp"

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

inline std::vector<uint64_t> prefix_of(const std::vector<uint64_t>& x) {
    std::vector<uint64_t> o(x.size() + 1, 0);
    for (size_t i = 0; i < x.size(); ++i) o[i + 1] = o[i] + x[i];
    return o;
}

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

template <typename T> __global__ void widen_text_kernel(const uint8_t* __restrict__ t, uint64_t cnt, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) out[i] = (T)t[i];
}

// ---- kernels of the distributed left-branching characters (MultiRun::left_chars): the text position SA[i-1] + LCP[i] of
//      every entry of a piece (prev_sa: SA of the entry before the piece; has_prev = 0 at global position 0 -> n = "none"),
//      then the fetched characters narrowed to bytes ('\0' where the position is past the end, alphabet.hpp:168)
template <typename T>
__global__ void lc_queries_kernel(const T* __restrict__ SA, const T* __restrict__ LCP, uint64_t cnt, uint64_t n, int has_prev, T prev_sa,
                                  T* __restrict__ q) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        uint64_t p = n;
        if (i || has_prev) {
            p = (uint64_t)(i ? SA[i - 1] : prev_sa) + (uint64_t)LCP[i];
            if (p > n) p = n;
        }
        q[i] = (T)p;
    }
}
template <typename T>
__global__ void lc_narrow_kernel(const T* __restrict__ ch, const T* __restrict__ q, uint64_t cnt, uint64_t n, uint8_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride)
        out[i] = (uint64_t)q[i] < n ? (uint8_t)ch[i] : (uint8_t)0;
}

// ---- kernels of the distributed ANSV (MultiRun::ansv).  Start positions travel as T with one added (0 = before
//      position 0, n + 1 = past the end), "none" as all ones.
template <typename T>
__global__ void ansv_owner_kernel(const T* __restrict__ start1, uint64_t cnt, BlkDist d, T* __restrict__ cls) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        uint64_t s = (uint64_t)start1[j];
        s = s ? s - 1 : 0;
        if (s >= d.n) s = d.n - 1;
        cls[j] = (T)d.rank_of(s);
    }
}
// nearest element of this block strictly beyond start (left: below it) with value < thr (strict) or <= thr
template <typename T>
__global__ void nsv_from_enc_kernel(Pyramid<T> P, uint64_t m, uint64_t off, const T* __restrict__ start1, const T* __restrict__ thr,
                                    uint64_t cnt, int strict, int left, T* __restrict__ out_idx, T* __restrict__ out_val) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const long long s = (long long)(uint64_t)start1[j] - 1 - (long long)off;          // block-relative, may be < 0 or >= m
        const T v = thr[j];
        uint64_t r = NSV_NONE;
        if (m) {
            if (left) {
                if (s > 0) {
                    if ((uint64_t)s >= m) {
                        const T x = P.lvl[0][m - 1];
                        r = (strict ? x < v : x <= v) ? m - 1 : (m > 1 ? nsv_search<T, true>(P, m - 1, v, strict != 0) : NSV_NONE);
                    } else r = nsv_search<T, true>(P, (uint64_t)s, v, strict != 0);
                }
            } else if (s < (long long)m - 1) {
                if (s < 0) {
                    const T x = P.lvl[0][0];
                    r = (strict ? x < v : x <= v) ? 0 : (m > 1 ? nsv_search<T, false>(P, 0, v, strict != 0) : NSV_NONE);
                } else r = nsv_search<T, false>(P, (uint64_t)s, v, strict != 0);
            }
        }
        out_idx[j] = r == NSV_NONE ? ~(T)0 : (T)(off + r);
        out_val[j] = r == NSV_NONE ? (T)0 : P.lvl[0][r];
    }
}
// open queries (idx == none): the nearest rank beyond the start's owner whose block minimum qualifies, P = none
template <typename T>
__global__ void ansv_target_kernel(const T* __restrict__ own, const T* __restrict__ thr, const T* __restrict__ idx, uint64_t cnt, RankMins mins,
                                   RankMins sizes, int P, int strict, int left, T* __restrict__ target) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        int t = P;
        if (idx[j] == ~(T)0) {
            const unsigned long long v = (unsigned long long)thr[j];
            const int o = (int)own[j];
            if (left) { for (int b = o - 1; b >= 0; --b) if (sizes.v[b] && (strict ? mins.v[b] < v : mins.v[b] <= v)) { t = b; break; } }
            else { for (int b = o + 1; b < P; ++b) if (sizes.v[b] && (strict ? mins.v[b] < v : mins.v[b] <= v)) { t = b; break; } }
        }
        target[j] = (T)t;
    }
}
template <typename T>
__global__ void ansv_merge_kernel(T* __restrict__ idx, T* __restrict__ val, const T* __restrict__ idx2, const T* __restrict__ val2,
                                  const T* __restrict__ target, uint64_t cnt, int P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride)
        if ((int)target[j] < P) { idx[j] = idx2[j]; val[j] = val2[j]; }
}
template <typename T>
__global__ void fill_t_kernel(T* __restrict__ a, uint64_t cnt, T v) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) a[j] = v;
}
// local tile ANSV results (block-relative uint64, NSV_NONE = not inside the block) -> idx (global T, all ones = open) and value found
template <typename T>
__global__ void ansv_local_to_idx_kernel(const uint64_t* __restrict__ loc, const T* __restrict__ block, uint64_t cnt, uint64_t off,
                                         T* __restrict__ idx, T* __restrict__ val) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const uint64_t r = loc[j];
        idx[j] = r == NSV_NONE ? ~(T)0 : (T)(off + r);
        val[j] = r == NSV_NONE ? (T)0 : block[r];
    }
}
// start positions (plus one) for the follow-up searches of furthest_eq
template <typename T>
__global__ void ansv_next_start_kernel(const T* __restrict__ idx, uint64_t cnt, T when_none1, T* __restrict__ start1) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride)
        start1[j] = idx[j] == ~(T)0 ? when_none1 : (T)(idx[j] + 1);
}
template <typename T>
__global__ void ansv_finish_kernel(const T* __restrict__ first, const T* __restrict__ far, int use_far, uint64_t cnt, uint64_t nonsv,
                                   uint64_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const T a = first[j];
        const T r = (use_far && a != ~(T)0) ? far[j] : a;
        out[j] = r == ~(T)0 ? nonsv : (uint64_t)r;
    }
}

// Distributed check, per block (see MultiRun::check).  For SA position p = off + i:
//   back[i] = ISA[SA[p]] (must be p), ch[i] = S[SA[p]], nx[i] = ISA[SA[p] + 1] (undefined when SA[p] + 1 == n).
// Queries of the LCP recurrence: LCP[p] = 0 if the first characters differ, 1 if the smaller suffix is one character
// long, else 1 + min(LCP[ISA[SA[p-1]+1] + 1 .. ISA[SA[p]+1]]).
template <typename T>
__global__ void check_queries_kernel(const T* __restrict__ SA, const T* __restrict__ ch, const T* __restrict__ nx, uint64_t cnt, uint64_t n,
                                     int has_prev, T prev_sa, T prev_ch, T prev_nx, T* __restrict__ qlo, T* __restrict__ qhi) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        T lo = 0, hi = 1;                                   // a harmless query where none is needed
        if (i > 0 || has_prev) {
            const uint64_t a = i ? (uint64_t)SA[i - 1] : (uint64_t)prev_sa, b = SA[i];
            const T ca = i ? ch[i - 1] : prev_ch, na = i ? nx[i - 1] : prev_nx;
            if (a < n && b < n && ca == ch[i] && a + 1 < n && b + 1 < n && na < nx[i]) { lo = (T)(na + 1); hi = (T)(nx[i] + 1); }
        }
        qlo[i] = lo; qhi[i] = hi;
    }
}
template <typename T>
__global__ void check_verdict_kernel(const T* __restrict__ SA, const T* __restrict__ back, const T* __restrict__ ch, const T* __restrict__ nx,
                                     const T* __restrict__ LCP, const T* __restrict__ mins, uint64_t cnt, uint64_t off, uint64_t n,
                                     int has_prev, T prev_sa, T prev_ch, T prev_nx, unsigned long long* __restrict__ err) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned e0 = 0, e1 = 0, e2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        const uint64_t b = SA[i], p = off + i;
        if (b >= n || (uint64_t)back[i] != p) { ++e0; continue; }
        if (p == 0) { if (LCP && LCP[0] != 0) atomicAdd(&err[3], 1ull); continue; }
        if (i == 0 && !has_prev) continue;
        const uint64_t a = i ? (uint64_t)SA[i - 1] : (uint64_t)prev_sa;
        if (a >= n) continue;                               // counted where it lives
        const T ca = i ? ch[i - 1] : prev_ch, cb = ch[i];
        const T na = i ? nx[i - 1] : prev_nx, nb = nx[i];
        bool ok = ca < cb;
        if (ca == cb) ok = (a + 1 == n) || (b + 1 < n && na < nb);
        if (!ok) { ++e1; continue; }
        if (LCP) {
            uint64_t want;
            if (ca != cb) want = 0;
            else if (a + 1 == n) want = 1;
            else want = 1 + (uint64_t)mins[i];
            if ((uint64_t)LCP[i] != want) ++e2;
        }
    }
    e0 = wave_reduce<uint32_t>(e0, OpSum()); e1 = wave_reduce<uint32_t>(e1, OpSum()); e2 = wave_reduce<uint32_t>(e2, OpSum());
    if (lane_id() == 0) {
        if (e0) atomicAdd(&err[0], (unsigned long long)e0);
        if (e1) atomicAdd(&err[1], (unsigned long long)e1);
        if (e2) atomicAdd(&err[2], (unsigned long long)e2);
    }
}

// ------------------------------------------------------------------------------------------------------------
// number of leading entries <= key of a non-decreasing array (one thread)
// ---- string sets (construct_ss on p ranks, suffix_array.hpp:267-363): for the positions base .. base + cnt of the text,
// slen = characters to the end of the string holding the position, soff = characters from its start (off: the nstr + 1
// global string offsets).  Positions past the end of the text count as strings of one character.
template <typename T>
__global__ void string_pos_kernel(const uint64_t* __restrict__ off, uint64_t nstr, uint64_t n, uint64_t base, uint64_t cnt, T* __restrict__ slen,
                                  T* __restrict__ soff) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const uint64_t i = base + j;
        if (i >= n) { if (slen) slen[j] = (T)1; if (soff) soff[j] = (T)0; continue; }
        uint64_t lo = 0, hi = nstr;              // largest t with off[t] <= i
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
        if (slen) slen[j] = (T)(off[lo + 1] - i);
        if (soff) soff[j] = (T)(i - off[lo]);
    }
}
// the ranks a rank answers for "the suffix h further" in a string set: none (all ones) when that suffix starts in another
// string, i.e. when the position lies fewer than h characters into its own string (shifting.hpp:374-418)
template <typename T>
__global__ void mask_by_string_kernel(const T* __restrict__ isa, const T* __restrict__ soff, uint64_t m, uint64_t h, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) out[j] = (uint64_t)soff[j] >= h ? isa[j] : ~(T)0;
}
template <typename T>
__global__ void finish_b2_masked_kernel(const T* __restrict__ ans, const T* __restrict__ q, uint64_t cnt, uint64_t n, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride)
        out[j] = ((uint64_t)q[j] < n && ans[j] != ~(T)0) ? (T)(ans[j] + 1) : (T)0;
}

// ---- suffix-tree node table over block-distributed SA / LCP (suffix_tree.hpp:43-223 for_each_parent, :440-499)
// For the LCP index i = off + j: the parent of leaf n + i and (when there is one) of internal node i, from the ANSV of LCP
// (left furthest_eq, right nearest_sm, suffix_tree.hpp:62) and the LCP values found there; q = the text position whose
// character labels the edge.  An index without an internal-node record gets parent = i and q2 = ST_NOREC.
constexpr uint64_t ST_NOREC = ~0ull;
template <typename T>
__global__ void st_parents_kernel(const T* __restrict__ LCP, const T* __restrict__ SA, uint64_t m, uint64_t off, uint64_t n,
                                  const uint64_t* __restrict__ lnsv, const uint64_t* __restrict__ rnsv, const T* __restrict__ lcp_l,
                                  const T* __restrict__ lcp_r, int has_next, T next_lcp, T* __restrict__ p1, uint64_t* __restrict__ q1,
                                  T* __restrict__ p2, uint64_t* __restrict__ q2) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const uint64_t i = off + j, ln = lnsv[j], rn = rnsv[j], sa = SA[j], li = LCP[j];
        const uint64_t lnext = j + 1 < m ? (uint64_t)LCP[j + 1] : (has_next ? (uint64_t)next_lcp : 0);
        const uint64_t lv = ln != NSV_NONE ? (uint64_t)lcp_l[j] : 0, rv = rn != NSV_NONE ? (uint64_t)lcp_r[j] : 0;
        uint64_t parent, lcp_val;
        if (i == 0) { lcp_val = n > 1 ? lnext : 0; parent = lcp_val > 0 ? 1 : 0; }
        else if (i == n - 1 || li >= lnext) {
            lcp_val = lv;
            if (ln != NSV_NONE && lcp_val == li) parent = ln; else { parent = i; lcp_val = li; }
        } else { parent = i + 1; lcp_val = lnext; }
        p1[j] = (T)parent; q1[j] = sa + lcp_val;
        bool rec = !(i == 0 || li == 0);
        if (rec) {
            if (rn == NSV_NONE) { if (lv == li) rec = false; else { parent = ln; lcp_val = lv; } }
            else if (lv >= rv) { if (lv == li) rec = false; else { parent = ln; lcp_val = lv; } }
            else { parent = rn; lcp_val = rv; }
        }
        p2[j] = rec ? (T)parent : (T)i;
        q2[j] = rec ? sa + lcp_val : ST_NOREC;
    }
}
// the positions as index words for the bulk fetch (past the end / no record: position 0, the answer is not used)
template <typename T>
__global__ void st_positions_kernel(const uint64_t* __restrict__ q, uint64_t m, uint64_t n, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) out[j] = q[j] < n ? (T)q[j] : (T)0;
}
template <typename T>
__global__ void st_nsv_positions_kernel(const uint64_t* __restrict__ q, uint64_t m, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) out[j] = q[j] != NSV_NONE ? (T)q[j] : (T)0;
}
// what travels to the owner of the parent's row: x = the LCP index the child stands for, y = column | leaf flag << 16
// (0xFFFF: no record)
template <typename T>
__global__ void st_payload_kernel(const uint64_t* __restrict__ q, const T* __restrict__ ch, uint64_t m, uint64_t off, uint64_t n, CodeTable tab,
                                  int leaf, T* __restrict__ x, T* __restrict__ y) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        x[j] = (T)(off + j);
        if (q[j] == ST_NOREC) y[j] = (T)0xFFFFu;
        else y[j] = (T)((q[j] < n ? (unsigned)tab.c[(unsigned)ch[j] & 255u] : 0u) | ((unsigned)leaf << 16));
    }
}
template <typename T>
__global__ void st_put_kernel(unsigned long long* __restrict__ nodes, uint64_t off, uint64_t row, const T* __restrict__ pos, const T* __restrict__ x,
                              const T* __restrict__ y, uint64_t cnt, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const unsigned yy = (unsigned)y[j];
        if ((yy & 0xFFFFu) == 0xFFFFu) continue;
        nodes[((uint64_t)pos[j] - off) * row + (yy & 0xFFFFu)] = (yy >> 16) ? n + (uint64_t)x[j] : (uint64_t)x[j];
    }
}

template <typename T> __global__ void upper_bound_kernel(const T* __restrict__ a, uint64_t n, uint64_t key, uint64_t* __restrict__ out) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if ((uint64_t)a[mid] <= key) lo = mid + 1; else hi = mid; }
    *out = lo;
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
            if (S[i].LCP) r.k2.borrow(c, S[i].LCP, cnt); else if (want_k2) MG_OP(g, c, r.k2.alloc(c, cnt, reserve_of(i)));
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
        if (r.k1.p && !r.k1.owned()) S[i].out_busy = false;
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

    explicit MultiRun(psacx_multi* mg) : g(mg), P(mg->nranks), L(mg->nlocal), trace_(getenv("PSACX_MULTI_TRACE") != nullptr) {
        if (const char* e = getenv("PSACX_MULTI_WIRE_PIECE")) wire_piece_ = std::max<size_t>(256, strtoull(e, nullptr, 10));
        if (const char* e = getenv("PSACX_MULTI_PIECES")) pieces_env_ = std::max(1, atoi(e));      // ranges per destination of the first round's shuffle (tests)
        one_stage_env_ = getenv("PSACX_ONE_STAGE") != nullptr;                                      // first round as one sort over both key words
        if (const char* e = getenv("PSACX_MULTI_SLAB")) slab_env_ = strtoull(e, nullptr, 10);       // unresolved suffixes per refinement slab (reduced-memory layout)
        if (const char* e = getenv("PSACX_MULTI_CHECK_CHUNKS")) check_chunks_env_ = strtoull(e, nullptr, 10);
        // PSACX_SLICE_SHAPE=wb,s1,step (tests: the levels of the slice inversion on small inputs): window bits, slice bits, slices per step; 0 = default
        if (const char* e = getenv("PSACX_SLICE_SHAPE")) { unsigned a = 0, b = 0; unsigned long long st = 0; if (sscanf(e, "%u,%u,%llu", &a, &b, &st) >= 1) { slice_wb_env_ = a; slice_s1_env_ = b; slice_step_env_ = st; } }
        solo_ = P == 1 && !mg->force_wire;
        t_last_ = t_phase_ = std::chrono::steady_clock::now();
    }
    // PSACX_MULTI_TRACE=1: wall time of every phase on stderr (all local streams drained at each mark)
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
    struct Msg { int peer; uint64_t off, cnt; };
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
            const uint64_t c = rec[i].cnt;
            std::vector<uint64_t> pos;
            for (int s = 0; s < SAMPLES && c; ++s) {
                const uint64_t lo = (uint64_t)(((unsigned __int128)c * s) / SAMPLES), hi = (uint64_t)(((unsigned __int128)c * (s + 1)) / SAMPLES);
                if (hi <= lo) continue;
                uint64_t z = ((uint64_t)rank(i) << 32 | (uint64_t)s) + 0x9E3779B97F4A7C15ull * (sort_calls_ + 1);      // splitmix64
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
                const uint64_t p = lo + z % (hi - lo);
                if (pos.empty() || pos.back() != p) pos.push_back(p);
            }
            std::vector<uint64_t> a, b;
            PSACX_TRY(fetch(i, rec[i].k1.p, pos, a));
            PSACX_TRY(fetch(i, rec[i].k2.p, pos, b));
            mine[i][0] = pos.size();
            for (size_t s = 0; s < pos.size(); ++s) { mine[i][1 + 3 * s] = a[s]; mine[i][2 + 3 * s] = b[s]; mine[i][3 + 3 * s] = pos[s]; }
            return PSACX_OK;
        }));
        std::vector<uint64_t> all;
        PSACX_TRY(gather(1 + 3 * SAMPLES, mine, all));
        struct Smp { uint64_t k1, k2, r, p; bool operator<(const Smp& o) const { return k1 != o.k1 ? k1 < o.k1 : k2 != o.k2 ? k2 < o.k2 : r != o.r ? r < o.r : p < o.p; }
                     bool operator==(const Smp& o) const { return k1 == o.k1 && k2 == o.k2 && r == o.r && p == o.p; } };
        std::vector<Smp> flat;
        for (int r = 0; r < P; ++r) {
            const uint64_t* row = &all[(size_t)r * (1 + 3 * SAMPLES)];
            for (uint64_t s = 0; s < row[0]; ++s) flat.push_back(Smp{row[1 + 3 * s], row[2 + 3 * s], (uint64_t)r, row[3 + 3 * s]});
        }
        std::sort(flat.begin(), flat.end());
        std::vector<Smp> spl;
        for (int d = 1; d < P && !flat.empty(); ++d) spl.push_back(flat[std::min(flat.size() - 1, flat.size() * d / P)]);
        std::sort(spl.begin(), spl.end());
        spl.erase(std::unique(spl.begin(), spl.end()), spl.end());
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
            const uint64_t gr = G[rank(i)];
            bounds[i].assign(P + 1, c2[i]);
            for (int d = 0; d < P; ++d) bounds[i][d] = std::min<uint64_t>(TP[d] > gr ? TP[d] - gr : 0, c2[i]);
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
            const int me = rank(i);
            const uint64_t g0 = held_from_[me], g1 = g0 + held_cnt_[me];
            if (g0 < TP[me] || g0 - TP[me] != head_[i] || g1 < TP[me + 1]) { mg_set_err(g, "re-balance in place: a rank does not hold the tail of its block"); return PSACX_EINVAL; }
            for (int d = 0; d < P; ++d) {
                if (d == me) continue;
                const uint64_t lo = std::max(g0, TP[d]), hi = std::min(g1, TP[d + 1]);
                if (lo < hi) sends[i].push_back(Msg{d, head_[i] + (lo - g0), hi - lo});
            }
            for (int r = 0; r < P; ++r) {
                if (r == me) continue;
                const uint64_t lo = std::max(held_from_[r], TP[me]), hi = std::min(held_from_[r] + held_cnt_[r], TP[me + 1]);
                if (lo < hi) recvs[i].push_back(Msg{r, lo - TP[me], hi - lo});
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

    // Both words of the packed 2k-character window of the suffixes gidx[i][0 .. cnt[i]) (global positions), computed by the
    // ranks that own those positions from their text blocks + halos (tbuf: block + 2k characters) and sent back in query
    // order: the remote form of window_word2() for the suffixes that tie on the leading bits of word 1.
    int dist_windows(const std::vector<DBuf<uint8_t>>& tbuf, uint32_t two_k, const CodeTable& tab, const KeyShape& ks, const std::vector<const T*>& gidx,
                     const std::vector<uint64_t>& cnt, std::vector<DBuf<T>>& w1, std::vector<DBuf<T>>& w2) {
        w1.clear(); w1.resize(L); w2.clear(); w2.resize(L);
        auto answer = [&](int i, const T* q, uint64_t qn, T* o1, T* o2) -> int {
            psacx_ctx* c = ctx(i);
            OP_PROLOGUE(c);
            if (qn) {
                hipLaunchKernelGGL((window_at_kernel<T, 256>), dim3(grid_for(c, qn, 256, 16)), dim3(256), 0, c->stream, tbuf[i].p, S[i].m + two_k, S[i].off, q, qn,
                                   tab, ks, o1, o2);
                PSACX_HIP(c, hipGetLastError());
            }
            return PSACX_OK;
        };
        if (solo_) {
            MG_OP(g, ctx(0), w1[0].alloc(ctx(0), cnt[0])); MG_OP(g, ctx(0), w2[0].alloc(ctx(0), cnt[0]));
            MG_OP(g, ctx(0), answer(0, gidx[0], cnt[0], w1[0].p, w2[0].p));
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
        std::vector<DBuf<T>> a1(L), a2(L);
        std::vector<std::vector<uint64_t>> back_bounds(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, a1[i].alloc(c, q[i][0].n)); MG_OP(g, c, a2[i].alloc(c, q[i][0].n));
            MG_OP(g, c, answer(i, q[i][0].p, q[i][0].n, a1[i].p, a2[i].p));
            back_bounds[i] = prefix_of(rc[i]);
            in[i] = {a1[i].p, a2[i].p};
            return PSACX_OK;
        }));
        PSACX_TRY(exchange<T>(2, in, back_bounds, got, rc2));
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, w1[i].alloc(c, cnt[i])); MG_OP(g, c, w2[i].alloc(c, cnt[i]));
            MG_OP(g, c, op_put(c, w1[i].p, routed[i].v.p, cnt[i], 0, got[i][0].p, 0));      // undo the routing permutation
            MG_OP(g, c, op_put(c, w2[i].p, routed[i].v.p, cnt[i], 0, got[i][1].p, 0));
            return PSACX_OK;
        }));
        return PSACX_OK;
    }

    // The first sort in two-word form (what the one-GPU engine does, construct.hpp "two stages"): the records are (word 1,
    // suffix) only.  When the leading `lead` = bits1 - lo1 bits of word 1 separate almost every suffix of the whole text,
    //   1. the shuffle goes by those leading bits alone -- splitters are prefix values and equal prefixes never part, so a
    //      group of suffixes that tie on them is whole on one rank -- and moves two words per record instead of three;
    //   2. the local sort is a prefix sort of two-word records on the leading bits (lead / 8 passes of 4w bytes per record
    //      instead of all digits of both words at 6w);
    //   3. the few suffixes that still tie are compacted, the full window of each is fetched from the rank that owns its text
    //      (dist_windows), the groups are ordered by it (in registers when every group is tiny, else by a radix sort of the
    //      compacted records) and written back; word 2 exists for those records only, which is all rebucket_first_kernel reads.
    // rec[i]: k1 and v filled, k2 allocated but unused until step 3.  Returns PSACX_RETRY_ before anything has moved when the
    // samples say the text is repetitive (many equal prefixes) or the prefixes cannot balance the ranks: the caller then runs
    // the three-word path.
    static constexpr int PSACX_RETRY_ = 1;
    int sort_first_two_word(std::vector<Rec<T>>& rec, const std::vector<uint64_t>& targets, unsigned bits1, unsigned bits2, unsigned lo1,
                            const std::vector<DBuf<uint8_t>>& tbuf, uint32_t two_k, const CodeTable& tab, const KeyShape& ks, bool trust, uint64_t spec_front) {
        ++sort_calls_;
        constexpr int SAMPLES = 8192;
        // Shuffle by key ranges (default with more than one rank): every destination's share of the prefix space is cut into QR
        // ranges, the block is partitioned once by (destination, range), range q of every destination travels in exchange q, and
        // the receiver sorts range q -- complete and final as soon as it has landed -- on its compute stream while ranges
        // q + 1 .. are still in flight on the second stream: the local sort runs under the shuffle (idxsort.hpp:58-62 sorts after
        // its Alltoallv has returned).  PSACX_MULTI_SHUFFLE_BY_POSITION=1: the earlier form (pieces of the block by position,
        // piece q + 1 partitioned while piece q travels, one local sort at the end).
        const bool by_range = !solo_;
        int QR = 1;
        if (by_range) { QR = std::max(1, std::min(4, 64 / P)); if (pieces_env_ > 0) QR = std::max(1, std::min(64 / P, pieces_env_)); }
        std::vector<uint64_t> spl;
        {
            std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(1 + SAMPLES, 0));
            PSACX_TRY(par([&](int i) -> int {
                const uint64_t c = rec[i].cnt;
                std::vector<uint64_t> pos;
                for (int s = 0; s < SAMPLES && c; ++s) {
                    const uint64_t lo = (uint64_t)(((unsigned __int128)c * s) / SAMPLES), hi = (uint64_t)(((unsigned __int128)c * (s + 1)) / SAMPLES);
                    if (hi <= lo) continue;
                    uint64_t z = ((uint64_t)rank(i) << 32 | (uint64_t)s) + 0x9E3779B97F4A7C15ull * (sort_calls_ + 1);      // splitmix64
                    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
                    const uint64_t p = lo + z % (hi - lo);
                    if (pos.empty() || pos.back() != p) pos.push_back(p);
                }
                std::vector<uint64_t> a;
                PSACX_TRY(fetch(i, rec[i].k1.p, pos, a));
                mine[i][0] = pos.size();
                for (size_t s = 0; s < pos.size(); ++s) mine[i][1 + s] = a[s] >> lo1;
                return PSACX_OK;
            }));
            std::vector<uint64_t> all, flat;
            PSACX_TRY(gather(1 + SAMPLES, mine, all));
            for (int r = 0; r < P; ++r) {
                const uint64_t* row = &all[(size_t)r * (1 + SAMPLES)];
                flat.insert(flat.end(), row + 1, row + 1 + row[0]);
            }
            std::sort(flat.begin(), flat.end());
            if (!trust && !flat.empty()) {
                // equal prefixes among a few thousand samples of a 2^lead space: a repetitive text, whose tie groups are long
                size_t dup = 0;
                for (size_t j = 1; j < flat.size(); ++j) dup += flat[j] == flat[j - 1];
                if (dup * 64 > flat.size()) return PSACX_RETRY_;
            }
            if (by_range) {
                // P * QR key ranges, QR consecutive ones per destination (equal splitters leave a range empty: the class numbers
                // must stay destination * QR + range)
                for (int cc = 1; cc < P * QR && !flat.empty(); ++cc) spl.push_back(flat[std::min(flat.size() - 1, flat.size() * cc / (size_t)(P * QR))]);
            } else {
                for (int d = 1; d < P && !flat.empty(); ++d) spl.push_back(flat[std::min(flat.size() - 1, flat.size() * d / P)]);
                spl.erase(std::unique(spl.begin(), spl.end()), spl.end());
            }
            if (!trust && !flat.empty() && P > 1) {
                // the share of the samples each destination would receive (destination = splitters <= prefix)
                std::vector<size_t> share(P, 0);
                for (uint64_t x : flat) share[std::min<size_t>((size_t)(std::upper_bound(spl.begin(), spl.end(), x) - spl.begin()) / (by_range ? QR : 1), P - 1)]++;
                for (int d = 0; d < P; ++d) if ((double)share[d] * P > 1.06 * (double)flat.size()) return PSACX_RETRY_;
            }
        }
        // The suffix a record stands for travels as a 32-bit entry while the text has at most 2^32 characters, else as a word.
        const bool v32 = sizeof(T) == 8 && n <= (1ull << 32);
        const size_t vb = v32 ? 4 : sizeof(T);
        // record j of local rank i stands for suffix: the spec short suffixes first on rank 0 (n - 1 - j), then the block in order
        auto payload_of = [&](int i, uint64_t a, uint64_t* spec_q, uint64_t* specn_q, uint64_t* voff_q) {
            const uint64_t front = rank(i) == 0 ? spec_front : 0;
            if (a == 0 && front) { *spec_q = front; *specn_q = n; *voff_q = 0; }          // (rank 0's block starts at position 0)
            else { *spec_q = 0; *specn_q = 0; *voff_q = S[i].off + a - front; }
        };
        bool sorted_already = false;
        if (by_range) {
            const int NC = P * QR;
            Splitters sp; std::memset(&sp, 0, sizeof(sp));
            sp.n = (uint32_t)spl.size();
            for (uint32_t s2 = 0; s2 < sp.n; ++s2) sp.k1[s2] = spl[s2];
            constexpr uint64_t SPAN = 256 * 32;
            std::vector<std::vector<uint64_t>> cnt_c(L, std::vector<uint64_t>((size_t)NC, 0));
            std::vector<DBuf<uint8_t>> cls(L);
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                const uint64_t cn = rec[i].cnt;
                MG_OP(g, c, cls[i].alloc(c, cn + 16));
                DBuf<unsigned long long> d_counts; MG_OP(g, c, d_counts.alloc(c, 64));
                MG_HIP(g, hipSetDevice(c->device));
                MG_HIP(g, hipMemsetAsync(d_counts.p, 0, 64 * 8, c->stream));
                if (cn) {
                    const uint64_t one = (cn + SPAN - 1) / SPAN * SPAN;           // the whole block as one "piece"
                    hipLaunchKernelGGL((classify_prefix_kernel<T>), dim3((unsigned)(one / SPAN)), dim3(256), 0, c->stream, (const T*)rec[i].k1.p, cn, lo1, sp, cls[i].p, one, d_counts.p);
                    MG_HIP(g, hipGetLastError());
                }
                MG_OP(g, c, ensure_pinned(c, 64 * 8 + 65536 + 32768));
                MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, d_counts.p, 64 * 8, hipMemcpyDeviceToHost, c->stream));
                MG_HIP(g, hipStreamSynchronize(c->stream));
                const unsigned long long* h = reinterpret_cast<const unsigned long long*>(c->pinned + 32768);
                for (int cc = 0; cc < NC; ++cc) cnt_c[i][cc] = h[cc];
                return PSACX_OK;
            }));
            std::vector<uint64_t> table;                       // table[r * NC + destination * QR + range]
            PSACX_TRY(gather(NC, cnt_c, table));
            std::vector<Rec<T>> grp(L), rcv(L);
            std::vector<std::vector<uint64_t>> rbase(L), soff(L);          // start of range q in the receive arrays; start of class c in the partitioned block
            int rc_alloc = PSACX_OK;
            for (int i = 0; i < L && rc_alloc == PSACX_OK; ++i) {
                const int me = rank(i);
                rbase[i].assign(QR + 1, 0);
                for (int q = 0; q < QR; ++q) { uint64_t t = 0; for (int r = 0; r < P; ++r) t += table[(size_t)r * NC + me * QR + q]; rbase[i][q + 1] = rbase[i][q] + t; }
                soff[i] = prefix_of(cnt_c[i]);
                rc_alloc = take3(i, grp[i], rec[i].cnt, false);
            }
            PSACX_TRY(agree(rc_alloc));
            for (int i = 0; i < L && rc_alloc == PSACX_OK; ++i) rc_alloc = take3(i, rcv[i], rbase[i][QR], false);
            PSACX_TRY(agree(rc_alloc));
            // one stable partition of the block by class; the suffix a record stands for is made up on the way
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                const uint64_t cn = rec[i].cnt;
                if (!cn) return PSACX_OK;
                SortScratch sc;
                auto layout = [&](Arena& ar) { sc.d_base = ar.take<unsigned long long>((size_t)RADIX); sc.desc_bytes = sort_desc_bytes(cn); sc.d_desc = ar.take<char>(sc.desc_bytes); };
                { Arena dry(nullptr); layout(dry); MG_OP(g, c, ensure_slab(c, dry.off + 4096)); }
                Arena ar(c->slab);
                layout(ar);
                uint64_t sq, snq, vq;
                payload_of(i, 0, &sq, &snq, &vq);
                MG_HIP(g, hipSetDevice(c->device));
                MG_OP(g, c, piece_partition<T>(c, sc.d_desc, sc.d_base, rec[i].k1.p, cls[i].p, cn, grp[i].k1.p, grp[i].v.p, v32, sq, snq, vq));
                return PSACX_OK;
            }));
            // the unpartitioned records are not needed any more: in the reduced-memory layout they sat in the rank's output arrays,
            // which now serve as the second record set of the range sorts
            for (int i = 0; i < L; ++i) { (void)hipSetDevice(ctx(i)->device); (void)hipStreamSynchronize(ctx(i)->stream); drop3(i, rec[i]); cls[i].release(); }
            std::vector<Rec<T>> alt(L);
            for (int i = 0; i < L && rc_alloc == PSACX_OK; ++i) rc_alloc = take3(i, alt[i], rbase[i][QR], false);
            PSACX_TRY(agree(rc_alloc));
            mark("    sort: classify + partition");
            // range q of every destination travels in exchanges 2 q (word 1) and 2 q + 1 (suffixes): all issued now, in order, on
            // the second streams.  The narrow suffix entries of range q land at the start of the range's own word-sized region,
            // so that the sort of an earlier range, which widens its entries in place, never touches a later range's input.
            std::vector<std::vector<hipEvent_t>> done(2 * QR, std::vector<hipEvent_t>(L, nullptr));
            auto drop_events = [&]() { for (auto& v : done) for (int i = 0; i < L; ++i) if (v[i]) { (void)hipSetDevice(ctx(i)->device); (void)hipEventDestroy(v[i]); v[i] = nullptr; } };
            int rc = PSACX_OK;
            for (int q = 0; q < 2 * QR && rc == PSACX_OK; ++q) for (int i = 0; i < L && rc == PSACX_OK; ++i)
                if (hipSetDevice(ctx(i)->device) != hipSuccess || hipEventCreateWithFlags(&done[q][i], hipEventDisableTiming) != hipSuccess) { mg_set_err(g, "two-word first sort: event creation failed"); rc = PSACX_EHIP; }
            const uint64_t wide = sizeof(T) / vb;                // narrow entries per word
            for (int q = 0; q < QR && rc == PSACX_OK; ++q) {
                for (int arr = 0; arr < 2 && rc == PSACX_OK; ++arr) {
                    std::vector<std::vector<Msg>> sends(L), recvs(L);
                    std::vector<std::vector<const void*>> in(L);
                    std::vector<std::vector<void*>> out(L);
                    for (int i = 0; i < L; ++i) {
                        const int me = rank(i);
                        for (int d = 0; d < P; ++d) sends[i].push_back(Msg{d, soff[i][(size_t)d * QR + q], cnt_c[i][(size_t)d * QR + q]});
                        uint64_t within = 0;
                        for (int r = 0; r < P; ++r) {
                            const uint64_t cn = table[(size_t)r * NC + me * QR + q];
                            recvs[i].push_back(Msg{r, (arr == 0 ? rbase[i][q] : rbase[i][q] * wide) + within, cn});
                            within += cn;
                        }
                        if (arr == 0) { in[i] = {grp[i].k1.p}; out[i] = {rcv[i].k1.p}; }
                        else { in[i] = {grp[i].v.p}; out[i] = {rcv[i].v.p}; }
                    }
                    rc = transfer(in, out, {arr == 0 ? sizeof(T) : vb}, sends, recvs, &done[2 * q + arr]);
                }
            }
            // the ranges, one after the other, as they arrive (a rank whose sort fails still waits for its messages and tells its peers:
            // every path below runs the waits, drops the events and agrees on the outcome)
            std::vector<std::vector<int32_t>> where(L, std::vector<int32_t>(QR, 0));
            for (int q = 0; q < QR && rc == PSACX_OK; ++q) {
                rc = (par([&](int i) -> int {
                    psacx_ctx* c = ctx(i);
                    MG_HIP(g, hipSetDevice(c->device));
                    for (int s2 = 0; s2 < L; ++s2) { MG_HIP(g, hipStreamWaitEvent(c->stream, done[2 * q][s2], 0)); MG_HIP(g, hipStreamWaitEvent(c->stream, done[2 * q + 1][s2], 0)); }
                    const uint64_t b0 = rbase[i][q], tq = rbase[i][q + 1] - b0;
                    if (!tq) return PSACX_OK;
                    MG_OP(g, c, op_pair_sort<T>(c, rcv[i].k1.p + b0, (T*)nullptr, rcv[i].v.p + b0, alt[i].k1.p + b0, (T*)nullptr, alt[i].v.p + b0, tq, bits1, 0, &where[i][q], lo1,
                                                false, 0, 0, v32));
                    return PSACX_OK;
                }));
            }
            // everything has arrived (and, with ranks in one process, has been pulled) before the partitioned copies go away
            for (int i = 0; i < L; ++i) {
                (void)hipSetDevice(ctx(i)->device);
                for (int q = 0; q < 2 * QR; ++q) for (int s2 = 0; s2 < L; ++s2) if (done[q][s2]) (void)hipStreamWaitEvent(ctx(i)->stream, done[q][s2], 0);
            }
            for (int i = 0; i < L; ++i) { (void)hipSetDevice(ctx(i)->device); (void)hipStreamSynchronize(ctx(i)->stream); }
            drop_events();
            PSACX_TRY(agree(rc));
            // the sorted ranges into one record set (a sort's result lies in the set its last executed pass wrote)
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_HIP(g, hipSetDevice(c->device));
                int in_alt = 0;
                for (int q = 0; q < QR; ++q) in_alt += where[i][q] != 0;
                const bool to_alt = in_alt * 2 > QR;
                for (int q = 0; q < QR; ++q) {
                    const uint64_t b0 = rbase[i][q], tq = rbase[i][q + 1] - b0;
                    if (!tq || (where[i][q] != 0) == to_alt) continue;
                    Rec<T>& from = to_alt ? rcv[i] : alt[i]; Rec<T>& to = to_alt ? alt[i] : rcv[i];
                    MG_HIP(g, hipMemcpyAsync(to.k1.p + b0, from.k1.p + b0, tq * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                    MG_HIP(g, hipMemcpyAsync(to.v.p + b0, from.v.p + b0, tq * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                }
                MG_HIP(g, hipStreamSynchronize(c->stream));
                drop3(i, grp[i]);
                if (to_alt) { drop3(i, rcv[i]); rec[i] = std::move(alt[i]); } else { drop3(i, alt[i]); rec[i] = std::move(rcv[i]); }
                rec[i].cnt = rbase[i][QR];
                return PSACX_OK;
            }));
            sorted_already = true;
            mark("    sort: shuffle by ranges + range sorts");
        }
        // prefix sort of (word 1, suffix) on the leading bits, then the ties
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            if (rec[i].cnt >= 1 && !sorted_already) {
                Rec<T> alt;
                PSACX_TRY(take3(i, alt, rec[i].cnt, false));
                int32_t where = 0;
                if (solo_) {
                    // the first pass makes up the payload (the suffix a record stands for), as on one GPU
                    MG_OP(g, c, op_pair_sort<T>(c, rec[i].k1.p, (T*)nullptr, rec[i].v.p, alt.k1.p, (T*)nullptr, alt.v.p, rec[i].cnt, bits1, 0, &where, lo1, true, spec_front, n));
                } else MG_OP(g, c, op_pair_sort<T>(c, rec[i].k1.p, (T*)nullptr, rec[i].v.p, alt.k1.p, (T*)nullptr, alt.v.p, rec[i].cnt, bits1, 0, &where, lo1, false, 0, 0, v32));
                if (where) swap3(rec[i], alt);
                drop3(i, alt);
            }
            return PSACX_OK;
        }));
        return first_sort_ties(rec, targets, bits1, bits2, lo1, tbuf, two_k, tab, ks, false);
    }

    // Stage 2 of a first round that sorted (word 1, suffix) on the leading bits of word 1 only (rec[i]: k1, v sorted; word 1 may have lost the
    // bits below the prefix: word1_gone): the suffixes that still tie are ordered by their full windows -- one rank with the text at hand: in
    // place (tie_resolve_kernel); else compacted, their windows fetched from the ranks that own the text (dist_windows), ordered and written
    // back -- and the records re-balanced to the block sizes.
    int first_sort_ties(std::vector<Rec<T>>& rec, const std::vector<uint64_t>& targets, unsigned bits1, unsigned bits2, unsigned lo1,
                        const std::vector<DBuf<uint8_t>>& tbuf, uint32_t two_k, const CodeTable& tab, const KeyShape& ks, bool word1_gone) {
        std::vector<uint64_t> ties(L, 0);
        bool general_ties = !solo_;
        const bool solo_packed = word1_gone;
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            PSACX_TRY(need_k2(i, rec[i]));
            if (solo_ && rec[i].cnt) {
                // one rank: the text is here, every tie group of at most 8 suffixes is ordered in place (tie_resolve_kernel, construct.hpp)
                constexpr int TB = 256, TI = sizeof(T) == 8 ? 32 : 16, TG = 8;
                DBuf<unsigned long long> big; MG_OP(g, c, big.alloc(c, 1));
                MG_HIP(g, hipSetDevice(c->device));
                MG_HIP(g, hipMemsetAsync(big.p, 0, 8, c->stream));
                const uint64_t nb = (rec[i].cnt + (uint64_t)TB * TI - 1) / ((uint64_t)TB * TI);
                hipLaunchKernelGGL((tie_resolve_kernel<T, TB, TI, TG>), dim3((unsigned)nb), dim3(TB), 0, c->stream, rec[i].k1.p, rec[i].v.p, rec[i].k2.p, rec[i].cnt, lo1,
                                   (const uint8_t*)tbuf[i].p, S[i].m + two_k, tab, ks, big.p, solo_packed);
                MG_HIP(g, hipGetLastError());
                MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, big.p, 8, hipMemcpyDeviceToHost, c->stream));
                MG_HIP(g, hipStreamSynchronize(c->stream));
                if (*reinterpret_cast<unsigned long long*>(c->pinned + 32768)) general_ties = true;
            }
            if (general_ties) MG_OP(g, c, op_compact_ties<T>(c, rec[i].k1.p, rec[i].v.p, rec[i].cnt, lo1, (T*)nullptr, (T*)nullptr, (T*)nullptr, &ties[i]));
            return PSACX_OK;
        }));
        mark("    sort: local prefix sort");
        if (!general_ties) { mark("    sort: ties"); return head_.empty() ? rebalance(rec, targets) : rebalance_in_place(rec, targets); }
        // Reduced-memory layout: the compacted ties, their windows and the second record set of their sort are eight arrays of as many
        // entries as there are ties -- on a repetitive text every suffix ties.  The records are then worked off in slabs of at most
        // `cap` records that end where a group of equal prefixes ends (groups are independent of each other; a group longer than a
        // slab is taken whole): the same steps on fewer records, every rank as many slabs as the one with the most.
        uint64_t cap = 0;
        if (diet && slab_cap) {
            std::vector<uint64_t> all;
            PSACX_TRY(gather1(ties, all));
            const uint64_t tcap = std::max<uint64_t>(slab_cap / 2, 64);
            for (uint64_t t : all) if (t > tcap) cap = tcap;
        }
        std::vector<uint64_t> at(L, 0), end(L, 0), tn(L, 0);
        for (;;) {
            if (!cap) for (int i = 0; i < L; ++i) { end[i] = rec[i].cnt; tn[i] = ties[i]; }
            else PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                const uint64_t cnt = rec[i].cnt;
                end[i] = cnt; tn[i] = 0;
                if (at[i] >= cnt) return PSACX_OK;
                if (cnt - at[i] > cap) {
                    DBuf<unsigned long long> cut; MG_OP(g, c, cut.alloc(c, 2));
                    unsigned long long* h = reinterpret_cast<unsigned long long*>(c->pinned + 32768);
                    MG_HIP(g, hipSetDevice(c->device));
                    auto ask = [&](uint64_t lo, uint64_t hi) -> int {
                        h[0] = 0; h[1] = ~0ull;
                        MG_HIP(g, hipMemcpyAsync(cut.p, h, 16, hipMemcpyHostToDevice, c->stream));
                        hipLaunchKernelGGL((prefix_cut_kernel<T>), dim3(grid_for(c, hi - lo, 256, 8)), dim3(256), 0, c->stream, (const T*)rec[i].k1.p, lo, hi, lo1, cut.p, cut.p + 1);
                        MG_HIP(g, hipGetLastError());
                        MG_HIP(g, hipMemcpyAsync(h, cut.p, 16, hipMemcpyDeviceToHost, c->stream));
                        MG_HIP(g, hipStreamSynchronize(c->stream));
                        return PSACX_OK;
                    };
                    PSACX_TRY(ask(at[i] + 1, at[i] + cap + 1));              // the last group start inside the slab ...
                    if (h[0]) end[i] = h[0];
                    else {                                                   // ... or, a group longer than the slab, the end of that group
                        PSACX_TRY(ask(at[i] + cap + 1, cnt));
                        if (h[1] != ~0ull) end[i] = h[1];
                    }
                }
                MG_OP(g, c, op_compact_ties<T>(c, rec[i].k1.p + at[i], rec[i].v.p + at[i], end[i] - at[i], lo1, (T*)nullptr, (T*)nullptr, (T*)nullptr, &tn[i]));
                return PSACX_OK;
            }));
            std::vector<DBuf<T>> tpos(L), tk1(L), tv(L), w1, w2;
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, tpos[i].alloc(c, tn[i])); MG_OP(g, c, tk1[i].alloc(c, tn[i])); MG_OP(g, c, tv[i].alloc(c, tn[i]));
                if (tn[i]) { uint64_t chk = 0; MG_OP(g, c, op_compact_ties<T>(c, rec[i].k1.p + at[i], rec[i].v.p + at[i], end[i] - at[i], lo1, tpos[i].p, tk1[i].p, tv[i].p, &chk)); }
                return PSACX_OK;
            }));
            {
                std::vector<const T*> q(L);
                for (int i = 0; i < L; ++i) q[i] = tv[i].p;
                PSACX_TRY(dist_windows(tbuf, two_k, tab, ks, q, tn, w1, w2));
            }
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                const uint64_t n_t = tn[i];
                if (!n_t) return PSACX_OK;
                MG_HIP(g, hipSetDevice(c->device));
                // every group is at most TG long: ordered in registers (tie_resolve_kernel reading both words from the arrays)
                constexpr int TB_ = 256, TI_ = 16, TG_ = 8;
                DBuf<unsigned long long> big; MG_OP(g, c, big.alloc(c, 1));
                MG_HIP(g, hipMemsetAsync(big.p, 0, 8, c->stream));
                const uint64_t nb = (n_t + (uint64_t)TB_ * TI_ - 1) / ((uint64_t)TB_ * TI_);
                hipLaunchKernelGGL((tie_resolve_kernel<T, TB_, TI_, TG_, true>), dim3((unsigned)nb), dim3(TB_), 0, c->stream, w1[i].p, tv[i].p, w2[i].p, n_t, lo1,
                                   (const uint8_t*)nullptr, (uint64_t)0, tab, ks, big.p);
                MG_HIP(g, hipGetLastError());
                MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, big.p, 8, hipMemcpyDeviceToHost, c->stream));
                MG_HIP(g, hipStreamSynchronize(c->stream));
                const T *s1 = w1[i].p, *s2 = w2[i].p, *sv = tv[i].p;
                DBuf<T> b1, b2, bv;
                if (*reinterpret_cast<unsigned long long*>(c->pinned + 32768)) {
                    // some group is long (repetitive text): a stable sort of all tied records by the full window; the groups come in
                    // ascending order of their prefix, so the sorted records go back to the same positions in order
                    tk1[i].release();                                    // (word 1 of the ties came back with the windows)
                    MG_OP(g, c, b1.alloc(c, n_t)); MG_OP(g, c, b2.alloc(c, n_t)); MG_OP(g, c, bv.alloc(c, n_t));
                    int32_t where = 0;
                    MG_OP(g, c, op_pair_sort<T>(c, w1[i].p, w2[i].p, tv[i].p, b1.p, b2.p, bv.p, n_t, bits1, bits2, &where));
                    if (where) { s1 = b1.p; s2 = b2.p; sv = bv.p; }
                }
                hipLaunchKernelGGL((scatter_prefix_ties_kernel<T>), dim3(grid_for(c, n_t, 256, 16)), dim3(256), 0, c->stream, (const T*)tpos[i].p, n_t, s1, s2, sv,
                                   rec[i].k1.p + at[i], rec[i].k2.p + at[i], rec[i].v.p + at[i]);
                MG_HIP(g, hipGetLastError());
                MG_HIP(g, hipStreamSynchronize(c->stream));          // (the compacted arrays go back to the cache when this scope ends)
                return PSACX_OK;
            }));
            if (!cap) break;
            std::vector<uint64_t> left(L), left_all;
            for (int i = 0; i < L; ++i) { at[i] = end[i]; left[i] = rec[i].cnt - at[i]; }
            PSACX_TRY(gather1(left, left_all));
            bool more = false;
            for (uint64_t x : left_all) more |= x != 0;
            if (!more) break;
            ++g->last_tie_slabs;
        }
        mark("    sort: ties");
        return head_.empty() ? rebalance(rec, targets) : rebalance_in_place(rec, targets);
    }

    // The first sort in ONE-word records (the one-GPU engine's prefix_sort_1w, engine.hpp, spread over the ranks).  A record is
    // (prefix of word 1 without its top digit) << sfield | suffix; the top digit is known from the record's place:
    //   1. every rank counts the top digits of its block straight from the text (top_digit_hist_kernel); one all-gather of the 256 counts
    //      gives every rank the exact size of every bucket on every rank -- no samples, no splitters;
    //   2. the 256 buckets are dealt to the ranks in order, whole, so that every rank's share is as close to its block as whole buckets
    //      allow (equal prefixes never part; the text's own distribution decides the balance: a text whose buckets cannot be dealt
    //      within the slack of the record arrays takes the two-word path with its sampled splitters);
    //   3. the pass on the top digit computes word 1 in registers and writes the one-word records bucket by bucket
    //      (key_scatter1w_kernel): 1 byte read + 8 written per record, nothing else is ever written on the sender;
    //   4. the buckets travel in QR groups per destination, each bucket's pieces from all senders landing back to back; a group is
    //      complete when it has landed and its LSD passes (8 + 8 bytes per record and pass, radix_scatter1w_kernel) run on the compute
    //      stream while the later groups are still in flight; the last pass writes word 1 and the suffixes as words.
    // The suffixes shorter than 2k (the last 2k - 1 positions of the text) are made on the host -- every rank knows the tail of the text
    // from the gather -- and placed at the head of their buckets, where the stable passes keep them in front of equal prefixes.
    // Needs 64-bit words and n <= 2^34 (the payload field takes bits_for(n - 1) bits, the prefix the rest + 8: fewer than 1/16 of the suffixes
    // of a random text tie).  Returns PSACX_RETRY_ before anything has moved.  *lo1_out: bits of word 1 below the sorted prefix.
    int sort_first_one_word(std::vector<Rec<T>>& rec, const std::vector<uint64_t>& targets, unsigned bits1, unsigned bits2,
                            const std::vector<DBuf<uint8_t>>& tbuf, uint32_t two_k, const CodeTable& tab, const KeyShape& ks, bool trust, uint64_t spec,
                            unsigned* lo1_out) {
        if constexpr (sizeof(T) != 8) { return PSACX_RETRY_; }
        else {
        constexpr int BLOCK = 512, ITEMS = 8, TILE0 = BLOCK * ITEMS, TILE = BLOCK * PSACX_1W_ITEMS;
        constexpr int TAILB = 128;                                   // bytes of every block's end that travel with the counts (2k <= 128)
        const unsigned nbits = bits_for(n - 1);
        if (bits1 < 24 || nbits > 40) return PSACX_RETRY_;
        // prefix bits that stay in the word: what the one-GPU rule asks for (bits_for(n - 1) + 3 leading bits, whole digits) as far as the word has room
        const unsigned want_lead = (nbits + 3 + RADIX_BITS - 1) / RADIX_BITS * RADIX_BITS;
        const unsigned low = std::min(std::min(64u - nbits, bits1 - (unsigned)RADIX_BITS), want_lead - (unsigned)RADIX_BITS);
        const unsigned lead = low + RADIX_BITS, sfield = 64 - low, lo1 = bits1 - lead;
        // Reduced-memory layout: a text that repeats itself, or one so long that few prefix bits fit beside the suffix (beyond 2^34 characters),
        // stays in one-word records -- its many ties are ordered slab by slab (first_sort_ties), while the three-word records of the
        // other forms would not fit the device at all (8.25 words per character against 3)
        const bool ties_ok = trust || diet;
        if (lead < nbits + 3 && !ties_ok) return PSACX_RETRY_;      // (too many suffixes would tie on the prefix)
        if (lead < nbits + 1) return PSACX_RETRY_;
        uint64_t min_m = sizes[0];
        for (int r = 1; r < P; ++r) min_m = std::min(min_m, sizes[r]);
        if (min_m < (uint64_t)TAILB || min_m < 2ull * two_k) return PSACX_RETRY_;
        ++sort_calls_;
        KeyShape ks0 = ks; ks0.spec = solo_ ? spec : 0;               // (one rank without the wire: the short suffixes are records of the kernel, as on one GPU)
        // ---- 1. top digits of every block
        std::vector<uint64_t> nrec(L), short_n(L);
        std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(RADIX + 2 + TAILB / 8, 0));
        struct Scr { unsigned long long* base0; char* desc; unsigned* tile_hist0; unsigned long long* slab_tot0; uint64_t ntiles; unsigned slab0; size_t desc_bytes; };
        std::vector<Scr> scr(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            const uint64_t m = S[i].m, end = S[i].off + m, first_short = n - spec;
            short_n[i] = solo_ ? 0 : std::min<uint64_t>(m, end > first_short ? end - first_short : 0);
            nrec[i] = m - short_n[i];
            Scr& q = scr[i];
            q.ntiles = (nrec[i] + TILE0 - 1) / TILE0;
            q.slab0 = slab_tiles_for(q.ntiles);
            // the scratch of the bucket passes on the receiving side lives in the same slab: sized now for the largest share a rank may accept
            const uint64_t cap_rec = m + m / 8 + 256 + (uint64_t)TILE;
            const uint64_t vt_ub = (cap_rec + TILE - 1) / TILE + (uint64_t)RADIX * 64 + 64;
            const size_t need_b = 256 + (((size_t)vt_ub * RADIX * sizeof(unsigned) + 255) & ~(size_t)255) + (((size_t)(vt_ub / 16 + RADIX) * RADIX * 8 + 255) & ~(size_t)255) +
                                  (size_t)RADIX * RADIX * 8 + 2 * (RADIX + 1) * 8 + 64 + (size_t)(vt_ub / 16 + RADIX) * sizeof(SlabInfo) + 4096;
            const uint64_t stride = std::max<uint64_t>(64, m >> 20), samples = m / stride;
            uint64_t slots = 1; while (slots < 4 * samples) slots <<= 1;
            const size_t need_a = 256 + std::max<size_t>((((size_t)q.ntiles * RADIX * sizeof(unsigned) + 255) & ~(size_t)255) + (q.ntiles / q.slab0 + 2) * RADIX * 8, slots * 8) + 4096;
            q.desc_bytes = std::max(need_a, need_b);
            MG_OP(g, c, ensure_slab(c, q.desc_bytes + (size_t)RADIX * 8 + 8192));
            MG_OP(g, c, ensure_pinned(c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
            Arena ar(c->slab);
            q.base0 = ar.take<unsigned long long>((size_t)RADIX);
            q.desc = ar.take<char>(q.desc_bytes);
            q.tile_hist0 = reinterpret_cast<unsigned*>(q.desc + 256);
            q.slab_tot0 = reinterpret_cast<unsigned long long*>(q.desc + 256 + (((size_t)q.ntiles * RADIX * sizeof(unsigned) + 255) & ~(size_t)255));
            MG_HIP(g, hipSetDevice(c->device));
            unsigned long long* h = reinterpret_cast<unsigned long long*>(c->pinned + 32768);
            h[RADIX] = 0; h[RADIX + 1] = 0;
            if (samples >= 1024 && !ties_ok) {
                // does the block repeat itself massively?  (prefix_dup_probe_kernel, sa_kernels.hpp: such a text keeps the two-word path)
                unsigned long long* table = reinterpret_cast<unsigned long long*>(q.desc + 256);
                unsigned long long* d_dups = reinterpret_cast<unsigned long long*>(q.desc + 128);
                MG_HIP(g, hipMemsetAsync(q.desc, 0, 256 + slots * 8, c->stream));
                hipLaunchKernelGGL((prefix_dup_probe_kernel<uint64_t>), dim3((unsigned)((samples + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t*)tbuf[i].p, m + two_k, tab, ks0, lo1,
                                   stride, samples, table, slots, d_dups);
                MG_HIP(g, hipGetLastError());
                MG_HIP(g, hipMemcpyAsync(h + RADIX, d_dups, 8, hipMemcpyDeviceToHost, c->stream));
                MG_HIP(g, hipStreamSynchronize(c->stream));
                h[RADIX + 1] = samples;
            }
            if (q.ntiles) {
                hipLaunchKernelGGL((top_digit_hist_kernel<uint64_t, BLOCK, ITEMS>), dim3((unsigned)q.ntiles), dim3(BLOCK), 0, c->stream, (const uint8_t*)tbuf[i].p, solo_ ? m : nrec[i],
                                   m + two_k, tab, ks0, q.tile_hist0);
                const uint64_t nslabs0 = (q.ntiles + q.slab0 - 1) / q.slab0;
                hipLaunchKernelGGL(radix_slab_scan_kernel<0>, dim3((unsigned)nslabs0), dim3(RADIX), 0, c->stream, q.tile_hist0, q.ntiles, q.slab_tot0, q.slab0);
                hipLaunchKernelGGL(radix_top_scan_kernel<0>, dim3(1), dim3(RADIX), 0, c->stream, q.slab_tot0, nslabs0, q.base0);
                MG_HIP(g, hipGetLastError());
                MG_HIP(g, hipMemcpyAsync(h, q.base0, RADIX * 8, hipMemcpyDeviceToHost, c->stream));
            } else std::memset(h, 0, RADIX * 8);
            MG_HIP(g, hipMemcpyAsync(h + RADIX + 2, tbuf[i].p + m - TAILB, TAILB, hipMemcpyDeviceToHost, c->stream));
            MG_HIP(g, hipStreamSynchronize(c->stream));
            const uint64_t total = solo_ ? m : nrec[i];
            for (int d = 0; d < RADIX; ++d) mine[i][d] = (d + 1 < RADIX ? h[d + 1] : total) - h[d];       // bucket sizes (the starts are their prefix sums)
            for (int w = RADIX; w < RADIX + 2 + TAILB / 8; ++w) mine[i][w] = h[w];
            return PSACX_OK;
        }));
        std::vector<uint64_t> table;                                 // table[r * W + b]
        const int W = RADIX + 2 + TAILB / 8;
        PSACX_TRY(gather(W, mine, table));
        // ---- 2. the short suffixes (host), the buckets' sizes, their owners
        {
            uint64_t dups = 0, smp = 0;
            for (int r = 0; r < P; ++r) { dups += table[(size_t)r * W + RADIX]; smp += table[(size_t)r * W + RADIX + 1]; }
            if (!ties_ok && smp && dups * 8 > smp) return PSACX_RETRY_;
        }
        std::vector<std::vector<uint64_t>> short_words(RADIX);
        if (!solo_ && spec) {
            const uint8_t* tail = reinterpret_cast<const uint8_t*>(&table[(size_t)(P - 1) * W + RADIX + 2]);       // text[n - TAILB .. n)
            for (uint64_t j = 0; j < spec; ++j) {                    // suffix n - 1 - j, j + 1 characters long: shortest first
                const uint64_t pos = n - 1 - j;
                uint64_t w1 = 0;
                for (unsigned t = 0; t < ks.c1; ++t) {
                    const uint64_t code = pos + t < n ? (uint64_t)tab.c[tail[(size_t)TAILB - 1 - j + t]] : 0ull;
                    w1 = (ks.lc >= 64 ? 0ull : (w1 << ks.lc)) | code;
                }
                const uint64_t prefix = lo1 >= 64 ? 0ull : (w1 >> lo1);
                short_words[(size_t)((prefix >> low) & (RADIX - 1))].push_back((prefix << sfield) | pos);
            }
        }
        std::vector<uint64_t> tot(RADIX, 0), PT(RADIX + 1, 0);
        for (int b = 0; b < RADIX; ++b) {
            tot[b] = short_words[b].size();
            for (int r = 0; r < P; ++r) tot[b] += table[(size_t)r * W + b];
            PT[b + 1] = PT[b] + tot[b];
        }
        if (PT[RADIX] != n) { mg_set_err(g, "one-word first sort: the top-digit counts do not add up to the text"); return PSACX_EDEVICE; }
        const std::vector<uint64_t> TP = prefix_of(targets);
        std::vector<int> cut(P + 1, 0);                              // rank d owns the buckets cut[d] .. cut[d + 1] - 1
        cut[P] = RADIX;
        // rank d starts at the first bucket boundary at or behind the start of its block: every rank then holds a little more than the tail
        // of its own block -- the head, at most one bucket, sits at the end of the rank before it and is received in front of the rank's own
        // records, for which the arrays leave room (head_ / room_; rebalance_in_place): no copy of the record arrays to re-balance them
        for (int d = 1; d < P; ++d) {
            int b = cut[d - 1];
            while (b < RADIX && PT[b] < TP[d]) ++b;
            cut[d] = b;
        }
        std::vector<uint64_t> Gs(P), cs(P), Hs(P), rooms(P);
        bool inplace = !solo_;
        for (int d = 0; d < P; ++d) {
            Gs[d] = PT[cut[d]]; cs[d] = PT[cut[d + 1]] - PT[cut[d]];
            Hs[d] = Gs[d] - TP[d]; rooms[d] = std::max(Hs[d] + cs[d], sizes[d]);
            if (rooms[d] > sizes[d] + sizes[d] / 8) {                 // (the slack of the reduced-memory layout's record arrays)
                if (!trust) return PSACX_RETRY_;
                inplace = false;
            }
        }
        if (!inplace) for (int d = 0; d < P; ++d) { Hs[d] = 0; rooms[d] = cs[d]; }
        *lo1_out = lo1;
        // ---- 3. arrays: the partitioned block (grp), two record arrays of the rank's share (A, B) and the suffixes of the last pass (vout).
        //      Reduced-memory layout: grp, the array that does not end up with word 1 and the suffixes are the rank's three output arrays.
        const int npass = (int)((low + RADIX_BITS - 1) / RADIX_BITS);
        std::vector<DBuf<T>> grp(L), A(L), B(L), vout(L);
        std::vector<uint64_t> share(L);
        int rc_alloc = PSACX_OK;
        for (int i = 0; i < L && rc_alloc == PSACX_OK; ++i) {
            psacx_ctx* c = ctx(i);
            const int me = rank(i);
            drop3(i, rec[i]);
            share[i] = cs[me];
            const uint64_t ng = solo_ ? S[i].m : nrec[i], room = rooms[me];
            const bool lend = diet && !S[i].out_busy && std::max(room, ng) <= S[i].out_cap;
            DBuf<T>& k1_final = (npass & 1) ? B[i] : A[i];          // the array the last pass writes word 1 into
            DBuf<T>& other = (npass & 1) ? A[i] : B[i];
            if (lend) {
                S[i].out_busy = true;
                other.borrow(c, S[i].ISA, room);
                vout[i].borrow(c, S[i].SA, room);
                if (S[i].LCP && !solo_) grp[i].borrow(c, S[i].LCP, ng);
            } else {
                rc_alloc = other.alloc(c, room, reserve_of(i));
                if (rc_alloc == PSACX_OK) rc_alloc = vout[i].alloc(c, room, reserve_of(i));
            }
            if (rc_alloc == PSACX_OK) rc_alloc = k1_final.alloc(c, room, reserve_of(i));
            if (rc_alloc == PSACX_OK && !solo_ && !grp[i].p) rc_alloc = grp[i].alloc(c, ng, reserve_of(i));
            if (rc_alloc != PSACX_OK) mg_set_err(g, "one-word first sort: record arrays: " + c->hip_err);
        }
        PSACX_TRY(agree(rc_alloc));
        // ---- 4. the pass on the top digit, word 1 computed on the spot (one rank without the wire: straight into A)
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            Scr& q = scr[i];
            if (!q.ntiles) return PSACX_OK;
            MG_HIP(g, hipSetDevice(c->device));
            MG_HIP(g, hipMemsetAsync(q.desc, 0, 256, c->stream));
            const uint64_t cnt = solo_ ? S[i].m : nrec[i];
            hipLaunchKernelGGL((key_scatter1w_kernel<BLOCK, ITEMS>), dim3((unsigned)q.ntiles), dim3(BLOCK), 0, c->stream, (const uint8_t*)tbuf[i].p, cnt, S[i].m + two_k, tab, ks0,
                               reinterpret_cast<uint64_t*>(solo_ ? A[i].p : grp[i].p), (int)(lo1 + low), q.base0, q.tile_hist0, q.slab_tot0, reinterpret_cast<unsigned*>(q.desc),
                               sort_chunk_for(cnt, true), q.slab0, lo1 | (sfield << 16), solo_ ? (uint64_t)0 : S[i].off);
            MG_HIP(g, hipGetLastError());
            return PSACX_OK;
        }));
        mark("    sort: keys + partition by the top digit");
        // ---- 5. where everything lands: bucket b of rank `me` = [short suffixes][sender 0] .. [sender P - 1]
        int QR = solo_ ? 1 : 4;
        if (pieces_env_ > 0) QR = std::max(1, std::min(16, pieces_env_));
        std::vector<std::vector<uint64_t>> boff(L, std::vector<uint64_t>(RADIX + 1, 0));       // start of bucket b in the rank's arrays
        std::vector<std::vector<uint64_t>> sstart(L, std::vector<uint64_t>(RADIX + 1, 0));     // start of bucket b in the sender's partitioned block
        for (int i = 0; i < L; ++i) {
            const int me = rank(i);
            uint64_t at = Hs[me];
            for (int b = 0; b <= RADIX; ++b) { boff[i][b] = at; if (b < RADIX && b >= cut[me] && b < cut[me + 1]) at += tot[b]; }
            for (int b = 0; b < RADIX; ++b) sstart[i][b + 1] = sstart[i][b] + mine[i][b];
        }
        // the buckets of a destination in QR ranges of about equal size (the same cuts on every rank: a sender must know the ranges of its destinations)
        auto range_cuts = [&](int d) -> std::vector<int> {
            const int nb = cut[d + 1] - cut[d];
            const int qr = std::max(1, std::min(QR, nb));
            const uint64_t sh = PT[cut[d + 1]] - PT[cut[d]];
            std::vector<int> rc(QR + 1, cut[d + 1]);
            rc[0] = cut[d];
            for (int q = 1; q < qr; ++q) {
                int b = rc[q - 1];
                const uint64_t want = PT[cut[d]] + (uint64_t)(((unsigned __int128)sh * q) / qr);
                while (b < cut[d + 1] && PT[b + 1] <= want) ++b;
                rc[q] = std::max(b, rc[q - 1]);
            }
            return rc;
        };
        std::vector<std::vector<int>> rcuts(P);
        for (int d = 0; d < P; ++d) rcuts[d] = range_cuts(d);
        std::vector<std::vector<hipEvent_t>> done(QR, std::vector<hipEvent_t>(L, nullptr));
        auto drop_events = [&]() { for (auto& v : done) for (int i = 0; i < L; ++i) if (v[i]) { (void)hipSetDevice(ctx(i)->device); (void)hipEventDestroy(v[i]); v[i] = nullptr; } };
        int rc = PSACX_OK;
        if (!solo_) {
            for (int q = 0; q < QR && rc == PSACX_OK; ++q) for (int i = 0; i < L && rc == PSACX_OK; ++i) {
                if (hipSetDevice(ctx(i)->device) != hipSuccess || hipEventCreateWithFlags(&done[q][i], hipEventDisableTiming) != hipSuccess) { mg_set_err(g, "one-word first sort: event creation failed"); rc = PSACX_EHIP; }
            }
            // the short suffixes at the head of their buckets (before the first exchange is issued: the copies are ordered on the compute streams,
            // which the range sorts wait on anyway)
            for (int i = 0; i < L && rc == PSACX_OK; ++i) {
                const int me = rank(i);
                (void)hipSetDevice(ctx(i)->device);
                for (int b = cut[me]; b < cut[me + 1] && rc == PSACX_OK; ++b)
                    if (!short_words[b].empty() && hipMemcpyAsync(A[i].p + boff[i][b], short_words[b].data(), short_words[b].size() * 8, hipMemcpyHostToDevice, ctx(i)->stream) != hipSuccess) {
                        mg_set_err(g, "one-word first sort: copy of the short suffixes failed"); rc = PSACX_EHIP;
                    }
            }
            // the messages from sender r to destination d in range q: one per bucket, neighbours joined where they are contiguous on both
            // sides (always on the sender's; on the receiver's when no other sender's records and no short suffix lie between them).  Sender
            // and receiver derive their lists from this one function.
            struct Piece { uint64_t soff, roff, cnt; };
            std::vector<std::vector<uint64_t>> bstart(P, std::vector<uint64_t>(RADIX + 1, 0));   // start of bucket b in rank d's arrays (as boff, for every rank)
            for (int d = 0; d < P; ++d) { uint64_t at = Hs[d]; for (int b = 0; b <= RADIX; ++b) { bstart[d][b] = at; if (b < RADIX && b >= cut[d] && b < cut[d + 1]) at += tot[b]; } }
            auto pieces = [&](int r, int d, int q) -> std::vector<Piece> {
                std::vector<Piece> out;
                uint64_t so = 0;
                for (int b = 0; b < rcuts[d][q]; ++b) so += table[(size_t)r * W + b];
                for (int b = rcuts[d][q]; b < rcuts[d][q + 1]; ++b) {
                    const uint64_t cn = table[(size_t)r * W + b];
                    uint64_t ro = bstart[d][b] + short_words[b].size();
                    for (int r2 = 0; r2 < r; ++r2) ro += table[(size_t)r2 * W + b];
                    if (cn) {
                        if (!out.empty() && out.back().soff + out.back().cnt == so && out.back().roff + out.back().cnt == ro) out.back().cnt += cn;
                        else out.push_back(Piece{so, ro, cn});
                    }
                    so += cn;
                }
                return out;
            };
            for (int q = 0; q < QR && rc == PSACX_OK; ++q) {
                std::vector<std::vector<Msg>> sends(L), recvs(L);
                std::vector<std::vector<const void*>> in(L);
                std::vector<std::vector<void*>> out(L);
                for (int i = 0; i < L; ++i) {
                    const int me = rank(i);
                    for (int d = 0; d < P; ++d) for (const Piece& pc : pieces(me, d, q)) sends[i].push_back(Msg{d, pc.soff, pc.cnt});
                    for (int r = 0; r < P; ++r) for (const Piece& pc : pieces(r, me, q)) recvs[i].push_back(Msg{r, pc.roff, pc.cnt});
                    in[i] = {grp[i].p}; out[i] = {A[i].p};
                }
                rc = transfer(in, out, {sizeof(T)}, sends, recvs, &done[q]);
            }
        }
        // ---- 6. the LSD passes inside the buckets of a range as soon as it has landed
        std::vector<std::vector<std::vector<unsigned long long>>> tabs(L, std::vector<std::vector<unsigned long long>>(QR));
        std::vector<uint64_t*> s1(L, nullptr);
        for (int q = 0; q < QR && rc == PSACX_OK; ++q) {
            rc = par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_HIP(g, hipSetDevice(c->device));
                if (!solo_) for (int s2 = 0; s2 < L; ++s2) MG_HIP(g, hipStreamWaitEvent(c->stream, done[q][s2], 0));
                std::vector<unsigned long long>& ht = tabs[i][q];
                ht.assign(2 * (RADIX + 1), 0);
                const int b0 = rcuts[rank(i)][q], b1 = rcuts[rank(i)][q + 1];
                uint64_t cntq = 0;
                for (int b = 0; b <= RADIX; ++b) ht[b] = boff[i][std::min(std::max(b, b0), b1)];
                cntq = ht[RADIX] - ht[0];
                if (!cntq) { if (!s1[i]) s1[i] = reinterpret_cast<uint64_t*>(((npass & 1) ? B[i] : A[i]).p); return PSACX_OK; }
                const OneWordLayout lay = onew_layout<TILE>(ht.data(), (share[i] + TILE - 1) / TILE);
                if (lay.need > scr[i].desc_bytes || lay.vtiles >= (1ull << 31)) { mg_set_err(g, "one-word first sort: scratch of the bucket passes too small"); return PSACX_EDEVICE; }
                uint64_t* res = nullptr;
                MG_OP(g, c, onew_bucket_passes(c, scr[i].desc, ht.data(), lay, reinterpret_cast<uint64_t*>(A[i].p), reinterpret_cast<uint64_t*>(B[i].p),
                                               reinterpret_cast<uint64_t*>(vout[i].p), sfield, low, lo1, cntq, &res));
                s1[i] = res;
                return PSACX_OK;
            });
        }
        // everything has arrived and every pass has run before the partitioned blocks and the tables go away
        for (int i = 0; i < L; ++i) {
            (void)hipSetDevice(ctx(i)->device);
            if (!solo_) for (int q = 0; q < QR; ++q) for (int s2 = 0; s2 < L; ++s2) if (done[q][s2]) (void)hipStreamWaitEvent(ctx(i)->stream, done[q][s2], 0);
        }
        for (int i = 0; i < L; ++i) { (void)hipSetDevice(ctx(i)->device); (void)hipStreamSynchronize(ctx(i)->stream); }
        drop_events();
        PSACX_TRY(agree(rc));
        for (int i = 0; i < L; ++i) {
            DBuf<T>& k1_final = (npass & 1) ? B[i] : A[i];
            if (s1[i] && reinterpret_cast<T*>(s1[i]) != k1_final.p) { mg_set_err(g, "one-word first sort: word 1 ended in the wrong array"); return PSACX_EDEVICE; }
            grp[i].release();
            ((npass & 1) ? A[i] : B[i]).release();
            rec[i] = Rec<T>();
            rec[i].k1 = std::move(k1_final); rec[i].v = std::move(vout[i]); rec[i].cnt = share[i];
            rec[i].k1.advance(Hs[rank(i)]); rec[i].v.advance(Hs[rank(i)]);
            rec[i].k1.n = share[i]; rec[i].v.n = share[i];
        }
        if (inplace) {
            head_.assign(L, 0); room_.assign(L, 0);
            for (int i = 0; i < L; ++i) { head_[i] = Hs[rank(i)]; room_[i] = rooms[rank(i)]; }
            held_from_ = Gs; held_cnt_ = cs;
        }
        g->last_one_word = true;
        mark("    sort: shuffle by buckets + bucket passes");
        const int rct = first_sort_ties(rec, targets, bits1, bits2, lo1, tbuf, two_k, tab, ks, true);
        head_.clear(); room_.clear();
        return rct;
        }
    }

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
        constexpr int PB = 512, PI = 16, TILE_BITS = 13;
        uint64_t max_m = 0;
        for (int r = 0; r < P; ++r) max_m = std::max(max_m, sizes[r]);
        const unsigned kb = bits_for(max_m > 1 ? max_m - 1 : 1);
        unsigned cap_bits = 0;
        while ((2u << cap_bits) * (unsigned)P <= (unsigned)SLICE_MAX_CLASSES) ++cap_bits;        // most slice bits with P * 2^bits classes
        unsigned wbmax = WBMAX;
        if (slice_wb_env_) wbmax = std::min<unsigned>(WBMAX, std::max(4u, slice_wb_env_));     // (tests: levels on small inputs)
        if (slice_s1_env_) cap_bits = std::min<unsigned>(cap_bits, slice_s1_env_);
        unsigned s1 = std::min<unsigned>(cap_bits, kb > wbmax ? kb - wbmax : 0);
        // a further level walks tiles of 2^13 pairs that must not straddle slices
        if (kb - s1 > wbmax && kb - s1 < (unsigned)TILE_BITS) s1 = kb > (unsigned)TILE_BITS ? kb - TILE_BITS : 0;
        const unsigned sb = kb - s1;
        const unsigned spo = (unsigned)((max_m + (1ull << sb) - 1) >> sb);
        const unsigned wb = std::min(sb, wbmax);
        const unsigned rbits = sb - wb;
        // levels of at most 9 bits each; a level's parent buckets (2^(shift + cb) pairs) must hold whole tiles, which only
        // binds the last level when a test shrinks the windows below a tile
        std::vector<unsigned> cbs;
        if (rbits) {
            const unsigned last_min = wb >= (unsigned)TILE_BITS ? 1u : std::min(rbits, (unsigned)TILE_BITS - wb);
            unsigned nl = (rbits + 8) / 9;
            cbs.assign(nl, 0);
            for (unsigned j = 0; j < nl; ++j) cbs[j] = rbits / nl + (j < rbits % nl ? 1 : 0);
            if (cbs.back() < last_min) {
                const unsigned rest = rbits - last_min;
                nl = 1 + (rest + 8) / 9;
                cbs.assign(nl, 0);
                for (unsigned j = 0; j + 1 < nl; ++j) cbs[j] = rest / (nl - 1) + (j < rest % (nl - 1) ? 1 : 0);
                cbs.back() = last_min;
            }
        }
        const unsigned levels2 = (unsigned)cbs.size();
        const unsigned C = (unsigned)P * spo;
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
        uint64_t G = spo;
        if (diet) G = std::max<uint64_t>(1, std::max<uint64_t>(slice, max_m / 8) >> sb);     // (one rank without the wire: the levels work on a step's part of the class array)
        if (slice_step_env_) G = slice_step_env_;
        G = std::min<uint64_t>(G, spo);
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
        if (sizeof(T) == 4 || (n <= (1ull << 32) && !getenv("PSACX_SLICE_WIDE"))) return isa_by_slices_t<uint32_t>(ids_in_isa);      // (PSACX_SLICE_WIDE: tests)
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

    // min(LCP[lo .. hi)) over the block-distributed LCP array for every query (bulk_rmq_v2, par_rmq.hpp:199-332)
    int dist_range_min(const std::vector<const T*>& lo, const std::vector<const T*>& hi, const std::vector<uint64_t>& cnt,
                       std::vector<DBuf<T>>& out) {
        out.clear(); out.resize(L);
        if (solo_) {
            MG_OP(g, ctx(0), out[0].alloc(ctx(0), cnt[0]));
            MG_OP(g, ctx(0), op_range_min<T>(ctx(0), S[0].LCP, S[0].m, lo[0], hi[0], cnt[0], S[0].off, out[0].p));
            return PSACX_OK;
        }
        // one min-pyramid of every rank's LCP block serves its block minimum and both batches of sub-queries
        std::vector<uint64_t> bm(L), mins;
        std::vector<Pyramid<T>> pyr(L);
        std::vector<DBuf<T>> pyr_mem(L);
        for (int i = 0; i < L; ++i) PSACX_TRY(block_pyramid(i, pyr[i], pyr_mem[i], &bm[i]));
        PSACX_TRY(gather1(bm, mins));
        // own1/lo1/hi1: the part inside the rank of lo; own2/lo2/hi2: the part inside the rank of hi - 1; ra/rb: whole ranks between
        std::vector<std::vector<DBuf<T>>> parts(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            parts[i].resize(8);
            for (int q = 0; q < 8; ++q) MG_OP(g, c, parts[i][q].alloc(c, cnt[i]));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (rmq_split_kernel<T>), cnt[i], lo[i], hi[i], cnt[i], make_dist(n, (unsigned)P), parts[i][0].p, parts[i][1].p, parts[i][2].p,
                          parts[i][3].p, parts[i][4].p, parts[i][5].p, parts[i][6].p, parts[i][7].p);
            return PSACX_OK;
        }));
        std::vector<std::vector<DBuf<T>>> answers(2);
        for (int half = 0; half < 2; ++half) {
            std::vector<Rec<T>> ra(L), rb(L);
            std::vector<std::vector<uint64_t>> bounds(L), b2(L), rc, rc2;
            std::vector<std::vector<const T*>> in(L);
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                const T* a = parts[i][3 * half + 1].p; const T* b = parts[i][3 * half + 2].p;
                // route by the owner of the sub-range's lower end: (a, b) and (a, slot) through the same stable pass
                DBuf<T> slot; MG_OP(g, c, slot.alloc(c, cnt[i]));
                MG_OP(g, c, psacx_op_iota(c, slot.p, cnt[i], 0));
                std::vector<uint64_t> bnd2;
                PSACX_TRY(route_by(i, parts[i][3 * half].p, a, b, cnt[i], ra[i], bounds[i]));
                PSACX_TRY(route_by(i, parts[i][3 * half].p, a, slot.p, cnt[i], rb[i], bnd2));
                rb[i].k2.release();                       // (only the slots of the second pass are read again)
                for (int q3 = 0; q3 < 3; ++q3) parts[i][3 * half + q3].release();      // this half's sub-queries are on their way
                in[i] = {ra[i].k2.p, ra[i].v.p};
                return PSACX_OK;
            }));
            std::vector<std::vector<DBuf<T>>> q, got;
            PSACX_TRY(exchange<T>(2, in, bounds, q, rc));
            ra.clear(); ra.resize(L);                     // (blocks go back to the rank's cache in stream order: engine.hpp pool)
            std::vector<DBuf<T>> res(L);
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, res[i].alloc(c, q[i][0].n));
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (range_min_kernel<T>), q[i][0].n, pyr[i], q[i][0].p, q[i][1].p, q[i][0].n, S[i].off, res[i].p);
                b2[i] = prefix_of(rc[i]);
                in[i] = {res[i].p};
                return PSACX_OK;
            }));
            PSACX_TRY(exchange<T>(1, in, b2, got, rc2));
            q.clear(); res.clear();
            answers[half].resize(L);
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, answers[half][i].alloc(c, cnt[i]));
                MG_OP(g, c, op_put(c, answers[half][i].p, rb[i].v.p, cnt[i], 0, got[i][0].p, 0));
                return PSACX_OK;
            }));
        }
        RankMins rm;
        for (int r = 0; r < 64; ++r) rm.v[r] = r < P ? mins[r] : ~0ull;
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, out[i].alloc(c, cnt[i]));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (rmq_combine_kernel<T>), cnt[i], answers[0][i].p, answers[1][i].p, parts[i][6].p, parts[i][7].p, cnt[i], rm, out[i].p);
            return PSACX_OK;
        }));
        return PSACX_OK;
    }

    // 64-ary min-pyramid over this rank's LCP block in its own buffer (levels >= 1; level 0 is the block), and the block minimum
    int block_pyramid(int i, Pyramid<T>& Pm, DBuf<T>& mem, uint64_t* block_min) {
        psacx_ctx* c = ctx(i);
        const uint64_t m = S[i].m;
        Pm = Pyramid<T>();
        *block_min = (uint64_t)(T)~(T)0;
        if (m == 0) return PSACX_OK;
        uint64_t total = 0, len = m;
        int nlev = 1;
        while (len > 128 && nlev < PYR_MAX) { len = (len + 63) / 64; total += (len + 63) & ~63ull; ++nlev; }
        MG_OP(g, c, mem.alloc(c, total + 64));
        Pm.lvl[0] = S[i].LCP; Pm.len[0] = m; Pm.nlev = 1;
        len = m;
        uint64_t at = 0;
        OP_PROLOGUE(c);
        while (len > 128 && Pm.nlev < PYR_MAX) {
            len = (len + 63) / 64;
            Pm.lvl[Pm.nlev] = mem.p + at; Pm.len[Pm.nlev] = len; at += (len + 63) & ~63ull;
            hipLaunchKernelGGL((pyramid_level_kernel<T>), dim3(grid_for(c, len * 64, 256, 8)), dim3(256), 0, c->stream, Pm.lvl[Pm.nlev - 1],
                               Pm.len[Pm.nlev - 1], Pm.lvl[Pm.nlev], len);
            MG_HIP(g, hipGetLastError());
            Pm.nlev++;
        }
        unsigned long long* d = reinterpret_cast<unsigned long long*>(mem.p + at);      // 64 spare entries at the end
        hipLaunchKernelGGL((top_min_kernel<T>), dim3(1), dim3(256), 0, c->stream, Pm.lvl[Pm.nlev - 1], Pm.len[Pm.nlev - 1], d);
        MG_HIP(g, hipGetLastError());
        MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, d, 8, hipMemcpyDeviceToHost, c->stream));
        MG_HIP(g, hipStreamSynchronize(c->stream));
        *block_min = *reinterpret_cast<uint64_t*>(c->pinned + 32768);
        return PSACX_OK;
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
        PSACX_TRY(dist_sort(rec, counts, id_bits, id_bits));
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
            MG_OP(g, c, op_rebucket_refine<T>(c, rec[i].k1.p, rec[i].k2.p, rec[i].v.p, plist[i], cnt[i], n, h, &bd[i], S[i].SA, S[i].Bsa.p,
                                              S[i].LCP, ids[i].p, qa[i].p, ql[i].p, qh[i].p, &nq[i], &nact[i], &nunf[i]));
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
                SIMPLE_LAUNCH(c, (lcp_apply_kernel<T>), nq[i], S[i].LCP, qa[i].p, nq[i], S[i].off, mins[i].p, h);
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
            const char* env_diet = getenv("PSACX_MULTI_DIET");
            for (int i = 0; i < L; ++i) {
                int same = 0;
                for (int j = 0; j < L; ++j) same += ctx(j)->device == ctx(i)->device;
                size_t fr = 0, tot = 0;
                MG_HIP(g, hipSetDevice(ctx(i)->device));
                MG_HIP(g, hipMemGetInfo(&fr, &tot));
                const double avail = ((double)fr + (double)ctx(i)->pool_bytes * same) / same;
                const bool tight = 14.0 * (double)S[i].m * sizeof(T) > 0.9 * avail;
                mine[i][257] = g->opt_layout == 2 || (g->opt_layout == 0 && ((env_diet && atoi(env_diet)) || tight)) ? 1 : 0;
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
            for (int r = 0; r < P; ++r)            // suffix_array.hpp:226-227
                if (sizes[r] != n / P + ((uint64_t)r < n % P ? 1 : 0)) { g->err = "The input string must be equally block decomposed accross all MPI processes."; return PSACX_EINVAL; }
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
        const char* env_tw = getenv("PSACX_MULTI_TWO_WORD");
        const int tw_mode = env_tw ? atoi(env_tw) : -1;
        bool two_word = !gsa && tw_mode != 0 && lead <= bits_w1 && lead + RADIX_BITS <= bits_w1 + bits_w2 && (tw_mode >= 1 || min_local >= (1ull << 21)) &&
                        !one_stage_env_;
        // One-word records dealt by the top digit of the prefix (sort_first_one_word): 64-bit words, blocks of at least 2^21 characters
        // (PSACX_MULTI_ONE_WORD: 0 = never, 1 = also for small blocks: tests)
        CodeTable tab; for (int ch = 0; ch < 256; ++ch) tab.c[ch] = codes_[ch];
        KeyShape ks; ks.lc = lc; ks.c1 = c1; ks.c2 = c2; ks.spec = 0;
        const char* env_ow = getenv("PSACX_MULTI_ONE_WORD");
        const int ow_mode = env_ow ? atoi(env_ow) : -1;
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
        const bool slices = !getenv("PSACX_MULTI_NO_SLICES") && sizes[0] <= (1ull << 32);      // SA -> ISA by destination slices (below)
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

    // ---------------------------------------------------------------- all nearest smaller values over a block-distributed array
    // ansv<T, left_type, right_type, global_indexing> (ansv.hpp:2042-2051; gansv_impl :1304-1740 keeps per-rank stacks
    // and exchanges unmatched prefix minima).  Here every element first searches its own block (the tile kernel of
    // ansv_tile.hpp); a search that leaves the block goes to the nearest further block whose all-gathered minimum
    // qualifies and is answered from that block's edge.  furthest_eq = nearest <=, then the first strictly smaller value
    // beyond it, then back to the first value <= (three searches, ansv_common.hpp:20-22).
    struct AnsvState { std::vector<const T*> block; std::vector<uint64_t> m; std::vector<Pyramid<T>> pyr; std::vector<DBuf<T>> pyr_mem; std::vector<uint64_t> mins; };

    int ansv_pyramid(int i, const T* block, uint64_t m, Pyramid<T>& Pm, DBuf<T>& mem, uint64_t* block_min) {
        psacx_ctx* c = ctx(i);
        Pm = Pyramid<T>();
        *block_min = ~0ull;
        if (m == 0) return PSACX_OK;
        uint64_t total = 0, len = m;
        while (len > 64) { len = (len + 63) / 64; total += (len + 63) & ~63ull; }
        MG_OP(g, c, mem.alloc(c, total + 64));
        Pm.lvl[0] = const_cast<T*>(block); Pm.len[0] = m; Pm.nlev = 1;
        len = m;
        uint64_t at = 0;
        OP_PROLOGUE(c);
        while (len > 64 && Pm.nlev < PYR_MAX) {
            len = (len + 63) / 64;
            Pm.lvl[Pm.nlev] = mem.p + at; Pm.len[Pm.nlev] = len; at += (len + 63) & ~63ull;
            hipLaunchKernelGGL((pyramid_level_kernel<T>), dim3(grid_for(c, len * 64, 256, 8)), dim3(256), 0, c->stream, Pm.lvl[Pm.nlev - 1],
                               Pm.len[Pm.nlev - 1], Pm.lvl[Pm.nlev], len);
            MG_HIP(g, hipGetLastError());
            Pm.nlev++;
        }
        unsigned long long* d = reinterpret_cast<unsigned long long*>(mem.p + at);
        hipLaunchKernelGGL((top_min_kernel<T>), dim3(1), dim3(256), 0, c->stream, Pm.lvl[Pm.nlev - 1], Pm.len[Pm.nlev - 1], d);
        MG_HIP(g, hipGetLastError());
        MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, d, 8, hipMemcpyDeviceToHost, c->stream));
        MG_HIP(g, hipStreamSynchronize(c->stream));
        *block_min = *reinterpret_cast<uint64_t*>(c->pinned + 32768);
        return PSACX_OK;
    }

    // queries (start1 = start + 1, thr) of every local rank sent to rank cls[j] (< P; P = nowhere), answered there from
    // that rank's block, answers back in query order.  idx / val: all ones / 0 where nothing was found or asked.
    int ansv_ask(AnsvState& A, const std::vector<const T*>& cls, const std::vector<const T*>& start1, const std::vector<const T*>& thr,
                 const std::vector<uint64_t>& cnt, bool strict, bool left, std::vector<DBuf<T>>& idx, std::vector<DBuf<T>>& val) {
        std::vector<Rec<T>> ra(L), rb(L);
        std::vector<std::vector<uint64_t>> bounds(L), b2(L), rc, rc2;
        std::vector<std::vector<const T*>> in(L);
        std::vector<DBuf<T>> slot(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, slot[i].alloc(c, cnt[i]));
            MG_OP(g, c, psacx_op_iota(c, slot[i].p, cnt[i], 0));
            std::vector<uint64_t> bnd2;
            PSACX_TRY(route_by(i, cls[i], start1[i], thr[i], cnt[i], ra[i], bounds[i]));
            PSACX_TRY(route_by(i, cls[i], start1[i], slot[i].p, cnt[i], rb[i], bnd2));
            in[i] = {ra[i].k2.p, ra[i].v.p};
            return PSACX_OK;
        }));
        // class P ("nowhere") is the tail of the routed arrays: it is simply not sent (bounds[P] = its start)
        std::vector<std::vector<DBuf<T>>> q, got;
        PSACX_TRY(exchange<T>(2, in, bounds, q, rc));
        std::vector<DBuf<T>> ri(L), rv(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            const uint64_t qn = q[i][0].n;
            MG_OP(g, c, ri[i].alloc(c, qn)); MG_OP(g, c, rv[i].alloc(c, qn));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (nsv_from_enc_kernel<T>), qn, A.pyr[i], A.m[i], S[i].off, q[i][0].p, q[i][1].p, qn, strict ? 1 : 0, left ? 1 : 0, ri[i].p, rv[i].p);
            b2[i] = prefix_of(rc[i]);
            in[i] = {ri[i].p, rv[i].p};
            return PSACX_OK;
        }));
        PSACX_TRY(exchange<T>(2, in, b2, got, rc2));
        idx.clear(); idx.resize(L); val.clear(); val.resize(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, idx[i].alloc(c, cnt[i])); MG_OP(g, c, val[i].alloc(c, cnt[i]));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (fill_t_kernel<T>), cnt[i], idx[i].p, cnt[i], (T)~(T)0);
            SIMPLE_LAUNCH(c, (fill_t_kernel<T>), cnt[i], val[i].p, cnt[i], (T)0);
            const uint64_t back = got[i][0].n;            // answers come back for the queries that were sent, in routed order
            MG_OP(g, c, op_put(c, idx[i].p, rb[i].v.p, back, 0, got[i][0].p, 0));
            MG_OP(g, c, op_put(c, val[i].p, rb[i].v.p, back, 0, got[i][1].p, 0));
            return PSACX_OK;
        }));
        return PSACX_OK;
    }

    // For every query the nearest element strictly beyond start (start1 - 1; -1 and n allowed) with value < thr (strict) or
    // <= thr, towards lower positions if left.  have_local: idx / val already hold the answers of the block that owns the
    // start (the tile kernel's pass); otherwise that block is asked first.
    int ansv_search(AnsvState& A, const std::vector<const T*>& start1, const std::vector<const T*>& thr, const std::vector<uint64_t>& cnt,
                    bool strict, bool left, bool have_local, std::vector<DBuf<T>>& idx, std::vector<DBuf<T>>& val) {
        const BlkDist bd = make_dist(n, (unsigned)P);
        std::vector<DBuf<T>> own(L);
        std::vector<const T*> cls(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, own[i].alloc(c, cnt[i]));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (ansv_owner_kernel<T>), cnt[i], start1[i], cnt[i], bd, own[i].p);
            cls[i] = own[i].p;
            return PSACX_OK;
        }));
        if (!have_local) PSACX_TRY(ansv_ask(A, cls, start1, thr, cnt, strict, left, idx, val));
        if (solo_) return PSACX_OK;
        RankMins rm, rs;
        for (int r = 0; r < 64; ++r) { rm.v[r] = r < P ? A.mins[r] : ~0ull; rs.v[r] = r < P ? sizes[r] : 0; }
        std::vector<DBuf<T>> target(L), edge(L);
        std::vector<const T*> tp(L), ep(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, target[i].alloc(c, cnt[i])); MG_OP(g, c, edge[i].alloc(c, cnt[i]));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (ansv_target_kernel<T>), cnt[i], own[i].p, thr[i], idx[i].p, cnt[i], rm, rs, P, strict ? 1 : 0, left ? 1 : 0, target[i].p);
            SIMPLE_LAUNCH(c, (fill_t_kernel<T>), cnt[i], edge[i].p, cnt[i], (T)(left ? n + 1 : 0));     // beyond the target's far edge
            tp[i] = target[i].p; ep[i] = edge[i].p;
            return PSACX_OK;
        }));
        std::vector<DBuf<T>> i2, v2;
        PSACX_TRY(ansv_ask(A, tp, ep, thr, cnt, strict, left, i2, v2));
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (ansv_merge_kernel<T>), cnt[i], idx[i].p, val[i].p, i2[i].p, v2[i].p, target[i].p, cnt[i], P);
            return PSACX_OK;
        }));
        return PSACX_OK;
    }

    int ansv(const std::vector<const T*>& block, const std::vector<uint64_t>& m_local, int left_type, int right_type, uint64_t nonsv,
             const std::vector<uint64_t*>& out_left, const std::vector<uint64_t*>& out_right) {
        if (left_type < 0 || left_type > 2 || right_type < 0 || right_type > 2) return PSACX_EINVAL;
        S.resize(L);
        AnsvState A;
        A.block = block; A.m = m_local; A.pyr.resize(L); A.pyr_mem.resize(L);
        PSACX_TRY(par([&](int i) -> int {
            S[i].c = ctx(i); S[i].r = rank(i); S[i].m = m_local[i];
            MG_OP(g, S[i].c, ensure_pinned(S[i].c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
            return PSACX_OK;
        }));
        {
            std::vector<uint64_t> all;
            PSACX_TRY(gather1(m_local, all));
            sizes = all; offs = prefix_of(sizes); n = offs[P];
            for (int r = 0; r < P; ++r)
                if (sizes[r] != n / P + ((uint64_t)r < n % P ? 1 : 0)) { g->err = "The input must be equally block decomposed accross all MPI processes."; return PSACX_EINVAL; }
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
            if (sizeof(T) == 4 && n > 0xFFFFFFFDull) return PSACX_ERANGE;
        }
        std::vector<uint64_t> bm(L);
        for (int i = 0; i < L; ++i) PSACX_TRY(ansv_pyramid(i, block[i], m_local[i], A.pyr[i], A.pyr_mem[i], &bm[i]));
        PSACX_TRY(gather1(bm, A.mins));
        // every element's own position (plus one) as the start of its first search
        std::vector<DBuf<T>> here(L);
        std::vector<const T*> herep(L);
        PSACX_TRY(par([&](int i) -> int {
            MG_OP(g, ctx(i), here[i].alloc(ctx(i), m_local[i]));
            MG_OP(g, ctx(i), psacx_op_iota(ctx(i), here[i].p, m_local[i], S[i].off + 1));
            herep[i] = here[i].p;
            return PSACX_OK;
        }));
        for (int side = 0; side < 2; ++side) {
            const bool left = side == 0;
            const int typ = left ? left_type : right_type;
            const std::vector<uint64_t*>& out = left ? out_left : out_right;
            // first search inside the own block by the tile kernel (it fills both sides; the other side's array is scratch)
            std::vector<DBuf<T>> idx(L), val(L);
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, idx[i].alloc(c, m_local[i])); MG_OP(g, c, val[i].alloc(c, m_local[i]));
                if (!m_local[i]) return PSACX_OK;
                DBuf<uint64_t> other; MG_OP(g, c, other.alloc(c, m_local[i]));
                const int t1 = typ == 0 ? 0 : 1;                       // strict, or nearest <=
                MG_HIP(g, hipSetDevice(c->device));
                if (left) launch_ansv_tiles<T>(c, A.pyr[i], m_local[i], t1, 0, NSV_NONE, out[i], other.p);
                else launch_ansv_tiles<T>(c, A.pyr[i], m_local[i], 0, t1, NSV_NONE, other.p, out[i]);
                MG_HIP(g, hipGetLastError());
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (ansv_local_to_idx_kernel<T>), m_local[i], out[i], block[i], m_local[i], S[i].off, idx[i].p, val[i].p);
                return PSACX_OK;
            }));
            PSACX_TRY(ansv_search(A, herep, block, m_local, typ == 0, left, true, idx, val));
            std::vector<DBuf<T>> far(L);
            if (typ == 2) {
                // s = first strictly smaller value beyond j (threshold: the value found at j), f = from s back towards i the first value <= it
                std::vector<DBuf<T>> st2(L), st3(L), si, sv, fv;
                std::vector<const T*> p2(L), p3(L), u(L);
                PSACX_TRY(par([&](int i) -> int {
                    psacx_ctx* c = ctx(i);
                    MG_OP(g, c, st2[i].alloc(c, m_local[i]));
                    OP_PROLOGUE(c);
                    SIMPLE_LAUNCH(c, (ansv_next_start_kernel<T>), m_local[i], idx[i].p, m_local[i], (T)(left ? n + 1 : 0), st2[i].p);
                    p2[i] = st2[i].p; u[i] = val[i].p;
                    return PSACX_OK;
                }));
                PSACX_TRY(ansv_search(A, p2, u, m_local, true, left, false, si, sv));
                PSACX_TRY(par([&](int i) -> int {
                    psacx_ctx* c = ctx(i);
                    MG_OP(g, c, st3[i].alloc(c, m_local[i]));
                    OP_PROLOGUE(c);
                    SIMPLE_LAUNCH(c, (ansv_next_start_kernel<T>), m_local[i], si[i].p, m_local[i], (T)(left ? 0 : n + 1), st3[i].p);
                    p3[i] = st3[i].p;
                    return PSACX_OK;
                }));
                PSACX_TRY(ansv_search(A, p3, u, m_local, false, !left, false, far, fv));
            }
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (ansv_finish_kernel<T>), m_local[i], idx[i].p, typ == 2 ? (const T*)far[i].p : (const T*)idx[i].p, typ == 2 ? 1 : 0,
                              m_local[i], nonsv, out[i]);
                return PSACX_OK;
            }));
        }
        for (int i = 0; i < L; ++i) { MG_HIP(g, hipSetDevice(ctx(i)->device)); MG_HIP(g, hipStreamSynchronize(ctx(i)->stream)); }
        return PSACX_OK;
    }

    // Left-branching characters of a block-distributed SA / LCP (suffix_array.hpp:211-212; the reference fills local_Lc
    // inside its LCP code, :1365-1383 and par_rmq.hpp:334-481; the result is by definition Lc[i] = S[SA[i-1] + LCP[i]],
    // desa.hpp:262-264, '\0' past the end and at i = 0): the last SA entry of every block goes to its right neighbour, the
    // text positions are fetched from their owners through the engine's bulk-RMA exchange (dist_take), piece by piece so
    // that a block that is a large share of its device fits.
    int left_chars(const std::vector<const uint8_t*>& text, const std::vector<uint64_t>& m_local, const std::vector<T*>& d_sa,
                   const std::vector<T*>& d_lcp, const std::vector<uint8_t*>& d_lc) {
        want_lcp = true;
        S.resize(L);
        for (int i = 0; i < L; ++i) { (void)hipSetDevice(ctx(i)->device); pool_flush(ctx(i)); }
        PSACX_TRY(par([&](int i) -> int {
            S[i].c = ctx(i); S[i].r = rank(i); S[i].m = m_local[i]; S[i].text = text[i];
            S[i].SA = d_sa[i]; S[i].ISA = nullptr; S[i].LCP = d_lcp[i];
            MG_OP(g, S[i].c, ensure_pinned(S[i].c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
            return PSACX_OK;
        }));
        uint64_t chunks = 1;
        {
            std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(2, 0));
            for (int i = 0; i < L; ++i) {
                int same = 0;
                for (int j = 0; j < L; ++j) same += ctx(j)->device == ctx(i)->device;
                size_t fr = 0, tot = 0;
                MG_HIP(g, hipSetDevice(ctx(i)->device));
                MG_HIP(g, hipMemGetInfo(&fr, &tot));
                // the widened text (1 word per character) stays; a piece wants about 12 words per entry
                const double avail = 0.8 * (double)fr / same - (double)m_local[i] * sizeof(T), need = 12.0 * (double)m_local[i] * sizeof(T);
                mine[i][0] = m_local[i];
                mine[i][1] = check_chunks_env_ ? check_chunks_env_ : need > avail ? (uint64_t)(need / std::max(avail, 1.0)) + 1 : 1;
            }
            std::vector<uint64_t> all;
            PSACX_TRY(gather(2, mine, all));
            sizes.assign(P, 0);
            for (int r = 0; r < P; ++r) { sizes[r] = all[(size_t)r * 2]; chunks = std::max(chunks, all[(size_t)r * 2 + 1]); }
            chunks = std::min<uint64_t>(chunks, 4096);
            offs = prefix_of(sizes); n = offs[P];
            for (int r = 0; r < P; ++r)
                if (sizes[r] != n / P + ((uint64_t)r < n % P ? 1 : 0)) { g->err = "The input string must be equally block decomposed accross all MPI processes."; return PSACX_EINVAL; }
            for (int i = 0; i < L; ++i) { S[i].off = offs[rank(i)]; ctx(i)->pool_cache_limit = 0; }
            if (n == 0) return PSACX_EINVAL;
        }
        std::vector<DBuf<T>> wide(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, wide[i].alloc(c, S[i].m));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (widen_text_kernel<T>), S[i].m, text[i], S[i].m, wide[i].p);
            return PSACX_OK;
        }));
        // SA of the entry before every block
        std::vector<psacx_boundary> edge;
        {
            std::vector<uint64_t> one(L);
            std::vector<const T*> a1(L), a2(L), a3(L);
            for (int i = 0; i < L; ++i) { one[i] = S[i].m ? 1 : 0; a1[i] = S[i].SA + (S[i].m ? S[i].m - 1 : 0); a2[i] = a1[i]; a3[i] = a1[i]; }
            PSACX_TRY(neighbours(a1, a2, a3, one, 1, edge));
        }
        std::vector<uint64_t> carry(L, 0);                       // SA of the last entry of the previous piece
        for (uint64_t q = 0; q < chunks; ++q) {
            std::vector<uint64_t> from(L), cnt(L);
            for (int i = 0; i < L; ++i) {
                from[i] = (uint64_t)(((unsigned __int128)S[i].m * q) / chunks);
                cnt[i] = (uint64_t)(((unsigned __int128)S[i].m * (q + 1)) / chunks) - from[i];
            }
            std::vector<DBuf<T>> qs(L), ch;
            std::vector<const T*> blk(L), gi(L);
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, qs[i].alloc(c, cnt[i]));
                const int has_prev = from[i] ? 1 : edge[i].has_prev;
                const uint64_t prev = from[i] ? carry[i] : edge[i].prev[0];
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (lc_queries_kernel<T>), cnt[i], S[i].SA + from[i], S[i].LCP + from[i], cnt[i], n, has_prev, (T)prev, qs[i].p);
                if (cnt[i]) { std::vector<uint64_t> o; PSACX_TRY(fetch(i, S[i].SA + from[i], {cnt[i] - 1}, o)); carry[i] = o[0]; }
                blk[i] = wide[i].p; gi[i] = qs[i].p;
                return PSACX_OK;
            }));
            PSACX_TRY(dist_take(blk, gi, cnt, ch));
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (lc_narrow_kernel<T>), cnt[i], (const T*)ch[i].p, (const T*)qs[i].p, cnt[i], n, d_lc[i] + from[i]);
                MG_HIP(g, hipStreamSynchronize(c->stream));
                return PSACX_OK;
            }));
        }
        return PSACX_OK;
    }

    // Suffix-tree node table of a block-distributed SA / LCP (construct_suffix_tree on p ranks, suffix_tree.hpp:413-499): rank r
    // receives the rows of the LCP indices of its block, nodes[i][(sigma + 1) columns], column c = the child reached through
    // the character with alphabet code c (0 = end of text), leaves numbered n + i, 0 = none.  Parents from the distributed
    // ANSV of LCP (suffix_tree.hpp:62), the LCP values at the parents and the edge characters S[SA[i] + lcp] through the bulk
    // fetch (dist_take), the cells to the owners of the parents' rows like bulk_permute's (index, value) pairs.
    // d_nodes == nullptr: only *sigma is computed (the size query of psacx_suffix_tree_*).
    int suffix_tree(const std::vector<const uint8_t*>& text, const std::vector<uint64_t>& m_local, const std::vector<T*>& d_sa,
                    const std::vector<T*>& d_lcp, const std::vector<unsigned long long*>* d_nodes, uint32_t* sigma) {
        want_lcp = true;
        // ---- alphabet over all blocks (alphabet.hpp:147-164: codes 1 .. sigma in byte order)
        CodeTable tab;
        {
            std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(256, 0));
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, ensure_pinned(c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
                DBuf<unsigned long long> h; MG_OP(g, c, h.alloc(c, 256));
                MG_HIP(g, hipSetDevice(c->device));
                MG_HIP(g, hipMemsetAsync(h.p, 0, 256 * 8, c->stream));
                if (m_local[i]) {
                    hipLaunchKernelGGL((char_hist_kernel<256>), dim3(grid_for(c, m_local[i] / 16 + 1, 256, 8)), dim3(256), 0, c->stream, text[i], m_local[i], h.p);
                    MG_HIP(g, hipGetLastError());
                }
                MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, h.p, 256 * 8, hipMemcpyDeviceToHost, c->stream));
                MG_HIP(g, hipStreamSynchronize(c->stream));
                std::memcpy(mine[i].data(), c->pinned + 32768, 256 * 8);
                return PSACX_OK;
            }));
            std::vector<uint64_t> all;
            PSACX_TRY(gather(256, mine, all));
            uint16_t next = 1;
            for (int ch = 0; ch < 256; ++ch) {
                uint64_t tot = 0;
                for (int r = 0; r < P; ++r) tot += all[(size_t)r * 256 + ch];
                tab.c[ch] = tot ? next++ : (uint16_t)0;
            }
            *sigma = next - 1u;
        }
        if (!d_nodes) return PSACX_OK;
        const uint64_t row = (uint64_t)*sigma + 1;
        // ---- ANSV of LCP: left furthest_eq, right nearest_sm
        std::vector<DBuf<uint64_t>> ln(L), rn(L);
        {
            std::vector<const T*> blk(L); std::vector<uint64_t*> ol(L), orr(L);
            for (int i = 0; i < L; ++i) {
                MG_OP(g, ctx(i), ln[i].alloc(ctx(i), m_local[i])); MG_OP(g, ctx(i), rn[i].alloc(ctx(i), m_local[i]));
                blk[i] = d_lcp[i]; ol[i] = ln[i].p; orr[i] = rn[i].p;
            }
            PSACX_TRY(ansv(blk, m_local, 2, 0, NSV_NONE, ol, orr));          // (sets sizes / offs / n)
        }
        S.resize(L);
        for (int i = 0; i < L; ++i) {
            S[i].c = ctx(i); S[i].r = rank(i); S[i].m = m_local[i]; S[i].text = text[i]; S[i].SA = d_sa[i]; S[i].ISA = nullptr; S[i].LCP = d_lcp[i];
            S[i].off = offs[rank(i)];
        }
        // ---- LCP at the two parents; the first LCP entry of the next block
        std::vector<DBuf<T>> lcp_l, lcp_r;
        {
            std::vector<DBuf<T>> pl(L), pr(L);
            std::vector<const T*> blk(L), g1(L), g2(L);
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, pl[i].alloc(c, S[i].m)); MG_OP(g, c, pr[i].alloc(c, S[i].m));
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (st_nsv_positions_kernel<T>), S[i].m, (const uint64_t*)ln[i].p, S[i].m, pl[i].p);
                SIMPLE_LAUNCH(c, (st_nsv_positions_kernel<T>), S[i].m, (const uint64_t*)rn[i].p, S[i].m, pr[i].p);
                blk[i] = S[i].LCP; g1[i] = pl[i].p; g2[i] = pr[i].p;
                return PSACX_OK;
            }));
            PSACX_TRY(dist_take(blk, g1, m_local, lcp_l));
            PSACX_TRY(dist_take(blk, g2, m_local, lcp_r));
        }
        std::vector<psacx_boundary> edge;
        {
            std::vector<const T*> a1(L);
            for (int i = 0; i < L; ++i) a1[i] = S[i].LCP;
            PSACX_TRY(neighbours(a1, a1, a1, m_local, 1, edge));
        }
        // ---- parents and edge positions, edge characters
        std::vector<DBuf<T>> p1(L), p2(L);
        std::vector<DBuf<uint64_t>> q1(L), q2(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, p1[i].alloc(c, S[i].m)); MG_OP(g, c, p2[i].alloc(c, S[i].m)); MG_OP(g, c, q1[i].alloc(c, S[i].m)); MG_OP(g, c, q2[i].alloc(c, S[i].m));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (st_parents_kernel<T>), S[i].m, (const T*)S[i].LCP, (const T*)S[i].SA, S[i].m, S[i].off, n, (const uint64_t*)ln[i].p, (const uint64_t*)rn[i].p,
                          (const T*)lcp_l[i].p, (const T*)lcp_r[i].p, (int)edge[i].has_next, (T)edge[i].next[0], p1[i].p, q1[i].p, p2[i].p, q2[i].p);
            MG_HIP(g, hipStreamSynchronize(c->stream));
            return PSACX_OK;
        }));
        for (int i = 0; i < L; ++i) { ln[i].release(); rn[i].release(); lcp_l[i].release(); lcp_r[i].release(); }
        std::vector<DBuf<T>> wide(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, wide[i].alloc(c, S[i].m));
            MG_HIP(g, hipSetDevice(c->device));
            MG_HIP(g, hipMemsetAsync((*d_nodes)[i], 0, S[i].m * row * sizeof(unsigned long long), c->stream));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (widen_text_kernel<T>), S[i].m, text[i], S[i].m, wide[i].p);
            return PSACX_OK;
        }));
        for (int which = 0; which < 2; ++which) {
            std::vector<DBuf<uint64_t>>& q = which ? q2 : q1;
            std::vector<DBuf<T>>& par_ = which ? p2 : p1;
            std::vector<DBuf<T>> qs(L), ch, x(L), y(L);
            std::vector<const T*> blk(L), gi(L);
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, qs[i].alloc(c, S[i].m));
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (st_positions_kernel<T>), S[i].m, (const uint64_t*)q[i].p, S[i].m, n, qs[i].p);
                blk[i] = wide[i].p; gi[i] = qs[i].p;
                return PSACX_OK;
            }));
            PSACX_TRY(dist_take(blk, gi, m_local, ch));
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, x[i].alloc(c, S[i].m)); MG_OP(g, c, y[i].alloc(c, S[i].m));
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (st_payload_kernel<T>), S[i].m, (const uint64_t*)q[i].p, (const T*)ch[i].p, S[i].m, S[i].off, n, tab, which == 0 ? 1 : 0, x[i].p, y[i].p);
                return PSACX_OK;
            }));
            // the cells to the owners of their rows: the same stable partition by owner for both payload words
            std::vector<const T*> pos(L), xs(L), ys(L);
            std::vector<uint64_t> tot(L);
            std::vector<Rec<T>> r1(L), r2(L);
            std::vector<std::vector<DBuf<T>>> got1, got2;
            if (solo_) { for (int i = 0; i < L; ++i) { pos[i] = par_[i].p; xs[i] = x[i].p; ys[i] = y[i].p; tot[i] = S[i].m; } }
            else {
                std::vector<std::vector<uint64_t>> b1(L), b2(L), rc;
                std::vector<std::vector<const T*>> in1(L), in2(L);
                for (int i = 0; i < L; ++i) {
                    PSACX_TRY(route(i, par_[i].p, x[i].p, S[i].m, r1[i], b1[i])); in1[i] = {r1[i].k2.p, r1[i].v.p};
                    PSACX_TRY(route(i, par_[i].p, y[i].p, S[i].m, r2[i], b2[i])); in2[i] = {r2[i].v.p};
                }
                PSACX_TRY(exchange<T>(2, in1, b1, got1, rc));
                PSACX_TRY(exchange<T>(1, in2, b2, got2, rc));
                for (int i = 0; i < L; ++i) { pos[i] = got1[i][0].p; xs[i] = got1[i][1].p; ys[i] = got2[i][0].p; tot[i] = got1[i][0].n; }
            }
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (st_put_kernel<T>), tot[i], (*d_nodes)[i], S[i].off, row, pos[i], xs[i], ys[i], tot[i], n);
                MG_HIP(g, hipStreamSynchronize(c->stream));
                return PSACX_OK;
            }));
        }
        return PSACX_OK;
    }

    // Distributed verification of block-distributed SA / ISA / LCP without gathering anything on one rank: what
    // d_check_sa does (check_suffix_array.hpp:207-267: SA a permutation whose inverse is ISA, S[SA[i-1]] <= S[SA[i]],
    // ties decided by the ranks of the suffixes one further) with the engine's own exchanges (bulk_rma for
    // ISA[SA[i]], S[SA[i]], ISA[SA[i]+1]), plus the LCP array through its recurrence
    //   LCP[i] = 0 | 1 | 1 + min(LCP[ISA[SA[i-1]+1]+1 .. ISA[SA[i]+1]])      (range minima: bulk_rmq_v2)
    // which has the true LCP array as its only solution.  errors[0..3] as psacx_check_dev_*, summed over all ranks.
    int check(const std::vector<const uint8_t*>& text, const std::vector<uint64_t>& m_local, const std::vector<T*>& d_sa,
              const std::vector<T*>& d_isa, const std::vector<T*>& d_lcp, bool with_lcp, uint64_t errors[4]) {
        want_lcp = with_lcp;
        S.resize(L);
        for (int i = 0; i < L; ++i) { (void)hipSetDevice(ctx(i)->device); pool_flush(ctx(i)); }     // the checker wants different sizes than the construction left cached
        PSACX_TRY(par([&](int i) -> int {
            S[i].c = ctx(i); S[i].r = rank(i); S[i].m = m_local[i]; S[i].text = text[i];
            S[i].SA = d_sa[i]; S[i].ISA = d_isa[i]; S[i].LCP = with_lcp ? d_lcp[i] : nullptr;
            MG_OP(g, S[i].c, ensure_pinned(S[i].c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
            return PSACX_OK;
        }));
        // The block is verified in `chunks` pieces of consecutive SA positions (every test is local to an entry and its
        // predecessor): one piece needs about 24 words per entry, so a block that large a share of the device is cut.
        uint64_t chunks = 1;
        {
            std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(2, 0));
            for (int i = 0; i < L; ++i) {
                int same = 0;
                for (int j = 0; j < L; ++j) same += ctx(j)->device == ctx(i)->device;
                size_t fr = 0, tot = 0;
                MG_HIP(g, hipSetDevice(ctx(i)->device));
                MG_HIP(g, hipMemGetInfo(&fr, &tot));
                const double avail = 0.8 * (double)fr / same, need = 24.0 * (double)m_local[i] * sizeof(T);
                mine[i][0] = m_local[i];
                mine[i][1] = check_chunks_env_ ? check_chunks_env_ : need > avail ? (uint64_t)(need / std::max(avail, 1.0)) + 1 : 1;
            }
            std::vector<uint64_t> all;
            PSACX_TRY(gather(2, mine, all));
            sizes.assign(P, 0);
            for (int r = 0; r < P; ++r) { sizes[r] = all[(size_t)r * 2]; chunks = std::max(chunks, all[(size_t)r * 2 + 1]); }
            chunks = std::min<uint64_t>(chunks, 4096);
            offs = prefix_of(sizes); n = offs[P];
            for (int r = 0; r < P; ++r)
                if (sizes[r] != n / P + ((uint64_t)r < n % P ? 1 : 0)) { g->err = "The input string must be equally block decomposed accross all MPI processes."; return PSACX_EINVAL; }
            for (int i = 0; i < L; ++i) { S[i].off = offs[rank(i)]; ctx(i)->pool_cache_limit = 0; }
            if (n == 0) return PSACX_EINVAL;
        }
        // the text as index words, once (S[SA[i]] travels through the same exchanges as the indices)
        std::vector<DBuf<T>> wide(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, wide[i].alloc(c, S[i].m));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (widen_text_kernel<T>), S[i].m, text[i], S[i].m, wide[i].p);
            return PSACX_OK;
        }));
        // (SA, S[SA], ISA[SA + 1]) of a range of SA positions of every rank
        auto triple = [&](const std::vector<uint64_t>& from, const std::vector<uint64_t>& cnt, std::vector<DBuf<T>>* back, std::vector<DBuf<T>>& ch,
                          std::vector<DBuf<T>>& nx) -> int {
            std::vector<const T*> blk(L), gi(L);
            std::vector<DBuf<T>> q1(L);
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, q1[i].alloc(c, cnt[i]));
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (add_scalar_kernel<T>), cnt[i], S[i].SA + from[i], cnt[i], (uint64_t)1, n, q1[i].p);
                return PSACX_OK;
            }));
            for (int i = 0; i < L; ++i) { blk[i] = S[i].ISA; gi[i] = S[i].SA + from[i]; }
            if (back) PSACX_TRY(dist_take(blk, gi, cnt, *back));
            for (int i = 0; i < L; ++i) blk[i] = wide[i].p;
            PSACX_TRY(dist_take(blk, gi, cnt, ch));
            for (int i = 0; i < L; ++i) { blk[i] = S[i].ISA; gi[i] = q1[i].p; }
            PSACX_TRY(dist_take(blk, gi, cnt, nx));
            return PSACX_OK;
        };
        // the last entry of every block: the predecessor of the next non-empty block's first entry
        std::vector<psacx_boundary> edge;
        {
            std::vector<uint64_t> from(L), one(L);
            std::vector<DBuf<T>> ch, nx;
            for (int i = 0; i < L; ++i) { one[i] = S[i].m ? 1 : 0; from[i] = S[i].m ? S[i].m - 1 : 0; }
            PSACX_TRY(triple(from, one, nullptr, ch, nx));
            std::vector<const T*> a1(L), a2(L), a3(L);
            for (int i = 0; i < L; ++i) { a1[i] = S[i].SA + from[i]; a2[i] = ch[i].p; a3[i] = nx[i].p; }
            PSACX_TRY(neighbours(a1, a2, a3, one, 3, edge));
        }
        std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(4, 0));
        std::vector<std::vector<uint64_t>> carry(L, std::vector<uint64_t>(3, 0));        // last entry of the previous piece
        for (uint64_t q = 0; q < chunks; ++q) {
            std::vector<uint64_t> from(L), cnt(L);
            for (int i = 0; i < L; ++i) {
                from[i] = (uint64_t)(((unsigned __int128)S[i].m * q) / chunks);
                cnt[i] = (uint64_t)(((unsigned __int128)S[i].m * (q + 1)) / chunks) - from[i];
            }
            std::vector<DBuf<T>> back, ch, nx, mins;
            PSACX_TRY(triple(from, cnt, &back, ch, nx));
            std::vector<psacx_boundary> bd(L);
            for (int i = 0; i < L; ++i) {
                std::memset(&bd[i], 0, sizeof(psacx_boundary));
                if (from[i] == 0) { bd[i].has_prev = edge[i].has_prev; for (int w = 0; w < 3; ++w) bd[i].prev[w] = edge[i].prev[w]; }
                else { bd[i].has_prev = 1; for (int w = 0; w < 3; ++w) bd[i].prev[w] = carry[i][w]; }
            }
            PSACX_TRY(par([&](int i) -> int {                     // this piece's last entry, for the next one
                if (!cnt[i]) return PSACX_OK;
                const T* arr[3] = {S[i].SA + from[i], ch[i].p, nx[i].p};
                for (int w = 0; w < 3; ++w) { std::vector<uint64_t> o; PSACX_TRY(fetch(i, arr[w], {cnt[i] - 1}, o)); carry[i][w] = o[0]; }
                return PSACX_OK;
            }));
            if (with_lcp) {
                std::vector<DBuf<T>> qlo(L), qhi(L);
                std::vector<const T*> lo(L), hi(L);
                PSACX_TRY(par([&](int i) -> int {
                    psacx_ctx* c = ctx(i);
                    MG_OP(g, c, qlo[i].alloc(c, cnt[i])); MG_OP(g, c, qhi[i].alloc(c, cnt[i]));
                    OP_PROLOGUE(c);
                    SIMPLE_LAUNCH(c, (check_queries_kernel<T>), cnt[i], S[i].SA + from[i], ch[i].p, nx[i].p, cnt[i], n, bd[i].has_prev, (T)bd[i].prev[0],
                                  (T)bd[i].prev[1], (T)bd[i].prev[2], qlo[i].p, qhi[i].p);
                    lo[i] = qlo[i].p; hi[i] = qhi[i].p;
                    return PSACX_OK;
                }));
                PSACX_TRY(dist_range_min(lo, hi, cnt, mins));
            }
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                DBuf<unsigned long long> e; MG_OP(g, c, e.alloc(c, 4));
                MG_HIP(g, hipSetDevice(c->device));
                MG_HIP(g, hipMemsetAsync(e.p, 0, 32, c->stream));
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (check_verdict_kernel<T>), cnt[i], S[i].SA + from[i], back[i].p, ch[i].p, nx[i].p, with_lcp ? (const T*)(S[i].LCP + from[i]) : (const T*)nullptr,
                              with_lcp ? (const T*)mins[i].p : (const T*)nullptr, cnt[i], S[i].off + from[i], n, bd[i].has_prev, (T)bd[i].prev[0], (T)bd[i].prev[1],
                              (T)bd[i].prev[2], e.p);
                MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, e.p, 32, hipMemcpyDeviceToHost, c->stream));
                MG_HIP(g, hipStreamSynchronize(c->stream));
                const uint64_t* h = reinterpret_cast<const uint64_t*>(c->pinned + 32768);
                for (int w = 0; w < 4; ++w) mine[i][w] += h[w];
                return PSACX_OK;
            }));
        }
        std::vector<uint64_t> all;
        PSACX_TRY(gather(4, mine, all));
        for (int q = 0; q < 4; ++q) { errors[q] = 0; for (int r = 0; r < P; ++r) errors[q] += all[(size_t)r * 4 + q]; }
        return PSACX_OK;
    }

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
