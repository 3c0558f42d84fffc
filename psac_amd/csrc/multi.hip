// multi.hip -- extern "C" surface of the multi-GPU construction (include/psacx.h, psacx_multi_*) over multi.hpp.
#include "multi.hpp"
#include "multi_first_round.hpp"     // members of MultiRun: the first round in two-word / one-word records
#include "multi_queries.hpp"         // members of MultiRun: ANSV, left-branching characters, suffix-tree table, checker
#include "multi_refine.hpp"          // members of MultiRun: the sort and the range minima of a refinement round

using namespace psacx;

namespace {
int make_rank(psacx_multi* g, int i, int grank, int device) {
    MRank& R = g->R[i];
    R.grank = grank;
    int rc = psacx_create(&R.ctx, device, nullptr);
    if (rc != PSACX_OK) { g->err = std::string("psacx_create: ") + psacx_strerror(rc); return rc; }
    MG_HIP(g, hipSetDevice(device));
    MG_HIP(g, hipStreamCreateWithFlags(&R.comm_stream, hipStreamNonBlocking));
    MG_HIP(g, hipEventCreateWithFlags(&R.ev_ready, hipEventDisableTiming));
    MG_HIP(g, hipEventCreateWithFlags(&R.ev_done, hipEventDisableTiming));
    return PSACX_OK;
}

template <typename T>
int run_dev(psacx_multi* g, const uint8_t* const* d_text, const uint64_t* m, uint32_t k, uint32_t flags, T* const* sa, T* const* isa,
            T* const* lcp, const uint64_t* str_off = nullptr, uint64_t nstr = 0) {
    if (!g || !d_text || !m || !sa || !isa) return PSACX_EINVAL;
    if ((flags & PSACX_LCP) && !lcp) return PSACX_EINVAL;
    g->err.clear();
    MultiRun<T> run(g);
    std::vector<const uint8_t*> t(g->nlocal); std::vector<uint64_t> mm(g->nlocal);
    std::vector<T*> a(g->nlocal), b(g->nlocal), c(g->nlocal, nullptr);
    for (int i = 0; i < g->nlocal; ++i) { t[i] = d_text[i]; mm[i] = m[i]; a[i] = sa[i]; b[i] = isa[i]; if (flags & PSACX_LCP) c[i] = lcp[i]; }
    return run.construct(t, mm, k, flags, a, b, c, str_off, nstr);
}

template <typename T>
int check_dev(psacx_multi* g, const uint8_t* const* d_text, const uint64_t* m, const T* const* sa, const T* const* isa, const T* const* lcp,
              uint64_t errors[4]) {
    if (!g || !d_text || !m || !sa || !isa || !errors) return PSACX_EINVAL;
    g->err.clear();
    MultiRun<T> run(g);
    std::vector<const uint8_t*> t(g->nlocal); std::vector<uint64_t> mm(g->nlocal);
    std::vector<T*> a(g->nlocal), b(g->nlocal), c(g->nlocal, nullptr);
    for (int i = 0; i < g->nlocal; ++i) { t[i] = d_text[i]; mm[i] = m[i]; a[i] = const_cast<T*>(sa[i]); b[i] = const_cast<T*>(isa[i]); if (lcp) c[i] = const_cast<T*>(lcp[i]); }
    return run.check(t, mm, a, b, c, lcp != nullptr, errors);
}

template <typename T>
int ansv_dev(psacx_multi* g, const T* const* d_in, const uint64_t* m, int lt, int rt, uint64_t nonsv, uint64_t* const* l, uint64_t* const* r) {
    if (!g || !d_in || !m || !l || !r) return PSACX_EINVAL;
    g->err.clear();
    MultiRun<T> run(g);
    std::vector<const T*> b(g->nlocal); std::vector<uint64_t> mm(g->nlocal); std::vector<uint64_t*> ol(g->nlocal), orr(g->nlocal);
    for (int i = 0; i < g->nlocal; ++i) { b[i] = d_in[i]; mm[i] = m[i]; ol[i] = l[i]; orr[i] = r[i]; }
    return run.ansv(b, mm, lt, rt, nonsv, ol, orr);
}

template <typename T>
int left_chars_dev(psacx_multi* g, const uint8_t* const* d_text, const uint64_t* m, const T* const* sa, const T* const* lcp, uint8_t* const* lc) {
    if (!g || !d_text || !m || !sa || !lcp || !lc) return PSACX_EINVAL;
    g->err.clear();
    MultiRun<T> run(g);
    std::vector<const uint8_t*> t(g->nlocal); std::vector<uint64_t> mm(g->nlocal);
    std::vector<T*> a(g->nlocal), c(g->nlocal); std::vector<uint8_t*> o(g->nlocal);
    for (int i = 0; i < g->nlocal; ++i) { t[i] = d_text[i]; mm[i] = m[i]; a[i] = const_cast<T*>(sa[i]); c[i] = const_cast<T*>(lcp[i]); o[i] = lc[i]; }
    return run.left_chars(t, mm, a, c, o);
}

template <typename T>
int suffix_tree_dev(psacx_multi* g, const uint8_t* const* d_text, const uint64_t* m, const T* const* sa, const T* const* lcp, uint64_t* const* nodes, uint32_t* sigma) {
    if (!g || !d_text || !m || !sigma) return PSACX_EINVAL;
    if (nodes && (!sa || !lcp)) return PSACX_EINVAL;
    g->err.clear();
    MultiRun<T> run(g);
    std::vector<const uint8_t*> t(g->nlocal); std::vector<uint64_t> mm(g->nlocal);
    std::vector<T*> a(g->nlocal, nullptr), c(g->nlocal, nullptr); std::vector<unsigned long long*> o(g->nlocal, nullptr);
    for (int i = 0; i < g->nlocal; ++i) {
        t[i] = d_text[i]; mm[i] = m[i];
        if (nodes) { a[i] = const_cast<T*>(sa[i]); c[i] = const_cast<T*>(lcp[i]); o[i] = reinterpret_cast<unsigned long long*>(nodes[i]); }
    }
    return run.suffix_tree(t, mm, a, c, nodes ? &o : nullptr, sigma);
}

// whole text on the host of a single process that owns every rank: blocks to the GPUs, results back in rank order
template <typename T>
int run_host(psacx_multi* g, const uint8_t* text, uint64_t n, uint32_t k, uint32_t flags, T* sa, T* isa, T* lcp, uint8_t* lc = nullptr,
             const uint64_t* str_off = nullptr, uint64_t nstr = 0) {
    if (!g || !text || !sa || !isa || n == 0) return PSACX_EINVAL;
    if ((flags & PSACX_LCP) && !lcp) return PSACX_EINVAL;
    if (lc && !(flags & PSACX_LCP)) return PSACX_EINVAL;        // (the reference fills Lc inside its LCP code)
    if (g->nlocal != g->nranks) { g->err = "the host-pointer form needs every rank in this process"; return PSACX_EINVAL; }
    if (sizeof(T) == 4 && n > 0xFFFFFFFEull) return PSACX_ERANGE;
    const int P = g->nranks;
    const bool want_lcp = (flags & PSACX_LCP) != 0;
    std::vector<uint64_t> m(P), off(P + 1, 0);
    for (int r = 0; r < P; ++r) { m[r] = n / P + ((uint64_t)r < n % P ? 1 : 0); off[r + 1] = off[r] + m[r]; }
    const uint64_t slack = m[0] / 8 + 256;
    std::vector<DBuf<uint8_t>> dt(P);
    std::vector<DBuf<T>> dsa(P), disa(P), dlcp(P);
    std::vector<const uint8_t*> tp(P); std::vector<T*> a(P), b(P), c(P, nullptr);
    for (int r = 0; r < P; ++r) {
        psacx_ctx* cx = g->R[r].ctx;
        // the result arrays get the slack that lets the reduced-memory layout use them as record arrays
        MG_OP(g, cx, dt[r].alloc(cx, m[r])); MG_OP(g, cx, dsa[r].alloc(cx, m[r], m[r] + slack)); MG_OP(g, cx, disa[r].alloc(cx, m[r], m[r] + slack));
        if (want_lcp) MG_OP(g, cx, dlcp[r].alloc(cx, m[r], m[r] + slack));
        if (m[r]) MG_OP(g, cx, staged_h2d(cx, dt[r].p, text + off[r], m[r]));
        tp[r] = dt[r].p; a[r] = dsa[r].p; b[r] = disa[r].p; c[r] = want_lcp ? dlcp[r].p : nullptr;
    }
    const uint64_t user_slack = g->out_slack;
    g->out_slack = slack;
    int rc = run_dev<T>(g, tp.data(), m.data(), k, flags, a.data(), b.data(), c.data(), str_off, nstr);
    g->out_slack = user_slack;
    if (rc != PSACX_OK) return rc;
    if (lc) {
        std::vector<DBuf<uint8_t>> dlc(P);
        std::vector<uint8_t*> o(P);
        std::vector<const T*> ca(P), cc(P);
        for (int r = 0; r < P; ++r) { psacx_ctx* cx = g->R[r].ctx; MG_OP(g, cx, dlc[r].alloc(cx, m[r])); o[r] = dlc[r].p; ca[r] = a[r]; cc[r] = c[r]; }
        rc = left_chars_dev<T>(g, tp.data(), m.data(), ca.data(), cc.data(), o.data());
        if (rc != PSACX_OK) return rc;
        for (int r = 0; r < P; ++r) {
            psacx_ctx* cx = g->R[r].ctx;
            MG_HIP(g, hipSetDevice(cx->device));
            if (m[r]) MG_OP(g, cx, staged_d2h(cx, lc + off[r], dlc[r].p, m[r]));
        }
    }
    for (int r = 0; r < P; ++r) {
        psacx_ctx* cx = g->R[r].ctx;
        MG_HIP(g, hipSetDevice(cx->device));
        if (!m[r]) continue;
        // (narrowed on the device, widened on the host, the three arrays chunk by chunk through one ring: engine.hpp: staged_d2h_jobs)
        D2hJob<T> jobs[3] = {{sa + off[r], dsa[r].p, m[r], n - 1, 0, 0, 0, 0}, {isa + off[r], disa[r].p, m[r], n - 1, 0, 0, 0, 0},
                             {want_lcp ? lcp + off[r] : (T*)nullptr, want_lcp ? dlcp[r].p : (const T*)nullptr, m[r], ~0ull, 0, 0, 0, 0}};
        MG_OP(g, cx, staged_d2h_jobs<T>(cx, jobs, want_lcp ? 3 : 2));
    }
    return PSACX_OK;
}

// host-pointer form of the node table: blocks of text / SA / LCP to the ranks, the rows back in rank order
template <typename T>
int suffix_tree_host(psacx_multi* g, const uint8_t* text, uint64_t n, const T* sa, const T* lcp, uint64_t* nodes, uint32_t* sigma) {
    if (!g || !text || !sigma || n == 0) return PSACX_EINVAL;
    if (nodes && (!sa || !lcp)) return PSACX_EINVAL;
    if (g->nlocal != g->nranks) { g->err = "the host-pointer form needs every rank in this process"; return PSACX_EINVAL; }
    const int P = g->nranks;
    std::vector<uint64_t> m(P), off(P + 1, 0);
    for (int r = 0; r < P; ++r) { m[r] = n / P + ((uint64_t)r < n % P ? 1 : 0); off[r + 1] = off[r] + m[r]; }
    std::vector<DBuf<uint8_t>> dt(P);
    std::vector<DBuf<T>> dsa(P), dlcp(P);
    std::vector<const uint8_t*> tp(P); std::vector<const T*> a(P), c(P);
    for (int r = 0; r < P; ++r) {
        psacx_ctx* cx = g->R[r].ctx;
        MG_OP(g, cx, dt[r].alloc(cx, m[r]));
        if (m[r]) MG_OP(g, cx, staged_h2d(cx, dt[r].p, text + off[r], m[r]));
        tp[r] = dt[r].p;
        if (nodes) {
            MG_OP(g, cx, dsa[r].alloc(cx, m[r])); MG_OP(g, cx, dlcp[r].alloc(cx, m[r]));
            if (m[r]) { MG_OP(g, cx, staged_h2d(cx, dsa[r].p, sa + off[r], m[r] * sizeof(T))); MG_OP(g, cx, staged_h2d(cx, dlcp[r].p, lcp + off[r], m[r] * sizeof(T))); }
        }
        a[r] = dsa[r].p; c[r] = dlcp[r].p;
    }
    int rc = suffix_tree_dev<T>(g, tp.data(), m.data(), (const T* const*)nullptr, (const T* const*)nullptr, (uint64_t* const*)nullptr, sigma);
    if (rc != PSACX_OK || !nodes) return rc;
    const uint64_t row = (uint64_t)*sigma + 1;
    std::vector<DBuf<uint64_t>> dn(P);
    std::vector<uint64_t*> o(P);
    for (int r = 0; r < P; ++r) { psacx_ctx* cx = g->R[r].ctx; MG_OP(g, cx, dn[r].alloc(cx, m[r] * row)); o[r] = dn[r].p; }
    rc = suffix_tree_dev<T>(g, tp.data(), m.data(), a.data(), c.data(), o.data(), sigma);
    if (rc != PSACX_OK) return rc;
    for (int r = 0; r < P; ++r) {
        psacx_ctx* cx = g->R[r].ctx;
        MG_HIP(g, hipSetDevice(cx->device));
        if (m[r]) MG_OP(g, cx, staged_d2h(cx, nodes + off[r] * row, dn[r].p, m[r] * row * sizeof(uint64_t)));
    }
    return PSACX_OK;
}

} // namespace

extern "C" {

int psacx_multi_create(psacx_multi** out, int ndev, const int* dev_ids) { return psacx_multi_create_ex(out, ndev, dev_ids, 0u); }

int psacx_multi_create_ex(psacx_multi** out, int ndev, const int* dev_ids, uint32_t flags) {
    if (!out || ndev < 1 || ndev > 64 || (flags & ~(PSACX_MULTI_FORCE_WIRE | PSACX_MULTI_NO_RCCL))) return PSACX_EINVAL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return PSACX_ENOGPU; }
    std::vector<int> devs(ndev);
    bool distinct = true;
    for (int i = 0; i < ndev; ++i) {
        devs[i] = dev_ids ? dev_ids[i] : i;
        if (devs[i] < 0 || devs[i] >= count) return PSACX_EINVAL;
        for (int j = 0; j < i; ++j) distinct = distinct && devs[j] != devs[i];
    }
    psacx_multi* g = new psacx_multi();
    g->nranks = g->nlocal = ndev; g->first = 0;
    g->R.resize(ndev);
    for (int i = 0; i < ndev; ++i) {
        const int rc = make_rank(g, i, i, devs[i]);
        if (rc != PSACX_OK) { psacx_multi_destroy(g); return rc; }
    }
    // one RCCL communicator over the devices; ranks that share a device (and a single rank) exchange by copies.
    // A communicator that cannot be built is an error, not a silent change of transport (the flag PSACX_MULTI_NO_RCCL asks
    // for peer copies between distinct devices explicitly).
    g->force_wire = (flags & PSACX_MULTI_FORCE_WIRE) != 0;
    g->use_rccl = distinct && (ndev > 1 || g->force_wire) && !(flags & PSACX_MULTI_NO_RCCL);
    if (g->use_rccl) {
        std::string err;
        std::vector<ncclComm_t> comms(ndev);
        if (!rccl().load(err)) { psacx_multi_destroy(g); return PSACX_MULTI_EPEER; }
        if (rccl().CommInitAll(comms.data(), ndev, devs.data()) != ncclSuccess) {
            (void)hipGetLastError();
            psacx_multi_destroy(g);
            return PSACX_MULTI_EPEER;
        }
        for (int i = 0; i < ndev; ++i) g->R[i].comm = comms[i];
        g->transport = PSACX_TR_RCCL;
    }
    *out = g;
    return PSACX_OK;
}

int psacx_multi_unique_id(void* id128) {
    if (!id128) return PSACX_EINVAL;
    std::string err;
    if (!rccl().load(err)) return PSACX_MULTI_EPEER;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return PSACX_MULTI_EPEER;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, 128);
    return PSACX_OK;
}

int psacx_multi_create_rank(psacx_multi** out, int rank, int nranks, int device, const void* id128) {
    return psacx_multi_create_rank_ex(out, rank, nranks, device, id128, 0u, 0);
}

int psacx_multi_create_rank_ex(psacx_multi** out, int rank, int nranks, int device, const void* id128, uint32_t flags, uint64_t shm_box_bytes) {
    if (!out || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks || device < 0 || (flags & ~(PSACX_MULTI_FORCE_WIRE | PSACX_MULTI_SHM))) return PSACX_EINVAL;
    if (nranks > 1 && !id128) return PSACX_EINVAL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return PSACX_ENOGPU; }
    if (device >= count) return PSACX_EINVAL;
    psacx_multi* g = new psacx_multi();
    g->nranks = nranks; g->nlocal = 1; g->first = rank;
    g->R.resize(1);
    int rc = make_rank(g, 0, rank, device);
    if (rc != PSACX_OK) { psacx_multi_destroy(g); return rc; }
    g->force_wire = (flags & PSACX_MULTI_FORCE_WIRE) != 0;
    if (id128 && (flags & PSACX_MULTI_SHM)) {
        // one process per rank on one host, exchanges staged through shared memory (shm_link.hpp): ranks may share a device
        std::string err;
        if (!g->shm.open(rank, nranks, id128, err, (size_t)shm_box_bytes)) { g->err = err; psacx_multi_destroy(g); return PSACX_MULTI_EPEER; }
        g->transport = PSACX_TR_SHM;
    } else if (id128) {
        std::string err;
        ncclUniqueId id;
        std::memcpy(&id, id128, 128);
        if (!rccl().load(err)) { psacx_multi_destroy(g); return PSACX_MULTI_EPEER; }
        if (hipSetDevice(device) != hipSuccess || rccl().CommInitRank(&g->R[0].comm, nranks, id, rank) != ncclSuccess) {
            psacx_multi_destroy(g);
            return PSACX_MULTI_EPEER;
        }
        g->use_rccl = true;
        g->transport = PSACX_TR_RCCL;
    }
    *out = g;
    return PSACX_OK;
}

void psacx_multi_destroy(psacx_multi* g) {
    if (!g) return;
    for (auto& R : g->R) {
        if (!R.ctx) continue;
        (void)hipSetDevice(R.ctx->device);
        (void)hipStreamSynchronize(R.ctx->stream);
        if (R.comm_stream) (void)hipStreamSynchronize(R.comm_stream);
        if (R.comm) (void)rccl().CommDestroy(R.comm);
        if (R.d_scal) (void)hipFree(R.d_scal);
        if (R.ev_ready) (void)hipEventDestroy(R.ev_ready);
        if (R.ev_done) (void)hipEventDestroy(R.ev_done);
        for (auto& e : R.ex_ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        if (R.comm_stream) (void)hipStreamDestroy(R.comm_stream);
        psacx_destroy(R.ctx);
    }
    g->shm.timeout_s = 10.0;
    g->shm.close_link();
    delete g;
}

int psacx_multi_nranks(const psacx_multi* g) { return g ? g->nranks : 0; }
int psacx_multi_nlocal(const psacx_multi* g) { return g ? g->nlocal : 0; }
int psacx_multi_uses_rccl(const psacx_multi* g) { return g && g->use_rccl ? 1 : 0; }
const char* psacx_multi_last_error(const psacx_multi* g) { return g ? g->err.c_str() : ""; }

int psacx_multi_get_stats(const psacx_multi* g, psacx_stats* out, uint64_t* bytes_sent, uint64_t* exchanges, uint64_t* gathers) {
    if (!g || !out) return PSACX_EINVAL;
    *out = g->stats;
    if (bytes_sent) *bytes_sent = g->bytes_sent;
    if (exchanges) *exchanges = g->n_exchanges;
    if (gathers) *gathers = g->n_gathers;
    return PSACX_OK;
}

int psacx_multi_transport(const psacx_multi* g) { return g ? g->transport : -1; }
int psacx_multi_last_form(const psacx_multi* g) { return g ? (g->last_two_word ? 1 : 0) | (g->last_reduced ? 2 : 0) | (g->last_slice_inversion ? 4 : 0) | (g->last_one_word ? 16 : 0) | ((int)std::min<uint32_t>(g->last_tie_slabs, 255u) << 8) : 0; }

int psacx_multi_get_wire(const psacx_multi* g, uint64_t* sends, uint64_t* recvs, uint64_t* allgathers, double* exchange_ms) {
    if (!g) return PSACX_EINVAL;
    if (sends) *sends = g->wire_sends;
    if (recvs) *recvs = g->wire_recvs;
    if (allgathers) *allgathers = g->wire_gathers;
    if (exchange_ms) for (int i = 0; i < g->nlocal; ++i) exchange_ms[i] = g->R[i].exchange_ms;
    return PSACX_OK;
}

int psacx_multi_get_phases(const psacx_multi* g, char* buf, uint64_t cap) {
    if (!g || !buf || cap == 0) return PSACX_EINVAL;
    std::string s;
    for (const auto& ph : g->phases) { char t[160]; snprintf(t, sizeof(t), "%s%s=%.3f", s.empty() ? "" : ";", ph.first.c_str(), ph.second); s += t; }
    if (s.size() + 1 > cap) return PSACX_ERANGE;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return PSACX_OK;
}

int psacx_multi_configure(psacx_multi* g, int option, uint64_t value) {
    if (!g) return PSACX_EINVAL;
    switch (option) {
    case PSACX_MULTI_OPT_LAYOUT: if (value > 2) return PSACX_EINVAL; g->opt_layout = (int)value; return PSACX_OK;
    case PSACX_MULTI_OPT_SLAB: g->opt_slab = value; return PSACX_OK;
    case PSACX_MULTI_OPT_OUTPUT_SLACK: g->out_slack = value; return PSACX_OK;
    case PSACX_MULTI_OPT_TRACE: g->opt_trace = value != 0; return PSACX_OK;
    case PSACX_MULTI_OPT_WIRE_PIECE: g->opt_wire_piece = value; return PSACX_OK;
    case PSACX_MULTI_OPT_PIECES: if (value > 64) return PSACX_EINVAL; g->opt_pieces = (int)value; return PSACX_OK;
    case PSACX_MULTI_OPT_CHECK_CHUNKS: g->opt_check_chunks = value; return PSACX_OK;
    case PSACX_MULTI_OPT_GLOBAL_REFINE_SORT: g->opt_global_refine_sort = value != 0; return PSACX_OK;
    case PSACX_MULTI_OPT_ONE_STAGE: g->opt_one_stage = value != 0; return PSACX_OK;
    case PSACX_MULTI_OPT_TWO_WORD: if (value > 3) return PSACX_EINVAL; g->opt_two_word = (int)value; return PSACX_OK;
    case PSACX_MULTI_OPT_ONE_WORD: if (value > 2) return PSACX_EINVAL; g->opt_one_word = (int)value; return PSACX_OK;
    case PSACX_MULTI_OPT_NO_SLICES: g->opt_no_slices = value != 0; return PSACX_OK;
    case PSACX_MULTI_OPT_SLICE_WIDE: g->opt_slice_wide = value != 0; return PSACX_OK;
    case PSACX_MULTI_OPT_SLICE_SHAPE: g->opt_slice_wb = (unsigned)(value & 0xFF); g->opt_slice_s1 = (unsigned)((value >> 8) & 0xFF); g->opt_slice_step = value >> 16; return PSACX_OK;
    default: return PSACX_EINVAL;
    }
}

int psacx_multi_configure_from_env(psacx_multi* g) {
    if (!g) return PSACX_EINVAL;
    auto num = [](const char* name, uint64_t* v) { const char* e = psacx_debug_env(name); if (!e) return false; *v = strtoull(e, nullptr, 10); return true; };
    uint64_t v = 0;
    g->opt_trace = psacx_debug_env("PSACX_MULTI_TRACE") != nullptr;
    g->opt_wire_piece = num("PSACX_MULTI_WIRE_PIECE", &v) ? v : 0;
    g->opt_pieces = num("PSACX_MULTI_PIECES", &v) ? (int)std::min<uint64_t>(v, 64) : 0;
    g->opt_check_chunks = num("PSACX_MULTI_CHECK_CHUNKS", &v) ? v : 0;
    g->opt_global_refine_sort = psacx_debug_env("PSACX_MULTI_GLOBAL_REFINE_SORT") != nullptr;
    g->opt_one_stage = psacx_debug_env("PSACX_ONE_STAGE") != nullptr;
    g->opt_two_word = num("PSACX_MULTI_TWO_WORD", &v) ? (int)std::min<uint64_t>(v, 2) + 1 : 0;
    g->opt_one_word = num("PSACX_MULTI_ONE_WORD", &v) ? (int)std::min<uint64_t>(v, 1) + 1 : 0;
    g->opt_no_slices = psacx_debug_env("PSACX_MULTI_NO_SLICES") != nullptr;
    g->opt_slice_wide = psacx_debug_env("PSACX_SLICE_WIDE") != nullptr;
    g->opt_slice_wb = g->opt_slice_s1 = 0; g->opt_slice_step = 0;
    if (const char* e = psacx_debug_env("PSACX_SLICE_SHAPE")) { unsigned a = 0, b = 0; unsigned long long st = 0; if (sscanf(e, "%u,%u,%llu", &a, &b, &st) >= 1) { g->opt_slice_wb = a; g->opt_slice_s1 = b; g->opt_slice_step = st; } }
    if (num("PSACX_MULTI_DIET", &v) && v) g->opt_layout = 2;
    if (num("PSACX_MULTI_SLAB", &v)) g->opt_slab = v;
    for (auto& R : g->R) if (R.ctx) (void)psacx_configure_from_env(R.ctx);
    return PSACX_OK;
}

int psacx_multi_get_memory(const psacx_multi* g, uint64_t* peak_bytes, int* reduced, uint32_t* slab_rounds) {
    if (!g) return PSACX_EINVAL;
    if (peak_bytes) for (int i = 0; i < g->nlocal; ++i) peak_bytes[i] = g->R[i].ctx->pool_peak;
    if (reduced) *reduced = g->last_reduced ? 1 : 0;
    if (slab_rounds) *slab_rounds = g->last_slab_rounds;
    return PSACX_OK;
}

int psacx_multi_construct_dev_u32(psacx_multi* g, const uint8_t* const* t, const uint64_t* m, uint32_t k, uint32_t f, uint32_t* const* sa,
                                  uint32_t* const* isa, uint32_t* const* lcp) { return run_dev<uint32_t>(g, t, m, k, f, sa, isa, lcp); }
int psacx_multi_construct_dev_u64(psacx_multi* g, const uint8_t* const* t, const uint64_t* m, uint32_t k, uint32_t f, uint64_t* const* sa,
                                  uint64_t* const* isa, uint64_t* const* lcp) { return run_dev<uint64_t>(g, t, m, k, f, sa, isa, lcp); }
int psacx_multi_construct_gsa_dev_u32(psacx_multi* g, const uint8_t* const* t, const uint64_t* m, const uint64_t* off, uint64_t nstr, uint32_t k, uint32_t f,
                                      uint32_t* const* sa, uint32_t* const* isa, uint32_t* const* lcp) {
    return off ? run_dev<uint32_t>(g, t, m, k, f, sa, isa, lcp, off, nstr) : PSACX_EINVAL;
}
int psacx_multi_construct_gsa_dev_u64(psacx_multi* g, const uint8_t* const* t, const uint64_t* m, const uint64_t* off, uint64_t nstr, uint32_t k, uint32_t f,
                                      uint64_t* const* sa, uint64_t* const* isa, uint64_t* const* lcp) {
    return off ? run_dev<uint64_t>(g, t, m, k, f, sa, isa, lcp, off, nstr) : PSACX_EINVAL;
}
int psacx_multi_construct_gsa_u32(psacx_multi* g, const uint8_t* t, uint64_t n, const uint64_t* off, uint64_t nstr, uint32_t k, uint32_t f, uint32_t* sa,
                                  uint32_t* isa, uint32_t* lcp) { return off ? run_host<uint32_t>(g, t, n, k, f, sa, isa, lcp, nullptr, off, nstr) : PSACX_EINVAL; }
int psacx_multi_construct_gsa_u64(psacx_multi* g, const uint8_t* t, uint64_t n, const uint64_t* off, uint64_t nstr, uint32_t k, uint32_t f, uint64_t* sa,
                                  uint64_t* isa, uint64_t* lcp) { return off ? run_host<uint64_t>(g, t, n, k, f, sa, isa, lcp, nullptr, off, nstr) : PSACX_EINVAL; }
int psacx_multi_construct_u32(psacx_multi* g, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp) {
    return run_host<uint32_t>(g, t, n, k, f, sa, isa, lcp);
}
int psacx_multi_construct_u64(psacx_multi* g, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp) {
    return run_host<uint64_t>(g, t, n, k, f, sa, isa, lcp);
}

int psacx_multi_check_dev_u32(psacx_multi* g, const uint8_t* const* t, const uint64_t* m, const uint32_t* const* sa, const uint32_t* const* isa,
                              const uint32_t* const* lcp, uint64_t errors[4]) { return check_dev<uint32_t>(g, t, m, sa, isa, lcp, errors); }
int psacx_multi_check_dev_u64(psacx_multi* g, const uint8_t* const* t, const uint64_t* m, const uint64_t* const* sa, const uint64_t* const* isa,
                              const uint64_t* const* lcp, uint64_t errors[4]) { return check_dev<uint64_t>(g, t, m, sa, isa, lcp, errors); }

int psacx_multi_construct_lc_u32(psacx_multi* g, const uint8_t* text, uint64_t n, uint32_t k, uint32_t flags, uint32_t* SA, uint32_t* ISA,
                                 uint32_t* LCP, uint8_t* Lc) { return Lc ? run_host<uint32_t>(g, text, n, k, flags, SA, ISA, LCP, Lc) : PSACX_EINVAL; }
int psacx_multi_construct_lc_u64(psacx_multi* g, const uint8_t* text, uint64_t n, uint32_t k, uint32_t flags, uint64_t* SA, uint64_t* ISA,
                                 uint64_t* LCP, uint8_t* Lc) { return Lc ? run_host<uint64_t>(g, text, n, k, flags, SA, ISA, LCP, Lc) : PSACX_EINVAL; }
int psacx_multi_left_chars_dev_u32(psacx_multi* g, const uint8_t* const* t, const uint64_t* m, const uint32_t* const* sa, const uint32_t* const* lcp,
                                   uint8_t* const* lc) { return left_chars_dev<uint32_t>(g, t, m, sa, lcp, lc); }
int psacx_multi_left_chars_dev_u64(psacx_multi* g, const uint8_t* const* t, const uint64_t* m, const uint64_t* const* sa, const uint64_t* const* lcp,
                                   uint8_t* const* lc) { return left_chars_dev<uint64_t>(g, t, m, sa, lcp, lc); }

int psacx_multi_suffix_tree_dev_u32(psacx_multi* g, const uint8_t* const* t, const uint64_t* m, const uint32_t* const* sa, const uint32_t* const* lcp,
                                    uint64_t* const* nodes, uint32_t* sigma) { return suffix_tree_dev<uint32_t>(g, t, m, sa, lcp, nodes, sigma); }
int psacx_multi_suffix_tree_dev_u64(psacx_multi* g, const uint8_t* const* t, const uint64_t* m, const uint64_t* const* sa, const uint64_t* const* lcp,
                                    uint64_t* const* nodes, uint32_t* sigma) { return suffix_tree_dev<uint64_t>(g, t, m, sa, lcp, nodes, sigma); }

int psacx_multi_suffix_tree_u32(psacx_multi* g, const uint8_t* text, uint64_t n, const uint32_t* sa, const uint32_t* lcp, uint64_t* nodes, uint32_t* sigma) {
    return suffix_tree_host<uint32_t>(g, text, n, sa, lcp, nodes, sigma);
}
int psacx_multi_suffix_tree_u64(psacx_multi* g, const uint8_t* text, uint64_t n, const uint64_t* sa, const uint64_t* lcp, uint64_t* nodes, uint32_t* sigma) {
    return suffix_tree_host<uint64_t>(g, text, n, sa, lcp, nodes, sigma);
}

int psacx_multi_ansv_dev_u32(psacx_multi* g, const uint32_t* const* in, const uint64_t* m, int lt, int rt, uint64_t nonsv, uint64_t* const* l,
                             uint64_t* const* r) { return ansv_dev<uint32_t>(g, in, m, lt, rt, nonsv, l, r); }
int psacx_multi_ansv_dev_u64(psacx_multi* g, const uint64_t* const* in, const uint64_t* m, int lt, int rt, uint64_t nonsv, uint64_t* const* l,
                             uint64_t* const* r) { return ansv_dev<uint64_t>(g, in, m, lt, rt, nonsv, l, r); }

/* the rank-local psacx_ctx of local rank i (its device, compute stream and workspace), e.g. for psacx_dev_alloc */
psacx_ctx* psacx_multi_ctx(psacx_multi* g, int i) { return (g && i >= 0 && i < g->nlocal) ? g->R[i].ctx : nullptr; }

} // extern "C"
