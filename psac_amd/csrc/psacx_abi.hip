// psacx_abi.hip -- extern "C" surface of libpsacx.so (include/psacx.h).
#include "engine.hpp"

namespace psacx {
int construct_dev_u32(psacx_ctx*, const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*);
int construct_dev_u64(psacx_ctx*, const uint8_t*, uint64_t, uint32_t, uint32_t, uint64_t*, uint64_t*, uint64_t*);
int construct_host_u32(psacx_ctx*, const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*);
int construct_gsa_host_u32(psacx_ctx*, const uint8_t*, uint64_t, const uint64_t*, uint64_t, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*);
int construct_gsa_dev_u32(psacx_ctx*, const uint8_t*, uint64_t, const uint64_t*, uint64_t, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*);
int construct_gsa_host_u64(psacx_ctx*, const uint8_t*, uint64_t, const uint64_t*, uint64_t, uint32_t, uint32_t, uint64_t*, uint64_t*, uint64_t*);
int construct_gsa_dev_u64(psacx_ctx*, const uint8_t*, uint64_t, const uint64_t*, uint64_t, uint32_t, uint32_t, uint64_t*, uint64_t*, uint64_t*);
int construct_lc_host_u32(psacx_ctx*, const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint8_t*);
int construct_lc_host_u64(psacx_ctx*, const uint8_t*, uint64_t, uint32_t, uint32_t, uint64_t*, uint64_t*, uint64_t*, uint8_t*);
int construct_lc_dev_u32(psacx_ctx*, const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint8_t*);
int construct_lc_dev_u64(psacx_ctx*, const uint8_t*, uint64_t, uint32_t, uint32_t, uint64_t*, uint64_t*, uint64_t*, uint8_t*);
int construct_host_u64(psacx_ctx*, const uint8_t*, uint64_t, uint32_t, uint32_t, uint64_t*, uint64_t*, uint64_t*);
int pair_sort_dev_u32(psacx_ctx*, uint32_t*, uint32_t*, uint32_t*, uint64_t, uint32_t);
int pair_sort_dev_u64(psacx_ctx*, uint64_t*, uint64_t*, uint64_t*, uint64_t, uint32_t);
int check_dev_u32(psacx_ctx*, const uint8_t*, uint64_t, const uint32_t*, const uint32_t*, const uint32_t*, uint64_t*);
int check_dev_u64(psacx_ctx*, const uint8_t*, uint64_t, const uint64_t*, const uint64_t*, const uint64_t*, uint64_t*);
int synth_text_dev(psacx_ctx*, uint8_t*, uint64_t, uint64_t, int, uint64_t, uint64_t);
int suffix_tree_host_u32(psacx_ctx*, const uint8_t*, uint64_t, const uint32_t*, const uint32_t*, uint64_t*, uint32_t*);
int suffix_tree_host_u64(psacx_ctx*, const uint8_t*, uint64_t, const uint64_t*, const uint64_t*, uint64_t*, uint32_t*);
int ansv_host_u32(psacx_ctx*, const uint32_t*, uint64_t, int, int, uint64_t, uint64_t*, uint64_t*);
int ansv_host_u64(psacx_ctx*, const uint64_t*, uint64_t, int, int, uint64_t, uint64_t*, uint64_t*);
int ansv_dev_u32(psacx_ctx*, const uint32_t*, uint64_t, int, int, uint64_t, uint64_t*, uint64_t*);
int ansv_dev_u64(psacx_ctx*, const uint64_t*, uint64_t, int, int, uint64_t, uint64_t*, uint64_t*);
}

using namespace psacx;

extern "C" {

int psacx_create(psacx_ctx** out, int device, void* stream) {
    if (!out || device < 0) return PSACX_EINVAL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return PSACX_ENOGPU; }
    if (device >= count) return PSACX_EINVAL;
    psacx_ctx* c = new psacx_ctx();
    c->device = device;
    std::memset(&c->stats, 0, sizeof(c->stats));
    if (hipSetDevice(device) != hipSuccess) { delete c; return PSACX_EHIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        c->n_cu = prop.multiProcessorCount;
    if (stream == PSACX_STREAM_DEFAULT) { c->stream = nullptr; c->own_stream = false; }
    else if (stream) { c->stream = reinterpret_cast<hipStream_t>(stream); c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return PSACX_EHIP; }
        c->own_stream = true;
    }
    *out = c;
    return PSACX_OK;
}

void psacx_destroy(psacx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& e : c->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (c->slab) (void)hipFree(c->slab);
    if (c->aux) (void)hipFree(c->aux);
    if (c->io) (void)hipFree(c->io);
    pool_flush(c);
    delete c->pool;
    for (int i = 0; i < psacx_ctx::STAGE_SLOTS; ++i) {
        if (c->stage[i]) (void)hipHostFree(c->stage[i]);
        if (c->stage_ev[i]) (void)hipEventDestroy(c->stage_ev[i]);
    }
    if (c->dstage) (void)hipFree(c->dstage);
    for (int i = 0; i < 2; ++i) if (c->copy_stream[i]) (void)hipStreamDestroy(c->copy_stream[i]);
    for (int i = 0; i < psacx_ctx::STAGE_SLOTS; ++i) if (c->narrow_ev[i]) (void)hipEventDestroy(c->narrow_ev[i]);
    if (c->early_ev) (void)hipEventDestroy(c->early_ev);
    if (c->early_stream) (void)hipStreamDestroy(c->early_stream);
    if (c->early_word) (void)hipFree(c->early_word);
    delete c->hpool;
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* psacx_strerror(int code) {
    switch (code) {
        case PSACX_OK: return "ok";
        case PSACX_EINVAL: return "invalid argument";
        case PSACX_ERANGE: return "input too long for the index type";
        case PSACX_ENOMEM: return "out of device memory";
        case PSACX_EHIP: return "HIP runtime error";
        case PSACX_EDEVICE: return "device-side failure (look-back timeout)";
        case PSACX_ENOGPU: return "no HIP device available";
        default: return "unknown error";
    }
}

const char* psacx_last_hip_error(const psacx_ctx* c) { return c ? c->hip_err.c_str() : ""; }

int psacx_configure(psacx_ctx* c, int option, uint64_t value) {
    if (!c) return PSACX_EINVAL;
    psacx::Knobs& k = c->knobs;
    switch (option) {
    case PSACX_OPT_RESET: k = psacx::Knobs(); return PSACX_OK;
    case PSACX_OPT_FORCE_DIET: k.force_diet = value != 0; return PSACX_OK;
    case PSACX_OPT_DIET_CAP: k.diet_cap = value; return PSACX_OK;
    case PSACX_OPT_ONE_STAGE: k.one_stage = value != 0; return PSACX_OK;
    case PSACX_OPT_TIES_RADIX: k.ties_radix = value != 0; return PSACX_OK;
    case PSACX_OPT_NO_ONE_WORD: k.no_one_word = value != 0; return PSACX_OK;
    case PSACX_OPT_ONE_WORD_ALWAYS: k.one_word_always = value != 0; return PSACX_OK;
    case PSACX_OPT_ONE_WORD_MIN: if (value > 62) return PSACX_EINVAL; k.one_word_min = value ? (unsigned)std::max<uint64_t>(16, value) : 24u; return PSACX_OK;
    case PSACX_OPT_WIDEN_LAST: k.widen_last = value != 0; return PSACX_OK;
    case PSACX_OPT_NO_DIGIT_BYTES: k.no_digit_bytes = value != 0; return PSACX_OK;
    case PSACX_OPT_NO_BUCKET_SORT: k.no_bucket_sort = value != 0; return PSACX_OK;
    case PSACX_OPT_ISA_UPDATE: if (value > 2) return PSACX_EINVAL; k.isa_update = (int)value; return PSACX_OK;
    case PSACX_OPT_GATHER: if (value > 2) return PSACX_EINVAL; k.gather = (int)value; return PSACX_OK;
    case PSACX_OPT_NO_HEAVY: k.no_heavy = value != 0; return PSACX_OK;
    case PSACX_OPT_NO_WHOLE: k.no_whole = value != 0; return PSACX_OK;
    case PSACX_OPT_NO_LAZY_RANKS: k.no_lazy_ranks = value != 0; return PSACX_OK;
    case PSACX_OPT_NO_EARLY_OUT: k.no_early_out = value != 0; return PSACX_OK;
    default: return PSACX_EINVAL;
    }
}

const char* psacx_debug_env(const char* name) { return name ? getenv(name) : nullptr; }

int psacx_configure_from_env(psacx_ctx* c) {
    if (!c) return PSACX_EINVAL;
    static const struct { const char* name; int option; } flags[] = {
        {"PSACX_FORCE_DIET", PSACX_OPT_FORCE_DIET}, {"PSACX_ONE_STAGE", PSACX_OPT_ONE_STAGE}, {"PSACX_TIES_RADIX", PSACX_OPT_TIES_RADIX},
        {"PSACX_NO_ONE_WORD", PSACX_OPT_NO_ONE_WORD}, {"PSACX_ONE_WORD_ALWAYS", PSACX_OPT_ONE_WORD_ALWAYS}, {"PSACX_WIDEN_LAST", PSACX_OPT_WIDEN_LAST},
        {"PSACX_NO_DIGIT_BYTES", PSACX_OPT_NO_DIGIT_BYTES}, {"PSACX_NO_BUCKET_SORT", PSACX_OPT_NO_BUCKET_SORT}, {"PSACX_NO_HEAVY", PSACX_OPT_NO_HEAVY},
        {"PSACX_NO_WHOLE", PSACX_OPT_NO_WHOLE}, {"PSACX_NO_LAZY_RANKS", PSACX_OPT_NO_LAZY_RANKS},
        {"PSACX_NO_EARLY_OUT", PSACX_OPT_NO_EARLY_OUT}};
    (void)psacx_configure(c, PSACX_OPT_RESET, 0);
    for (const auto& f : flags) if (psacx_debug_env(f.name)) (void)psacx_configure(c, f.option, 1);
    if (const char* e = psacx_debug_env("PSACX_DIET_CAP")) (void)psacx_configure(c, PSACX_OPT_DIET_CAP, strtoull(e, nullptr, 10));
    if (const char* e = psacx_debug_env("PSACX_ONE_WORD_MIN")) (void)psacx_configure(c, PSACX_OPT_ONE_WORD_MIN, strtoull(e, nullptr, 10));
    if (const char* e = psacx_debug_env("PSACX_ISA_UPDATE")) (void)psacx_configure(c, PSACX_OPT_ISA_UPDATE, e[0] == 's' ? 1 : 2);
    if (const char* e = psacx_debug_env("PSACX_GATHER")) (void)psacx_configure(c, PSACX_OPT_GATHER, e[0] == 'f' ? 1 : 2);
    return PSACX_OK;
}

int psacx_trim(psacx_ctx* c) {
    if (!c) return PSACX_EINVAL;
    PSACX_HIP(c, hipSetDevice(c->device));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    if (c->slab) { PSACX_HIP(c, hipFree(c->slab)); c->slab = nullptr; c->slab_bytes = 0; }
    if (c->aux) { PSACX_HIP(c, hipFree(c->aux)); c->aux = nullptr; c->aux_bytes = 0; }
    if (c->io) { PSACX_HIP(c, hipFree(c->io)); c->io = nullptr; c->io_bytes = 0; }
    if (c->dstage) { PSACX_HIP(c, hipFree(c->dstage)); c->dstage = nullptr; }
    // ... and what the host-pointer path keeps on the host side: the ring of pinned buffers and the threads that widen the results
    // (a process that drives several contexts one after the other would otherwise hold them for every context it ever used)
    for (int i = 0; i < psacx_ctx::STAGE_SLOTS; ++i) {
        if (c->stage[i]) { (void)hipHostFree(c->stage[i]); c->stage[i] = nullptr; }
        if (c->stage_ev[i]) { (void)hipEventDestroy(c->stage_ev[i]); c->stage_ev[i] = nullptr; }
    }
    delete c->hpool; c->hpool = nullptr;
    pool_flush(c);
    return PSACX_OK;
}

int psacx_construct_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp) {
    return construct_host_u32(c, t, n, k, f, sa, isa, lcp);
}
int psacx_construct_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp) {
    return construct_host_u64(c, t, n, k, f, sa, isa, lcp);
}
int psacx_construct_dev_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp) {
    return construct_dev_u32(c, t, n, k, f, sa, isa, lcp);
}
int psacx_construct_dev_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp) {
    return construct_dev_u64(c, t, n, k, f, sa, isa, lcp);
}

int psacx_construct_lc_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp, uint8_t* lc) {
    return construct_lc_host_u32(c, t, n, k, f, sa, isa, lcp, lc);
}
int psacx_construct_lc_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp, uint8_t* lc) {
    return construct_lc_host_u64(c, t, n, k, f, sa, isa, lcp, lc);
}
int psacx_construct_lc_dev_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp, uint8_t* lc) {
    return construct_lc_dev_u32(c, t, n, k, f, sa, isa, lcp, lc);
}
int psacx_construct_lc_dev_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp, uint8_t* lc) {
    return construct_lc_dev_u64(c, t, n, k, f, sa, isa, lcp, lc);
}

int psacx_construct_gsa_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* off, uint64_t m, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp) {
    return construct_gsa_host_u32(c, t, n, off, m, k, f, sa, isa, lcp);
}
int psacx_construct_gsa_dev_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* off, uint64_t m, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp) {
    return construct_gsa_dev_u32(c, t, n, off, m, k, f, sa, isa, lcp);
}
int psacx_construct_gsa_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* off, uint64_t m, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp) {
    return construct_gsa_host_u64(c, t, n, off, m, k, f, sa, isa, lcp);
}
int psacx_construct_gsa_dev_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* off, uint64_t m, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp) {
    return construct_gsa_dev_u64(c, t, n, off, m, k, f, sa, isa, lcp);
}

int psacx_profile(psacx_ctx* c, int on) {
    if (!c) return PSACX_EINVAL;
    c->profile_ops = on != 0;
    if (on) std::memset(&c->stats, 0, sizeof(c->stats));
    return PSACX_OK;
}

int psacx_get_stats(const psacx_ctx* c, psacx_stats* out) {
    if (!c || !out) return PSACX_EINVAL;
    *out = c->stats;
    return PSACX_OK;
}

int psacx_pair_sort_dev_u32(psacx_ctx* c, uint32_t* b1, uint32_t* b2, uint32_t* idx, uint64_t n, uint32_t bits) {
    return pair_sort_dev_u32(c, b1, b2, idx, n, bits);
}
int psacx_pair_sort_dev_u64(psacx_ctx* c, uint64_t* b1, uint64_t* b2, uint64_t* idx, uint64_t n, uint32_t bits) {
    return pair_sort_dev_u64(c, b1, b2, idx, n, bits);
}

int psacx_check_dev_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint32_t* sa, const uint32_t* isa, const uint32_t* lcp, uint64_t* e) {
    return check_dev_u32(c, t, n, sa, isa, lcp, e);
}
int psacx_check_dev_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* sa, const uint64_t* isa, const uint64_t* lcp, uint64_t* e) {
    return check_dev_u64(c, t, n, sa, isa, lcp, e);
}

int psacx_synth_text_dev(psacx_ctx* c, uint8_t* d_text, uint64_t n, uint64_t first, int kind, uint64_t seed, uint64_t period) {
    return synth_text_dev(c, d_text, n, first, kind, seed, period);
}
int psacx_rand_dna(uint8_t* out, uint64_t n, int seed) {
    if (!out) return PSACX_EINVAL;
    srand(1337u * (unsigned)seed);
    for (uint64_t i = 0; i < n; ++i) out[i] = (uint8_t)"ACGT"[rand() % 4];
    return PSACX_OK;
}

int psacx_ansv_dev_u32(psacx_ctx* c, const uint32_t* in, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* l, uint64_t* r) {
    return ansv_dev_u32(c, in, n, lt, rt, nonsv, l, r);
}
int psacx_ansv_dev_u64(psacx_ctx* c, const uint64_t* in, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* l, uint64_t* r) {
    return ansv_dev_u64(c, in, n, lt, rt, nonsv, l, r);
}
int psacx_ansv_u32(psacx_ctx* c, const uint32_t* in, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* l, uint64_t* r) {
    return ansv_host_u32(c, in, n, lt, rt, nonsv, l, r);
}
int psacx_ansv_u64(psacx_ctx* c, const uint64_t* in, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* l, uint64_t* r) {
    return ansv_host_u64(c, in, n, lt, rt, nonsv, l, r);
}

int psacx_suffix_tree_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint32_t* sa, const uint32_t* lcp, uint64_t* nodes, uint32_t* sg) {
    return suffix_tree_host_u32(c, t, n, sa, lcp, nodes, sg);
}
int psacx_suffix_tree_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* sa, const uint64_t* lcp, uint64_t* nodes, uint32_t* sg) {
    return suffix_tree_host_u64(c, t, n, sa, lcp, nodes, sg);
}

int psacx_dev_alloc(psacx_ctx* c, void** out, uint64_t bytes) {
    if (!c || !out) return PSACX_EINVAL;
    PSACX_HIP(c, hipSetDevice(c->device));
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) { c->hip_err = std::string("hipMalloc: ") + hipGetErrorString(e); (void)hipGetLastError(); return PSACX_ENOMEM; }
    return PSACX_OK;
}
int psacx_dev_free(psacx_ctx* c, void* p) {
    if (!c) return PSACX_EINVAL;
    PSACX_HIP(c, hipSetDevice(c->device));
    PSACX_HIP(c, hipFree(p));
    return PSACX_OK;
}
int psacx_copy_h2d(psacx_ctx* c, void* dst, const void* src, uint64_t bytes) {
    if (!c) return PSACX_EINVAL;
    PSACX_HIP(c, hipSetDevice(c->device));
    PSACX_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    return PSACX_OK;
}
int psacx_copy_d2h(psacx_ctx* c, void* dst, const void* src, uint64_t bytes) {
    if (!c) return PSACX_EINVAL;
    PSACX_HIP(c, hipSetDevice(c->device));
    PSACX_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    return PSACX_OK;
}
int psacx_sync(psacx_ctx* c) {
    if (!c) return PSACX_EINVAL;
    PSACX_HIP(c, hipSetDevice(c->device));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    return PSACX_OK;
}

} // extern "C"
