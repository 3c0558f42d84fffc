// dist_ops.hip -- extern "C" surface of the step-level ops (include/psacx_ops.h) over dist_ops.hpp.
#include "dist_ops.hpp"

using namespace psacx;

extern "C" {

int psacx_op_char_hist(psacx_ctx* c, const uint8_t* text, uint64_t n, uint64_t* hist) {
    OP_PROLOGUE(c);
    PSACX_HIP(c, hipMemsetAsync(hist, 0, 256 * 8, c->stream));
    if (n) {
        hipLaunchKernelGGL((char_hist_kernel<256>), dim3(grid_for(c, n / 16 + 1, 256, 8)), dim3(256), 0, c->stream, text, n,
                           reinterpret_cast<unsigned long long*>(hist));
        PSACX_HIP(c, hipGetLastError());
    }
    return PSACX_OK;
}

#define PSACX_OPS_IMPL(T, S)                                                                                   \
    int psacx_op_make_keys_##S(psacx_ctx* c, const uint8_t* t, uint64_t m, uint64_t tl, const uint16_t* codes, \
                               uint32_t l, uint32_t c1, uint32_t c2, T* k1, T* k2) {                           \
        return op_make_keys<T>(c, t, m, tl, codes, l, c1, c2, k1, k2);                                          \
    }                                                                                                          \
    int psacx_op_iota_##S(psacx_ctx* c, T* out, uint64_t m, uint64_t start) {                                  \
        OP_PROLOGUE(c); SIMPLE_LAUNCH(c, (iota_from_kernel<T>), m, out, m, start); return PSACX_OK;            \
    }                                                                                                          \
    int psacx_op_pair_sort_##S(psacx_ctx* c, T* k1, T* k2, T* v, T* a1, T* a2, T* av, uint64_t n, uint32_t b1, \
                               uint32_t b2, int32_t* where) {                                                  \
        return op_pair_sort<T>(c, k1, k2, v, a1, a2, av, n, b1, b2, where);                                     \
    }                                                                                                          \
    int psacx_op_split_by_##S(psacx_ctx* c, const T* k1, const T* k2, const T* v, uint64_t n, const uint64_t* a, \
                              const uint64_t* b, const uint64_t* r, const uint64_t* x, uint32_t ns, uint64_t me, \
                              T* o1, T* o2, T* ov, uint64_t* cs) {                                            \
        return op_split_by<T>(c, k1, k2, v, n, a, b, r, x, ns, me, o1, o2, ov, cs);                              \
    }                                                                                                          \
    int psacx_op_put_perm_##S(psacx_ctx* c, T* b, const T* g, uint64_t cnt, uint64_t off, const T* v, T* s1,   \
                              T* s2, T* s3, T* s4) {                                                           \
        return op_put_perm<T>(c, b, g, cnt, off, v, s1, s2, s3, s4);                                            \
    }                                                                                                          \
    int psacx_op_pair_bounds_##S(psacx_ctx* c, const T* s1, const T* s2, uint64_t n, const uint64_t* q1,       \
                                 const uint64_t* q2, uint32_t nq, int us, uint64_t* lb, uint64_t* ub) {        \
        return op_pair_bounds<T>(c, s1, s2, n, q1, q2, nq, us, lb, ub);                                         \
    }                                                                                                          \
    int psacx_op_owners_##S(psacx_ctx* c, const T* g, uint64_t cnt, uint64_t n, uint32_t P, T* out) {          \
        OP_PROLOGUE(c); SIMPLE_LAUNCH(c, (owners_kernel<T>), cnt, g, cnt, make_dist(n, P), out); return PSACX_OK; \
    }                                                                                                          \
    int psacx_op_take_##S(psacx_ctx* c, const T* b, const T* g, uint64_t cnt, uint64_t off, uint64_t n, T* o) { \
        OP_PROLOGUE(c); SIMPLE_LAUNCH(c, (take_kernel<T>), cnt, b, g, cnt, off, n, o); return PSACX_OK;         \
    }                                                                                                          \
    int psacx_op_put_##S(psacx_ctx* c, T* b, const T* g, uint64_t cnt, uint64_t off, const T* v, int64_t d) {  \
        OP_PROLOGUE(c); SIMPLE_LAUNCH(c, (put_kernel<T>), cnt, b, g, cnt, off, v, d); return PSACX_OK;          \
    }                                                                                                          \
    int psacx_op_add_scalar_##S(psacx_ctx* c, const T* in, uint64_t cnt, uint64_t s, uint64_t cap, T* out) {   \
        OP_PROLOGUE(c); SIMPLE_LAUNCH(c, (add_scalar_kernel<T>), cnt, in, cnt, s, cap, out); return PSACX_OK;   \
    }                                                                                                          \
    int psacx_op_finish_b2_##S(psacx_ctx* c, const T* a, const T* q, uint64_t cnt, uint64_t n, T* out) {       \
        OP_PROLOGUE(c); SIMPLE_LAUNCH(c, (finish_b2_kernel<T>), cnt, a, q, cnt, n, out); return PSACX_OK;       \
    }                                                                                                          \
    int psacx_op_last_head_##S(psacx_ctx* c, int mode, const T* s1, const T* s2, const T* s3, uint64_t cnt,    \
                               uint64_t n, uint32_t l, uint32_t c1, uint32_t c2, const psacx_boundary* bd,     \
                               uint64_t* out) {                                                                \
        return op_last_head<T>(c, mode, s1, s2, s3, cnt, n, l, c1, c2, bd, out);                                \
    }                                                                                                          \
    int psacx_op_rebucket_first_##S(psacx_ctx* c, const T* s1, const T* s2, const T* sa, uint64_t cnt,         \
                                    uint64_t n, uint32_t l, uint32_t c1, uint32_t c2, const psacx_boundary* bd, \
                                    T* bsa, T* lcp, uint64_t* nact, uint64_t* nunf) {                          \
        return op_rebucket_first<T>(c, s1, s2, sa, cnt, n, l, c1, c2, bd, bsa, lcp, nact, nunf);                \
    }                                                                                                          \
    int psacx_op_rebucket_refine_##S(psacx_ctx* c, const T* t1, const T* t2, const T* tv, const T* pos,        \
                                     uint64_t cnt, uint64_t n, uint64_t h, const psacx_boundary* bd, T* sab,   \
                                     T* bsab, T* lcpb, T* ids, T* qa, T* ql, T* qh, uint64_t* nq,              \
                                     uint64_t* nact, uint64_t* nunf) {                                         \
        return op_rebucket_refine<T>(c, t1, t2, tv, pos, cnt, n, h, bd, sab, bsab, lcpb, ids, qa, ql, qh, nq, nact, nunf); \
    }                                                                                                          \
    int psacx_op_compact_##S(psacx_ctx* c, const T* ids, const T* pos, uint64_t cnt, uint64_t off,             \
                             uint64_t pid, uint64_t nid, T* out, uint64_t* n_out) {                            \
        return op_compact<T>(c, ids, pos, cnt, off, pid, nid, out, n_out);                                      \
    }                                                                                                          \
    int psacx_op_block_min_##S(psacx_ctx* c, const T* b, uint64_t m, uint64_t* out) {                          \
        return op_block_min<T>(c, b, m, out);                                                                   \
    }                                                                                                          \
    int psacx_op_range_min_##S(psacx_ctx* c, const T* b, uint64_t m, const T* lo, const T* hi, uint64_t cnt,   \
                               uint64_t off, T* out) {                                                         \
        return op_range_min<T>(c, b, m, lo, hi, cnt, off, out);                                                 \
    }                                                                                                          \
    int psacx_op_rmq_split_##S(psacx_ctx* c, const T* lo, const T* hi, uint64_t cnt, uint64_t n, uint32_t P,   \
                               T* o1, T* l1, T* h1, T* o2, T* l2, T* h2, T* ra, T* rb) {                       \
        OP_PROLOGUE(c);                                                                                        \
        SIMPLE_LAUNCH(c, (rmq_split_kernel<T>), cnt, lo, hi, cnt, make_dist(n, P), o1, l1, h1, o2, l2, h2, ra, rb); \
        return PSACX_OK;                                                                                       \
    }                                                                                                          \
    int psacx_op_rmq_combine_##S(psacx_ctx* c, const T* a1, const T* a2, const T* ra, const T* rb,             \
                                 uint64_t cnt, const uint64_t* mins, uint32_t P, T* out) {                     \
        OP_PROLOGUE(c);                                                                                        \
        if (P > 64) return PSACX_EINVAL;                                                                       \
        RankMins rm; for (uint32_t i = 0; i < 64; ++i) rm.v[i] = i < P ? mins[i] : ~0ull;                      \
        SIMPLE_LAUNCH(c, (rmq_combine_kernel<T>), cnt, a1, a2, ra, rb, cnt, rm, out);                          \
        return PSACX_OK;                                                                                       \
    }                                                                                                          \
    int psacx_op_nsv_from_##S(psacx_ctx* c, const T* b, uint64_t m, uint64_t off, const int64_t* start,        \
                              const T* thr, uint64_t cnt, int strict, int left, T* oi, T* ov) {                \
        return op_nsv_from<T>(c, b, m, off, reinterpret_cast<const long long*>(start), thr, cnt, strict, left, oi, ov); \
    }                                                                                                          \
    int psacx_op_lcp_apply_##S(psacx_ctx* c, T* b, const T* at, uint64_t cnt, uint64_t off, const T* mins,     \
                               uint64_t h) {                                                                   \
        OP_PROLOGUE(c); SIMPLE_LAUNCH(c, (lcp_apply_kernel<T>), cnt, b, at, cnt, off, mins, h); return PSACX_OK; \
    }

PSACX_OPS_IMPL(uint32_t, u32)
PSACX_OPS_IMPL(uint64_t, u64)

} // extern "C"
