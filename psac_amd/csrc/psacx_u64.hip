// 64-bit index instantiation of the engine.
#include "construct.hpp"
namespace psacx {
int construct_dev_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp) {
    return construct_dispatch<uint64_t>(c, t, n, k, f, sa, isa, lcp);
}
int construct_host_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp) {
    return construct_host<uint64_t>(c, t, n, k, f, sa, isa, lcp);
}
int construct_lc_dev_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp, uint8_t* lc) {
    if (!lc) return PSACX_EINVAL;
    return construct_dispatch<uint64_t>(c, t, n, k, f | PSACX_LCP, sa, isa, lcp, lc);
}
int construct_lc_host_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp, uint8_t* lc) {
    if (!lc) return PSACX_EINVAL;
    return construct_host<uint64_t>(c, t, n, k, f | PSACX_LCP, sa, isa, lcp, lc);
}
int construct_gsa_dev_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* off, uint64_t m, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp) {
    return construct_gsa_dispatch<uint64_t>(c, t, n, off, m, k, f, sa, isa, lcp);
}
int construct_gsa_host_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* off, uint64_t m, uint32_t k, uint32_t f, uint64_t* sa, uint64_t* isa, uint64_t* lcp) {
    return construct_gsa_host<uint64_t>(c, t, n, off, m, k, f, sa, isa, lcp);
}
int pair_sort_dev_u64(psacx_ctx* c, uint64_t* b1, uint64_t* b2, uint64_t* idx, uint64_t n, uint32_t bits) {
    return pair_sort_dev<uint64_t>(c, b1, b2, idx, n, bits);
}
}
