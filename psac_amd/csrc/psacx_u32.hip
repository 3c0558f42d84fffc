// 32-bit index instantiation of the engine (n <= 2^32 - 2).
#include "construct.hpp"
namespace psacx {
int construct_dev_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp) {
    return construct_dispatch<uint32_t>(c, t, n, k, f, sa, isa, lcp);
}
int construct_host_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp) {
    return construct_host<uint32_t>(c, t, n, k, f, sa, isa, lcp);
}
int construct_lc_dev_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp, uint8_t* lc) {
    if (!lc) return PSACX_EINVAL;
    return construct_dispatch<uint32_t>(c, t, n, k, f | PSACX_LCP, sa, isa, lcp, lc);
}
int construct_lc_host_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp, uint8_t* lc) {
    if (!lc) return PSACX_EINVAL;
    return construct_host<uint32_t>(c, t, n, k, f | PSACX_LCP, sa, isa, lcp, lc);
}
int construct_gsa_dev_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* off, uint64_t m, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp) {
    return construct_gsa_dispatch<uint32_t>(c, t, n, off, m, k, f, sa, isa, lcp);
}
int construct_gsa_host_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* off, uint64_t m, uint32_t k, uint32_t f, uint32_t* sa, uint32_t* isa, uint32_t* lcp) {
    return construct_gsa_host<uint32_t>(c, t, n, off, m, k, f, sa, isa, lcp);
}
int pair_sort_dev_u32(psacx_ctx* c, uint32_t* b1, uint32_t* b2, uint32_t* idx, uint64_t n, uint32_t bits) {
    return pair_sort_dev<uint32_t>(c, b1, b2, idx, n, bits);
}
}
