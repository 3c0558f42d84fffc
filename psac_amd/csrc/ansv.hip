// ansv.hip -- all nearest smaller values over an integer array (the LCP array).
// Placeholder translation unit: the kernels land in a later commit of this round.
#include "engine.hpp"
namespace psacx {
int ansv_host_u32(psacx_ctx*, const uint32_t*, uint64_t, int, int, uint64_t, uint64_t*, uint64_t*) { return PSACX_EINVAL; }
int ansv_host_u64(psacx_ctx*, const uint64_t*, uint64_t, int, int, uint64_t, uint64_t*, uint64_t*) { return PSACX_EINVAL; }
}
