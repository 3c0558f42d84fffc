// check.hip -- device-side verification of SA / ISA / LCP (the CLI's -c at sizes where a host
// check is impractical).  Follows check_SA (/root/reference/include/check_suffix_array.hpp:56-88:
// range, ISA[SA[i]] == i, order through the first character and the ranks of the suffixes one
// further) and d_check_sa's idea of a scalable checker (:207-267).  LCP entries are verified by
// direct character comparison, which is linear in sum(LCP): meant for texts with short repeats.
#include "engine.hpp"

namespace psacx {

// err[0]: SA out of range / not inverse of ISA, err[1]: order violations, err[2]: LCP mismatches,
// err[3]: LCP[0] != 0
template <typename T>
__global__ void check_kernel(const uint8_t* __restrict__ text, uint64_t n, const T* __restrict__ SA,
                             const T* __restrict__ ISA, const T* __restrict__ LCP, unsigned long long* __restrict__ err) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned e0 = 0, e1 = 0, e2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t b = SA[i];
        if (b >= n || (uint64_t)ISA[b] != i) { ++e0; continue; }
        if (i == 0) { if (LCP && LCP[0] != 0) atomicAdd(&err[3], 1ull); continue; }
        const uint64_t a = SA[i - 1];
        if (a >= n) continue;                       // counted by the thread that owns i - 1
        const uint8_t ca = text[a], cb = text[b];
        bool ok = ca < cb;
        if (ca == cb) ok = (a + 1 == n) || (b + 1 < n && ISA[a + 1] < ISA[b + 1]);
        if (!ok) ++e1;
        if (LCP) {
            const uint64_t l = LCP[i];
            uint64_t h = 0;
            while (h < l && a + h < n && b + h < n && text[a + h] == text[b + h]) ++h;
            const bool more = (a + h < n && b + h < n && text[a + h] == text[b + h]);
            if (h != l || more) ++e2;
        }
    }
    e0 = wave_reduce<uint32_t>(e0, OpSum()); e1 = wave_reduce<uint32_t>(e1, OpSum()); e2 = wave_reduce<uint32_t>(e2, OpSum());
    if (lane_id() == 0) {
        if (e0) atomicAdd(&err[0], (unsigned long long)e0);
        if (e1) atomicAdd(&err[1], (unsigned long long)e1);
        if (e2) atomicAdd(&err[2], (unsigned long long)e2);
    }
}

template <typename T>
int check_dev(psacx_ctx* c, const uint8_t* text, uint64_t n, const T* sa, const T* isa, const T* lcp, uint64_t* errors) {
    if (!c || !text || !sa || !isa || !errors || n == 0) return PSACX_EINVAL;
    PSACX_HIP(c, hipSetDevice(c->device));
    PSACX_TRY(ensure_slab(c, 4096));
    unsigned long long* d = reinterpret_cast<unsigned long long*>(c->slab);
    PSACX_HIP(c, hipMemsetAsync(d, 0, 32, c->stream));
    hipLaunchKernelGGL((check_kernel<T>), dim3(grid_for(c, n, 256, 16)), dim3(256), 0, c->stream, text, n, sa, isa, lcp, d);
    PSACX_HIP(c, hipGetLastError());
    PSACX_HIP(c, hipMemcpyAsync(errors, d, 32, hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    return PSACX_OK;
}

int check_dev_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint32_t* sa, const uint32_t* isa, const uint32_t* lcp, uint64_t* e) {
    return check_dev<uint32_t>(c, t, n, sa, isa, lcp, e);
}
int check_dev_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* sa, const uint64_t* isa, const uint64_t* lcp, uint64_t* e) {
    return check_dev<uint64_t>(c, t, n, sa, isa, lcp, e);
}

} // namespace psacx
