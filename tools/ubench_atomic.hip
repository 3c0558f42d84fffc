// Throughput of returning device-scope atomicAdd on MI355X over NB counters (stride PAD words), one atomic per thread:
// what a reservation per (window, bucket) costs (heavy_keys.hpp).  hipcc --offload-arch=gfx950 -O3 -o tools/ubench_atomic tools/ubench_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE>
__global__ void k(unsigned long long* ctr, unsigned nb, unsigned pad, uint64_t* out, unsigned reps) {
    uint64_t acc = 0;
    for (unsigned r = 0; r < reps; ++r) {
        const unsigned b = (threadIdx.x + r * 37u + blockIdx.x * 11u) % nb;
        if (MODE == 0) acc += atomicAdd(&ctr[(size_t)b * pad], 16ull);
        else if (MODE == 1) { atomicAdd(&ctr[(size_t)b * pad], 16ull); }                       // not returning
        else acc += __hip_atomic_fetch_add(&ctr[(size_t)b * pad], 16ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (acc == 0x1234567) out[0] = acc;
}
int main() {
    unsigned long long* ctr; uint64_t* out;
    CK(hipMalloc(&ctr, 4096 * 64 * 8)); CK(hipMalloc(&out, 64)); CK(hipMemset(ctr, 0, 4096 * 64 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = 65536, reps = 1;
    for (int mode = 0; mode < 3; ++mode)
        for (unsigned nb : {1024u, 256u, 4096u})
            for (unsigned pad : {1u, 8u, 16u, 32u}) {
                auto fn = [&] { if (mode == 0) k<0><<<grid, 1024>>>(ctr, nb, pad, out, reps); else if (mode == 1) k<1><<<grid, 1024>>>(ctr, nb, pad, out, reps); else k<2><<<grid, 1024>>>(ctr, nb, pad, out, reps); };
                fn(); CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
                const double n = (double)grid * 1024 * reps;
                printf("%-22s %4u counters, stride %3u B: %8.3f ms for %.0f M atomics = %6.2f G/s\n", mode == 0 ? "returning, agent" : mode == 1 ? "no return, agent" : "returning, workgroup", nb, pad * 8, ms, n / 1e6, n / ms / 1e6);
            }
    return 0;
}
