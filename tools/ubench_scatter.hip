// Micro-benchmark: permutation scatter out[p[i]] = v[i] on MI355X, plain vs windowed by
// destination range (does the L2 / Infinity Cache merge the 4-byte writes?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t mask) {
    uint32_t x = (i * 0x9E3779B1u) & mask;
    x ^= x >> 13; x = (x * 0x85EBCA6Bu) & mask; x ^= x >> 11; x = (x * 0xC2B2AE35u) & mask; x ^= x >> 15;
    return x & mask;
}
__global__ void make_perm(uint32_t* p, uint32_t* v, uint64_t n, uint32_t mask) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { p[i] = perm((uint32_t)i, mask); v[i] = (uint32_t)i + 1; }
}
__global__ void scatter_plain(const uint32_t* __restrict__ p, const uint32_t* __restrict__ v, uint64_t n, uint32_t* __restrict__ out) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[p[i]] = v[i];
}
__global__ void scatter_window(const uint32_t* __restrict__ p, const uint32_t* __restrict__ v, uint64_t n, uint32_t* __restrict__ out, uint32_t lo, uint32_t hi) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { uint32_t d = p[i]; if (d >= lo && d < hi) out[d] = v[i]; }
}
// vectorised reads: 4 items per thread
__global__ void scatter_window4(const uint4* __restrict__ p, const uint4* __restrict__ v, uint64_t n4, uint32_t* __restrict__ out, uint32_t lo, uint32_t hi) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        uint4 d = p[i];
        bool a = d.x >= lo && d.x < hi, b = d.y >= lo && d.y < hi, c = d.z >= lo && d.z < hi, e = d.w >= lo && d.w < hi;
        if (a | b | c | e) { uint4 x = v[i]; if (a) out[d.x] = x.x; if (b) out[d.y] = x.y; if (c) out[d.z] = x.z; if (e) out[d.w] = x.w; }
    }
}
__global__ void gather_plain(const uint32_t* __restrict__ p, const uint32_t* __restrict__ src, uint64_t n, uint32_t* __restrict__ out) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = src[p[i]];
}
__global__ void copy4(const uint4* __restrict__ a, uint4* __restrict__ b, uint64_t n4) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}
int main(int argc, char** argv) {
    int logn = argc > 1 ? atoi(argv[1]) : 28;
    uint64_t n = 1ull << logn; uint32_t mask = (uint32_t)(n - 1);
    uint32_t *p, *v, *out;
    CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&out, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    make_perm<<<4096, 256>>>(p, v, n, mask); CK(hipDeviceSynchronize());
    auto timeit = [&](const char* name, auto fn, int reps) { fn(); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0); for (int r = 0; r < reps; ++r) fn(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); printf("%-40s %8.3f ms\n", name, ms / reps); };
    timeit("copy 1 GiB (uint4)", [&] { copy4<<<8192, 256>>>((const uint4*)p, (uint4*)out, n / 4); }, 5);
    for (int g : {2048, 4096, 16384}) { char nm[64]; snprintf(nm, 64, "scatter plain grid %d", g); timeit(nm, [&] { scatter_plain<<<g, 256>>>(p, v, n, out); }, 3); }
    timeit("gather plain", [&] { gather_plain<<<4096, 256>>>(p, v, n, out); }, 3);
    for (int W : {2, 4, 8, 16, 32, 64}) {
        char nm[64]; snprintf(nm, 64, "scatter windowed W=%d", W);
        timeit(nm, [&] { for (int w = 0; w < W; ++w) { uint32_t lo = (uint32_t)((n / W) * w), hi = (uint32_t)((n / W) * (w + 1)); scatter_window<<<4096, 256>>>(p, v, n, out, lo, hi); } }, 2);
        snprintf(nm, 64, "scatter windowed4 W=%d", W);
        timeit(nm, [&] { for (int w = 0; w < W; ++w) { uint32_t lo = (uint32_t)((n / W) * w), hi = (uint32_t)((n / W) * (w + 1)); scatter_window4<<<4096, 256>>>((const uint4*)p, (const uint4*)v, n / 4, out, lo, hi); } }, 2);
    }
    // verify last result
    std::vector<uint32_t> h(1024); CK(hipMemcpy(h.data(), out, 4096, hipMemcpyDeviceToHost));
    printf("out[0..3] = %u %u %u %u\n", h[0], h[1], h[2], h[3]);
    return 0;
}
