// Upper bound for a radix scatter pass on MI355X: stream 3 arrays in, write 3 arrays out where each
// tile of TILE records is split into NB runs of TILE/NB consecutive records that land at NB
// far-apart frontiers (exactly the write pattern of an 8-bit digit pass with uniform digits), but
// with no ranking, no look-back and no LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void pattern_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const uint32_t* __restrict__ c,
                               uint32_t* __restrict__ oa, uint32_t* __restrict__ ob, uint32_t* __restrict__ oc, uint64_t n, int nb) {
    constexpr int TILE = BLOCK * ITEMS;
    const uint64_t tile = blockIdx.x;
    const uint64_t base = tile * TILE;
    const unsigned run = TILE / nb;            // records per bin per tile
    const uint64_t bin_stride = n / nb;
    uint32_t x[ITEMS], y[ITEMS], z[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { unsigned p = threadIdx.x + j * BLOCK; x[j] = a[base + p]; y[j] = b[base + p]; z[j] = c[base + p]; }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        unsigned p = threadIdx.x + j * BLOCK;
        unsigned bin = p / run, r = p % run;
        uint64_t d = (uint64_t)bin * bin_stride + tile * run + r;
        oa[d] = x[j]; ob[d] = y[j]; oc[d] = z[j];
    }
}
// the same pattern for two-word records (the prefix sort / inversion levels): 6144-record tiles as used there
template <int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void pattern2_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                               uint32_t* __restrict__ oa, uint32_t* __restrict__ ob, uint64_t n, int nb) {
    constexpr int TILE = BLOCK * ITEMS;
    const uint64_t tile = blockIdx.x;
    const uint64_t base = tile * TILE;
    const unsigned run = TILE / nb;
    const uint64_t bin_stride = n / nb;
    uint32_t x[ITEMS], y[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { unsigned p = threadIdx.x + j * BLOCK; x[j] = a[base + p]; y[j] = b[base + p]; }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        unsigned p = threadIdx.x + j * BLOCK;
        unsigned bin = p / run, r = p % run;
        uint64_t d = (uint64_t)bin * bin_stride + tile * run + r;
        if (d < n) { oa[d] = x[j]; ob[d] = y[j]; }
    }
}
__global__ void copy3(const uint4* a, const uint4* b, const uint4* c, uint4* oa, uint4* ob, uint4* oc, uint64_t n4) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { oa[i] = a[i]; ob[i] = b[i]; oc[i] = c[i]; }
}
__global__ void copy1(const uint4* a, uint4* oa, uint64_t n4) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { oa[i] = a[i]; }
}
__global__ void read1(const uint4* a, uint4* oa, uint64_t n4) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x; uint4 acc = {0,0,0,0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345678 && acc.y == 1) oa[0] = acc;
}
__global__ void write1(uint4* oa, uint64_t n4) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x; uint4 v = {1,2,3,4};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) oa[i] = v;
}
int main() {
    const uint64_t n = 1ull << 28;
    uint32_t *a, *b, *c, *oa, *ob, *oc;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
    CK(hipMalloc(&oa, n * 4)); CK(hipMalloc(&ob, n * 4)); CK(hipMalloc(&oc, n * 4));
    CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 2, n * 4)); CK(hipMemset(c, 3, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double gbytes, auto fn) { fn(); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0); for (int r = 0; r < 5; ++r) fn(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5; printf("%-44s %7.3f ms  %7.0f GB/s\n", name, ms, gbytes / (ms * 1e-3)); };
    const double GB3 = 6.0 * n * 4 / 1e9;
    for (int g : {2048, 8192, 32768}) { char nm[64]; snprintf(nm, 64, "read 1 GiB grid %d", g); timeit(nm, n * 4 / 1e9, [&] { read1<<<g, 256>>>((const uint4*)a, (uint4*)oa, n / 4); }); }
    for (int g : {2048, 8192, 32768}) { char nm[64]; snprintf(nm, 64, "write 1 GiB grid %d", g); timeit(nm, n * 4 / 1e9, [&] { write1<<<g, 256>>>((uint4*)oa, n / 4); }); }
    for (int g : {2048, 8192, 32768, 262144}) { char nm[64]; snprintf(nm, 64, "copy 1 array grid %d", g); timeit(nm, 2.0 * n * 4 / 1e9, [&] { copy1<<<g, 256>>>((const uint4*)a, (uint4*)oa, n / 4); }); }
    for (int g : {2048, 8192, 32768}) { char nm[64]; snprintf(nm, 64, "copy 3 arrays grid %d", g); timeit(nm, GB3, [&] { copy3<<<g, 256>>>((const uint4*)a, (const uint4*)b, (const uint4*)c, (uint4*)oa, (uint4*)ob, (uint4*)oc, n / 4); }); }
    for (int nb : {1, 16, 64, 128, 256}) {
        char nm[64];
        snprintf(nm, 64, "pattern 256x16 tile 4096, %d bins", nb); timeit(nm, GB3, [&] { pattern_kernel<256, 16><<<(unsigned)(n / 4096), 256>>>(a, b, c, oa, ob, oc, n, nb); });
        snprintf(nm, 64, "pattern 512x16 tile 8192, %d bins", nb); timeit(nm, GB3, [&] { pattern_kernel<512, 16><<<(unsigned)(n / 8192), 512>>>(a, b, c, oa, ob, oc, n, nb); });
        snprintf(nm, 64, "pattern 1024x16 tile 16384, %d bins", nb); timeit(nm, GB3, [&] { pattern_kernel<1024, 16><<<(unsigned)(n / 16384), 1024>>>(a, b, c, oa, ob, oc, n, nb); });
    }
    const double GB2 = 4.0 * n * 4 / 1e9;
    const uint64_t n2 = n / 6144 * 6144;
    for (int nb : {1, 256}) {
        char nm[64];
        snprintf(nm, 64, "two-word pattern 512x12 tile 6144, %d bins", nb);
        timeit(nm, GB2 * n2 / n, [&] { pattern2_kernel<512, 12><<<(unsigned)(n2 / 6144), 512>>>(a, b, oa, ob, n2, nb); });
    }
    return 0;
}
