// The write pattern of a 512-way partition level (sa_kernels.hpp: partition_packed_kernel): tile t of 8192 eight-byte pairs writes 512 runs of
// 16 pairs; run c of a tile lands in front c of the tile's region.  Level 1 (the level fused into rebucket_first_kernel): one region of n pairs,
// fronts n / 512 apart (64 MiB at 2^32); level 2: regions of 2^23 pairs, fronts 2^14 pairs = 128 KiB apart.  skew = pairs added to every
// front's distance: do fronts a power of two apart alias in the memory channels as the 256 fronts of the top-digit pass do (ubench_fronts.hip)?
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench_fronts2 tools/ubench_fronts2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(512) void fronts(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t region, uint64_t skew, int rd) {
    const uint64_t tile = blockIdx.x;
    const uint64_t tiles_per_region = region / 8192, front_gap = region / 512;
    const uint64_t rbase = (tile / tiles_per_region) * (region + 512 * skew), tr = tile % tiles_per_region;
    uint64_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = rd ? in[tile * 8192 + threadIdx.x + j * 512] : (tile << 13) | (threadIdx.x + j * 512);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const unsigned p = threadIdx.x + j * 512;          // staged position: run p / 16, place p % 16
        out[rbase + (uint64_t)(p >> 4) * (front_gap + skew) + tr * 16 + (p & 15)] = v[j];
    }
}
int main() {
    const uint64_t n = 1ull << 32;
    uint64_t *out, *in;
    CK(hipMalloc((void**)&out, n * 8 + ((size_t)8 << 30))); CK(hipMalloc((void**)&in, n * 8));
    CK(hipMemset(in, 1, n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rd = 0; rd < 2; ++rd)
        for (uint64_t region : {n, n >> 9}) {
            for (uint64_t skew : {0ull, 8ull, 32ull, 64ull, 128ull, 512ull, 2048ull, 2064ull, 20000ull}) {
                if ((region + 512 * skew) * (n / region) > n + (1ull << 30)) continue;
                auto fn = [&] { hipLaunchKernelGGL(fronts, dim3((unsigned)(n / 8192)), dim3(512), 0, 0, (const uint64_t*)in, out, region, skew, rd); };
                fn(); CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0)); for (int r = 0; r < 3; ++r) fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
                printf("%s 512 fronts %10.3f KiB + %6llu B apart (regions of 2^%d pairs): %7.3f ms, %6.0f GB/s %s\n", rd ? "copy " : "write", region / 512 * 8 / 1024.0,
                       (unsigned long long)(skew * 8), 63 - __builtin_clzll(region), ms, n * 8 * (rd ? 2 : 1) / (ms * 1e-3) / 1e9, rd ? "moved" : "written");
            }
        }
    return 0;
}
