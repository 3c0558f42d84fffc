// Why does the pass on the top digit (1 byte read + 8 written per record, 256 output fronts 128 MB apart at 2^32 records) take longer than a
// bucket pass (8 + 8 bytes, 256 fronts inside a 128 MB bucket)?  Write-only pattern: tile t of 4096 eight-byte records writes 256 runs of 16
// records; run d of tile t lands at  region(t) + d * front_gap + (t mod tiles_per_region) * 16,  i.e. 256 fronts `front_gap` records apart that
// advance as the tiles of a region go by.  far: one region of n records (front_gap = n / 256); near: regions of 2^24 records (front_gap = 2^16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(512) void fronts(uint64_t* __restrict__ out, uint64_t n, uint64_t region, int read_text, const uint8_t* __restrict__ text, uint64_t skew) {
    const uint64_t tile = blockIdx.x;
    const uint64_t tiles_per_region = region / 4096, front_gap = region / 256;
    const uint64_t rbase = (tile / tiles_per_region) * region, tr = tile % tiles_per_region;
    uint64_t acc = 0;
    if (read_text) acc = text[tile * 4096 + threadIdx.x * 8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned p = threadIdx.x + j * 512;          // staged position: run p / 16, place p % 16
        const uint64_t at = rbase + (uint64_t)(p >> 4) * (front_gap + skew) + tr * 16 + (p & 15);          // skew: records added to every front's distance
        out[at] = (tile << 12) | p | acc;
    }
}
int main() {
    const uint64_t n = 1ull << 32;
    uint64_t* out; uint8_t* text;
    CK(hipMalloc((void**)&out, n * 8 + ((size_t)4 << 30))); CK(hipMalloc((void**)&text, n));
    CK(hipMemset(text, 1, n));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rd = 0; rd < 2; ++rd)
        for (uint64_t region : {n, n / 16, n / 256, n / 4096, n / 16384}) {
            for (uint64_t skew : {0ull, 20000ull, 40000ull, 65536ull + 16ull, 131072ull + 48ull, 262144ull + 272ull, 1000003ull}) {
                if (skew && region != n && region != n / 2) continue;
                hipLaunchKernelGGL(fronts, dim3((unsigned)(n / 4096)), dim3(512), 0, 0, out, n, region, rd, (const uint8_t*)text, skew);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(fronts, dim3((unsigned)(n / 4096)), dim3(512), 0, 0, out, n, region, rd, (const uint8_t*)text, skew);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
                printf("%s 256 fronts %8.2f MiB + %6llu B apart (regions of 2^%d records): %7.3f ms, %6.0f GB/s written\n", rd ? "text read +" : "write only ", region / 256 * 8 / 1048576.0,
                       (unsigned long long)(skew * 8), 63 - __builtin_clzll(region), ms, n * 8 / (ms * 1e-3) / 1e9);
            }
        }
    return 0;
}
