// Does the 256 MiB Infinity Cache carry a ping-pong between two buffers?  The bucket passes of the one-word prefix sort read what
// the pass before wrote; run bucket by bucket (128 MiB per bucket at 2^32 records) instead of pass by pass the data of a bucket
// would still sit in the die-level cache when the next pass asks for it.  This measures the best case: plain 16-byte copies
// A -> B, B -> A, ... over buffers of S bytes, and a read-only sweep of a buffer that was just written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void copy1(const uint4* __restrict__ a, uint4* __restrict__ oa, uint64_t n4) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) oa[i] = a[i];
}
// tile-ordered copy: workgroup b copies the b-th 64 KiB piece (the access order of a scatter pass: tiles in start order)
__global__ __launch_bounds__(512) void copy_tiles(const uint4* __restrict__ a, uint4* __restrict__ oa, uint64_t n4) {
    const uint64_t base = (uint64_t)blockIdx.x * 4096;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const uint64_t i = base + (uint64_t)j * 512 + threadIdx.x; if (i < n4) oa[i] = a[i]; }
}
__global__ __launch_bounds__(512) void read1(const uint4* __restrict__ a, uint4* __restrict__ oa, uint64_t n4) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x; uint4 acc = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { const uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345678 && acc.y == 1) oa[0] = acc;
}

int main() {
    const size_t MAXB = (size_t)4 << 30;
    char *A, *B;
    CK(hipMalloc((void**)&A, MAXB)); CK(hipMalloc((void**)&B, MAXB));
    CK(hipMemset(A, 1, MAXB)); CK(hipMemset(B, 2, MAXB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes[] = {(size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20, (size_t)96 << 20, (size_t)128 << 20, (size_t)192 << 20,
                            (size_t)256 << 20, (size_t)512 << 20, (size_t)1 << 30, (size_t)4 << 30};
    for (size_t S : sizes) {
        const uint64_t n4 = S / 16;
        const int reps = (int)std::max<size_t>(4, ((size_t)8 << 30) / S);
        // (1) ping-pong copy, grid-stride
        for (int form = 0; form < 2; ++form) {
            const unsigned grid = form ? (unsigned)((n4 + 4095) / 4096) : 256 * 8;
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(form ? copy_tiles : copy1, dim3(grid), dim3(512), 0, 0, (const uint4*)A, (uint4*)B, n4);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) {
                const uint4* src = (const uint4*)((r & 1) ? B : A); uint4* dst = (uint4*)((r & 1) ? A : B);
                hipLaunchKernelGGL(form ? copy_tiles : copy1, dim3(grid), dim3(512), 0, 0, src, dst, n4);
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("ping-pong copy %-10s S = %5zu MiB: %8.3f us per copy, %7.1f GB/s (read + write)\n", form ? "tiles" : "gridstride", S >> 20, ms * 1e3 / reps,
                   2.0 * S * reps / (ms * 1e-3) / 1e9);
        }
        // (2) write B (copy from a far-away part of A so that the source never hits), then read B back
        {
            float ms_r = 0;
            const int reps2 = std::min(reps, 64);
            for (int r = 0; r < reps2; ++r) {
                const size_t off = ((size_t)r * S) % (MAXB - S + 1) & ~(size_t)255;
                hipLaunchKernelGGL(copy1, dim3(256 * 8), dim3(512), 0, 0, (const uint4*)(A + off), (uint4*)B, n4);
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(read1, dim3(256 * 8), dim3(512), 0, 0, (const uint4*)B, (uint4*)A, n4);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms_r += ms;
            }
            printf("read after write          S = %5zu MiB: %8.3f us per read, %7.1f GB/s\n", S >> 20, ms_r * 1e3 / reps2, (double)S * reps2 / (ms_r * 1e-3) / 1e9);
        }
    }
    return 0;
}
