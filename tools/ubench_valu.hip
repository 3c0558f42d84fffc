// How many cycles does a wave64 VALU instruction occupy its SIMD on MI355X?  (The kernels of this engine count their vector instructions
// per tile; whether 800 of them are 3200 or 1600 SIMD cycles decides if a pass is bound by issue or by HBM.)
// Every wave runs ITER x 16 independent 32-bit integer operations; waves per SIMD = 1, 2, 4, 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int KIND>
__global__ __launch_bounds__(256) void spin(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = seed + threadIdx.x * 16 + k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == 0) a[k] = a[k] * 3u + 1u;                 // v_mad_u32_u24 / v_mul_lo + add
            else if (KIND == 1) a[k] = (a[k] ^ seed) + (uint32_t)i;   // xor + add
            else a[k] = a[k] < seed ? a[k] + 7u : a[k] - 3u;      // cmp + cndmask + ...
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s ^= a[k];
    if (s == 0x12345678u) out[0] = s;
}
int main() {
    uint32_t* d; CK(hipMalloc((void**)&d, 4096));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("CUs %d, clock %d kHz\n", cus, p.clockRate);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 1 << 14;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = cus * wps;             // 256 threads = 4 waves per block = 1 wave per SIMD per block
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double winstr = (double)blocks * 4 * iters * 16 * 2;          // xor + add per element
        printf("waves per SIMD %d: %.3f ms, %.2f G wave-instructions/s, per SIMD %.3f wave-instr per ns (at 2.4 GHz: %.2f cycles per instruction)\n", wps, ms,
               winstr / ms / 1e6, winstr / ms / 1e6 / (cus * 4), 2.4 / (winstr / ms / 1e6 / (cus * 4)));
    }
    return 0;
}
