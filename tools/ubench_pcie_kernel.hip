// tools/ubench_pcie_kernel.hip -- can a kernel that narrows 64-bit entries to 32 bits and stores them straight into pinned host memory beat
// the narrow kernel + bounce buffer + hipMemcpyAsync of engine.hpp: staged_d2h_entries (52 GB/s at 2^32 entries in the product)?
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_pcie_kernel tools/ubench_pcie_kernel.hip && tools/ubench_pcie_kernel
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// m entries (a multiple of 4): dst[i] = (uint32_t)src[i]; a lane takes four entries a step
__global__ __launch_bounds__(256) void narrow_to(const uint64_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t m) {
    typedef uint64_t v2 __attribute__((ext_vector_type(2)));
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < m / 4; q += stride) {
        const v2 a = reinterpret_cast<const v2*>(src)[2 * q], b = reinterpret_cast<const v2*>(src)[2 * q + 1];
        v4 o; o[0] = (uint32_t)a[0]; o[1] = (uint32_t)a[1]; o[2] = (uint32_t)b[0]; o[3] = (uint32_t)b[1];
        __builtin_nontemporal_store(o, reinterpret_cast<v4*>(dst) + q);
    }
}
int main() {
    const size_t CH = (size_t)64 << 20;               // bytes per chunk on the wire
    const uint64_t per = CH / 4;                      // entries per chunk
    const int NCH = 64, NS = 4;
    uint64_t* d = nullptr; CK(hipMalloc((void**)&d, (size_t)NCH * per * 8));
    CK(hipMemset(d, 1, (size_t)NCH * per * 8));
    char* bounce = nullptr; CK(hipMalloc((void**)&bounce, NS * CH));
    char* h[NS]; for (int i = 0; i < NS; ++i) CK(hipHostMalloc((void**)&h[i], CH, hipHostMallocDefault));
    hipStream_t s[3]; for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    hipEvent_t ev[NS]; for (int i = 0; i < NS; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    // (a) what the product does: narrow into a bounce buffer on one stream, copy out alternating between two streams
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (int q = 0; q < NCH; ++q) {
            const int b = q % NS;
            hipLaunchKernelGGL(narrow_to, dim3(2048), dim3(256), 0, s[2], d + (uint64_t)q * per, reinterpret_cast<uint32_t*>(bounce + (size_t)b * CH), per);
            CK(hipEventRecord(ev[b], s[2]));
            CK(hipStreamWaitEvent(s[q & 1], ev[b], 0));
            CK(hipMemcpyAsync(h[b], bounce + (size_t)b * CH, CH, hipMemcpyDeviceToHost, s[q & 1]));
        }
        CK(hipDeviceSynchronize());
        if (rep) printf("narrow kernel -> bounce -> hipMemcpyAsync on two streams: %.1f GB/s on the wire\n", NCH * (double)CH / (now() - t0) / 1e9);
    }
    // (b) the kernel stores into pinned host memory itself
    for (unsigned grid : {64u, 128u, 256u, 512u, 1024u, 2048u, 8192u})
        for (int ns = 1; ns <= 2; ++ns) {
            CK(hipDeviceSynchronize());
            double t0 = now();
            for (int q = 0; q < NCH; ++q)
                hipLaunchKernelGGL(narrow_to, dim3(grid), dim3(256), 0, s[q % ns], d + (uint64_t)q * per, reinterpret_cast<uint32_t*>(h[q % NS]), per);
            CK(hipDeviceSynchronize());
            printf("kernel stores into pinned host memory, grid %5u, %d stream(s): %.1f GB/s on the wire\n", grid, ns, NCH * (double)CH / (now() - t0) / 1e9);
        }
    return 0;
}
