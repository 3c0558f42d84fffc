// What a random 8-byte fetch / store costs on MI355X as a function of how far apart the addresses of neighbouring requests lie:
// requests j = 0 .. cnt-1 go to A[class(j) * span + hash(j) % span], class(j) = j / (cnt / (len / span)) -- i.e. the requests have been
// partitioned by address into classes of `span` entries and are random inside their class.  span = len: the random fetch of
// gather_keys_kernel; span = 2^23 (64 MiB of 8-byte entries): after ONE 512-way partition level at len = 2^32; span = 2^14: after two.
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t addr_of(uint64_t j, uint64_t per_class, unsigned span_bits) {
    const uint64_t cls = j / per_class;
    return (cls << span_bits) | (mix(j) & ((1ull << span_bits) - 1));
}
template <int W>
__global__ void gather_kernel(const uint64_t* __restrict__ A, uint64_t cnt, uint64_t per_class, unsigned span_bits, uint64_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const uint64_t a = addr_of(j, per_class, span_bits);
        if (W == 8) out[j] = A[a] + 1;
        else reinterpret_cast<uint32_t*>(out)[j] = reinterpret_cast<const uint32_t*>(A)[a] + 1;
    }
}
__global__ void scatter_kernel(uint64_t* __restrict__ A, uint64_t cnt, uint64_t per_class, unsigned span_bits, const uint64_t* __restrict__ in) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) A[addr_of(j, per_class, span_bits)] = in[j];
}
int main(int argc, char** argv) {
    const int lg_len = argc > 1 ? atoi(argv[1]) : 32;           // entries of A
    const int lg_cnt = argc > 2 ? atoi(argv[2]) : 30;           // requests
    const uint64_t len = 1ull << lg_len, cnt = 1ull << lg_cnt;
    uint64_t *A, *io;
    CK(hipMalloc(&A, len * 8)); CK(hipMalloc(&io, cnt * 8));
    CK(hipMemset(A, 1, len * 8)); CK(hipMemset(io, 2, cnt * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("A: 2^%d 8-byte entries (%.1f GiB), 2^%d requests\n", lg_len, len * 8 / 1073741824.0, lg_cnt);
    for (int sb : {lg_len, 26, 23, 20, 17, 14}) {
        if (sb > lg_len) continue;
        const uint64_t classes = len >> sb, per_class = (cnt + classes - 1) / classes;
        for (int kind = 0; kind < 3; ++kind) {
            const int grid = 256 * 8 * 4;
            auto fn = [&] {
                if (kind == 0) gather_kernel<8><<<grid, 256>>>(A, cnt, per_class, (unsigned)sb, io);
                else if (kind == 1) gather_kernel<4><<<grid, 256>>>(A, cnt, per_class, (unsigned)sb + 1, io);       // the same bytes as 32-bit entries
                else scatter_kernel<<<grid, 256>>>(A, cnt, per_class, (unsigned)sb, io);
            };
            fn(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            printf("span 2^%-2d (%8.1f MiB) %-9s %8.3f ms  %6.1f ps per request  %6.1f G requests/s\n", sb, (8ull << sb) / 1048576.0,
                   kind == 0 ? "gather8" : kind == 1 ? "gather4" : "scatter8", ms, ms * 1e9 / cnt, cnt / (ms * 1e6));
        }
    }
    return 0;
}
