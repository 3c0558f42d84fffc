// tools/ubench_pcie.hip -- what bounds the host-pointer path (engine.hpp: staged_d2h_entries): device -> pinned host copy rate with one and
// two streams, and the rate at which host threads widen 32-bit entries into a pageable array of 64-bit words.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_pcie tools/ubench_pcie.hip -lpthread && tools/ubench_pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t CH = (size_t)64 << 20, NCH = 64;
    char* d = nullptr; hipMalloc((void**)&d, CH * 4);
    char* h[4]; for (int i = 0; i < 4; ++i) hipHostMalloc((void**)&h[i], CH, hipHostMallocDefault);
    hipStream_t s[2]; hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
    for (int ns = 1; ns <= 2; ++ns) {
        double t0 = now();
        for (size_t q = 0; q < NCH; ++q) hipMemcpyAsync(h[q & 3], d + (q & 3) * CH, CH, hipMemcpyDeviceToHost, s[q % ns]);
        hipStreamSynchronize(s[0]); hipStreamSynchronize(s[1]);
        double dt = now() - t0;
        printf("D2H pinned, %d stream(s): %.1f GB/s\n", ns, NCH * CH / dt / 1e9);
    }
    {
        double t0 = now();
        for (size_t q = 0; q < NCH; ++q) hipMemcpyAsync(d + (q & 3) * CH, h[q & 3], CH, hipMemcpyHostToDevice, s[0]);
        hipStreamSynchronize(s[0]);
        printf("H2D pinned, 1 stream: %.1f GB/s\n", NCH * CH / (now() - t0) / 1e9);
    }
    {   // one big pinned buffer instead of chunks
        char* big = nullptr; char* dbig = nullptr;
        if (hipHostMalloc((void**)&big, (size_t)2 << 30, hipHostMallocDefault) == hipSuccess && hipMalloc((void**)&dbig, (size_t)2 << 30) == hipSuccess) {
            double t0 = now();
            hipMemcpyAsync(big, dbig, (size_t)2 << 30, hipMemcpyDeviceToHost, s[0]); hipStreamSynchronize(s[0]);
            printf("D2H pinned, one 2 GiB copy: %.1f GB/s\n", (double)((size_t)2 << 30) / (now() - t0) / 1e9);
        }
    }
    const size_t N = (size_t)1 << 30;      // 2^30 entries: 4 GiB in, 8 GiB out
    uint32_t* in = (uint32_t*)malloc(N * 4); uint64_t* out = (uint64_t*)malloc(N * 8);
    memset(in, 1, N * 4); memset(out, 0, N * 8);
    for (int nt : {8, 16, 32, 48, 64, 96}) {
        for (int ntmp = 0; ntmp < 2; ++ntmp) {
            double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back([=]() {
                const size_t a = N * t / nt, b = N * (t + 1) / nt;
                if (ntmp) for (size_t i = a; i < b; ++i) __builtin_nontemporal_store((uint64_t)in[i], out + i);
                else for (size_t i = a; i < b; ++i) out[i] = in[i];
            });
            for (auto& x : th) x.join();
            double dt = now() - t0;
            printf("widen 32 -> 64 bits, %d threads, %s stores: %.1f GB/s written\n", nt, ntmp ? "streaming" : "plain", N * 8 / dt / 1e9);
        }
    }
    return 0;
}
