// tools/ubench_numa.hip -- does it matter on which NUMA node of the host the pinned staging buffers of the host-pointer path lie?
// For every node: bind the allocating thread's memory policy to it, hipHostMalloc the ring, time device -> host copies on two streams and
// host -> device copies; prints what sysfs says about the device's own node.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_numa tools/ubench_numa.hip && tools/ubench_numa
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cctype>
#include <string>
#include <unistd.h>
#include <sys/syscall.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static long set_policy(int mode, unsigned long mask) { return syscall(SYS_set_mempolicy, mode, mask ? &mask : nullptr, mask ? sizeof(mask) * 8 : 0); }
int main() {
    char bdf[64] = {0};
    CK(hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0));
    for (char* p = bdf; *p; ++p) *p = (char)tolower(*p);
    std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
    int node = -2; if (FILE* f = fopen(path.c_str(), "r")) { if (fscanf(f, "%d", &node) != 1) node = -2; fclose(f); }
    printf("device 0: PCI %s, %s says node %d\n", bdf, path.c_str(), node);
    const size_t CH = (size_t)64 << 20; const int NCH = 64;
    char* d = nullptr; CK(hipMalloc((void**)&d, CH * 4));
    hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    for (int nd = 0; nd < 2; ++nd) {
        const long rc = set_policy(2 /* MPOL_BIND */, 1ul << nd);
        char* h[4]; for (int i = 0; i < 4; ++i) { CK(hipHostMalloc((void**)&h[i], CH, hipHostMallocDefault)); memset(h[i], 1, CH); }
        set_policy(0, 0);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            double t0 = now();
            for (int q = 0; q < NCH; ++q) CK(hipMemcpyAsync(h[q & 3], d + (q & 3) * CH, CH, hipMemcpyDeviceToHost, s[q & 1]));
            CK(hipDeviceSynchronize());
            const double d2h = NCH * (double)CH / (now() - t0) / 1e9;
            t0 = now();
            for (int q = 0; q < NCH; ++q) CK(hipMemcpyAsync(d + (q & 3) * CH, h[q & 3], CH, hipMemcpyHostToDevice, s[q & 1]));
            CK(hipDeviceSynchronize());
            const double h2d = NCH * (double)CH / (now() - t0) / 1e9;
            if (rep) printf("pinned buffers on node %d (set_mempolicy %ld): device -> host %.1f GB/s, host -> device %.1f GB/s\n", nd, rc, d2h, h2d);
        }
        for (int i = 0; i < 4; ++i) CK(hipHostFree(h[i]));
    }
    return 0;
}
