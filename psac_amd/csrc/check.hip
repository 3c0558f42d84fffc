// check.hip -- device-side verification of SA / ISA / LCP (the CLI's -c at sizes where a host
// check is impractical).  Follows check_SA (/root/reference/include/check_suffix_array.hpp:56-88:
// range, ISA[SA[i]] == i, order through the first character and the ranks of the suffixes one
// further) and d_check_sa's idea of a scalable checker (:207-267).  LCP entries are verified by
// direct character comparison, which is linear in sum(LCP): meant for texts with short repeats.
#include "engine.hpp"

namespace psacx {

// err[0]: SA out of range / not inverse of ISA, err[1]: order violations, err[2]: LCP mismatches,
// err[3]: LCP[0] != 0
template <typename T>
__global__ void check_kernel(const uint8_t* __restrict__ text, uint64_t n, const T* __restrict__ SA,
                             const T* __restrict__ ISA, const T* __restrict__ LCP, unsigned long long* __restrict__ err) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned e0 = 0, e1 = 0, e2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t b = SA[i];
        if (b >= n || (uint64_t)ISA[b] != i) { ++e0; continue; }
        if (i == 0) { if (LCP && LCP[0] != 0) atomicAdd(&err[3], 1ull); continue; }
        const uint64_t a = SA[i - 1];
        if (a >= n) continue;                       // counted by the thread that owns i - 1
        const uint8_t ca = text[a], cb = text[b];
        bool ok = ca < cb;
        if (ca == cb) ok = (a + 1 == n) || (b + 1 < n && ISA[a + 1] < ISA[b + 1]);
        if (!ok) ++e1;
        if (LCP) {
            const uint64_t l = LCP[i];
            uint64_t h = 0;
            while (h < l && a + h < n && b + h < n && text[a + h] == text[b + h]) ++h;
            const bool more = (a + h < n && b + h < n && text[a + h] == text[b + h]);
            if (h != l || more) ++e2;
        }
    }
    e0 = wave_reduce<uint32_t>(e0, OpSum()); e1 = wave_reduce<uint32_t>(e1, OpSum()); e2 = wave_reduce<uint32_t>(e2, OpSum());
    if (lane_id() == 0) {
        if (e0) atomicAdd(&err[0], (unsigned long long)e0);
        if (e1) atomicAdd(&err[1], (unsigned long long)e1);
        if (e2) atomicAdd(&err[2], (unsigned long long)e2);
    }
}

template <typename T>
int check_dev(psacx_ctx* c, const uint8_t* text, uint64_t n, const T* sa, const T* isa, const T* lcp, uint64_t* errors) {
    if (!c || !text || !sa || !isa || !errors || n == 0) return PSACX_EINVAL;
    PSACX_HIP(c, hipSetDevice(c->device));
    PSACX_TRY(ensure_slab(c, 4096));
    unsigned long long* d = reinterpret_cast<unsigned long long*>(c->slab);
    PSACX_HIP(c, hipMemsetAsync(d, 0, 32, c->stream));
    hipLaunchKernelGGL((check_kernel<T>), dim3(grid_for(c, n, 256, 16)), dim3(256), 0, c->stream, text, n, sa, isa, lcp, d);
    PSACX_HIP(c, hipGetLastError());
    PSACX_HIP(c, hipMemcpyAsync(errors, d, 32, hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    return PSACX_OK;
}

// Synthetic benchmark texts of SURVEY.md section 8(d), generated where they are used: character g of
// DNA(n, seed) is "ACGT"[z & 3], of ASCII128(n, seed) z & 127, with z the g-th output (counting from 1) of
// splitmix64 started at `seed`; TANDEM repeats the first `period` characters of DNA(period, seed); MUTATED is that repeat with one
// position in 200 (chosen by a second stream over the absolute position) given a character of its own: repeated reads with mutations.
// tests/inputs.py defines the same streams on the host.
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t g) {
    uint64_t z = seed + (g + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void synth_text_kernel(uint8_t* __restrict__ out, uint64_t n, uint64_t first, int kind, uint64_t seed, uint64_t period) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16;
    for (uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i0 < n; i0 += stride) {
        uint8_t b[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint64_t g = first + i0 + j;
            const uint64_t g_abs = g;
            if (kind == 2 || kind == 3) g %= period;
            uint64_t z = splitmix64_at(seed, g);
            if (kind == 3) {           // one position in 200 carries its own character instead of the repeat's
                const uint64_t m = splitmix64_at(seed ^ 0xA5A5A5A5A5A5A5A5ull, g_abs);
                if (m % 200 == 0) z = m >> 8;
            }
            b[j] = kind == 1 ? (uint8_t)(z & 127) : (uint8_t)"ACGT"[z & 3];
        }
        if (i0 + 16 <= n && ((uintptr_t)(out + i0) & 15) == 0) {
            uint4 v;
            v.x = b[0] | (b[1] << 8) | (b[2] << 16) | ((unsigned)b[3] << 24);
            v.y = b[4] | (b[5] << 8) | (b[6] << 16) | ((unsigned)b[7] << 24);
            v.z = b[8] | (b[9] << 8) | (b[10] << 16) | ((unsigned)b[11] << 24);
            v.w = b[12] | (b[13] << 8) | (b[14] << 16) | ((unsigned)b[15] << 24);
            *reinterpret_cast<uint4*>(out + i0) = v;
        } else {
            for (int j = 0; j < 16 && i0 + j < n; ++j) out[i0 + j] = b[j];
        }
    }
}

int synth_text_dev(psacx_ctx* c, uint8_t* d_text, uint64_t n, uint64_t first, int kind, uint64_t seed, uint64_t period) {
    if (!c || !d_text || kind < 0 || kind > 3 || (kind >= 2 && period == 0)) return PSACX_EINVAL;
    if (n == 0) return PSACX_OK;
    PSACX_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(synth_text_kernel, dim3(grid_for(c, n / 16 + 1, 256, 16)), dim3(256), 0, c->stream, d_text, n, first, kind, seed, period);
    PSACX_HIP(c, hipGetLastError());
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    return PSACX_OK;
}

int check_dev_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint32_t* sa, const uint32_t* isa, const uint32_t* lcp, uint64_t* e) {
    return check_dev<uint32_t>(c, t, n, sa, isa, lcp, e);
}
int check_dev_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* sa, const uint64_t* isa, const uint64_t* lcp, uint64_t* e) {
    return check_dev<uint64_t>(c, t, n, sa, isa, lcp, e);
}

} // namespace psacx
