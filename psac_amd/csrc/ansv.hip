// ansv.hip -- all nearest smaller values over an integer array (the LCP array).
//
// Stands in for ansv<T,left_type,right_type,global_indexing>() (/root/reference/include/
// ansv.hpp:2042-2051; sequential semantics ansv.hpp:48-65; tie rules ansv_common.hpp:20-22;
// caller suffix_tree.hpp:43-63 with left = furthest_eq, right = nearest_sm).
//
// The reference walks a monotone stack per rank and exchanges unmatched prefix minima.  On the
// GPU the array is cut into tiles; inside a tile every search is a binary descent over window
// minima held in registers and LDS, and the few searches that leave a tile share one walk of a
// global 64-ary min-pyramid per distinct value (ansv_tile.hpp).
#include "engine.hpp"
#include "nsv.hpp"
#include "ansv_wave.hpp"

namespace psacx {

// type 0 nearest_sm, 1 nearest_eq, 2 furthest_eq
template <typename T, bool LEFT>
__device__ __forceinline__ uint64_t nsv_typed(const Pyramid<T>& P, uint64_t n, uint64_t i, int type) {
    const T v = P.lvl[0][i];
    if (type == 0) return nsv_search<T, LEFT>(P, i, v, true);
    const uint64_t j = nsv_search<T, LEFT>(P, i, v, false);
    if (type == 1 || j == NSV_NONE) return j;
    // furthest_eq: the far end of the run of values equal to in[j] that nothing smaller interrupts
    const T u = P.lvl[0][j];
    const uint64_t s = nsv_search<T, LEFT>(P, j, u, true);         // first strictly smaller beyond j
    if (LEFT) {
        // leftmost element <= u in (s, j]: search rightwards from s (or from before index 0)
        if (s == NSV_NONE) { if (P.lvl[0][0] <= u) return 0; return nsv_search<T, false>(P, 0, u, false); }
        return nsv_search<T, false>(P, s, u, false);
    } else {
        if (s == NSV_NONE) { if (P.lvl[0][n - 1] <= u) return n - 1; return nsv_search<T, true>(P, n - 1, u, false); }
        return nsv_search<T, true>(P, s, u, false);
    }
}

// dev: in / left / right are device pointers (results stay in HBM); otherwise host pointers, staged here
template <typename T>
int ansv_run(psacx_ctx* c, const T* in, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* left, uint64_t* right, bool dev) {
    if (!c || !in || !left || !right || n == 0 || lt < 0 || lt > 2 || rt < 0 || rt > 2) return PSACX_EINVAL;
    PSACX_HIP(c, hipSetDevice(c->device));
    Pyramid<T> P;
    T* d_in = nullptr; uint64_t *d_l = nullptr, *d_r = nullptr;
    auto layout = [&](Arena& a) {
        if (dev) { d_in = const_cast<T*>(in); d_l = left; d_r = right; }
        else { d_in = a.take<T>(n); d_l = a.take<uint64_t>(n); d_r = a.take<uint64_t>(n); }
        nsv_pyramid_layout<T>(a, d_in, n, P);
    };
    { Arena dry(nullptr); layout(dry); PSACX_TRY(ensure_slab(c, dry.off + 4096)); }
    Arena ar(c->slab);
    layout(ar);
    if (!dev) PSACX_HIP(c, hipMemcpyAsync(d_in, in, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    ProfScope* ps = new ProfScope(c, TC_TOTAL);
    for (int L = 1; L < P.nlev; ++L) {
        hipLaunchKernelGGL((pyramid_level_kernel<T>), dim3(grid_for(c, P.len[L] * 64, 256, 8)), dim3(256), 0, c->stream,
                           P.lvl[L - 1], P.len[L - 1], P.lvl[L], P.len[L]);
    }
    launch_ansv_tiles<T>(c, P, n, lt, rt, nonsv, d_l, d_r);
    delete ps;
    PSACX_HIP(c, hipGetLastError());
    if (!dev) {
        PSACX_HIP(c, hipMemcpyAsync(left, d_l, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        PSACX_HIP(c, hipMemcpyAsync(right, d_r, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    }
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    if (c->profile) prof_collect(c);
    return PSACX_OK;
}

// ---------------------------------------------------------------------------------------------
// Suffix-tree node table (/root/reference/include/suffix_tree.hpp:43-223 for_each_parent and
// :440-499 construct_suffix_tree, one rank): one row of sigma + 1 cells per LCP index (= internal
// node); cell c holds the child reached through the character with alphabet code c (0 = end of
// text).  Leaves are numbered n + i.  Parents come from the ANSV of LCP: left = furthest_eq,
// right = nearest_sm (suffix_tree.hpp:62).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void st_nodes_kernel(const T* __restrict__ LCP, uint64_t n, const T* __restrict__ SA, const uint8_t* __restrict__ text,
                                CodeTable tab, uint64_t row, const uint64_t* __restrict__ lnsv, const uint64_t* __restrict__ rnsv,
                                unsigned long long* __restrict__ nodes) {
    // lnsv / rnsv: ANSV of LCP with left = furthest_eq, right = nearest_sm (suffix_tree.hpp:62), NSV_NONE where none
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t ln = lnsv[i], rn = rnsv[i];
        const uint64_t sa = SA[i];
        const uint64_t li = LCP[i];
        // ---- the leaf n + i (suffix_tree.hpp:72-143)
        uint64_t parent, lcp_val;
        if (i == 0) {
            lcp_val = n > 1 ? (uint64_t)LCP[1] : 0;
            parent = lcp_val > 0 ? 1 : 0;
        } else if (i == n - 1 || li >= (uint64_t)LCP[i + 1]) {
            lcp_val = ln != NSV_NONE ? (uint64_t)LCP[ln] : 0;
            if (ln != NSV_NONE && lcp_val == li) parent = ln;
            else { parent = i; lcp_val = li; }
        } else {
            parent = i + 1; lcp_val = LCP[i + 1];
        }
        uint64_t ci = sa + lcp_val;
        nodes[parent * row + (ci < n ? tab.c[text[ci]] : 0)] = n + i;
        // ---- the internal node i (suffix_tree.hpp:146-222)
        if (i == 0 || li == 0) continue;
        const uint64_t lv = LCP[ln];                  // exists because LCP[0] = 0
        if (rn == NSV_NONE) {
            if (lv == li) continue;                   // duplicate of the node further left
            parent = ln; lcp_val = lv;
        } else {
            const uint64_t rv = LCP[rn];
            if (lv >= rv) { if (lv == li) continue; parent = ln; lcp_val = lv; }
            else { parent = rn; lcp_val = rv; }
        }
        ci = sa + lcp_val;
        nodes[parent * row + (ci < n ? tab.c[text[ci]] : 0)] = i;
    }
}

template <typename T>
int suffix_tree_host(psacx_ctx* c, const uint8_t* text, uint64_t n, const T* sa, const T* lcp, uint64_t* nodes, uint32_t* sigma) {
    if (!c || !text || !sigma || n == 0) return PSACX_EINVAL;
    PSACX_HIP(c, hipSetDevice(c->device));
    // alphabet on the host (alphabet.hpp:147-164): codes 1..sigma in byte order
    bool used[256] = {false};
    for (uint64_t i = 0; i < n; ++i) used[text[i]] = true;
    CodeTable tab;
    uint16_t next = 1;
    for (int ch = 0; ch < 256; ++ch) tab.c[ch] = used[ch] ? next++ : (uint16_t)0;
    *sigma = next - 1u;
    if (!nodes) return PSACX_OK;                      // size query
    if (!sa || !lcp) return PSACX_EINVAL;
    const uint64_t row = (uint64_t)*sigma + 1;
    Pyramid<T> P;
    T* d_lcp = nullptr; T* d_sa = nullptr; uint8_t* d_text = nullptr; unsigned long long* d_nodes = nullptr;
    uint64_t *d_ln = nullptr, *d_rn = nullptr;
    auto layout = [&](Arena& a) {
        d_lcp = a.take<T>(n); d_sa = a.take<T>(n); d_text = a.take<uint8_t>(n); d_nodes = a.take<unsigned long long>(n * row);
        d_ln = a.take<uint64_t>(n); d_rn = a.take<uint64_t>(n);
        P.lvl[0] = d_lcp; P.len[0] = n; P.nlev = 1;
        uint64_t len = n;
        while (len > 64 && P.nlev < PYR_MAX) { len = (len + 63) / 64; P.lvl[P.nlev] = a.take<T>(len); P.len[P.nlev] = len; P.nlev++; }
    };
    { Arena dry(nullptr); layout(dry); PSACX_TRY(ensure_slab(c, dry.off + 4096)); }
    Arena ar(c->slab);
    layout(ar);
    PSACX_HIP(c, hipMemcpyAsync(d_lcp, lcp, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    PSACX_HIP(c, hipMemcpyAsync(d_sa, sa, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    PSACX_HIP(c, hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, c->stream));
    PSACX_HIP(c, hipMemsetAsync(d_nodes, 0, n * row * sizeof(unsigned long long), c->stream));
    for (int L = 1; L < P.nlev; ++L) {
        hipLaunchKernelGGL((pyramid_level_kernel<T>), dim3(grid_for(c, P.len[L] * 64, 256, 8)), dim3(256), 0, c->stream,
                           P.lvl[L - 1], P.len[L - 1], P.lvl[L], P.len[L]);
        PSACX_HIP(c, hipGetLastError());
    }
    launch_ansv_tiles<T>(c, P, n, 2, 0, NSV_NONE, d_ln, d_rn);
    PSACX_HIP(c, hipGetLastError());
    hipLaunchKernelGGL((st_nodes_kernel<T>), dim3(grid_for(c, n, 256, 16)), dim3(256), 0, c->stream, d_lcp, n, d_sa, d_text, tab, row,
                       d_ln, d_rn, d_nodes);
    PSACX_HIP(c, hipGetLastError());
    PSACX_HIP(c, hipMemcpyAsync(nodes, d_nodes, n * row * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    PSACX_HIP(c, hipStreamSynchronize(c->stream));
    return PSACX_OK;
}

int suffix_tree_host_u32(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint32_t* sa, const uint32_t* lcp, uint64_t* nodes, uint32_t* sg) {
    return suffix_tree_host<uint32_t>(c, t, n, sa, lcp, nodes, sg);
}
int suffix_tree_host_u64(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* sa, const uint64_t* lcp, uint64_t* nodes, uint32_t* sg) {
    return suffix_tree_host<uint64_t>(c, t, n, sa, lcp, nodes, sg);
}

int ansv_host_u32(psacx_ctx* c, const uint32_t* in, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* l, uint64_t* r) {
    return ansv_run<uint32_t>(c, in, n, lt, rt, nonsv, l, r, false);
}
int ansv_host_u64(psacx_ctx* c, const uint64_t* in, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* l, uint64_t* r) {
    return ansv_run<uint64_t>(c, in, n, lt, rt, nonsv, l, r, false);
}
int ansv_dev_u32(psacx_ctx* c, const uint32_t* in, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* l, uint64_t* r) {
    return ansv_run<uint32_t>(c, in, n, lt, rt, nonsv, l, r, true);
}
int ansv_dev_u64(psacx_ctx* c, const uint64_t* in, uint64_t n, int lt, int rt, uint64_t nonsv, uint64_t* l, uint64_t* r) {
    return ansv_run<uint64_t>(c, in, n, lt, rt, nonsv, l, r, true);
}

} // namespace psacx
