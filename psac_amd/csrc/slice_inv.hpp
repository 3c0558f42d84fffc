// slice_inv.hpp -- SA -> ISA on block-distributed arrays by destination slices (bulk_permute_inplace,
// /root/reference/include/bulk_permute.hpp:14-73, whose MPI_Alltoallv moves (index, value) pairs to the owners, :60-61).
//
// ISA[SA[i]] = B[i] - 1 is the inverse of a permutation of the WHOLE text.  A rank holds m of its (position, rank)
// pairs -- a subset, not a permutation -- so the histogram-free partition of the one-GPU engine (sa_kernels.hpp:
// partition_pairs_kernel) does not apply to them directly.  It does apply on the receiving side once the pairs travel
// by (owner, slice): a slice is 2^sb consecutive positions of an owner's block, and because SA is a permutation the
// pairs that ALL ranks hold for one slice are exactly a permutation of that slice.  So
//   1. every rank counts its pairs per (owner, slice) class (slice_hist_kernel: one read of SA) and partitions them into
//      those classes (slice_partition_kernel: keys leave as 32-bit block-relative positions, values as ranks);
//   2. the classes travel to their owners slice by slice (multi.hpp: MultiRun::transfer), each slice landing at its own
//      aligned place of the receive arrays;
//   3. the owner runs the remaining partition levels (pairs_partition_kernel: reservation by one atomic per tile and class,
//      no histogram) down to windows of 2^wb positions and scatters each window inside LDS (pairs_window_kernel),
//      writing ISA as full lines.
// With one rank, step 2 vanishes and the classes are the slices of the own block: the one-GPU scheme.
#pragma once
#include "dev_common.hpp"

namespace psacx {

constexpr int SLICE_MAX_CLASSES = 512;

// class of a global position: (owner, slice of the owner's block)
struct SliceMap {
    uint64_t div, mod;       // mxx::blk_dist of the n positions: the first `mod` owners hold div + 1
    unsigned P, sb, spo;     // owners, log2 of the slice length, slices per owner
    int dshift;              // >= 0: every block is 2^dshift long (owner = g >> dshift)
    __host__ __device__ unsigned owner(uint64_t g) const {
        if (dshift >= 0) return (unsigned)(g >> dshift);
        const uint64_t big = (div + 1) * mod;
        if (g < big) return (unsigned)(g / (div + 1));
        return (unsigned)(mod + (g - big) / (div ? div : 1));
    }
    __host__ __device__ uint64_t off(unsigned r) const { return div * r + (r < mod ? r : mod); }
    __host__ __device__ uint64_t size(unsigned r) const { return div + (r < mod ? 1 : 0); }
};

template <typename T>
__global__ __launch_bounds__(512) void slice_hist_kernel(const T* __restrict__ SA, uint64_t n, SliceMap map, unsigned long long* __restrict__ counts) {
    __shared__ unsigned lh[SLICE_MAX_CLASSES];
    for (int i = threadIdx.x; i < SLICE_MAX_CLASSES; i += 512) lh[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 512;
    for (uint64_t i = (uint64_t)blockIdx.x * 512 + threadIdx.x; i < n; i += stride) {
        const uint64_t g = SA[i];
        const unsigned o = map.owner(g);
        atomicAdd(&lh[o * map.spo + (unsigned)((g - map.off(o)) >> map.sb)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SLICE_MAX_CLASSES; i += 512) if (lh[i]) atomicAdd(&counts[i], (unsigned long long)lh[i]);
}

// First level: (SA[i], B[i]) -> (block-relative position, B[i] - 1) grouped by class.  cursors[c] starts at the first slot
// of class c (the exclusive scan of the counts) and ends at the first slot of class c + 1; the order inside a class is the
// order of the reservations, which is irrelevant to an inversion.
template <typename T, typename V, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void slice_partition_kernel(const T* __restrict__ SA, const T* __restrict__ B, uint64_t n, SliceMap map,
                                                                unsigned long long* __restrict__ cursors, uint32_t* __restrict__ key_out,
                                                                V* __restrict__ val_out, T* __restrict__ b_copy = nullptr) {
    // b_copy (optional): the bucket ids B are written there as well (reduced-memory layout: they were parked in the ISA array,
    // which the inversion is about to overwrite)
    constexpr int TILE = BLOCK * ITEMS;
    static_assert(BLOCK >= SLICE_MAX_CLASSES, "one thread per class");
    __shared__ V stage[TILE];
    __shared__ unsigned short cstage[TILE];
    __shared__ unsigned cnt[SLICE_MAX_CLASSES];
    __shared__ unsigned bstart[SLICE_MAX_CLASSES];
    __shared__ unsigned long long gbase[SLICE_MAX_CLASSES];
    __shared__ unsigned scan_tmp[BLOCK / WAVE + 1];
    const unsigned tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    const uint64_t remain = n - base;
    const unsigned count = remain < (uint64_t)TILE ? (unsigned)remain : (unsigned)TILE;
    for (int i = tid; i < SLICE_MAX_CLASSES; i += BLOCK) cnt[i] = 0;
    __syncthreads();
    uint32_t key[ITEMS]; V val[ITEMS];
    unsigned short cls[ITEMS];
    unsigned slot[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned loc = tid + i * BLOCK;
        if (loc < count) {
            const uint64_t g = SA[base + loc];
            const unsigned o = map.owner(g);
            const uint64_t rel = g - map.off(o);
            key[i] = (uint32_t)rel;
            cls[i] = (unsigned short)(o * map.spo + (unsigned)(rel >> map.sb));
            const T bb = B[base + loc];
            val[i] = (V)((uint64_t)bb - 1u);
            if (b_copy) b_copy[base + loc] = bb;
        } else { key[i] = 0; cls[i] = 0; val[i] = 0; }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) slot[i] = (tid + i * BLOCK) < count ? atomicAdd(&cnt[cls[i]], 1u) : 0u;
    __syncthreads();
    const unsigned tot = tid < SLICE_MAX_CLASSES ? cnt[tid] : 0u;
    unsigned total;
    const unsigned bs = block_scan_exclusive<BLOCK, unsigned>(tot, OpSum(), 0u, scan_tmp, &total);
    if (tid < SLICE_MAX_CLASSES) {
        bstart[tid] = bs;
        if (tot) gbase[tid] = atomicAdd(&cursors[tid], (unsigned long long)tot) - bs;
    }
    __syncthreads();
    // keys first (the stage is shared with the values); the class of every staged slot is kept beside them
    uint32_t* const kstage = reinterpret_cast<uint32_t*>(stage);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        slot[i] += bstart[cls[i]];
        if (tid + i * BLOCK < count) { kstage[slot[i]] = key[i]; cstage[slot[i]] = cls[i]; }
    }
    __syncthreads();
    unsigned long long dest[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (p < count) {
            dest[j] = gbase[cstage[p]] + p;
            key_out[dest[j]] = kstage[p];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (tid + i * BLOCK < count) stage[slot[i]] = val[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (p < count) val_out[dest[j]] = stage[p];
    }
}

// Ranks beyond 2^32 on the wire as 32 bits (packed with the position into one 64-bit entry, slice_partition_packed_kernel with
// rel): a sender's ranks are bucket heads at or before its own positions, so e = (last position of the sender's block) - rank
// is >= 0, and below 2^32 as long as no bucket reaches further back than 2^32 positions from the sender's block end (checked
// before the form is chosen).  The owner knows the sender of every record of a received slice from where it lies: the slice's
// region holds the senders' segments in rank order.  seg[s * (P + 1) + r] = start of sender r's segment inside slice s of the
// step (seg[.. + P] = its end), base[r] = last position of sender r's block.
struct SliceDecode {
    const uint64_t* seg; const uint64_t* base; unsigned P, sb;
    __device__ __forceinline__ uint64_t rank_of(uint64_t idx, uint32_t e) const {
        const uint64_t* row = seg + (idx >> sb) * (P + 1);
        const uint64_t w = idx & ((1ull << sb) - 1);
        unsigned lo = 0, hi = P;                  // sender r with row[r] <= w < row[r + 1] (empty segments share a start: the last one)
        while (hi - lo > 1) { const unsigned mid = (lo + hi) >> 1; if (row[mid] <= w) lo = mid; else hi = mid; }
        return base[lo] - (uint64_t)e;
    }
};

// A further level on 32-bit keys: 2^cb destination classes per parent bucket of 2^(shift + cb) positions.  The input holds
// every parent bucket complete and at its own place (keys minus koff), so a parent receives exactly its size in pairs and
// a tile reserves room in a class with one atomic (see partition_pairs_kernel).  cb <= 9 at run time.
// DEC: the input is the packed wire form (key_in viewed as uint64_t*, val_in unused), decoded on the way in (SliceDecode)
template <typename V, int BLOCK, int ITEMS, bool DEC = false>
__global__ __launch_bounds__(BLOCK) void pairs_partition_kernel(const uint32_t* __restrict__ key_in, const V* __restrict__ val_in,
                                                                uint32_t* __restrict__ key_out, V* __restrict__ val_out, uint64_t n, unsigned shift,
                                                                unsigned cb, unsigned* __restrict__ cursors, uint32_t koff, SliceDecode dec = SliceDecode()) {
    constexpr int NMAX = 512;
    static_assert(BLOCK >= NMAX, "one thread per class");
    constexpr int TILE = BLOCK * ITEMS;
    __shared__ V stage[TILE];
    __shared__ unsigned cnt[NMAX];
    __shared__ unsigned bstart[NMAX];
    __shared__ uint64_t gbase[NMAX];
    __shared__ unsigned scan_tmp[BLOCK / WAVE + 1];
    const unsigned ncls = 1u << cb;
    const unsigned tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    const uint64_t remain = n - base;
    const unsigned count = remain < (uint64_t)TILE ? (unsigned)remain : (unsigned)TILE;
    for (int i = tid; i < NMAX; i += BLOCK) cnt[i] = 0;
    __syncthreads();
    uint32_t key[ITEMS]; V val[ITEMS];
    unsigned slot[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned loc = tid + i * BLOCK;
        if (loc < count) {
            if (DEC) {
                const uint64_t x = reinterpret_cast<const uint64_t*>(key_in)[base + loc];
                key[i] = (uint32_t)x - koff; val[i] = (V)dec.rank_of(base + loc, (uint32_t)(x >> 32));
            } else { key[i] = key_in[base + loc] - koff; val[i] = val_in[base + loc]; }
        } else { key[i] = 0; val[i] = 0; }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned d = (key[i] >> shift) & (ncls - 1);
        slot[i] = (tid + i * BLOCK) < count ? atomicAdd(&cnt[d], 1u) : 0u;
    }
    __syncthreads();
    const unsigned tot = tid < ncls ? cnt[tid] : 0u;
    unsigned total;
    const unsigned bs = block_scan_exclusive<BLOCK, unsigned>(tot, OpSum(), 0u, scan_tmp, &total);
    if (tid < ncls) {
        bstart[tid] = bs;
        if (tot) {
            const uint32_t k0 = DEC ? (uint32_t)reinterpret_cast<const uint64_t*>(key_in)[base] : key_in[base];
            const uint64_t parent = (uint64_t)(k0 - koff) >> shift >> cb;        // same for the whole tile
            const uint64_t gq = (parent << cb) | tid;
            const unsigned at = atomicAdd(&cursors[gq], tot);
            gbase[tid] = (gq << shift) + at - bs;
        }
    }
    __syncthreads();
    uint32_t* const kstage = reinterpret_cast<uint32_t*>(stage);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned d = (key[i] >> shift) & (ncls - 1);
        slot[i] += bstart[d];
        if (tid + i * BLOCK < count) kstage[slot[i]] = key[i];
    }
    __syncthreads();
    uint64_t dest[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (p < count) {
            const uint32_t x = kstage[p];
            dest[j] = gbase[(x >> shift) & (ncls - 1)] + p;
            key_out[dest[j]] = x;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (tid + i * BLOCK < count) stage[slot[i]] = val[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (p < count) val_out[dest[j]] = stage[p];
    }
}

// One workgroup per window of 2^wb (<= 2^WBMAX) destinations: the window's pairs are scattered inside LDS and leave as
// consecutive entries of `out` (out[0] = destination koff).
template <typename V, typename TO, int BLOCK, int WBMAX, bool DEC = false>
__global__ __launch_bounds__(BLOCK) void pairs_window_kernel(const uint32_t* __restrict__ key, const V* __restrict__ val, uint64_t n, unsigned wb,
                                                             uint32_t koff, TO* __restrict__ out, SliceDecode dec = SliceDecode()) {
    __shared__ V win[1u << WBMAX];
    const unsigned W = 1u << wb;
    const uint64_t base = (uint64_t)blockIdx.x * W;
    const uint64_t remain = n - base;
    const unsigned count = remain < (uint64_t)W ? (unsigned)remain : W;
    for (unsigned p = threadIdx.x; p < count; p += BLOCK) {
        if (DEC) {
            const uint64_t x = reinterpret_cast<const uint64_t*>(key)[base + p];
            win[((uint32_t)x - koff) & (W - 1)] = (V)dec.rank_of(base + p, (uint32_t)(x >> 32));
        } else win[(key[base + p] - koff) & (W - 1)] = val[base + p];
    }
    __syncthreads();
    for (unsigned p = threadIdx.x; p < count; p += BLOCK) out[base + p] = (TO)win[p];
}

// ---- the same three kernels on packed pairs (ranks below 2^32: position in the low half of a 64-bit entry, rank in the high
// half): one array on the wire and in every level, one LDS staging round, runs twice as long (see partition_packed_kernel).
template <typename T, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void slice_partition_packed_kernel(const T* __restrict__ SA, const T* __restrict__ B, uint64_t n, SliceMap map,
                                                                       unsigned long long* __restrict__ cursors, uint64_t* __restrict__ out,
                                                                       T* __restrict__ b_copy = nullptr, int rel = 0, uint64_t rel_base = 0) {
    // rel: the rank leaves as rel_base - rank (SliceDecode), rel_base = last position of this rank's block
    constexpr int TILE = BLOCK * ITEMS;
    static_assert(BLOCK >= SLICE_MAX_CLASSES, "one thread per class");
    __shared__ uint64_t stage[TILE];
    __shared__ unsigned short cstage[TILE];
    __shared__ unsigned cnt[SLICE_MAX_CLASSES];
    __shared__ unsigned bstart[SLICE_MAX_CLASSES];
    __shared__ unsigned long long gbase[SLICE_MAX_CLASSES];
    __shared__ unsigned scan_tmp[BLOCK / WAVE + 1];
    const unsigned tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    const uint64_t remain = n - base;
    const unsigned count = remain < (uint64_t)TILE ? (unsigned)remain : (unsigned)TILE;
    for (int i = tid; i < SLICE_MAX_CLASSES; i += BLOCK) cnt[i] = 0;
    __syncthreads();
    uint64_t rec[ITEMS];
    unsigned short cls[ITEMS];
    unsigned slot[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned loc = tid + i * BLOCK;
        if (loc < count) {
            const uint64_t g = SA[base + loc];
            const unsigned o = map.owner(g);
            const uint64_t rel_pos = g - map.off(o);
            const T bb = B[base + loc];
            const uint64_t rk = (uint64_t)bb - 1u;
            rec[i] = (uint64_t)(uint32_t)rel_pos | ((uint64_t)(uint32_t)(rel ? rel_base - rk : rk) << 32);
            cls[i] = (unsigned short)(o * map.spo + (unsigned)(rel_pos >> map.sb));
            if (b_copy) b_copy[base + loc] = bb;
        } else { rec[i] = 0; cls[i] = 0; }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) slot[i] = (tid + i * BLOCK) < count ? atomicAdd(&cnt[cls[i]], 1u) : 0u;
    __syncthreads();
    const unsigned tot = tid < SLICE_MAX_CLASSES ? cnt[tid] : 0u;
    unsigned total;
    const unsigned bs = block_scan_exclusive<BLOCK, unsigned>(tot, OpSum(), 0u, scan_tmp, &total);
    if (tid < SLICE_MAX_CLASSES) {
        bstart[tid] = bs;
        if (tot) gbase[tid] = atomicAdd(&cursors[tid], (unsigned long long)tot);
    }
    __syncthreads();
    // staged by class; the class of every staged slot is kept beside it
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (tid + i * BLOCK < count) { const unsigned at = slot[i] + bstart[cls[i]]; stage[at] = rec[i]; cstage[at] = cls[i]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (p < count) { const unsigned cc = cstage[p]; out[gbase[cc] + (p - bstart[cc])] = stage[p]; }
    }
}

template <int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void pairs_partition_packed_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t n, unsigned shift,
                                                                       unsigned cb, unsigned* __restrict__ cursors, uint32_t koff) {
    constexpr int NMAX = 512;
    static_assert(BLOCK >= NMAX, "one thread per class");
    constexpr int TILE = BLOCK * ITEMS;
    __shared__ uint64_t stage[TILE];
    __shared__ unsigned cnt[NMAX];
    __shared__ unsigned bstart[NMAX];
    __shared__ uint64_t gbase[NMAX];
    __shared__ unsigned scan_tmp[BLOCK / WAVE + 1];
    const unsigned ncls = 1u << cb;
    const unsigned tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    const uint64_t remain = n - base;
    const unsigned count = remain < (uint64_t)TILE ? (unsigned)remain : (unsigned)TILE;
    for (int i = tid; i < NMAX; i += BLOCK) cnt[i] = 0;
    __syncthreads();
    uint64_t rec[ITEMS];
    unsigned slot[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned loc = tid + i * BLOCK;
        if (loc < count) { const uint64_t x = in[base + loc]; rec[i] = (x & 0xFFFFFFFF00000000ull) | (uint64_t)((uint32_t)x - koff); }
        else rec[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned d = ((uint32_t)rec[i] >> shift) & (ncls - 1);
        slot[i] = (tid + i * BLOCK) < count ? atomicAdd(&cnt[d], 1u) : 0u;
    }
    __syncthreads();
    const unsigned tot = tid < ncls ? cnt[tid] : 0u;
    unsigned total;
    const unsigned bs = block_scan_exclusive<BLOCK, unsigned>(tot, OpSum(), 0u, scan_tmp, &total);
    if (tid < ncls) {
        bstart[tid] = bs;
        if (tot) {
            const uint64_t parent = (uint64_t)((uint32_t)in[base] - koff) >> shift >> cb;        // same for the whole tile
            const uint64_t gq = (parent << cb) | tid;
            const unsigned at = atomicAdd(&cursors[gq], tot);
            gbase[tid] = (gq << shift) + at - bs;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const unsigned d = ((uint32_t)rec[i] >> shift) & (ncls - 1);
        if (tid + i * BLOCK < count) stage[slot[i] + bstart[d]] = rec[i];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned p = tid + j * BLOCK;
        if (p < count) {
            const uint64_t x = stage[p];
            out[gbase[((uint32_t)x >> shift) & (ncls - 1)] + p] = x;
        }
    }
}

template <typename TO, int BLOCK, int WBMAX>
__global__ __launch_bounds__(BLOCK) void pairs_window_packed_kernel(const uint64_t* __restrict__ pairs, uint64_t n, unsigned wb, uint32_t koff, TO* __restrict__ out) {
    __shared__ uint32_t win[1u << WBMAX];
    const unsigned W = 1u << wb;
    const uint64_t base = (uint64_t)blockIdx.x * W;
    const uint64_t remain = n - base;
    const unsigned count = remain < (uint64_t)W ? (unsigned)remain : W;
    for (unsigned p = threadIdx.x; p < count; p += BLOCK) { const uint64_t x = pairs[base + p]; win[((uint32_t)x - koff) & (W - 1)] = (uint32_t)(x >> 32); }
    __syncthreads();
    for (unsigned p = threadIdx.x; p < count; p += BLOCK) out[base + p] = (TO)win[p];
}

} // namespace psacx
