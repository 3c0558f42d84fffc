// multi_kernels.hpp -- the small kernels of the multi-GPU engine's query side (multi_queries.hpp) and of its string-set path.
#pragma once
#include "dist_ops.hpp"
#include "ansv_wave.hpp"

namespace psacx {

// ---- kernels of the distributed left-branching characters (MultiRun::left_chars): the text position SA[i-1] + LCP[i] of
//      every entry of a piece (prev_sa: SA of the entry before the piece; has_prev = 0 at global position 0 -> n = "none"),
//      then the fetched characters narrowed to bytes ('\0' where the position is past the end, alphabet.hpp:168)
template <typename T>
__global__ void lc_queries_kernel(const T* __restrict__ SA, const T* __restrict__ LCP, uint64_t cnt, uint64_t n, int has_prev, T prev_sa,
                                  T* __restrict__ q) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        uint64_t p = n;
        if (i || has_prev) {
            p = (uint64_t)(i ? SA[i - 1] : prev_sa) + (uint64_t)LCP[i];
            if (p > n) p = n;
        }
        q[i] = (T)p;
    }
}
template <typename T>
__global__ void lc_narrow_kernel(const T* __restrict__ ch, const T* __restrict__ q, uint64_t cnt, uint64_t n, uint8_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride)
        out[i] = (uint64_t)q[i] < n ? (uint8_t)ch[i] : (uint8_t)0;
}

// ---- kernels of the distributed ANSV (MultiRun::ansv).  Start positions travel as T with one added (0 = before
//      position 0, n + 1 = past the end), "none" as all ones.
template <typename T>
__global__ void ansv_owner_kernel(const T* __restrict__ start1, uint64_t cnt, BlkDist d, T* __restrict__ cls) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        uint64_t s = (uint64_t)start1[j];
        s = s ? s - 1 : 0;
        if (s >= d.n) s = d.n - 1;
        cls[j] = (T)d.rank_of(s);
    }
}
// nearest element of this block strictly beyond start (left: below it) with value < thr (strict) or <= thr
template <typename T>
__global__ void nsv_from_enc_kernel(Pyramid<T> P, uint64_t m, uint64_t off, const T* __restrict__ start1, const T* __restrict__ thr,
                                    uint64_t cnt, int strict, int left, T* __restrict__ out_idx, T* __restrict__ out_val) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const long long s = (long long)(uint64_t)start1[j] - 1 - (long long)off;          // block-relative, may be < 0 or >= m
        const T v = thr[j];
        uint64_t r = NSV_NONE;
        if (m) {
            if (left) {
                if (s > 0) {
                    if ((uint64_t)s >= m) {
                        const T x = P.lvl[0][m - 1];
                        r = (strict ? x < v : x <= v) ? m - 1 : (m > 1 ? nsv_search<T, true>(P, m - 1, v, strict != 0) : NSV_NONE);
                    } else r = nsv_search<T, true>(P, (uint64_t)s, v, strict != 0);
                }
            } else if (s < (long long)m - 1) {
                if (s < 0) {
                    const T x = P.lvl[0][0];
                    r = (strict ? x < v : x <= v) ? 0 : (m > 1 ? nsv_search<T, false>(P, 0, v, strict != 0) : NSV_NONE);
                } else r = nsv_search<T, false>(P, (uint64_t)s, v, strict != 0);
            }
        }
        out_idx[j] = r == NSV_NONE ? ~(T)0 : (T)(off + r);
        out_val[j] = r == NSV_NONE ? (T)0 : P.lvl[0][r];
    }
}
// open queries (idx == none): the nearest rank beyond the start's owner whose block minimum qualifies, P = none
template <typename T>
__global__ void ansv_target_kernel(const T* __restrict__ own, const T* __restrict__ thr, const T* __restrict__ idx, uint64_t cnt, RankMins mins,
                                   RankMins sizes, int P, int strict, int left, T* __restrict__ target) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        int t = P;
        if (idx[j] == ~(T)0) {
            const unsigned long long v = (unsigned long long)thr[j];
            const int o = (int)own[j];
            if (left) { for (int b = o - 1; b >= 0; --b) if (sizes.v[b] && (strict ? mins.v[b] < v : mins.v[b] <= v)) { t = b; break; } }
            else { for (int b = o + 1; b < P; ++b) if (sizes.v[b] && (strict ? mins.v[b] < v : mins.v[b] <= v)) { t = b; break; } }
        }
        target[j] = (T)t;
    }
}
template <typename T>
__global__ void ansv_merge_kernel(T* __restrict__ idx, T* __restrict__ val, const T* __restrict__ idx2, const T* __restrict__ val2,
                                  const T* __restrict__ target, uint64_t cnt, int P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride)
        if ((int)target[j] < P) { idx[j] = idx2[j]; val[j] = val2[j]; }
}
template <typename T>
__global__ void fill_t_kernel(T* __restrict__ a, uint64_t cnt, T v) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) a[j] = v;
}
// local tile ANSV results (block-relative uint64, NSV_NONE = not inside the block) -> idx (global T, all ones = open) and value found
template <typename T>
__global__ void ansv_local_to_idx_kernel(const uint64_t* __restrict__ loc, const T* __restrict__ block, uint64_t cnt, uint64_t off,
                                         T* __restrict__ idx, T* __restrict__ val) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const uint64_t r = loc[j];
        idx[j] = r == NSV_NONE ? ~(T)0 : (T)(off + r);
        val[j] = r == NSV_NONE ? (T)0 : block[r];
    }
}
// start positions (plus one) for the follow-up searches of furthest_eq
template <typename T>
__global__ void ansv_next_start_kernel(const T* __restrict__ idx, uint64_t cnt, T when_none1, T* __restrict__ start1) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride)
        start1[j] = idx[j] == ~(T)0 ? when_none1 : (T)(idx[j] + 1);
}
template <typename T>
__global__ void ansv_finish_kernel(const T* __restrict__ first, const T* __restrict__ far, int use_far, uint64_t cnt, uint64_t nonsv,
                                   uint64_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const T a = first[j];
        const T r = (use_far && a != ~(T)0) ? far[j] : a;
        out[j] = r == ~(T)0 ? nonsv : (uint64_t)r;
    }
}

// Distributed check, per block (see MultiRun::check).  For SA position p = off + i:
//   back[i] = ISA[SA[p]] (must be p), ch[i] = S[SA[p]], nx[i] = ISA[SA[p] + 1] (undefined when SA[p] + 1 == n).
// Queries of the LCP recurrence: LCP[p] = 0 if the first characters differ, 1 if the smaller suffix is one character
// long, else 1 + min(LCP[ISA[SA[p-1]+1] + 1 .. ISA[SA[p]+1]]).
template <typename T>
__global__ void check_queries_kernel(const T* __restrict__ SA, const T* __restrict__ ch, const T* __restrict__ nx, uint64_t cnt, uint64_t n,
                                     int has_prev, T prev_sa, T prev_ch, T prev_nx, T* __restrict__ qlo, T* __restrict__ qhi) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        T lo = 0, hi = 1;                                   // a harmless query where none is needed
        if (i > 0 || has_prev) {
            const uint64_t a = i ? (uint64_t)SA[i - 1] : (uint64_t)prev_sa, b = SA[i];
            const T ca = i ? ch[i - 1] : prev_ch, na = i ? nx[i - 1] : prev_nx;
            if (a < n && b < n && ca == ch[i] && a + 1 < n && b + 1 < n && na < nx[i]) { lo = (T)(na + 1); hi = (T)(nx[i] + 1); }
        }
        qlo[i] = lo; qhi[i] = hi;
    }
}
template <typename T>
__global__ void check_verdict_kernel(const T* __restrict__ SA, const T* __restrict__ back, const T* __restrict__ ch, const T* __restrict__ nx,
                                     const T* __restrict__ LCP, const T* __restrict__ mins, uint64_t cnt, uint64_t off, uint64_t n,
                                     int has_prev, T prev_sa, T prev_ch, T prev_nx, unsigned long long* __restrict__ err) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned e0 = 0, e1 = 0, e2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        const uint64_t b = SA[i], p = off + i;
        if (b >= n || (uint64_t)back[i] != p) { ++e0; continue; }
        if (p == 0) { if (LCP && LCP[0] != 0) atomicAdd(&err[3], 1ull); continue; }
        if (i == 0 && !has_prev) continue;
        const uint64_t a = i ? (uint64_t)SA[i - 1] : (uint64_t)prev_sa;
        if (a >= n) continue;                               // counted where it lives
        const T ca = i ? ch[i - 1] : prev_ch, cb = ch[i];
        const T na = i ? nx[i - 1] : prev_nx, nb = nx[i];
        bool ok = ca < cb;
        if (ca == cb) ok = (a + 1 == n) || (b + 1 < n && na < nb);
        if (!ok) { ++e1; continue; }
        if (LCP) {
            uint64_t want;
            if (ca != cb) want = 0;
            else if (a + 1 == n) want = 1;
            else want = 1 + (uint64_t)mins[i];
            if ((uint64_t)LCP[i] != want) ++e2;
        }
    }
    e0 = wave_reduce<uint32_t>(e0, OpSum()); e1 = wave_reduce<uint32_t>(e1, OpSum()); e2 = wave_reduce<uint32_t>(e2, OpSum());
    if (lane_id() == 0) {
        if (e0) atomicAdd(&err[0], (unsigned long long)e0);
        if (e1) atomicAdd(&err[1], (unsigned long long)e1);
        if (e2) atomicAdd(&err[2], (unsigned long long)e2);
    }
}

// ------------------------------------------------------------------------------------------------------------
// number of leading entries <= key of a non-decreasing array (one thread)
// ---- string sets (construct_ss on p ranks, suffix_array.hpp:267-363): for the positions base .. base + cnt of the text,
// slen = characters to the end of the string holding the position, soff = characters from its start (off: the nstr + 1
// global string offsets).  Positions past the end of the text count as strings of one character.
template <typename T>
__global__ void string_pos_kernel(const uint64_t* __restrict__ off, uint64_t nstr, uint64_t n, uint64_t base, uint64_t cnt, T* __restrict__ slen,
                                  T* __restrict__ soff) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const uint64_t i = base + j;
        if (i >= n) { if (slen) slen[j] = (T)1; if (soff) soff[j] = (T)0; continue; }
        uint64_t lo = 0, hi = nstr;              // largest t with off[t] <= i
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
        if (slen) slen[j] = (T)(off[lo + 1] - i);
        if (soff) soff[j] = (T)(i - off[lo]);
    }
}
// the ranks a rank answers for "the suffix h further" in a string set: none (all ones) when that suffix starts in another
// string, i.e. when the position lies fewer than h characters into its own string (shifting.hpp:374-418)
template <typename T>
__global__ void mask_by_string_kernel(const T* __restrict__ isa, const T* __restrict__ soff, uint64_t m, uint64_t h, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) out[j] = (uint64_t)soff[j] >= h ? isa[j] : ~(T)0;
}
template <typename T>
__global__ void finish_b2_masked_kernel(const T* __restrict__ ans, const T* __restrict__ q, uint64_t cnt, uint64_t n, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride)
        out[j] = ((uint64_t)q[j] < n && ans[j] != ~(T)0) ? (T)(ans[j] + 1) : (T)0;
}

// ---- suffix-tree node table over block-distributed SA / LCP (suffix_tree.hpp:43-223 for_each_parent, :440-499)
// For the LCP index i = off + j: the parent of leaf n + i and (when there is one) of internal node i, from the ANSV of LCP
// (left furthest_eq, right nearest_sm, suffix_tree.hpp:62) and the LCP values found there; q = the text position whose
// character labels the edge.  An index without an internal-node record gets parent = i and q2 = ST_NOREC.
constexpr uint64_t ST_NOREC = ~0ull;
template <typename T>
__global__ void st_parents_kernel(const T* __restrict__ LCP, const T* __restrict__ SA, uint64_t m, uint64_t off, uint64_t n,
                                  const uint64_t* __restrict__ lnsv, const uint64_t* __restrict__ rnsv, const T* __restrict__ lcp_l,
                                  const T* __restrict__ lcp_r, int has_next, T next_lcp, T* __restrict__ p1, uint64_t* __restrict__ q1,
                                  T* __restrict__ p2, uint64_t* __restrict__ q2) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const uint64_t i = off + j, ln = lnsv[j], rn = rnsv[j], sa = SA[j], li = LCP[j];
        const uint64_t lnext = j + 1 < m ? (uint64_t)LCP[j + 1] : (has_next ? (uint64_t)next_lcp : 0);
        const uint64_t lv = ln != NSV_NONE ? (uint64_t)lcp_l[j] : 0, rv = rn != NSV_NONE ? (uint64_t)lcp_r[j] : 0;
        uint64_t parent, lcp_val;
        if (i == 0) { lcp_val = n > 1 ? lnext : 0; parent = lcp_val > 0 ? 1 : 0; }
        else if (i == n - 1 || li >= lnext) {
            lcp_val = lv;
            if (ln != NSV_NONE && lcp_val == li) parent = ln; else { parent = i; lcp_val = li; }
        } else { parent = i + 1; lcp_val = lnext; }
        p1[j] = (T)parent; q1[j] = sa + lcp_val;
        bool rec = !(i == 0 || li == 0);
        if (rec) {
            if (rn == NSV_NONE) { if (lv == li) rec = false; else { parent = ln; lcp_val = lv; } }
            else if (lv >= rv) { if (lv == li) rec = false; else { parent = ln; lcp_val = lv; } }
            else { parent = rn; lcp_val = rv; }
        }
        p2[j] = rec ? (T)parent : (T)i;
        q2[j] = rec ? sa + lcp_val : ST_NOREC;
    }
}
// the positions as index words for the bulk fetch (past the end / no record: position 0, the answer is not used)
template <typename T>
__global__ void st_positions_kernel(const uint64_t* __restrict__ q, uint64_t m, uint64_t n, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) out[j] = q[j] < n ? (T)q[j] : (T)0;
}
template <typename T>
__global__ void st_nsv_positions_kernel(const uint64_t* __restrict__ q, uint64_t m, T* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) out[j] = q[j] != NSV_NONE ? (T)q[j] : (T)0;
}
// what travels to the owner of the parent's row: x = the LCP index the child stands for, y = column | leaf flag << 16
// (0xFFFF: no record)
template <typename T>
__global__ void st_payload_kernel(const uint64_t* __restrict__ q, const T* __restrict__ ch, uint64_t m, uint64_t off, uint64_t n, CodeTable tab,
                                  int leaf, T* __restrict__ x, T* __restrict__ y) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        x[j] = (T)(off + j);
        if (q[j] == ST_NOREC) y[j] = (T)0xFFFFu;
        else y[j] = (T)((q[j] < n ? (unsigned)tab.c[(unsigned)ch[j] & 255u] : 0u) | ((unsigned)leaf << 16));
    }
}
template <typename T>
__global__ void st_put_kernel(unsigned long long* __restrict__ nodes, uint64_t off, uint64_t row, const T* __restrict__ pos, const T* __restrict__ x,
                              const T* __restrict__ y, uint64_t cnt, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += stride) {
        const unsigned yy = (unsigned)y[j];
        if ((yy & 0xFFFFu) == 0xFFFFu) continue;
        nodes[((uint64_t)pos[j] - off) * row + (yy & 0xFFFFu)] = (yy >> 16) ? n + (uint64_t)x[j] : (uint64_t)x[j];
    }
}

template <typename T> __global__ void upper_bound_kernel(const T* __restrict__ a, uint64_t n, uint64_t key, uint64_t* __restrict__ out) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if ((uint64_t)a[mid] <= key) lo = mid + 1; else hi = mid; }
    *out = lo;
}

} // namespace psacx
