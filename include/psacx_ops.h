/* psacx_ops.h -- step-level C ABI used by the block-distributed construction
 * (psac_amd/dist.py, one process per GPU).  Every function works on one rank's block of
 * a globally partitioned array; the exchanges between ranks (RCCL all-to-all, all-gather)
 * are done by the host between these calls.  Device pointers unless noted; `T` is
 * uint32_t (_u32) or uint64_t (_u64).  Return codes as in psacx.h.
 *
 * Each op names the loop of the reference (paths under /root/reference/include) it is the
 * single-rank part of.
 */
#ifndef PSACX_OPS_H
#define PSACX_OPS_H

#include "psacx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* last record of the nearest non-empty lower rank / first record of the nearest higher one
 * (mxx::right_shift of the last tuple, bucketing.hpp:77,100), the global position of local
 * element 0 and the bucket-id carry of all lower ranks (exscan(max), bucketing.hpp:39) */
typedef struct psacx_boundary {
    uint64_t off, base;
    int32_t has_prev, has_next;
    uint64_t prev[3], next[3];
} psacx_boundary;

/* alphabet.hpp:49-59: hist256 (device, 256 x uint64) += byte counts of text[0..n) */
int psacx_op_char_hist(psacx_ctx*, const uint8_t* text, uint64_t n, uint64_t* hist256);

#define PSACX_OPS_FOR(T, S)                                                                            \
    /* kmer.hpp:119-177 + shifting.hpp:33-122: packed 2k-character windows (l bits per character,     \
       code 0 = past the end) of text_halo[i ..] for i < m; codes256: host table */                    \
    int psacx_op_make_keys_##S(psacx_ctx*, const uint8_t* text_halo, uint64_t m, uint64_t text_len,    \
                               const uint16_t* codes256, uint32_t l, uint32_t c1, uint32_t c2, T* k1,  \
                               T* k2);                                                                 \
    int psacx_op_iota_##S(psacx_ctx*, T* out, uint64_t m, uint64_t start);                             \
    /* idxsort.hpp:23-83 on one rank: stable sort of (k1,k2,v) by the low bits1 / bits2 bits, radix    \
       passes ping-pong between the records and the scratch set (a1,a2,av); *where = 0 if the sorted   \
       records end in (k1,k2,v), 1 if in (a1,a2,av).  The input survives when at most one pass runs */ \
    int psacx_op_pair_sort_##S(psacx_ctx*, T* k1, T* k2, T* v, T* a1, T* a2, T* av, uint64_t n,        \
                               uint32_t bits1, uint32_t bits2, int32_t* where);                        \
    /* sample-sort shuffle (the splitter step of mxx::sort, idxsort.hpp:60-62): records are grouped,   \
       stably, by the number of splitters (k1,k2,rank,index tuples, host arrays, at most 63) that do    \
       not sort after them; result in (o1,o2,ov), class_start[0..nsplit+1] on the host */              \
    int psacx_op_split_by_##S(psacx_ctx*, const T* k1, const T* k2, const T* v, uint64_t n,            \
                              const uint64_t* sk1, const uint64_t* sk2, const uint64_t* srank,         \
                              const uint64_t* sidx, uint32_t nsplit, uint64_t my_rank, T* o1, T* o2,   \
                              T* ov, uint64_t* class_start);                                           \
    /* lower / upper bound of each query pair in the sorted (s1,s2); queries and results on the host */ \
    int psacx_op_pair_bounds_##S(psacx_ctx*, const T* s1, const T* s2, uint64_t n, const uint64_t* q1, \
                                 const uint64_t* q2, uint32_t nq, int use_second, uint64_t* lb,        \
                                 uint64_t* ub);                                                        \
    /* mxx::blk_dist::rank_of (bulk_permute.hpp:23) */                                                 \
    int psacx_op_owners_##S(psacx_ctx*, const T* gidx, uint64_t cnt, uint64_t n, uint32_t P, T* out);  \
    /* out[j] = block[min(gidx[j], n-1) - off]   /   block[gidx[j] - off] = vals[j] + delta */         \
    int psacx_op_take_##S(psacx_ctx*, const T* block, const T* gidx, uint64_t cnt, uint64_t off,       \
                          uint64_t n, T* out);                                                         \
    int psacx_op_put_##S(psacx_ctx*, T* block, const T* gidx, uint64_t cnt, uint64_t off,              \
                         const T* vals, int64_t delta);                                                \
    /* the same when gidx is a permutation of [off, off + cnt) (bulk_permute.hpp:14-73 for a whole    \
       block): destination-partition passes + LDS window scatter instead of cnt random stores;        \
       only delta = -1 is supported; s1..s4: scratch arrays of cnt entries */                          \
    int psacx_op_put_perm_##S(psacx_ctx*, T* block, const T* gidx, uint64_t cnt, uint64_t off,         \
                              const T* vals, T* s1, T* s2, T* s3, T* s4);                              \
    /* out = min(in + s, cap), formed in 64 bits (SA + h of suffix_array.hpp:978 cannot wrap) */      \
    int psacx_op_add_scalar_##S(psacx_ctx*, const T* in, uint64_t cnt, uint64_t s, uint64_t cap,       \
                                T* out);                                                               \
    /* suffix_array.hpp:972-996: out = q < n ? ans + 1 : 0 */                                          \
    int psacx_op_finish_b2_##S(psacx_ctx*, const T* ans, const T* q, uint64_t cnt, uint64_t n, T* out); \
    /* bucketing.hpp:57-123 on one block.  mode 0: first round (s3 = suffix starts, packed windows   \
       described by l,c1,c2), mode 1: refinement (s3 = SA positions of the list entries).             \
       last_head: id of the last bucket head inside the block (0 = none) */                            \
    int psacx_op_last_head_##S(psacx_ctx*, int mode, const T* s1, const T* s2, const T* s3,            \
                               uint64_t cnt, uint64_t n, uint32_t l, uint32_t c1, uint32_t c2,         \
                               const psacx_boundary* bd, uint64_t* last_head);                         \
    /* first round: bucket ids (bsa), k-mer LCP (lcp, may be NULL), counts of active positions and    \
       of buckets with more than one member (suffix_array.hpp:1353-1396, bucketing.hpp:98-118) */      \
    int psacx_op_rebucket_first_##S(psacx_ctx*, const T* s1, const T* s2, const T* sa, uint64_t cnt,   \
                                    uint64_t n, uint32_t l, uint32_t c1, uint32_t c2,                  \
                                    const psacx_boundary* bd, T* bsa, T* lcp, uint64_t* nact,          \
                                    uint64_t* nunf);                                                   \
    /* refinement (suffix_array.hpp:1092-1157, :1444-1476): writes sa_block / bsa_block at            \
       pos - off, ids_out per entry, sets LCP = h where the reference does, and appends the range-min \
       queries (q_at, q_lo, q_hi; capacity cnt) of the other new boundaries */                         \
    int psacx_op_rebucket_refine_##S(psacx_ctx*, const T* t1, const T* t2, const T* tv, const T* pos,  \
                                     uint64_t cnt, uint64_t n, uint64_t h, const psacx_boundary* bd,   \
                                     T* sa_block, T* bsa_block, T* lcp_block, T* ids_out, T* q_at,     \
                                     T* q_lo, T* q_hi, uint64_t* nq, uint64_t* nact, uint64_t* nunf);  \
    /* suffix_array.hpp:925-965: positions whose id equals a neighbour's; pos may be NULL (entry j   \
       sits at SA position off + j).  Returns the number written to pos_out (capacity cnt) */          \
    int psacx_op_compact_##S(psacx_ctx*, const T* ids, const T* pos, uint64_t cnt, uint64_t off,       \
                             uint64_t prev_id, uint64_t next_id, T* pos_out, uint64_t* n_out);         \
    /* par_rmq.hpp:199-332, the part one rank answers */                                               \
    int psacx_op_block_min_##S(psacx_ctx*, const T* block, uint64_t m, uint64_t* out);                 \
    int psacx_op_range_min_##S(psacx_ctx*, const T* block, uint64_t m, const T* lo, const T* hi,       \
                               uint64_t cnt, uint64_t off, T* out);                                    \
    int psacx_op_rmq_split_##S(psacx_ctx*, const T* lo, const T* hi, uint64_t cnt, uint64_t n,         \
                               uint32_t P, T* own1, T* lo1, T* hi1, T* own2, T* lo2, T* hi2, T* ra,    \
                               T* rb);                                                                 \
    int psacx_op_rmq_combine_##S(psacx_ctx*, const T* a1, const T* a2, const T* ra, const T* rb,       \
                                 uint64_t cnt, const uint64_t* rank_mins_host, uint32_t P, T* out);    \
    /* one search step of the distributed ANSV (ansv.hpp:2042-2051 over a block-distributed array): for  \
       every query the nearest element of this block strictly beyond global position start[j] (left != 0: \
       towards lower positions) with value < thr[j] (strict != 0) or <= thr[j]; start may lie outside the  \
       block.  idx = global position or all ones, val = its value. */                                      \
    int psacx_op_nsv_from_##S(psacx_ctx*, const T* block, uint64_t m, uint64_t off, const int64_t* start,  \
                              const T* thr, uint64_t cnt, int strict, int left, T* idx, T* val);           \
    /* suffix_array.hpp:1503-1505: block[at - off] = h + mins */                                       \
    int psacx_op_lcp_apply_##S(psacx_ctx*, T* block, const T* at, uint64_t cnt, uint64_t off,          \
                               const T* mins, uint64_t h);

PSACX_OPS_FOR(uint32_t, u32)
PSACX_OPS_FOR(uint64_t, u64)

#ifdef __cplusplus
}
#endif
#endif /* PSACX_OPS_H */
