/* psacx.h -- C ABI of the MI355X-native suffix-array / ISA / LCP engine.
 *
 * This is the drop-in boundary for the SA+LCP path of patflick/psac.  Each
 * entry point names the reference interface (file:line under /root/reference)
 * it stands in for.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a negative PSACX_E* code;
 *     psacx_strerror() turns a code into text.  Nothing throws across the ABI
 *     (the reference throws std::runtime_error, suffix_array.hpp:226-227; the
 *     C++ mirror in include/suffix_array.hpp re-throws from these codes).
 *   - a psacx_ctx owns one HIP device, one stream and a reusable HBM
 *     workspace.  It is not re-entrant; use one ctx per host thread / per GPU.
 *   - "_dev" entry points take DEVICE pointers (text resident in HBM, results
 *     left in HBM); the plain ones take HOST pointers and stage over PCIe.
 *   - index type: _u32 needs n <= 2^32 - 2 (idxsort.hpp:39), _u64 needs n < 2^62.
 */
#ifndef PSACX_H
#define PSACX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psacx_ctx psacx_ctx;

enum {
    PSACX_OK = 0,
    PSACX_EINVAL = -1,     /* bad argument (null pointer, n == 0, k too large ...) */
    PSACX_ERANGE = -2,     /* n does not fit the index type */
    PSACX_ENOMEM = -3,     /* HBM workspace allocation failed */
    PSACX_EHIP = -4,       /* a HIP runtime call failed (psacx_last_hip_error) */
    PSACX_EDEVICE = -5,    /* a kernel reported an internal error (look-back timeout) */
    PSACX_ENOGPU = -6      /* no usable HIP device */
};

/* flags for psacx_construct_* */
enum {
    PSACX_LCP = 1u,        /* also build the LCP array (template flag _CONSTRUCT_LCP,
                              suffix_array.hpp:170) */
    PSACX_NO_FAST = 2u,    /* fast_resolval = false (suffix_array.hpp:470): keep sorting
                              every suffix each round instead of only unresolved buckets */
    PSACX_PROFILE = 4u     /* bracket every kernel class with HIP events (psacx_get_stats) */
};

#define PSACX_MAX_ROUNDS 72

typedef struct psacx_round {
    uint64_t h;                    /* prefix length already sorted when the round started */
    uint64_t active;               /* suffixes that took part in this round's sort */
    uint64_t unfinished_buckets;   /* as printed by suffix_array.hpp:416 / :961 */
    uint64_t unfinished_elements;
    uint32_t sort_passes;          /* radix passes executed */
    uint32_t sort_passes_skipped;  /* digits found constant by the histogram */
} psacx_round;

typedef struct psacx_stats {
    uint32_t sigma;                /* alphabet.hpp:147-155 */
    uint32_t bits_per_char;        /* alphabet.hpp:154 */
    uint32_t k;                    /* kmer.hpp:26-40 */
    uint32_t n_rounds;
    psacx_round rounds[PSACX_MAX_ROUNDS];
    /* event timers, milliseconds, filled when PSACX_PROFILE was set */
    double ms_total;               /* whole construct call, device side */
    double ms_alphabet, ms_kmer, ms_sort_hist, ms_sort_scatter, ms_rebucket, ms_isa_scatter,
           ms_gather, ms_compact, ms_rmq_build, ms_finalize;
    /* the radix scatter-pass kernel, counted per form: [0] = single-sweep look-back form
       (radix_scatter_kernel, small inputs), [1] = three-kernel form (radix_scatter3_kernel,
       large inputs; ms_sort_scatter3 times only that kernel, its per-tile histogram and
       offset scans are in ms_sort_tilehist), [2] = three-kernel form over two-word records
       (B1, idx): the prefix sort that opens the first round (ms_sort_scatter2) */
    double ms_sort_scatter3, ms_sort_tilehist, ms_sort_scatter2;
    uint64_t scatter_launches[3];  /* kernels launched */
    uint64_t scatter_records[3];   /* records moved by them (sum over launches) */
    uint64_t scatter_bytes[3];     /* algorithmic bytes: read + write of the record, 2 * 3w ([2]: 2 * 2w) per record */
    uint64_t hist_bytes;           /* algorithmic bytes of the histogram kernels: 2w per record */
    uint64_t workspace_bytes;      /* HBM held by the ctx */
    uint64_t onew_passes;          /* bucket passes run over one-word records (the MSD-first prefix sort of the first round); 0: that sort ran
                                      in the two-array form, whose passes are counted in scatter_*[2] alone */
    uint64_t heavy_rounds;         /* refinement rounds (or slabs of one) whose records were split into heavy and light ones (psac_amd/csrc/heavy_keys.hpp) */
    uint64_t heavy_records;        /* ... records that carried their bucket's heavy rank and skipped the sort */
    uint64_t light_records;        /* ... records that were sorted */
    uint64_t level_gathers;        /* refinement rounds (or slabs) whose ranks h further came through partition levels (construct.hpp: gather_by_levels) */
    /* host-pointer calls (psacx_construct_u32 / _u64), host wall clock, milliseconds: [0] text to the device, [1] the construction,
       [2] what was left of SA and LCP on their early way out when the construction returned (below), [3] ISA (and SA, LCP if they did not
       leave early) into the caller's arrays, [4] Lc, [5] the call; SA and LCP leaving early (from the end of the first round):
       [6] when that began, [7] = [8] when both were through, since the call began */
    double ms_host[9];
} psacx_stats;

/* life cycle ------------------------------------------------------------- */

/* Replaces suffix_array(const mxx::comm&) (suffix_array.hpp:174): binds the
 * engine to HIP device `device` (>= 0).  `stream` may be NULL (the ctx makes
 * its own), an existing hipStream_t passed as void*, or PSACX_STREAM_DEFAULT for
 * the device's default (null) stream, e.g. when sharing PyTorch's current stream. */
#define PSACX_STREAM_DEFAULT ((void*)(intptr_t)-1)
int psacx_create(psacx_ctx** out, int device, void* stream);
void psacx_destroy(psacx_ctx* ctx);
const char* psacx_strerror(int code);
/* text of the last HIP error seen by this ctx ("" if none) */
const char* psacx_last_hip_error(const psacx_ctx* ctx);
/* release the cached HBM workspace and the host-side staging of the host-pointer path: pinned ring, widening threads (all re-made on the next call that needs them) */
int psacx_trim(psacx_ctx* ctx);

/* Options of a context.  Every stage of the construction has one or more forms; the engine picks by text size, free
 * HBM and what it finds in the text.  An option pins the choice of one stage -- for the parity suite, which runs every form, and for A/B
 * timings.  None changes SA / ISA / LCP.  Values: 0 = the engine decides (default), 1 = on, or as listed.  An option stays set until it
 * is set again; psacx_configure(ctx, PSACX_OPT_RESET, 0) restores all defaults.  (The reference has no counterpart: its forms are
 * template parameters and #defines, e.g. suffix_array.hpp:170, :470.)
 *   FORCE_DIET        reduced-memory layout although the normal one fits
 *   DIET_CAP          at most this many records of room for the refinement rounds of the reduced layout (0 = what fits)
 *   ONE_STAGE         first round as one sort over both key words
 *   TIES_RADIX        ties of the two-stage first round through compaction + radix sort
 *   NO_ONE_WORD       the prefix sort of the first round in (word 1, suffix) passes, not one-word records
 *   ONE_WORD_ALWAYS   no repetition probe before the one-word prefix sort
 *   ONE_WORD_MIN      log2 of the smallest text that takes the one-word form (0 = 24)
 *   WIDEN_LAST        the last pass of the one-word prefix sort writes two arrays
 *   NO_DIGIT_BYTES    tile histograms of the bucket passes from the records
 *   NO_BUCKET_SORT    refinement rounds never sort inside LDS
 *   ISA_UPDATE        1 = one store per record, 2 = partition levels
 *   GATHER            ranks h further of a refinement round: 1 = one fetch per record, 2 = partition levels
 *   NO_HEAVY          no split of a round's records into heavy and light ones
 *   NO_WHOLE          no text-order rounds
 *   NO_LAZY_RANKS     the heavy runs of a split round always take the rank of their head and store it
 *   NO_EARLY_OUT      host-pointer calls: SA and LCP leave the device only when the construction has returned (default: from the moment the
 *                     first round has written them, under the SA -> ISA inversion; copied again if refinement rounds follow)             */
enum {
    PSACX_OPT_RESET = 0, PSACX_OPT_FORCE_DIET, PSACX_OPT_DIET_CAP, PSACX_OPT_ONE_STAGE, PSACX_OPT_TIES_RADIX, PSACX_OPT_NO_ONE_WORD,
    PSACX_OPT_ONE_WORD_ALWAYS, PSACX_OPT_ONE_WORD_MIN, PSACX_OPT_WIDEN_LAST, PSACX_OPT_NO_DIGIT_BYTES, PSACX_OPT_NO_BUCKET_SORT,
    PSACX_OPT_ISA_UPDATE, PSACX_OPT_GATHER, PSACX_OPT_NO_HEAVY, PSACX_OPT_NO_WHOLE, PSACX_OPT_NO_LAZY_RANKS, PSACX_OPT_NO_EARLY_OUT, PSACX_OPT_COUNT
};
int psacx_configure(psacx_ctx* ctx, int option, uint64_t value);
/* Debug shim, the ONLY place where the library looks at the environment, and only when called: resets the options of ctx and sets those
 * named by PSACX_<OPTION> variables (PSACX_FORCE_DIET=1, PSACX_DIET_CAP=<records>, PSACX_ISA_UPDATE=stores|levels, PSACX_GATHER=fetch|levels, ...).
 * The Python tests call it before every construction (psac_amd.ENV_KNOBS); products call psacx_configure. */
int psacx_configure_from_env(psacx_ctx* ctx);
/* value of environment variable `name` as the shim sees it (NULL if unset): shared with psacx_multi_configure_from_env */
const char* psacx_debug_env(const char* name);

/* construction ------------------------------------------------------------
 * Replace suffix_array<char,index_t,LCP>::construct(begin, end, fast_resolval, k)
 * (suffix_array.hpp:469-486 -> :365-466) for one rank holding the whole text.
 *   text  n bytes (any byte values; the alphabet is detected, alphabet.hpp:213-218)
 *   k     0 = auto (kmer.hpp:26-40), else upper bound on the k-mer length
 *   SA    out, n entries: local_SA        (suffix_array.hpp:204)
 *   ISA   out, n entries: local_B, 0-based inverse SA (suffix_array.hpp:206, :460-464)
 *   LCP   out, n entries or NULL: local_LCP (suffix_array.hpp:209); LCP[0] = 0
 */
int psacx_construct_u32(psacx_ctx* ctx, const uint8_t* text, uint64_t n, uint32_t k,
                        uint32_t flags, uint32_t* SA, uint32_t* ISA, uint32_t* LCP);
int psacx_construct_u64(psacx_ctx* ctx, const uint8_t* text, uint64_t n, uint32_t k,
                        uint32_t flags, uint64_t* SA, uint64_t* ISA, uint64_t* LCP);
int psacx_construct_dev_u32(psacx_ctx* ctx, const uint8_t* d_text, uint64_t n, uint32_t k,
                            uint32_t flags, uint32_t* d_SA, uint32_t* d_ISA, uint32_t* d_LCP);
int psacx_construct_dev_u64(psacx_ctx* ctx, const uint8_t* d_text, uint64_t n, uint32_t k,
                            uint32_t flags, uint64_t* d_SA, uint64_t* d_ISA, uint64_t* d_LCP);

/* suffix_array<char_t, index_t, true, true>::construct (suffix_array.hpp:170, :469-486): SA, ISA
 * and LCP as above (PSACX_LCP is implied) plus the left-branching characters local_Lc
 * (suffix_array.hpp:211-212; built at :1365-1383 and par_rmq.hpp:334-481; consumed by
 * desa.hpp:408, tldt.hpp:437):
 *   Lc    out, n bytes: Lc[i] = text[SA[i-1] + LCP[i]], 0 when that position is past the end
 *         (alphabet.hpp:168 decodes the end marker to '\0') and for i = 0.
 */
int psacx_construct_lc_u32(psacx_ctx* ctx, const uint8_t* text, uint64_t n, uint32_t k, uint32_t flags,
                           uint32_t* SA, uint32_t* ISA, uint32_t* LCP, uint8_t* Lc);
int psacx_construct_lc_u64(psacx_ctx* ctx, const uint8_t* text, uint64_t n, uint32_t k, uint32_t flags,
                           uint64_t* SA, uint64_t* ISA, uint64_t* LCP, uint8_t* Lc);
int psacx_construct_lc_dev_u32(psacx_ctx* ctx, const uint8_t* d_text, uint64_t n, uint32_t k, uint32_t flags,
                               uint32_t* d_SA, uint32_t* d_ISA, uint32_t* d_LCP, uint8_t* d_Lc);
int psacx_construct_lc_dev_u64(psacx_ctx* ctx, const uint8_t* d_text, uint64_t n, uint32_t k, uint32_t flags,
                               uint64_t* d_SA, uint64_t* d_ISA, uint64_t* d_LCP, uint8_t* d_Lc);

/* Generalized suffix array of a set of strings: suffix_array<>::construct_ss(simple_dstringset&,
 * alphabet) (suffix_array.hpp:267-363; string set stringset.hpp:33-81; k-mers kmer.hpp:269-355; shifts
 * shifting.hpp:374-418; bucket rules bucketing.hpp:130-143; tests test/test_gsa.cpp).
 *   text     the m strings back to back WITHOUT separators, n bytes in total
 *   offsets  m + 1 ascending offsets, offsets[0] = 0, offsets[m] = n, no empty string
 *   SA       every suffix of every string, positions counted in `text`; a suffix ends with its
 *            string, an end sorts below every character, equal suffixes come in text order
 *   ISA      inverse of SA;  LCP (with PSACX_LCP): common prefix inside the strings
 */
int psacx_construct_gsa_u32(psacx_ctx* ctx, const uint8_t* text, uint64_t n, const uint64_t* offsets, uint64_t m,
                            uint32_t k, uint32_t flags, uint32_t* SA, uint32_t* ISA, uint32_t* LCP);
int psacx_construct_gsa_u64(psacx_ctx* ctx, const uint8_t* text, uint64_t n, const uint64_t* offsets, uint64_t m,
                            uint32_t k, uint32_t flags, uint64_t* SA, uint64_t* ISA, uint64_t* LCP);
int psacx_construct_gsa_dev_u32(psacx_ctx* ctx, const uint8_t* d_text, uint64_t n, const uint64_t* d_offsets, uint64_t m,
                                uint32_t k, uint32_t flags, uint32_t* d_SA, uint32_t* d_ISA, uint32_t* d_LCP);
int psacx_construct_gsa_dev_u64(psacx_ctx* ctx, const uint8_t* d_text, uint64_t n, const uint64_t* d_offsets, uint64_t m,
                                uint32_t k, uint32_t flags, uint64_t* d_SA, uint64_t* d_ISA, uint64_t* d_LCP);

/* psacx_profile(ctx, 1): zero the statistics and let the step-level ops of psacx_ops.h add their
 * radix-pass event times and byte counts to them (psacx_get_stats reads the running totals);
 * psacx_profile(ctx, 0) stops it. */
int psacx_profile(psacx_ctx* ctx, int on);

/* statistics of the last construct call on this ctx (iteration log of
 * suffix_array.hpp:416, section timers of suffix_array.hpp:52-63) */
int psacx_get_stats(const psacx_ctx* ctx, psacx_stats* out);

/* verification on the device ---------------------------------------------------
 * Replaces check_SA / check_lcp / d_check_sa (check_suffix_array.hpp:56-88, :106-126, :207-267)
 * for buffers resident in HBM.  errors[0] = SA entries out of range or ISA[SA[i]] != i,
 * errors[1] = suffix-order violations, errors[2] = LCP entries that differ from a direct
 * character comparison (d_LCP may be NULL), errors[3] = LCP[0] != 0.  All zero = correct.
 * The LCP check costs sum(LCP) character reads: use it on texts without long repeats. */
int psacx_check_dev_u32(psacx_ctx* ctx, const uint8_t* d_text, uint64_t n, const uint32_t* d_SA,
                        const uint32_t* d_ISA, const uint32_t* d_LCP, uint64_t errors[4]);
int psacx_check_dev_u64(psacx_ctx* ctx, const uint8_t* d_text, uint64_t n, const uint64_t* d_SA,
                        const uint64_t* d_ISA, const uint64_t* d_LCP, uint64_t errors[4]);

/* benchmark inputs ----------------------------------------------------------------
 * psacx_rand_dna: the reference's generator rand_dna(size, seed) (alphabet.hpp:32-45, used by psac -r,
 * src/psac.cpp:89-93): srand(1337 * seed), then "ACGT"[rand() % 4] per character (glibc rand; host memory).
 * psacx_synth_text_dev: the synthetic texts of the benchmark configurations written straight
 * into HBM: characters first .. first + n of kind 0 = DNA(seed), 1 = ASCII128(seed) (splitmix64 streams, the same
 * definition as tests/inputs.py), 2 = TANDEM: DNA(seed) repeated with `period`, 3 = MUTATED: that repeat with one position in
 * 200 replaced (long shared prefixes that end somewhere: the repeated reads of a sequencing run, the human genome of pbs_run.sh:36). */
int psacx_rand_dna(uint8_t* out, uint64_t n, int seed);
int psacx_synth_text_dev(psacx_ctx* ctx, uint8_t* d_text, uint64_t n, uint64_t first, int kind, uint64_t seed,
                         uint64_t period);

/* the rank-pair sort on its own -------------------------------------------
 * Replaces idxsort_vectors(vec1, vec2, comm) (idxsort.hpp:23-83) at one rank:
 * sorts records (b1[i], b2[i], i) by (b1, b2); on return b1/b2 hold the sorted
 * keys and idx the permutation.  Device pointers, n entries each.  key_bits =
 * number of significant low bits in each key word (0 = all). */
int psacx_pair_sort_dev_u32(psacx_ctx* ctx, uint32_t* d_b1, uint32_t* d_b2, uint32_t* d_idx,
                            uint64_t n, uint32_t key_bits);
int psacx_pair_sort_dev_u64(psacx_ctx* ctx, uint64_t* d_b1, uint64_t* d_b2, uint64_t* d_idx,
                            uint64_t n, uint32_t key_bits);

/* all nearest smaller values ----------------------------------------------
 * Replaces ansv<T,left_type,right_type,global_indexing>(in, left, right, comm)
 * (ansv.hpp:2042-2051) at one rank.  type: 0 nearest_sm, 1 nearest_eq,
 * 2 furthest_eq (ansv_common.hpp:20-22).  Results are global indices, `nonsv`
 * where no such element exists.  Host pointers. */
int psacx_ansv_u32(psacx_ctx* ctx, const uint32_t* in, uint64_t n, int left_type, int right_type,
                   uint64_t nonsv, uint64_t* left_nsv, uint64_t* right_nsv);
int psacx_ansv_u64(psacx_ctx* ctx, const uint64_t* in, uint64_t n, int left_type, int right_type,
                   uint64_t nonsv, uint64_t* left_nsv, uint64_t* right_nsv);
/* same with the input (e.g. the LCP array psacx_construct_dev_* left in HBM) and both results in device memory */
int psacx_ansv_dev_u32(psacx_ctx* ctx, const uint32_t* d_in, uint64_t n, int left_type, int right_type,
                       uint64_t nonsv, uint64_t* d_left_nsv, uint64_t* d_right_nsv);
int psacx_ansv_dev_u64(psacx_ctx* ctx, const uint64_t* d_in, uint64_t n, int left_type, int right_type,
                       uint64_t nonsv, uint64_t* d_left_nsv, uint64_t* d_right_nsv);

/* suffix tree topology -------------------------------------------------------
 * Replaces construct_suffix_tree(sa, begin, end, comm) (suffix_tree.hpp:413-499, parents by
 * for_each_parent :43-223) at one rank.  nodes: n x (sigma + 1) table, row i = internal node of LCP
 * index i, cell c = child through the character with alphabet code c (0 = end of text), leaves are
 * n + i, 0 = no child.  Host pointers.  Call with nodes == NULL first to learn sigma. */
int psacx_suffix_tree_u32(psacx_ctx* ctx, const uint8_t* text, uint64_t n, const uint32_t* SA,
                          const uint32_t* LCP, uint64_t* nodes, uint32_t* sigma);
int psacx_suffix_tree_u64(psacx_ctx* ctx, const uint8_t* text, uint64_t n, const uint64_t* SA,
                          const uint64_t* LCP, uint64_t* nodes, uint32_t* sigma);

/* several GPUs -------------------------------------------------------------------
 * The reference's suffix_array<> IS distributed: every MPI rank holds one block of the text and of SA / ISA / LCP
 * (suffix_array.hpp:183-194, :217-228; src/psac.cpp:85-93 block-decomposes the input).  A psacx_multi stands for the
 * communicator: nranks ranks, one GPU each, of which nlocal live in this process.
 *   psacx_multi_create       one process (one host thread) drives ndev GPUs: ranks 0..ndev-1 = dev_ids[0..ndev-1]
 *                            (NULL: devices 0..ndev-1).  Distinct devices share one RCCL communicator
 *                            (ncclCommInitAll); a device listed more than once carries several ranks that exchange
 *                            by device-to-device copies (how the tests run P ranks on a one-GPU box).
 *   psacx_multi_create_rank  one process per GPU, psac's own deployment (one MPI rank per device): every process
 *                            passes the same 128-byte id, made by psacx_multi_unique_id on one of them and
 *                            broadcast by the host (MPI_Bcast in psac, torch.distributed in bench.py).
 * Exchanges replace mxx's collectives: grouped ncclSend / ncclRecv for MPI_Alltoallv (idxsort.hpp:60-62 via
 * mxx::sort, bulk_permute.hpp:60-61, bulk_rma.hpp:20-49, par_rmq.hpp:273-293), one small all-gather for the
 * scalars (bucketing.hpp:39,70,117), on a second stream per GPU.
 * psacx_multi_construct_dev_*: d_text[i] / m[i] / d_SA[i] ... are the block of local rank i (device pointers on that
 * rank's GPU); the blocks must follow mxx::blk_dist (the first n mod p ranks hold one character more,
 * suffix_array.hpp:226-227: PSACX_EINVAL otherwise).  psacx_multi_construct_*: the whole text and results on the
 * host of the single process that owns every rank (psac --gpus N).
 * Errors: the codes above, -7 for an RCCL failure; psacx_multi_last_error gives the text. */
typedef struct psacx_multi psacx_multi;
int psacx_multi_create(psacx_multi** out, int ndev, const int* dev_ids);
int psacx_multi_unique_id(void* id128);
int psacx_multi_create_rank(psacx_multi** out, int rank, int nranks, int device, const void* id128);
void psacx_multi_destroy(psacx_multi* mg);
int psacx_multi_nranks(const psacx_multi* mg);
int psacx_multi_nlocal(const psacx_multi* mg);
int psacx_multi_uses_rccl(const psacx_multi* mg);
/* How the ranks reach each other: 0 = device-to-device copies inside one process (ranks may share a device), 1 = RCCL
 * (grouped ncclSend / ncclRecv + ncclAllGather, the replacement of mxx's MPI_Alltoallv / MPI_Allgather, idxsort.hpp:60-62,
 * bucketing.hpp:39), 2 = one process per rank on one host staged through POSIX shared memory: psacx_multi_create_rank
 * with PSACX_MULTI_TRANSPORT=shm in the environment -- psac's own deployment without a GPU-aware MPI (src/psac.cpp:85-93),
 * and the way two processes can share one GPU, which RCCL refuses.  id128 is then any 128 bytes all ranks agree on.
 * A communicator that cannot be built makes psacx_multi_create / _create_rank fail with -7; nothing falls back silently.
 * PSACX_MULTI_FORCE_WIRE=1 (test switch): data a rank addresses to itself and scalars that are already on this host still
 * travel through ncclSend / ncclRecv / ncclAllGather. */
int psacx_multi_transport(const psacx_multi* mg);
/* after a call: ncclSend / ncclRecv / ncclAllGather calls this process really issued, and exchange_ms[i] = time the
 * exchanges occupied local rank i's second stream (HIP events).  Any pointer may be null. */
int psacx_multi_get_wire(const psacx_multi* mg, uint64_t* sends, uint64_t* recvs, uint64_t* allgathers, double* exchange_ms);
/* host wall time of the phases of the last construction as "name=ms;name=ms;..." (the section timers of
 * suffix_array.hpp:52-63 for this engine); PSACX_ERANGE if buf is too small */
int psacx_multi_get_phases(const psacx_multi* mg, char* buf, uint64_t cap);
/* which forms the last construction took: bit 0 = first round in two-word form (records (B1, idx), ties repaired from the
 * text owners; idxsort.hpp:23-83 moves (B1, B2, idx)), bit 1 = reduced-memory layout, bit 2 = SA -> ISA slice by slice
 * through the destination-partition levels (bulk_permute.hpp:14-73),
 * bit 4 = first round in one-word records dealt to the ranks by the top digit of the prefix (8 bytes per record on the wire),
 * bits 8..15 = slabs beyond the first in which the ties of that round were ordered (reduced-memory layout, repetitive text: the
 * records of one slab at a time take the place of idxsort.hpp:41-45's whole second record set) */
int psacx_multi_last_form(const psacx_multi* mg);
const char* psacx_multi_last_error(const psacx_multi* mg);
psacx_ctx* psacx_multi_ctx(psacx_multi* mg, int local_rank);
int psacx_multi_construct_dev_u32(psacx_multi* mg, const uint8_t* const* d_text, const uint64_t* m, uint32_t k, uint32_t flags,
                                  uint32_t* const* d_SA, uint32_t* const* d_ISA, uint32_t* const* d_LCP);
int psacx_multi_construct_dev_u64(psacx_multi* mg, const uint8_t* const* d_text, const uint64_t* m, uint32_t k, uint32_t flags,
                                  uint64_t* const* d_SA, uint64_t* const* d_ISA, uint64_t* const* d_LCP);
int psacx_multi_construct_u32(psacx_multi* mg, const uint8_t* text, uint64_t n, uint32_t k, uint32_t flags, uint32_t* SA,
                              uint32_t* ISA, uint32_t* LCP);
int psacx_multi_construct_u64(psacx_multi* mg, const uint8_t* text, uint64_t n, uint32_t k, uint32_t flags, uint64_t* SA,
                              uint64_t* ISA, uint64_t* LCP);
/* Distributed verification of block-distributed results without gathering them on one rank: d_check_sa
 * (check_suffix_array.hpp:207-267: SA a permutation with inverse ISA, S[SA[i-1]] <= S[SA[i]], ties decided by the
 * ranks of the suffixes one further) through the engine's own exchanges, plus -- beyond the reference, whose
 * distributed checker leaves LCP out -- every LCP entry through the recurrence LCP[i] = 0 | 1 | 1 + min(LCP[ISA[SA[i-1]
 * +1]+1 .. ISA[SA[i]+1]]).  errors[0..3] as psacx_check_dev_*, summed over all ranks (every rank gets the totals).
 * d_LCP may be NULL. */
int psacx_multi_check_dev_u32(psacx_multi* mg, const uint8_t* const* d_text, const uint64_t* m, const uint32_t* const* d_SA,
                              const uint32_t* const* d_ISA, const uint32_t* const* d_LCP, uint64_t errors[4]);
int psacx_multi_check_dev_u64(psacx_multi* mg, const uint8_t* const* d_text, const uint64_t* m, const uint64_t* const* d_SA,
                              const uint64_t* const* d_ISA, const uint64_t* const* d_LCP, uint64_t errors[4]);
/* Left-branching characters of block-distributed results (suffix_array<char_t, index_t, true, true>::local_Lc on p ranks,
 * suffix_array.hpp:211-212; filled by :1365-1383 and par_rmq.hpp:334-481 in the reference; by definition
 * Lc[i] = S[SA[i-1] + LCP[i]], desa.hpp:262-264, '\0' past the end and at i = 0): d_Lc[i] receives m[i] bytes for the
 * block of local rank i.  SA and LCP as psacx_multi_construct_dev_* left them. */
/* the host-pointer form (psacx_multi_construct_* + Lc[0..n)): flags must carry PSACX_LCP */
int psacx_multi_construct_lc_u32(psacx_multi* mg, const uint8_t* text, uint64_t n, uint32_t k, uint32_t flags, uint32_t* SA,
                                 uint32_t* ISA, uint32_t* LCP, uint8_t* Lc);
int psacx_multi_construct_lc_u64(psacx_multi* mg, const uint8_t* text, uint64_t n, uint32_t k, uint32_t flags, uint64_t* SA,
                                 uint64_t* ISA, uint64_t* LCP, uint8_t* Lc);
int psacx_multi_left_chars_dev_u32(psacx_multi* mg, const uint8_t* const* d_text, const uint64_t* m, const uint32_t* const* d_SA,
                                   const uint32_t* const* d_LCP, uint8_t* const* d_Lc);
int psacx_multi_left_chars_dev_u64(psacx_multi* mg, const uint8_t* const* d_text, const uint64_t* m, const uint64_t* const* d_SA,
                                   const uint64_t* const* d_LCP, uint8_t* const* d_Lc);
/* ansv<T, left_type, right_type, global_indexing>(in, left_nsv, right_nsv, comm) (ansv.hpp:2042-2051) over a
 * block-distributed array (blocks as mxx::blk_dist, e.g. the LCP blocks psacx_multi_construct_dev_* left in HBM):
 * d_left[i] / d_right[i] receive, per element of local rank i's block, the GLOBAL index of its nearest smaller value on
 * that side (types as psacx_ansv_*), nonsv where there is none.  Device pointers; results are uint64. */
int psacx_multi_ansv_dev_u32(psacx_multi* mg, const uint32_t* const* d_in, const uint64_t* m, int left_type, int right_type,
                             uint64_t nonsv, uint64_t* const* d_left, uint64_t* const* d_right);
int psacx_multi_ansv_dev_u64(psacx_multi* mg, const uint64_t* const* d_in, const uint64_t* m, int left_type, int right_type,
                             uint64_t nonsv, uint64_t* const* d_left, uint64_t* const* d_right);
/* construct_ss (suffix_array.hpp:267-363: the generalized suffix array of a string set) on p ranks: the strings lie back to
 * back without separators in the block-distributed text (kmer.hpp:269-355 cuts every k-mer at its string's end,
 * shifting.hpp:374-418 answers "the suffix h further" with 0 beyond it, bucketing.hpp:130-143 the bucket rules); offsets =
 * the nstr + 1 ascending GLOBAL offsets of the strings (host array, the same on every rank; offsets[0] = 0,
 * offsets[nstr] = n, no empty string).  Results as psacx_construct_gsa_*, block-distributed as psacx_multi_construct_dev_*. */
int psacx_multi_construct_gsa_dev_u32(psacx_multi* mg, const uint8_t* const* d_text, const uint64_t* m, const uint64_t* offsets, uint64_t nstr,
                                      uint32_t k, uint32_t flags, uint32_t* const* d_SA, uint32_t* const* d_ISA, uint32_t* const* d_LCP);
int psacx_multi_construct_gsa_dev_u64(psacx_multi* mg, const uint8_t* const* d_text, const uint64_t* m, const uint64_t* offsets, uint64_t nstr,
                                      uint32_t k, uint32_t flags, uint64_t* const* d_SA, uint64_t* const* d_ISA, uint64_t* const* d_LCP);
/* the host-pointer form (every rank in this process): the signature of psacx_construct_gsa_* */
int psacx_multi_construct_gsa_u32(psacx_multi* mg, const uint8_t* text, uint64_t n, const uint64_t* offsets, uint64_t nstr, uint32_t k,
                                  uint32_t flags, uint32_t* SA, uint32_t* ISA, uint32_t* LCP);
int psacx_multi_construct_gsa_u64(psacx_multi* mg, const uint8_t* text, uint64_t n, const uint64_t* offsets, uint64_t nstr, uint32_t k,
                                  uint32_t flags, uint64_t* SA, uint64_t* ISA, uint64_t* LCP);
/* construct_suffix_tree(sa, begin, end, comm) on p ranks (suffix_tree.hpp:413-499; parents by for_each_parent :43-223 from
 * the ANSV of LCP :62, cells sent to the owners of their rows like bulk_permute's pairs): d_nodes[i] receives the rows of the
 * LCP indices of local rank i's block, m[i] x (sigma + 1) cells of 64 bits, row-major; cell (j, c) = the child of internal
 * node off_i + j through the character with alphabet code c (0 = end of text), leaves numbered n + index, 0 = none -- the
 * table psacx_suffix_tree_* builds on one rank, block-distributed by rows.  *sigma = number of distinct characters of the
 * whole text; d_nodes == NULL only queries it.  SA and LCP as psacx_multi_construct_dev_* left them. */
int psacx_multi_suffix_tree_dev_u32(psacx_multi* mg, const uint8_t* const* d_text, const uint64_t* m, const uint32_t* const* d_SA,
                                    const uint32_t* const* d_LCP, uint64_t* const* d_nodes, uint32_t* sigma);
int psacx_multi_suffix_tree_dev_u64(psacx_multi* mg, const uint8_t* const* d_text, const uint64_t* m, const uint64_t* const* d_SA,
                                    const uint64_t* const* d_LCP, uint64_t* const* d_nodes, uint32_t* sigma);
/* the host-pointer form (the signature of psacx_suffix_tree_*; needs every rank in this process): nodes[n x (sigma + 1)] */
int psacx_multi_suffix_tree_u32(psacx_multi* mg, const uint8_t* text, uint64_t n, const uint32_t* SA, const uint32_t* LCP,
                                uint64_t* nodes, uint32_t* sigma);
int psacx_multi_suffix_tree_u64(psacx_multi* mg, const uint8_t* text, uint64_t n, const uint64_t* SA, const uint64_t* LCP,
                                uint64_t* nodes, uint32_t* sigma);
/* statistics of the last call (sigma, k, the per-round log) and what this process moved: payload bytes sent to other
 * ranks, number of all-to-all exchanges and of scalar all-gathers */
int psacx_multi_get_stats(const psacx_multi* mg, psacx_stats* out, uint64_t* bytes_sent, uint64_t* exchanges, uint64_t* gathers);
/* Memory layout of the distributed construction.  psac plans "6 words per character" for its distributed sort
 * (idxsort.hpp:43, suffix_array.hpp:751).  The normal layout here keeps every intermediate array of a phase at once
 * (fastest; up to ~14 words per character beside the three result arrays).  The reduced-memory layout lets the records of
 * the first round alternate between the rank's three result arrays and ONE allocated set of three arrays, runs SA -> ISA in
 * chunks, and works a refinement round with more unresolved suffixes than SLAB on some rank off in slabs of whole buckets
 * (results identical; the per-round counters of such a round may run ahead of the one-step log).
 *   PSACX_MULTI_OPT_LAYOUT        0 = choose by the free device memory of every rank (default), 1 = normal, 2 = reduced
 *   PSACX_MULTI_OPT_SLAB          unresolved suffixes per slab and rank (0 = block size / 16)
 *   PSACX_MULTI_OPT_OUTPUT_SLACK  the d_SA / d_ISA / d_LCP arrays handed to psacx_multi_construct_dev_* hold this many
 *                                 elements MORE than the block (m / 8 + 256 lets every rank use them as record arrays
 *                                 despite the sample sort's imbalance; without slack a rank falls back to allocating)
 * Forms of single stages (tests, A/B runs, traces; none changes the result; 0 = the engine decides unless listed):
 *   PSACX_MULTI_OPT_TRACE          1 = wall time of every phase on stderr (all local streams drained at each mark)
 *   PSACX_MULTI_OPT_WIRE_PIECE     largest message in bytes (0 = 2^28; a single 2^31-byte ncclSend arrives damaged on this stack)
 *   PSACX_MULTI_OPT_PIECES         ranges per destination of the first round's shuffle
 *   PSACX_MULTI_OPT_CHECK_CHUNKS   chunks of the distributed checker
 *   PSACX_MULTI_OPT_GLOBAL_REFINE_SORT  1 = refinement rounds sort all their records across the ranks
 *   PSACX_MULTI_OPT_ONE_STAGE      1 = first round as one sort over both key words
 *   PSACX_MULTI_OPT_TWO_WORD       first round in two-word records: 1 = never, 2 = also below 2^21 records per rank, 3 = additionally whatever the samples say
 *   PSACX_MULTI_OPT_ONE_WORD       first round in one-word records dealt by top digit: 1 = never, 2 = also for small blocks
 *   PSACX_MULTI_OPT_NO_SLICES      1 = SA -> ISA without destination slices
 *   PSACX_MULTI_OPT_SLICE_WIDE     1 = slice inversion on full words although 32-bit entries would do
 *   PSACX_MULTI_OPT_SLICE_SHAPE    window bits | slice bits << 8 | slices per step << 16 of the slice inversion (0 = by size)
 * psacx_multi_configure_from_env: the debug shim of psacx_configure_from_env for these options (PSACX_MULTI_DIET, PSACX_MULTI_SLAB,
 * PSACX_MULTI_TRACE, PSACX_MULTI_WIRE_PIECE, PSACX_MULTI_PIECES, PSACX_MULTI_CHECK_CHUNKS, PSACX_MULTI_GLOBAL_REFINE_SORT, PSACX_ONE_STAGE,
 * PSACX_MULTI_TWO_WORD, PSACX_MULTI_ONE_WORD, PSACX_MULTI_NO_SLICES, PSACX_SLICE_WIDE, PSACX_SLICE_SHAPE=wb,s1,step); options set through
 * psacx_multi_configure before it (layout, slab, slack) are kept unless a variable names them.  It also forwards to
 * psacx_configure_from_env for the rank contexts. */
#define PSACX_MULTI_OPT_LAYOUT 1
#define PSACX_MULTI_OPT_SLAB 2
#define PSACX_MULTI_OPT_OUTPUT_SLACK 3
#define PSACX_MULTI_OPT_TRACE 4
#define PSACX_MULTI_OPT_WIRE_PIECE 5
#define PSACX_MULTI_OPT_PIECES 6
#define PSACX_MULTI_OPT_CHECK_CHUNKS 7
#define PSACX_MULTI_OPT_GLOBAL_REFINE_SORT 8
#define PSACX_MULTI_OPT_ONE_STAGE 9
#define PSACX_MULTI_OPT_TWO_WORD 10
#define PSACX_MULTI_OPT_ONE_WORD 11
#define PSACX_MULTI_OPT_NO_SLICES 12
#define PSACX_MULTI_OPT_SLICE_WIDE 13
#define PSACX_MULTI_OPT_SLICE_SHAPE 14
int psacx_multi_configure_from_env(psacx_multi* mg);
/* Creation with explicit transport choices (psacx_multi_create / psacx_multi_create_rank = flags 0, shm_box_bytes 0):
 *   PSACX_MULTI_FORCE_WIRE  no shortcut for data a rank sends to itself or for scalars already on this host: every ncclSend / ncclRecv /
 *                           ncclAllGather is really issued (a single rank then drives RCCL too)
 *   PSACX_MULTI_NO_RCCL     peer copies between distinct devices of one process instead of a communicator
 *   PSACX_MULTI_SHM         (create_rank) one process per rank on one host, exchanges staged through POSIX shared memory: ranks may share a device;
 *                           shm_box_bytes = size of a rank's mailbox (0 = 32 MiB) */
#define PSACX_MULTI_FORCE_WIRE 1u
#define PSACX_MULTI_NO_RCCL 2u
#define PSACX_MULTI_SHM 4u
int psacx_multi_create_ex(psacx_multi** out, int ndev, const int* dev_ids, uint32_t flags);
int psacx_multi_create_rank_ex(psacx_multi** out, int rank, int nranks, int device, const void* id128, uint32_t flags, uint64_t shm_box_bytes);
int psacx_multi_configure(psacx_multi* mg, int option, uint64_t value);
/* after a construction: peak_bytes[i] = high-water mark of the device memory local rank i's engine had in use at once
 * (every array it allocated; the caller's text and result arrays are not in it; free blocks the rank keeps cached for
 * reuse beyond that are returned to the device whenever an allocation does not fit), *reduced = 1 if the reduced-memory
 * layout ran, *slab_rounds = refinement rounds worked off in more than one slab.  Any pointer may be null. */
int psacx_multi_get_memory(const psacx_multi* mg, uint64_t* peak_bytes, int* reduced, uint32_t* slab_rounds);

/* device memory helpers for hosts without their own HIP bindings ------------ */
int psacx_dev_alloc(psacx_ctx* ctx, void** out, uint64_t bytes);
int psacx_dev_free(psacx_ctx* ctx, void* p);
int psacx_copy_h2d(psacx_ctx* ctx, void* dst, const void* src, uint64_t bytes);
int psacx_copy_d2h(psacx_ctx* ctx, void* dst, const void* src, uint64_t bytes);
int psacx_sync(psacx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PSACX_H */
